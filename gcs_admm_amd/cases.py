"""Loading GCS cases.  The numeric content of a case (``As, bs, n`` of the reference's test_data modules,
test_data/test1.py:26-33, plus the edge list this repo's build_graph derives from it) ships with the product as
``test_data/<name>.json``; the reference's RESULT records for the same cases are test fixtures (tests/golden) and are
merged in only when that directory is present."""
from __future__ import annotations

import json
import os

import numpy as np

from .graph import GcsGraph, graph_from_sets

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_CASES = os.path.join(_ROOT, "test_data")
_GOLDEN = os.path.join(_ROOT, "tests", "golden")


def load_fixture(name: str, directory: str | None = None):
    """Returns (case dict, GcsGraph).  The case file carries the edge list, so no
    overlap tests are run.  ``golden_v3`` / ``golden_classic`` (the reference's result records) are added when the
    test fixtures are available."""
    with open(os.path.join(directory or _CASES, f"{name}.json")) as f:
        case = json.load(f)
    gold = os.path.join(_GOLDEN, f"{name}.json")
    if directory is None and os.path.exists(gold):
        with open(gold) as f:
            case.update({k: v for k, v in json.load(f).items() if k.startswith("golden_")})
    keys = case["keys"]
    As = {k: np.array(a, float) for k, a in zip(keys, case["As"])}
    bs = {k: np.array(b, float) for k, b in zip(keys, case["bs"])}
    edges = [(u, w) for u, w in case["edges"]]
    return case, graph_from_sets(As, bs, case["n"], edges=edges)


def fixture_sets(name: str):
    """``As, bs, n, N, M`` of a committed fixture, as a reference-style case module exposes them."""
    with open(os.path.join(_CASES, f"{name}.json")) as f:
        case = json.load(f)
    keys = case["keys"]
    As = {k: np.array(a, float) for k, a in zip(keys, case["As"])}
    bs = {k: np.array(b, float) for k, b in zip(keys, case["bs"])}
    return As, bs, case["n"], case.get("N", 0), case.get("M", 0)
