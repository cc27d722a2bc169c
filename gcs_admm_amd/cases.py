"""Loading GCS cases: the committed JSON fixtures (tests/golden) and
reference-style ``test_data`` modules (``As, bs, n``; test_data/test1.py:26-33)."""
from __future__ import annotations

import json
import os

import numpy as np

from .graph import GcsGraph, graph_from_sets

_GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def load_fixture(name: str, directory: str | None = None):
    """Returns (case dict, GcsGraph).  The fixture carries the edge list, so no
    overlap tests are run."""
    with open(os.path.join(directory or _GOLDEN, f"{name}.json")) as f:
        case = json.load(f)
    keys = case["keys"]
    As = {k: np.array(a, float) for k, a in zip(keys, case["As"])}
    bs = {k: np.array(b, float) for k, b in zip(keys, case["bs"])}
    edges = [(u, w) for u, w in case["edges"]]
    return case, graph_from_sets(As, bs, case["n"], edges=edges)


def fixture_sets(name: str):
    """``As, bs, n, N, M`` of a committed fixture, as a reference-style case module exposes them."""
    with open(os.path.join(_GOLDEN, f"{name}.json")) as f:
        case = json.load(f)
    keys = case["keys"]
    As = {k: np.array(a, float) for k, a in zip(keys, case["As"])}
    bs = {k: np.array(b, float) for k, b in zip(keys, case["bs"])}
    return As, bs, case["n"], case.get("N", 0), case.get("M", 0)
