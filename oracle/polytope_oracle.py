"""CPU restatement of the reference's region-overlap decision and of the two auxiliary LPs of graph
construction -- TEST INFRASTRUCTURE ONLY (imported by tests/ and tools/bench_overlap.py's CPU leg; the
product path is gcs_admm_amd/scene.py + csrc/polytope_lp.hip and never touches this file).

  * ``overlap``            -- reference utils.py:49-65 ``check_overlap``: a program with the rows of both
                              regions, "do they share a point?" = LP feasibility (Drake + MOSEK there,
                              scipy / HiGHS here; closed sets, so touching regions overlap)
  * ``edges``              -- reference utils.py:31-82 ``build_graph``: all ordered pairs v1 != v2 in
                              double-loop order whose regions overlap
  * ``chebyshev`` / ``bounding_box`` -- the LPs the device path adds (largest inscribed ball; min / max of
                              every coordinate), solved with HiGHS

Pinned by the edge lists of the reference's own cases (tests/golden/*.json ``edges``, produced by the
reference's build_graph semantics on its test_data) -- tests/test_oracle_golden.py / tests/test_graph.py.
"""
from __future__ import annotations

import numpy as np
from scipy.optimize import linprog


def overlap(A1, b1, A2, b2) -> bool:
    A = np.vstack([A1, A2]); b = np.hstack([b1, b2])
    res = linprog(np.zeros(A.shape[1]), A_ub=A, b_ub=b, bounds=[(None, None)] * A.shape[1], method="highs")
    return bool(res.status == 0)


def overlap_radius(A1, b1, A2, b2) -> float:
    """Radius of the largest ball inscribed in the intersection (negative: by how much the rows must be
    relaxed to make it non-empty): the margin of the decision, used to skip ill-posed random cases."""
    A = np.vstack([A1, A2]); b = np.hstack([b1, b2])
    n = A.shape[1]
    nrm = np.linalg.norm(A, axis=1)
    c = np.zeros(n + 1); c[-1] = -1.0
    res = linprog(c, A_ub=np.hstack([A, nrm[:, None]]), b_ub=b, bounds=[(None, None)] * n + [(None, 1e6)], method="highs")
    return float(res.x[-1])


def edges(As, bs):
    keys = list(As.keys())
    return [(u, w) for u in keys for w in keys if u != w and overlap(As[u], bs[u], As[w], bs[w])]


def chebyshev(A, b):
    n = A.shape[1]
    nrm = np.linalg.norm(A, axis=1)
    c = np.zeros(n + 1); c[-1] = -1.0
    res = linprog(c, A_ub=np.hstack([A, nrm[:, None]]), b_ub=b, bounds=[(None, None)] * n + [(None, 1e6)], method="highs")
    return res.x[:n], float(res.x[-1])


def bounding_box(A, b):
    n = A.shape[1]
    lo = np.empty(n); hi = np.empty(n)
    for k in range(n):
        c = np.zeros(n); c[k] = 1.0
        lo[k] = linprog(c, A_ub=A, b_ub=b, bounds=[(None, None)] * n, method="highs").x[k]
        hi[k] = linprog(-c, A_ub=A, b_ub=b, bounds=[(None, None)] * n, method="highs").x[k]
    return lo, hi
