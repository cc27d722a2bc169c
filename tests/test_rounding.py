"""Rounding + convex restriction (gcs_admm_amd/rounding.py) against the reference's records: the rounded
path lengths of benchmark1-4 (3.236065 / 7.413745 / 60.177021 / 32.627198, recomputed from x_v_rounded,
y_v_rounded of the v3 records).  The relaxed activations fed in come from the CPU oracle's run, so this
test needs no GPU.  Path *sets* are not compared: ties exist (benchmark3's record passes through a region
the equally short most-probable path skips)."""
from gcs_admm_amd import IPM_TOL
import numpy as np
import pytest

from conftest import BENCHMARKS
from gcs_admm_amd.cases import fixture_sets, load_fixture
from gcs_admm_amd.rounding import rounding, solve_path_restriction


@pytest.mark.parametrize("name", BENCHMARKS)
def test_rounded_length_matches_reference_record(oracle_lib, name):
    case, g = load_fixture(name)
    As, bs, n, _, _ = fixture_sets(name)
    o = oracle_lib.Oracle(g, ipm_tol=IPM_TOL)
    o.run(nthreads=4)
    V, E = g.keys, g.edges_as_keys()
    y_e = {e: float(o.zedge[2 * n, i]) for i, e in enumerate(E)}
    I_out = {v: [e for e in E if e[0] == v] for v in V}
    cost, xv, yv = rounding(y_e, V, E, I_out, As, bs, n)
    gold = case["golden_v3"]
    gx, gy = np.array(gold["x_v_rounded"]), np.array(gold["y_v_rounded"])
    glen = sum(np.linalg.norm(gx[i][:n] - gx[i][n:]) for i in range(len(V)) if gy[i] == 1)
    assert abs(cost - glen) <= 1e-5 * glen
    # the result is a feasible s-t chain: active vertices contain their segment, consecutive ends meet
    assert yv['s'] == 1 and yv['t'] == 1
    for v in V:
        if yv[v]:
            for half in (xv[v][:n], xv[v][n:]):
                assert np.all(As[v] @ half <= bs[v] + 1e-6)
    # and it is no longer than the path the reference's own record holds, re-optimised
    ref_path_cost, _ = solve_path_restriction(As, bs, n, ['s'] + _order_path(gx, gy, V, n)[1:])
    assert cost <= ref_path_cost * (1 + 1e-6)


def _order_path(gx, gy, V, n):
    """order the active vertices of a record by chaining segment ends"""
    act = [i for i in range(len(V)) if gy[i] == 1]
    cur = V.index('s'); path = ['s']; left = set(act) - {cur}
    while left:
        end = gx[cur][n:]
        nxt = min(left, key=lambda i: np.linalg.norm(gx[i][:n] - end))
        path.append(V[nxt]); left.remove(nxt); cur = nxt
    return path
