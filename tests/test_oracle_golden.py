"""The CPU oracle (oracle/gcs_oracle.c) pinned against the reference's own committed result
records (benchmark_data/admm_solver_v3_benchmark{1..4}.pkl -> tests/golden/*.json, see
tests/golden/make_golden.py) and against analytic known answers.

Tolerances: stop iteration exact; every trace entry within
|x - golden| <= 2e-4 + 1e-3 |golden| (the goldens carry MOSEK's own ~1e-4 noise); cost rel 2e-4;
y_v abs 2e-3; x_v only where y_v ~ 1, abs 1e-3."""
from gcs_admm_amd import IPM_TOL
import numpy as np
import pytest

from conftest import BENCHMARKS
from gcs_admm_amd.cases import load_fixture


@pytest.mark.parametrize("name", BENCHMARKS)
def test_oracle_reproduces_reference_record(oracle_lib, name):
    case, g = load_fixture(name)
    gold = case["golden_v3"]
    o = oracle_lib.Oracle(g, ipm_tol=IPM_TOL)
    r = o.run(nthreads=4)
    assert r["status"] == 0 and r["inner_failures"] == 0
    assert r["iterations"] == gold["iterations"]                      # stop iteration: exact
    for key in ("pri_res_seq", "dual_res_seq"):
        mine, ref = r[key], np.array(gold[key])
        assert len(mine) == len(ref)
        assert np.all(np.abs(mine - ref) <= 2e-4 + 1e-3 * np.abs(ref)), key
    assert np.array_equal(r["rho_seq"], np.array(gold["rho_seq"]))     # rho never changed in the four records
    assert abs(r["cost"] - gold["cost"]) <= 2e-4 * gold["cost"]
    yv_ref = np.array(gold["y_v_sol"])
    assert np.max(np.abs(o.yv - yv_ref)) <= 2e-3
    xv_ref = np.array(gold["x_v_sol"])
    act = yv_ref >= 1 - 1e-6
    assert np.max(np.abs(o.xv[act] - xv_ref[act])) <= 1e-3
    # the relaxation optimum the monolithic solve reports is an upper bound the ADMM iterate approaches
    assert r["cost"] <= case["golden_classic"]["cost"] * (1 + 1e-3)


@pytest.mark.parametrize("name,pri1,stop,cost", [
    ("test1", 1.086278, 136, 0.420709), ("test2", 2.074849, 122, 2.694316), ("test3", None, 126, 1.898347)])
def test_oracle_known_answers(oracle_lib, name, pri1, stop, cost):
    """Appendix B: pri_1^2 = dual_1^2 = |s|^2 + |t|^2/2 + 1 when s and t each lie in one region;
    stop iterations / costs of the cases without a reference record (loose: +-2 iterations)."""
    case, g = load_fixture(name)
    o = oracle_lib.Oracle(g, ipm_tol=IPM_TOL)
    r = o.run()
    if pri1 is not None:
        # the analytic value assumes exact solves; the weakest words of a solve (y_e of an inactive edge) sit at mu / 1e-4 from theirs, so
        # at ipm_tol = 3e-9 the first residual is 1.6e-4 below it (5e-5 at 1e-9) -- the reference's own records are 1.25e-4 (benchmark1:
        # 3.999875 against 4) and 4.1e-4 (benchmark4: 20.312191 against 20.3126) below theirs
        assert abs(r["pri_res_seq"][1] - pri1) <= 2e-4 and abs(r["dual_res_seq"][1] - pri1) <= 2e-4
    assert abs(r["iterations"] - stop) <= 2
    assert abs(r["cost"] - cost) <= 1e-4


def test_oracle_invariants(oracle_lib):
    """mu of the two copies of a word cancel; an incoming edge's foreign word sits at its target;
    y_s = y_t = 1; edge activations in [0,1] (Appendix B, 'Other KATs')."""
    case, g = load_fixture("benchmark2")
    o = oracle_lib.Oracle(g, ipm_tol=IPM_TOL)
    for it in range(12):
        z0, m0 = o.zedge.copy(), o.mu.copy()
        assert o.vertex_step(1.0, 1.0) == 0
        for e in range(g.num_edges):
            ih = g.edge_inc_head[e]
            # head's copy of z_{e,tail}[:n] is only penalised: equals zedge - mu exactly
            assert np.allclose(o.copy[:g.n, ih], z0[:g.n, e] - m0[:g.n, ih], atol=1e-12)
        o.edge_step(1.0)
        assert np.max(np.abs(o.mu[:, g.edge_inc_tail] + o.mu[:, g.edge_inc_head])) <= 1e-12
        assert o.yv[g.src] == 1.0 and o.yv[g.dst] == 1.0
        assert o.zedge[2 * g.n].min() >= -1e-9 and o.zedge[2 * g.n].max() <= 1 + 1e-9


def test_oracle_matches_full_form_reference(oracle_lib):
    """The reduced 'arrow' form of the oracle against the reference's FULL form (every variable and
    constraint of admm_solver_v3.py:352-466) solved by an independent dense interior-point code."""
    import ref_dense
    case, g = load_fixture("benchmark1")
    dense = ref_dense.admm_v3(case, max_it=12)
    o = oracle_lib.Oracle(g, ipm_tol=IPM_TOL)
    r = o.run(max_it=12)
    k = 13
    assert np.allclose(r["pri_res_seq"][:k], dense["pri"][:k], rtol=1e-4, atol=1e-4)
    assert np.allclose(r["dual_res_seq"][:k], dense["dual"][:k], rtol=1e-4, atol=1e-4)


def test_rho_adaptation_branch(oracle_lib):
    """Quirk Q4: rho adapts only while it < 100 and rescales mu; the four reference records never
    exercise it, so drive it with a large initial rho."""
    case, g = load_fixture("benchmark1")
    o = oracle_lib.Oracle(g, ipm_tol=IPM_TOL)
    r = o.run(rho=64.0, max_it=130, eps_abs=1e-9, eps_rel=1e-9)
    rho = r["rho_seq"]
    assert r["iterations"] == 131 and len(rho) == 131
    assert len(set(rho)) > 1                       # adapted
    assert np.all(rho[99:] == rho[99])             # frozen: the last change can happen at it = 99
    ratio = rho[1:] / rho[:-1]
    assert set(np.unique(ratio)).issubset({0.5, 1.0, 2.0})
