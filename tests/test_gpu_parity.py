"""GPU parity tests proper: the HIP path, called through the C ABI (gcs_admm_amd.solver ->
libgcsadmm.so), against the CPU oracle on the same inputs, against the reference's committed
records, and -- at the benchmark's full size -- through size-independent properties.

Stated tolerances (f64):
  * edge step / control: same arithmetic on both sides -> 1e-12 relative.
  * one vertex step from identical state: coupled words within 2e-3 absolute (observed worst ~6e-4), median over steps
    below 1e-5.  Both sides run the same interior-point iteration to barrier parameter 1e-9, which
    resolves the sub-problem's minimiser to ~1e-4 in its weakly determined components (the
    reference's own MOSEK solutions carry ~5e-4 there); two implementations
    that differ in operation order inherit that scale in the worst case.
  * whole runs: stop iteration identical; residual traces within |a-b| <= 2e-4 + 1e-3|b|; cost 2e-4 rel.
"""
from gcs_admm_amd import IPM_TOL
import numpy as np
import pytest

from conftest import BENCHMARKS
from gcs_admm_amd.cases import load_fixture
from gcs_admm_amd.graph import lattice_boxes

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_gpu():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


def _solver(g, dtype="f64", program="auto"):
    from gcs_admm_amd.solver import DeviceSolver
    return DeviceSolver(g, dtype, device=0, program=program)


PROGRAMS = ["wavefront", "workgroup"]


def _generic_mask(g):
    deg = np.diff(g.inc_ptr)
    din = np.array([int((g.inc_out[g.inc_ptr[v]:g.inc_ptr[v + 1]] == 0).sum()) for v in range(g.num_vertices)])
    gen = (din > 0) & (deg - din > 0)
    gen[g.src] = False; gen[g.dst] = False
    return gen


@pytest.mark.parametrize("program", PROGRAMS)
@pytest.mark.parametrize("name", ["benchmark1", "benchmark4", "test_autogen2", "benchmark3"])
def test_step_by_step_against_oracle(torch_gpu, oracle_lib, name, program):
    torch = torch_gpu
    case, g = load_fixture(name)
    o = oracle_lib.Oracle(g, ipm_tol=IPM_TOL)
    d = _solver(g, program=program)
    d.reset()
    diffs = []
    for it in range(30):
        # identical state on both sides
        d.zedge.copy_(torch.from_numpy(o.zedge)); d.mu.copy_(torch.from_numpy(o.mu))
        d.vertex_step()
        assert o.vertex_step(1.0, 1.0) == 0
        copy = d.copy.cpu().numpy()
        assert np.isfinite(copy).all()
        diffs.append(np.abs(copy - o.copy).max())
        assert np.abs(d.yv.cpu().numpy() - o.yv).max() <= 5e-4
        # special vertices (s, t, no-flow) are closed form: tight
        spec = ~_generic_mask(g)
        for v in np.nonzero(spec)[0]:
            sl = slice(g.inc_ptr[v], g.inc_ptr[v + 1])
            assert np.abs(copy[:, sl] - o.copy[:, sl]).max() <= 1e-12 if sl.stop > sl.start else True
        # edge step from the ORACLE's copies on both sides: same arithmetic
        d.copy.copy_(torch.from_numpy(o.copy))
        sums_dev = d.edge_step().cpu().numpy()
        sums_ref = o.edge_step(1.0)
        assert np.allclose(sums_dev, sums_ref, rtol=1e-12, atol=1e-300)
        assert np.array_equal(d.zedge.cpu().numpy(), o.zedge)
        assert np.allclose(d.mu.cpu().numpy(), o.mu, rtol=0, atol=1e-15)
    diffs = np.array(diffs)
    assert diffs.max() <= 2e-3 and np.median(diffs) <= 1e-5, (diffs.max(), np.median(diffs))
    cb = d.read_control()
    assert cb.status == -1 and cb.it == 1       # control was never run here


@pytest.mark.parametrize("program", PROGRAMS)
@pytest.mark.parametrize("name", BENCHMARKS)
def test_full_run_against_oracle_and_reference_record(torch_gpu, oracle_lib, name, program):
    case, g = load_fixture(name)
    gold = case["golden_v3"]
    d = _solver(g, program=program)
    res = d.solve()
    ora = oracle_lib.Oracle(g, ipm_tol=IPM_TOL).run()
    assert res["status"] == "converged" and res["inner_failures"] == 0
    assert res["iterations"] == ora["iterations"] == gold["iterations"]
    for key in ("pri_res_seq", "dual_res_seq"):
        for ref in (ora[key], np.array(gold[key])):
            assert np.all(np.abs(res[key] - ref) <= 2e-4 + 1e-3 * np.abs(ref)), key
    assert np.array_equal(res["rho_seq"], np.array(gold["rho_seq"]))
    assert abs(res["cost"] - ora["cost"]) <= 2e-4 * abs(ora["cost"])
    assert abs(res["cost"] - gold["cost"]) <= 2e-4 * gold["cost"]
    yv = d.yv.cpu().numpy()
    assert np.max(np.abs(yv - np.array(gold["y_v_sol"]))) <= 2e-3
    act = np.array(gold["y_v_sol"]) >= 1 - 1e-6
    assert np.max(np.abs(d.xv.cpu().numpy()[act] - np.array(gold["x_v_sol"])[act])) <= 1e-3


def test_rho_adaptation_and_mu_rescale(torch_gpu, oracle_lib):
    case, g = load_fixture("benchmark1")
    d = _solver(g)
    res = d.solve(rho=64.0, max_it=130, eps_abs=1e-9, eps_rel=1e-9)
    ora = oracle_lib.Oracle(g, ipm_tol=IPM_TOL).run(rho=64.0, max_it=130, eps_abs=1e-9, eps_rel=1e-9)
    assert res["status"] == "max_it" and res["iterations"] == ora["iterations"] == 131
    assert np.array_equal(res["rho_seq"], ora["rho_seq"]) and len(set(res["rho_seq"])) > 1
    assert np.all(np.abs(res["pri_res_seq"] - ora["pri_res_seq"]) <= 2e-4 + 1e-3 * np.abs(ora["pri_res_seq"]))


def test_small_cases_known_answers(torch_gpu):
    for name, pri1, stop, cost in (("test1", 1.086278, 136, 0.420709), ("test2", 2.074849, 122, 2.694316)):
        case, g = load_fixture(name)
        res = _solver(g).solve()
        assert abs(res["pri_res_seq"][1] - pri1) <= 2e-4 and abs(res["dual_res_seq"][1] - pri1) <= 2e-4      # (tests/test_oracle_golden.py: why 2e-4)
        assert abs(res["iterations"] - stop) <= 2 and abs(res["cost"] - cost) <= 1e-4


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_lattice_10k_properties(torch_gpu, oracle_lib, dtype):
    """BASELINE config 3 size (100x100 cells + s + t): properties that need no oracle run, plus an
    oracle spot check of the first iterations on the same graph (the oracle does 10k vertices in
    well under a second per iteration)."""
    torch = torch_gpu
    g = lattice_boxes(100, 100, seed=0)
    assert g.num_vertices == 10002
    d = _solver(g, dtype)
    d.reset(max_it=40)
    o = oracle_lib.Oracle(g, ipm_tol=IPM_TOL)
    n = g.n
    tail = torch.from_numpy(g.edge_inc_tail.astype(np.int64)).cuda()
    head = torch.from_numpy(g.edge_inc_head.astype(np.int64)).cuda()
    # (f32 state: the two duals of a word are updated separately and rounded to f32 each iteration; the bound is a few ulp of the
    #  largest dual per iteration, ~30 on this lattice, not a constant)
    tol = 1e-12 if dtype == "f64" else 1e-4
    for it in range(6):
        d.vertex_step()
        copy = d.copy.double()
        assert torch.isfinite(copy).all()
        sums = d.edge_step().clone()
        # the five norms recomputed with torch from the state the kernel left behind
        z = d.zedge.double(); mu = d.mu.double()
        r = torch.cat([copy[:, tail] - z, copy[:, head] - z], 1)
        assert torch.allclose(sums[0], (r * r).sum(), rtol=1e-9 if dtype == "f64" else 1e-4)
        assert torch.allclose(sums[2], (copy * copy).sum(), rtol=1e-9)
        assert torch.allclose(sums[3], (z * z).sum(), rtol=1e-9)
        assert torch.allclose(sums[4], (mu * mu).sum(), rtol=1e-9)
        # mu of the two copies of a word cancel; activations stay in [0, 1]; s and t are on
        assert (mu[:, tail] + mu[:, head]).abs().max().item() <= tol
        assert z[2 * n].min().item() >= -1e-6 and z[2 * n].max().item() <= 1 + 1e-6
        assert d.yv[g.src].item() == 1.0 and d.yv[g.dst].item() == 1.0
        d.control()
        if dtype == "f64":
            assert o.vertex_step(1.0, 1.0) == 0
            s_ref = o.edge_step(1.0)
            assert np.allclose(sums.cpu().numpy(), s_ref, rtol=1e-3, atol=1e-6)
    cb = d.read_control()
    assert cb.it == 7 and cb.status == -1 and cb.inner_failures == 0


def test_bad_arguments_are_reported(torch_gpu):
    import ctypes as C
    from gcs_admm_amd import solver
    lib = solver.load_library()
    h = C.c_void_p()
    assert lib.gcsadmm_create(None, C.byref(h)) == 1
    assert b"null" in lib.gcsadmm_last_error(None)
    case, g = load_fixture("test1")
    d = _solver(g)
    with pytest.raises(solver.GcsAdmmError, match="reset"):
        d.vertex_step()


def test_partitioned_handles_match_single(torch_gpu):
    """Two vertex partitions of one lattice as two handles on the same GPU, exchanging halo columns and
    summing the five norms by hand (the RCCL transport is exercised by bench.py --gpus N; the kernels'
    ghost-column / ownership-mask paths are what this covers)."""
    torch = torch_gpu
    from gcs_admm_amd.partition import build_partition, strip_owner
    from gcs_admm_amd.solver import DeviceSolver
    g = lattice_boxes(12, 16, seed=5)
    single = _solver(g)
    single.reset(max_it=60)
    owner = strip_owner(g, 2)
    parts = [build_partition(g, owner, r, 2) for r in range(2)]
    devs = [DeviceSolver(p.graph, "f64", device=0, num_incidences=p.num_incidences, inc_counted=p.inc_counted,
                         edge_counted=p.edge_counted, nx_global=p.nx_global, nmu_global=p.nmu_global) for p in parts]
    for d in devs:
        d.reset(max_it=60)
    idx = lambda a: torch.as_tensor(a, device="cuda")
    for it in range(25):
        single.vertex_step(); s_ref = single.edge_step().clone(); single.control()
        for d in devs:
            d.vertex_step()
        for r in range(2):
            o = 1 - r
            devs[r].copy.index_copy_(1, idx(parts[r].recv_idx[o]), devs[o].copy.index_select(1, idx(parts[o].send_idx[r])))
        sums = [d.edge_step().clone() for d in devs]
        tot = sums[0] + sums[1]
        assert torch.allclose(tot, s_ref, rtol=1e-3, atol=1e-9)      # vertex solves agree to solver accuracy
        for d in devs:
            d.control(tot)
    cbs = [d.read_control() for d in devs] + [single.read_control()]
    assert len({cb.it for cb in cbs}) == 1 and len({cb.status for cb in cbs}) == 1
    # the owned columns of the partitions reproduce the single-handle state
    full = single.zedge.cpu().numpy()
    for p, d in zip(parts, devs):
        assert np.allclose(d.zedge.cpu().numpy(), full[:, p.edge_global], rtol=0, atol=5e-4)


@pytest.mark.parametrize("n", [1, 3, 4, 5, 6, 7, 8])
def test_other_space_dimensions(torch_gpu, oracle_lib, n):
    """The sub-problem takes any space dimension (admm_solver_v3.py:363-377); BASELINE config 5 is a GCS in R^6.  The workgroup
    program instantiated for n = 1 .. 8 (n = 2 has its own tests; 7, 8 since round 4) against the oracle, step by step and over a short run: box
    lattices, and a chain of intervals for n = 1."""
    torch = torch_gpu
    if n == 1:
        from conftest import interval_chain
        from gcs_admm_amd.graph import graph_from_sets
        g = graph_from_sets(*interval_chain(8))
    else:
        g = lattice_boxes(6, 5, n=n, seed=1)
    o = oracle_lib.Oracle(g, ipm_tol=IPM_TOL)
    d = _solver(g)
    d.reset(max_it=50)
    worst = 0.0
    for it in range(10):
        d.zedge.copy_(torch.from_numpy(o.zedge)); d.mu.copy_(torch.from_numpy(o.mu))
        d.vertex_step()
        assert o.vertex_step(1.0, 1.0) == 0
        copy = d.copy.cpu().numpy()
        assert np.isfinite(copy).all()
        worst = max(worst, np.abs(copy - o.copy).max())
        o.edge_step(1.0)
    assert worst <= 2e-3
    d2 = _solver(g)
    res = d2.solve(max_it=40, eps_abs=0.0, eps_rel=0.0)
    ora = oracle_lib.Oracle(g, ipm_tol=IPM_TOL).run(max_it=40, eps_abs=0.0, eps_rel=0.0)
    assert res["iterations"] == ora["iterations"] == 41 and res["inner_failures"] == 0
    assert np.all(np.abs(res["pri_res_seq"] - ora["pri_res_seq"]) <= 2e-4 + 1e-3 * np.abs(ora["pri_res_seq"]))


def test_lattice_10k_trace_against_oracle(torch_gpu, oracle_lib):
    """BASELINE config 3 (10k-vertex lattice), f64: residual traces of the HIP loop against the oracle over 60
    iterations.  At this level the two implementations agree far better than the per-step worst case
    (observed 2e-9 relative); asserted at 1e-6."""
    g = lattice_boxes(100, 100, seed=0)
    d = _solver(g)
    res = d.solve(max_it=60, eps_abs=0.0, eps_rel=0.0)
    ora = oracle_lib.Oracle(g, ipm_tol=IPM_TOL).run(max_it=60, eps_abs=0.0, eps_rel=0.0, nthreads=32)
    assert res["inner_failures"] == 0 and ora["inner_failures"] == 0
    for key in ("pri_res_seq", "dual_res_seq"):
        a, b = res[key][1:], ora[key][1:]
        assert np.max(np.abs(a - b) / b) <= 1e-6, key
    # same graph with f32 state: same loop to storage precision
    d32 = _solver(g, "f32")
    r32 = d32.solve(max_it=60, eps_abs=0.0, eps_rel=0.0)
    assert np.max(np.abs(r32["pri_res_seq"][1:] - res["pri_res_seq"][1:]) / res["pri_res_seq"][1:]) <= 1e-4


def test_high_degree_vertex_and_degree_limit(torch_gpu, oracle_lib):
    from conftest import star_case
    from gcs_admm_amd.graph import graph_from_sets
    from gcs_admm_amd import solver
    As, bs, n = star_case(24)
    g = graph_from_sets(As, bs, n)
    assert np.diff(g.inc_ptr).max() >= 40
    d = _solver(g, program="wavefront")
    res = d.solve(max_it=80, eps_abs=0.0, eps_rel=0.0)
    ora = oracle_lib.Oracle(g, ipm_tol=IPM_TOL).run(max_it=80, eps_abs=0.0, eps_rel=0.0)
    assert res["inner_failures"] == 0
    for key in ("pri_res_seq", "dual_res_seq"):
        assert np.all(np.abs(res[key] - ora[key]) <= 2e-4 + 1e-3 * np.abs(ora[key])), key
    # more than 63 incident edges at one vertex (beyond one wavefront): that vertex runs the workgroup program
    # (admm_solver_v3.py:371 takes any I_v_in[v] + I_v_out[v])
    As, bs, n = star_case(40)
    g2 = graph_from_sets(As, bs, n)
    assert np.diff(g2.inc_ptr).max() > 63
    for program in ("wavefront", "auto"):
        d2 = _solver(g2, program=program)
        q = d2.query()
        assert q["num_workgroup_vertices"] >= 1 and (program != "wavefront" or q["num_waves"] >= 1)
        res = d2.solve(max_it=60, eps_abs=0.0, eps_rel=0.0)
        ora = oracle_lib.Oracle(g2, ipm_tol=IPM_TOL).run(max_it=60, eps_abs=0.0, eps_rel=0.0)
        assert res["inner_failures"] == 0
        for key in ("pri_res_seq", "dual_res_seq"):
            assert np.all(np.abs(res[key] - ora[key]) <= 2e-4 + 1e-3 * np.abs(ora[key])), key


def test_degenerate_graphs(torch_gpu, oracle_lib):
    """no edges at all; s and t in regions that do not touch (no s-t path): defined behaviour, no crash"""
    from gcs_admm_amd.graph import convert_pt_to_polytope, graph_from_sets
    A = np.vstack([np.eye(2), -np.eye(2)])
    As = {}; bs = {}
    As['s'], bs['s'] = convert_pt_to_polytope(np.array([0.0, 0.0]))
    As['t'], bs['t'] = convert_pt_to_polytope(np.array([5.0, 0.0]))
    g = graph_from_sets(As, bs, 2)
    assert g.num_edges == 0
    res = _solver(g).solve(max_it=30)
    ora0 = oracle_lib.Oracle(g, ipm_tol=IPM_TOL).run(max_it=30)
    # nmu = 0 makes eps_dual = 0 and the strict test `dual < eps_dual` (admm_solver_v3.py:712) can never pass
    assert res["status"] == "max_it" and res["iterations"] == ora0["iterations"] == 31 and res["cost"] == 0.0
    As[0], bs[0] = A, np.array([1.0, 1.0, 1.0, 1.0])          # holds s
    As[1], bs[1] = A, np.array([6.0, 1.0, -4.0, 1.0])         # holds t, disjoint from region 0
    g = graph_from_sets(As, bs, 2)
    d = _solver(g)
    res = d.solve(max_it=200)
    ora = oracle_lib.Oracle(g, ipm_tol=IPM_TOL).run(max_it=200)
    assert res["iterations"] == ora["iterations"] and res["status"] in ("converged", "max_it")
    assert np.isfinite(res["pri_res_seq"]).all() and np.isfinite(d.copy.cpu().numpy()).all()
    assert np.all(np.abs(res["pri_res_seq"] - ora["pri_res_seq"]) <= 2e-4 + 1e-3 * np.abs(ora["pri_res_seq"]))


@pytest.mark.parametrize("knobs", [dict(wave_align=2, wave_slots=7), dict(wave_align=1, wave_slots=7),
                                   dict(wave_align=2, wave_slots=7, wave_generic_rows=1),
                                   dict(wave_align=1, wave_slots=3, wave_generic_rows=1),
                                   dict(wave_align=1, wave_store_dl=2), dict(wave_align=2, wave_store_dl=2, wave_generic_rows=1),
                                   dict(wave_align=2, wave_slots=7, wave_generic_rows=2), dict(wave_align=1, wave_store_dl=2, wave_generic_rows=2),
                                   dict(program="workgroup")])
def test_packing_and_reduction_modes(torch_gpu, oracle_lib, knobs):
    """every schedule of the vertex step (wavefront program: dense packing + chained wave shifts, row-aligned packing + DPP
    row shifts, generic / 4-facet / canonical-box variant (the default on this lattice), 3 or 7 vertices per wavefront, update pass from stored directions or
    recomputed rows; workgroup program) gives the oracle's vertex step.  The schedule is part of the graph descriptor
    (include/gcsadmm.h), not of the process environment."""
    torch = torch_gpu
    from gcs_admm_amd.solver import DeviceSolver
    g = lattice_boxes(14, 11, seed=7)
    o = oracle_lib.Oracle(g, ipm_tol=IPM_TOL)
    kw = dict(program="wavefront"); kw.update(knobs)
    d = DeviceSolver(g, "f64", device=0, **kw)
    q = d.query()
    assert (q["num_workgroup_vertices"] > 0) == (kw["program"] == "workgroup") and (q["num_waves"] > 0) == (kw["program"] == "wavefront")
    d.reset()
    for it in range(12):
        d.zedge.copy_(torch.from_numpy(o.zedge)); d.mu.copy_(torch.from_numpy(o.mu))
        d.vertex_step()
        assert o.vertex_step(1.0, 1.0) == 0
        assert np.abs(d.copy.cpu().numpy() - o.copy).max() <= 1e-6
        assert np.abs(d.yv.cpu().numpy() - o.yv).max() <= 1e-6
        o.edge_step(1.0)


@pytest.mark.parametrize("seed", range(6))
def test_random_scenes_fuzz(torch_gpu, oracle_lib, seed):
    """random polygon scenes (3-7 facets, degrees 2-16), random penalty and dual scale: the vertex step of the
    device equals the oracle's from random (not just reachable) states"""
    torch = torch_gpu
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from scale_demo import polygon_scene
    from gcs_admm_amd.graph import graph_from_sets
    rng = np.random.default_rng(100 + seed)
    As, bs = polygon_scene(5 + seed % 3, seed=seed, m=3 + seed % 5)
    g = graph_from_sets(As, bs, 2)
    o = oracle_lib.Oracle(g, ipm_tol=IPM_TOL)
    d = _solver(g)
    rho = float([0.25, 1.0, 4.0][seed % 3])
    d.reset(rho=rho)
    gen = _generic_mask(g)
    for it in range(8):
        if it >= 4:   # a random state around the current one
            o.zedge += 0.05 * rng.normal(size=o.zedge.shape); o.mu += 0.02 * rng.normal(size=o.mu.shape)
        d.zedge.copy_(torch.from_numpy(o.zedge)); d.mu.copy_(torch.from_numpy(o.mu))
        d.vertex_step()
        assert o.vertex_step(rho, 1.0) == 0
        diff = np.abs(d.copy.cpu().numpy() - o.copy)
        # same bar as the step-by-step fixture tests: both solvers stop at mu <= 1e-9, weakly determined
        # components (flat directions of a sub-problem) differ by up to ~1e-5, the bulk by far less; worst single entry 2e-3 as on
        # reachable states.  (Round 3 had to allow 6e-3 here: from a RANDOM state one warm solve in some thousands jammed against the
        # bound y_e <= 1 -- steps of 1e-6 at mu ~ 1e-7 -- and left through the precision-exhausted rule with an error of a few 1e-3.
        # A warm solve no longer leaves through that rule: it is repeated cold)
        assert diff.max() <= 2e-3 and np.median(diff) <= 1e-7 and np.quantile(diff, 0.99) <= 1e-4
        assert np.abs(d.yv.cpu().numpy()[gen] - o.yv[gen]).max() <= 5e-4
        o.edge_step(1.0)
