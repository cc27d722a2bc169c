"""GPU tests of the BASELINE.json configurations at their FULL sizes (the parity tests of test_gpu_parity.py use the
reference's own cases and small lattices):

  config 2  benchmark4, f64, tolerance 1e-6 (iterations-to-eps run)            test_benchmark4_tol_1e6_against_oracle
  config 3  10k-vertex lattice, f32 state, reference defaults to its stop      test_lattice_10k_f32_runs_to_the_same_stop
  config 4  100k-vertex lattice (316 x 317), one handle and 8 partitions       test_lattice_100k_*
  config 5  50k-vertex lattice in R^6 (223 x 224)                             test_lattice_r6_50k

Where the oracle can afford the size it is run beside the device (same inner solver, tight tolerances); everything else
is checked through properties that need no second implementation: the five norms recomputed with torch from the state
the kernels left behind, the mu-pair invariant, activations in [0, 1], s / t switched on, no inner failure.
Reference loop: admm_solver_v3.py:655-733."""
from gcs_admm_amd import IPM_TOL
import numpy as np
import pytest

from gcs_admm_amd.cases import load_fixture
from gcs_admm_amd.graph import lattice_boxes

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_gpu():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


def _solver(g, dtype="f64", **kw):
    from gcs_admm_amd.solver import DeviceSolver
    return DeviceSolver(g, dtype, device=0, **kw)


def _check_state_properties(torch, g, d, sums, dtype, it=0):
    """the five norms recomputed from the state, mu-pair invariant, activations in [0, 1], terminals on"""
    n = g.n
    tail = torch.from_numpy(g.edge_inc_tail.astype(np.int64)).cuda()
    head = torch.from_numpy(g.edge_inc_head.astype(np.int64)).cuda()
    copy = d.copy.double(); z = d.zedge.double(); mu = d.mu.double()
    assert torch.isfinite(copy).all()
    r = torch.cat([copy[:, tail] - z, copy[:, head] - z], 1)
    assert torch.allclose(sums[0], (r * r).sum(), rtol=1e-9 if dtype == "f64" else 1e-4)
    assert torch.allclose(sums[2], (copy * copy).sum(), rtol=1e-9)
    assert torch.allclose(sums[3], (z * z).sum(), rtol=1e-9)
    assert torch.allclose(sums[4], (mu * mu).sum(), rtol=1e-9)
    # the two duals of a coupled word cancel: exactly in f64 arithmetic up to round-off, to storage precision (a few f32
    # ulps of the largest dual: the 100k lattice has coordinates ~300) with f32 state
    # (f32: each dual update rounds both duals of a pair to storage independently, so the pair sum drifts by about an ulp per iteration)
    pair_tol = 1e-12 if dtype == "f64" else 1.2e-7 * (it + 4) * max(1.0, mu.abs().max().item())
    assert (mu[:, tail] + mu[:, head]).abs().max().item() <= pair_tol
    assert z[2 * n].min().item() >= -1e-6 and z[2 * n].max().item() <= 1 + 1e-6
    assert d.yv[g.src].item() == 1.0 and d.yv[g.dst].item() == 1.0


@pytest.fixture(scope="module")
def lattice_100k():
    g = lattice_boxes(316, 317, seed=0)
    assert g.num_vertices == 100174
    return g


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_lattice_100k_single_handle(torch_gpu, oracle_lib, lattice_100k, dtype):
    """BASELINE config 4's graph on one GPU: properties every iteration; f64: the residual trace of the first
    iterations against the oracle at 1e-6 relative (same inner solver on both sides)."""
    torch = torch_gpu
    g = lattice_100k
    d = _solver(g, dtype)
    d.reset(max_it=40)
    iters = 10 if dtype == "f64" else 6
    for it in range(iters):
        d.vertex_step()
        sums = d.edge_step().clone()
        _check_state_properties(torch, g, d, sums, dtype, it)
        d.control()
    cb = d.read_control()
    assert cb.it == iters + 1 and cb.status == -1 and cb.inner_failures == 0
    if dtype == "f64":
        ora = oracle_lib.Oracle(g, ipm_tol=IPM_TOL).run(max_it=iters, eps_abs=0.0, eps_rel=0.0, nthreads=16)
        tr = d.trace[:iters].cpu().numpy()
        assert ora["inner_failures"] == 0
        for col, key in ((1, "pri_res_seq"), (2, "dual_res_seq")):
            a, b = tr[:, col], ora[key][1:iters + 1]
            assert np.max(np.abs(a - b) / b) <= 1e-6, key


def test_lattice_100k_eight_partitions_match_single(torch_gpu, lattice_100k):
    """BASELINE config 4 as it is sharded: 8 row strips = 8 handles (here on one GPU), halo columns copied by hand, the
    five norms summed over the partitions -- against the single handle.  Same stop decisions, sums to 1e-3 (vertex
    solves agree to solver accuracy), edge copies to 5e-4."""
    torch = torch_gpu
    from gcs_admm_amd.partition import build_partition, strip_owner
    from gcs_admm_amd.solver import DeviceSolver
    g = lattice_100k
    world = 8
    single = _solver(g, "f32")
    single.reset(max_it=60)
    owner = strip_owner(g, world)
    parts = [build_partition(g, owner, r, world) for r in range(world)]
    assert all(len(p.send_idx) <= 2 for p in parts)           # row strips: at most two neighbours
    assert sum(p.graph.num_vertices for p in parts) == g.num_vertices
    devs = [DeviceSolver(p.graph, "f32", device=0, num_incidences=p.num_incidences, inc_counted=p.inc_counted,
                         edge_counted=p.edge_counted, nx_global=p.nx_global, nmu_global=p.nmu_global) for p in parts]
    for d in devs:
        d.reset(max_it=60)
    idx = {(r, o): (torch.as_tensor(parts[r].recv_idx[o], device="cuda"), torch.as_tensor(parts[o].send_idx[r], device="cuda"))
           for r in range(world) for o in parts[r].recv_idx}
    for it in range(25):
        single.vertex_step(); s_ref = single.edge_step().clone(); single.control()
        for d in devs:
            d.vertex_step()
        for (r, o), (rix, six) in idx.items():
            devs[r].copy.index_copy_(1, rix, devs[o].copy.index_select(1, six))
        tot = torch.zeros(5, dtype=torch.float64, device="cuda")
        for d in devs:
            tot += d.edge_step()
        assert torch.allclose(tot, s_ref, rtol=1e-3, atol=1e-9)
        for d in devs:
            d.control(tot)
    cbs = [d.read_control() for d in devs] + [single.read_control()]
    assert len({cb.it for cb in cbs}) == 1 and len({cb.status for cb in cbs}) == 1 and cbs[0].it == 26
    assert all(cb.inner_failures == 0 for cb in cbs)
    full = single.zedge.cpu().numpy()
    for p, d in zip(parts, devs):
        assert np.allclose(d.zedge.cpu().numpy(), full[:, p.edge_global], rtol=0, atol=5e-4)


def test_lattice_r6_50k(torch_gpu, oracle_lib):
    """BASELINE config 5 at full size (223 x 224 boxes in R^6, f32 state): the workgroup program, BOX instantiation.  24 iterations:
    the state properties on every one (20 of them with warm-started solves at full size), no inner failure, and the residual trace
    of the first four -- one cold and three warm iterations -- against the oracle at 1e-6 relative (the oracle takes ~2 s per
    iteration here)."""
    torch = torch_gpu
    g = lattice_boxes(223, 224, n=6, seed=0)
    assert g.num_vertices == 49954 and g.n == 6
    d = _solver(g, "f32")
    q = d.query()
    assert q["num_workgroup_vertices"] == g.num_vertices - 2 and q["num_waves"] == 0
    n_it, n_cmp = 24, 4
    d.reset(max_it=n_it + 5)
    first = None
    iters = []
    for it in range(n_it):
        d.vertex_step()
        sums = d.edge_step().clone()
        _check_state_properties(torch, g, d, sums, "f32", it)
        if it == 0:
            first = sums.cpu().numpy()
        d.control()
        cb = d.read_control()
        assert cb.inner_failures == 0 and cb.status == -1, (it, cb.inner_failures, cb.status)
        iters.append(cb.inner_iters / (g.num_vertices - 2))
    assert cb.it == n_it + 1
    assert iters[-1] < 0.6 * iters[0], iters          # the later solves really restart from their records (about 3 against 9 iterations)
    tr = d.trace[:n_cmp].cpu().numpy()
    o = oracle_lib.Oracle(g, ipm_tol=IPM_TOL)
    ora = o.run(max_it=n_cmp, eps_abs=0.0, eps_rel=0.0, nthreads=16)
    assert ora["inner_failures"] == 0
    # f32 storage of the state: the residuals of a 50k-vertex lattice agree to ~1e-6 relative (the S10k / S100k bound)
    assert np.allclose(tr[:, 1], ora["trace"][:n_cmp, 1], rtol=2e-6, atol=1e-9), (tr[:, 1], ora["trace"][:n_cmp, 1])
    assert np.allclose(tr[:, 2], ora["trace"][:n_cmp, 2], rtol=2e-6, atol=1e-9), (tr[:, 2], ora["trace"][:n_cmp, 2])
    o1 = oracle_lib.Oracle(g, ipm_tol=IPM_TOL)
    assert o1.vertex_step(1.0, 1.0, nthreads=16) == 0
    assert np.allclose(first, o1.edge_step(1.0), rtol=1e-4, atol=1e-8)


def test_benchmark4_tol_1e6_against_oracle(torch_gpu, oracle_lib):
    """BASELINE config 2's second half: benchmark4, f64, eps_abs = eps_rel = 1e-6, MAX_IT lifted.  Iterations to the stop
    within 1 % of the oracle's, relaxed cost within 2.5e-4 of the relaxation optimum the reference's monolithic solve
    reports (classic_solver record, 32.629444)."""
    case, g = load_fixture("benchmark4")
    classic = case["golden_classic"]["cost"]
    d = _solver(g)
    res = d.solve(chunk=500, max_it=40000, eps_abs=1e-6, eps_rel=1e-6)
    ora = oracle_lib.Oracle(g, ipm_tol=IPM_TOL).run(max_it=40000, eps_abs=1e-6, eps_rel=1e-6, nthreads=16)
    assert res["status"] == "converged" and ora["status"] == 0 and res["inner_failures"] == 0
    assert abs(res["iterations"] - ora["iterations"]) <= 0.01 * ora["iterations"], (res["iterations"], ora["iterations"])
    assert abs(res["cost"] - classic) <= 2.5e-4 * classic
    assert abs(res["cost"] - ora["cost"]) <= 1e-5 * classic


def test_lattice_10k_f32_runs_to_the_same_stop(torch_gpu, oracle_lib):
    """BASELINE config 3 under the reference's stop rule: f32 state, f64 state and the oracle stop at the same iteration,
    with the same relaxed cost (f32: to storage precision)."""
    g = lattice_boxes(100, 100, seed=0)
    r64 = _solver(g, "f64").solve(chunk=100)
    r32 = _solver(g, "f32").solve(chunk=100)
    ora = oracle_lib.Oracle(g, ipm_tol=IPM_TOL).run(nthreads=16)
    assert r64["status"] == r32["status"] == "converged" and ora["status"] == 0
    assert r64["iterations"] == r32["iterations"] == ora["iterations"]
    assert r64["inner_failures"] == 0 and r32["inner_failures"] == 0
    assert abs(r64["cost"] - ora["cost"]) <= 1e-6 * ora["cost"]
    assert abs(r32["cost"] - r64["cost"]) <= 1e-4 * r64["cost"]


def test_inner_failure_keeps_previous_copy(torch_gpu):
    """an inner solve that hits its iteration limit keeps the vertex's previous copy columns and is counted
    (admm_solver_v3.py:524-538 intent): with ipm_max_iter = 2 every generic vertex fails, the state stays finite and the
    generic vertices' columns stay what they were (zero)."""
    torch = torch_gpu
    for program in ("wavefront", "workgroup"):
        g = lattice_boxes(8, 7, seed=3)
        d = _solver(g, program=program)
        d.reset(max_it=10, ipm_max_iter=2)
        for it in range(3):
            d.vertex_step()
            d.edge_step()
            d.control()
        cb = d.read_control()
        n_generic = g.num_vertices - 2 - int(((np.diff(g.inc_ptr) == 0)).sum())
        assert cb.inner_failures >= n_generic - 4 and cb.status == -1
        assert torch.isfinite(d.copy).all() and torch.isfinite(d.zedge).all() and torch.isfinite(d.mu).all()
        deg = np.diff(g.inc_ptr)
        din = np.array([int((g.inc_out[g.inc_ptr[v]:g.inc_ptr[v + 1]] == 0).sum()) for v in range(g.num_vertices)])
        generic = [v for v in range(g.num_vertices) if v not in (g.src, g.dst) and din[v] > 0 and deg[v] - din[v] > 0]
        # EVERY generic vertex is looked at: a vertex whose solve failed in all three steps still holds the zeros it started with
        kept = [v for v in generic
                if d.copy[:, int(g.inc_ptr[v]):int(g.inc_ptr[v + 1])].abs().max().item() == 0.0 and d.yv[v].item() == 0.0]
        assert len(kept) >= cb.inner_failures - 0 and len(kept) >= len(generic) - 4, (len(kept), len(generic), cb.inner_failures)


def _region_row():
    """three boxes in a row, the outer two are the terminals: s = [0,1] x [0,1], t = [2,3] x [0,1], between them [0.8,2.2] x [0,1].
    Shortest path: leave s and enter t at the faces x = 1 and x = 2 -> length 1 (with point terminals at the box centres: 2)"""
    from gcs_admm_amd.graph import graph_from_sets
    A = np.vstack([np.eye(2), -np.eye(2)])
    box = lambda x0, x1, y0, y1: (A, np.array([x1, y1, -x0, -y0], float))
    As, bs = {}, {}
    As['s'], bs['s'] = box(0, 1, 0, 1); As['t'], bs['t'] = box(2, 3, 0, 1); As[0], bs[0] = box(0.8, 2.2, 0, 1)
    return graph_from_sets(As, bs, 2)


def _region_scene(seed, half=0.35):
    """a 4 x 4 polygon scene (tools/scale_demo.py) whose terminals are boxes of half-width `half` about the points they were"""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from scale_demo import polygon_scene
    from gcs_admm_amd.graph import graph_from_sets
    As, bs = polygon_scene(4, seed=seed, m=3 + seed % 4)
    A = np.vstack([np.eye(2), -np.eye(2)])
    for key in ('s', 't'):
        pt = 0.5 * (bs[key][:2] - bs[key][2:])
        As[key], bs[key] = A, np.hstack([pt + half, -pt + half])
    return graph_from_sets(As, bs, 2)


@pytest.mark.parametrize("program,columns,dtype", [("workgroup", "incidence", "f64"), ("wavefront", "edge", "f64"), ("workgroup", "edge", "f32")])
def test_region_terminals_vertex_step_against_oracle(torch_gpu, oracle_lib, program, columns, dtype):
    """'s' / 't' that are regions (admm_solver_v3.py:415-464 with delta_sv / delta_tv; the reference's own cases make them points,
    utils.py:12-28): the terminal kernel (csrc/terminal_region.h) against the oracle's solve_terminal_region, vertex step by vertex step
    along an oracle run and from perturbed states, on the terminals' own columns to 1e-5 (f32 state: 2e-4) and everywhere to the fixture bound"""
    from oracle.oracle import Oracle
    torch = torch_gpu
    rng = np.random.default_rng(3)
    for g in (_region_row(), _region_scene(1), _region_scene(2, half=0.2)):
        o = Oracle(g, ipm_tol=IPM_TOL)
        d = _solver(g, dtype, program=program, columns=columns)
        d.reset()
        tcols = np.concatenate([np.arange(g.inc_ptr[v], g.inc_ptr[v + 1]) for v in (g.src, g.dst)])
        for it in range(10):
            if it >= 6:
                o.zedge += 0.03 * rng.normal(size=o.zedge.shape); o.mu += 0.02 * rng.normal(size=o.mu.shape)
            perm = torch.from_numpy(d.col_of).cuda()      # incidence column -> state column (edge-major handles)
            d.zedge.copy_(torch.from_numpy(o.zedge)); d.mu[:, perm] = torch.from_numpy(o.mu).to(d.mu.dtype).cuda()
            d.vertex_step()
            assert o.vertex_step(1.0, 1.0) == 0
            got = d.copy[:, perm].double().cpu().numpy()
            assert np.abs(got[:, tcols] - o.copy[:, tcols]).max() < (1e-5 if dtype == "f64" else 2e-4), (it, np.abs(got[:, tcols] - o.copy[:, tcols]).max())
            assert np.abs(got - o.copy).max() < 2e-3
            xv = d.xv.cpu().numpy(); zv = d.zv.cpu().numpy()
            for v in (g.src, g.dst):
                assert np.abs(xv[v] - o.xv[v]).max() < (1e-5 if dtype == "f64" else 2e-4) and np.array_equal(xv[v], zv[v]) and d.yv[v].item() == 1.0
            assert d.read_control().inner_failures == 0
            o.edge_step(1.0)
        d.close()


def _region_star(n, spokes, seed=0):
    """a source that is a big box in R^n overlapping `spokes` small boxes on a ring; the target a point in the last of them (edges given: no
    |V|^2 overlap tests).  The source's sub-problem has `spokes` live blocks: at n = 6 its work arrays exceed the LDS budget of the terminal
    kernel (HBM workspace, 256 threads), at n = 3 they fit (LDS, one wavefront for few rows)"""
    from gcs_admm_amd.graph import convert_pt_to_polytope, graph_from_sets
    rng = np.random.default_rng(seed)
    A = np.vstack([np.eye(n), -np.eye(n)])
    As, bs = {}, {}
    As['s'], bs['s'] = A, np.hstack([np.full(n, 1.0), np.full(n, 1.0)])                      # [-1, 1]^n
    edges = []
    for k in range(spokes):
        c = np.zeros(n); ang = 2 * np.pi * k / spokes
        c[0], c[1] = 1.1 * np.cos(ang), 1.1 * np.sin(ang)
        c[2:] = rng.uniform(-0.3, 0.3, n - 2)
        h = rng.uniform(0.3, 0.4, n)
        As[k], bs[k] = A, np.hstack([c + h, -c + h])
        edges += [('s', k), (k, 's')]
    last = spokes - 1
    ctr = 0.5 * (bs[last][:n] - bs[last][n:])
    As['t'], bs['t'] = convert_pt_to_polytope(ctr + 0.1)
    edges += [(last, 't'), ('t', last)]
    for k in range(spokes):                                                                   # neighbours on the ring overlap
        edges += [(k, (k + 1) % spokes), ((k + 1) % spokes, k)]
    keys = ['s', 't'] + list(range(spokes))
    return graph_from_sets({k: As[k] for k in keys}, {k: bs[k] for k in keys}, n, edges=edges)


@pytest.mark.parametrize("n,spokes", [(3, 8), (6, 30), (8, 6)])
def test_region_terminal_other_dimensions_and_degrees(torch_gpu, oracle_lib, n, spokes):
    """the terminal kernel outside n = 2: a source box with 8 live edges in R^3, 30 in R^6 (work arrays in the HBM workspace, 256 threads)
    and 6 in R^8, vertex steps along an oracle run against the oracle"""
    from oracle.oracle import Oracle
    torch = torch_gpu
    g = _region_star(n, spokes, seed=n)
    assert int(np.diff(g.inc_ptr)[g.src]) == 2 * spokes
    o = Oracle(g, ipm_tol=IPM_TOL)
    d = _solver(g)
    d.reset()
    tcols = np.arange(g.inc_ptr[g.src], g.inc_ptr[g.src + 1])
    for it in range(8):
        d.zedge.copy_(torch.from_numpy(o.zedge)); d.mu.copy_(torch.from_numpy(o.mu))
        d.vertex_step()
        assert o.vertex_step(1.0, 1.0) == 0 and d.read_control().inner_failures == 0
        got = d.copy.cpu().numpy()
        assert np.abs(got[:, tcols] - o.copy[:, tcols]).max() < 1e-5, (it, np.abs(got[:, tcols] - o.copy[:, tcols]).max())
        assert np.abs(got - o.copy).max() < 2e-3
        assert abs(o.copy[2 * n, tcols[spokes:]].sum() - 1.0) < 1e-9 and d.yv[g.src].item() == 1.0      # the unit of flow leaves the source
        o.edge_step(1.0)
    d.close()


def test_region_terminals_whole_run(torch_gpu, oracle_lib):
    """the loop with region terminals to the reference's stop rule: same stop iteration and residual trace as the oracle; on the row of
    three boxes the cost is the known answer (length 1 between the faces x = 1 and x = 2, + 2 edges x 1e-4), where point terminals at
    the box centres give 2"""
    from oracle.oracle import Oracle
    g = _region_row()
    o = Oracle(g, ipm_tol=IPM_TOL)
    ro = o.run(max_it=1000)
    d = _solver(g)
    res = d.solve()
    assert res["status"] == "converged" and res["iterations"] == ro["iterations"]
    assert abs(res["cost"] - 1.0002) < 1e-2 and abs(res["cost"] - ro["cost"]) < 1e-6
    for key in ("pri_res_seq", "dual_res_seq"):
        a, b = np.asarray(res[key]), np.asarray(ro[key])
        assert np.abs(a - b).max() <= 1e-5 * max(1.0, np.abs(b).max())
    g2 = _region_scene(1)
    o2 = Oracle(g2, ipm_tol=IPM_TOL)
    r2 = o2.run(max_it=1000)
    for program in ("workgroup", "wavefront"):
        d2 = _solver(g2, program=program)
        res2 = d2.solve()
        assert res2["iterations"] == r2["iterations"] and abs(res2["cost"] - r2["cost"]) < 1e-5 * max(1.0, abs(r2["cost"])), (program, res2["iterations"], r2["iterations"])
        d2.close()


def test_region_terminals_on_a_vertex_partition(torch_gpu):
    """a lattice whose terminals are boxes of half-width 0.3, as three row strips (three handles on one GPU, halo columns copied by hand,
    norms summed) against the single handle: the strip that owns a terminal launches the terminal kernel, the others do not; same stop
    decisions, sums to 1e-9, edge copies to 1e-8 (f64)"""
    torch = torch_gpu
    from gcs_admm_amd.partition import build_partition, strip_owner
    from gcs_admm_amd.solver import DeviceSolver
    g = lattice_boxes(12, 11, seed=4)
    for v in (g.src, g.dst):
        g.poly_b[g.poly_ptr[v]:g.poly_ptr[v + 1]] += 0.3
    world = 3
    single = _solver(g)
    single.reset(max_it=60)
    owner = strip_owner(g, world)
    parts = [build_partition(g, owner, r, world) for r in range(world)]
    devs = [DeviceSolver(p.graph, "f64", device=0, num_incidences=p.num_incidences, inc_counted=p.inc_counted,
                         edge_counted=p.edge_counted, nx_global=p.nx_global, nmu_global=p.nmu_global) for p in parts]
    for d in devs:
        d.reset(max_it=60)
    idx = {(r, o): (torch.as_tensor(parts[r].recv_idx[o], device="cuda"), torch.as_tensor(parts[o].send_idx[r], device="cuda"))
           for r in range(world) for o in parts[r].recv_idx}
    for it in range(20):
        single.vertex_step(); s_ref = single.edge_step().clone(); single.control()
        for d in devs:
            d.vertex_step()
        for (r, o), (rix, six) in idx.items():
            devs[r].copy.index_copy_(1, rix, devs[o].copy.index_select(1, six))
        tot = torch.zeros(5, dtype=torch.float64, device="cuda")
        for d in devs:
            tot += d.edge_step()
        assert torch.allclose(tot, s_ref, rtol=1e-9, atol=1e-12), (it, tot, s_ref)
        for d in devs:
            d.control(tot)
    cbs = [d.read_control() for d in devs] + [single.read_control()]
    assert len({cb.it for cb in cbs}) == 1 and len({cb.status for cb in cbs}) == 1 and all(cb.inner_failures == 0 for cb in cbs)
    assert single.yv[g.src].item() == 1.0 and single.yv[g.dst].item() == 1.0
    xs = single.xv[g.src].cpu().numpy()
    assert np.abs(xs[:2] - g.interior[g.src]).max() <= 0.3 + 2e-6 and np.abs(xs[:2] - g.interior[g.src]).max() > 1e-3      # inside the box (0.3 + the 1e-6 it had), off its centre
    full = single.zedge.cpu().numpy()
    for p, d in zip(parts, devs):
        assert np.allclose(d.zedge.cpu().numpy(), full[:, p.edge_global], rtol=0, atol=1e-8)


def test_region_terminal_refusals(torch_gpu):
    """a region terminal with no edge on its live side cannot carry the unit of flow (the reference's program is infeasible there):
    refused at create; the prox kernel (v1 x-update) keeps point terminals"""
    from gcs_admm_amd import solver
    from gcs_admm_amd.graph import convert_pt_to_polytope, graph_from_sets
    A = np.vstack([np.eye(2), -np.eye(2)])
    As, bs = {}, {}
    As['s'], bs['s'] = A, np.array([0.5, 0.5, 0.5, 0.5])
    As['t'], bs['t'] = convert_pt_to_polytope(np.array([3.0, 0.0]))
    As[0], bs[0] = A, np.array([4.0, 1.0, 1.0, 1.0])
    g = graph_from_sets(As, bs, 2, edges=[(0, 's'), (0, 't')])            # nothing leaves s
    with pytest.raises(solver.GcsAdmmError, match="live side"):
        _solver(g)


def test_partitioned_loop_behind_the_abi_single_rank(torch_gpu):
    """gcsadmm_run_partitioned with a REAL RCCL communicator of one rank (what a one-GPU box can exercise: run-time binding
    of librccl, communicator creation, the 6-double all-reduce on the caller's stream, the control step fed from it):
    the trace equals the single-handle loop's bit for bit."""
    from gcs_admm_amd.solver import DeviceSolver
    g = lattice_boxes(20, 18, seed=2)
    a = DeviceSolver(g, "f64", device=0)
    a.reset(max_it=40)
    a.enqueue(30)
    b = DeviceSolver(g, "f64", device=0)
    b.attach_comm(0, 1, b.unique_id(), {}, {})
    b.reset(max_it=40)
    b.enqueue_partitioned(30)
    ca, cb = a.read_control(), b.read_control()
    assert ca.it == cb.it == 31 and ca.status == cb.status and cb.inner_failures == 0
    assert np.array_equal(a.trace[:30].cpu().numpy(), b.trace[:30].cpu().numpy())
    assert np.array_equal(a.zedge.cpu().numpy(), b.zedge.cpu().numpy())


@pytest.mark.parametrize("columns,terminals", [("incidence", "points"), ("edge", "points"), ("edge", "boxes")])
def test_overlapped_partitioned_loop_equals_serial(torch_gpu, columns, terminals):
    """The overlapped schedule of gcsadmm_run_partitioned (boundary wavefronts first, the halo exchange on a second stream behind an
    event while the interior wavefronts are solved, the edge step waiting for both) against the serial one and against the plain
    loop: the same bits.  One rank (a real one-rank RCCL communicator), so the split is forced: the first quarter of the wavefronts
    plays the boundary -- the two launches, their slowest-first re-ordering inside each part (every eighth step), the second stream
    and the two events are what is exercised; the exchange itself moves nothing here.  "boxes": the terminals are regions, their kernel
    forks from and joins whichever stream carries the boundary part."""
    from gcs_admm_amd.solver import DeviceSolver
    g = lattice_boxes(40, 40, seed=3)
    if terminals == "boxes":
        for v in (g.src, g.dst):
            g.poly_b[g.poly_ptr[v]:g.poly_ptr[v + 1]] += 0.3
    steps = 40
    out = {}
    for mode in ("plain", 2, 1):
        d = DeviceSolver(g, "f32", device=0, program="wavefront", columns=columns)
        if mode != "plain":
            d.attach_comm(0, 1, d.unique_id(), {}, {})
            nb = d.set_overlap(mode)
            assert (nb > 0) == (mode == 1) and nb < d.query()["num_waves"]
        d.reset(max_it=steps + 10)
        (d.enqueue if mode == "plain" else d.enqueue_partitioned)(steps)
        cb = d.read_control()
        assert cb.it == steps + 1 and cb.status == -1 and cb.inner_failures == 0
        out[mode] = [t.cpu().numpy().copy() for t in (d.trace[:steps], d.copy, d.mu, d.zedge, d.xv, d.zv, d.yv)]
        d.close()
    for a, b, c in zip(out["plain"], out[2], out[1]):
        assert np.array_equal(a, b) and np.array_equal(b, c)


@pytest.mark.parametrize("columns", ["incidence", "edge"])
def test_halo_pack_unpack_two_handles(torch_gpu, columns):
    """the pack / unpack kernels and halo lists of the C ABI with the transfer done by hand: two partitions of one lattice
    as two handles on this GPU; packed send buffers are copied device-to-device into the peer's receive buffer (what
    ncclSend / ncclRecv do across GPUs), then unpacked into the ghost columns.  Against the single handle.  Both column
    numberings of the state (edge-major: the ghost of a cut edge is the column of its remote side)."""
    torch = torch_gpu
    import ctypes as C
    import os
    from gcs_admm_amd.partition import build_partition, strip_owner
    from gcs_admm_amd.solver import DeviceSolver
    hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    g = lattice_boxes(12, 16, seed=5)
    single = DeviceSolver(g, "f32", device=0)
    single.reset(max_it=60)
    owner = strip_owner(g, 2)
    parts = [build_partition(g, owner, r, 2) for r in range(2)]
    devs = [DeviceSolver(p.graph, "f32", device=0, num_incidences=p.num_incidences, inc_counted=p.inc_counted,
                         edge_counted=p.edge_counted, nx_global=p.nx_global, nmu_global=p.nmu_global, columns=columns) for p in parts]
    for r, (p, d) in enumerate(zip(parts, devs)):
        d.attach_comm(r, 2, None, p.send_idx, p.recv_idx)
        d.reset(max_it=60)
    bufs = [d.halo_buffers() for d in devs]
    assert bufs[0][2] == bufs[1][2] == g.c * len(parts[0].send_idx[1])
    for it in range(25):
        single.vertex_step(); s_ref = single.edge_step().clone(); single.control()
        for d in devs:
            d.vertex_step(); d.halo_pack()
        torch.cuda.synchronize()
        for r in range(2):
            assert hip.hipMemcpy(C.c_void_p(bufs[r][1]), C.c_void_p(bufs[1 - r][0]), C.c_size_t(4 * bufs[r][2]), 3) == 0   # DeviceToDevice
        tot = torch.zeros(5, dtype=torch.float64, device="cuda")
        for d in devs:
            d.halo_unpack()
            tot += d.edge_step()
        assert torch.allclose(tot, s_ref, rtol=1e-3, atol=1e-9)
        for d in devs:
            d.control(tot)
    full = single.zedge.cpu().numpy()
    for p, d in zip(parts, devs):
        assert np.allclose(d.zedge.cpu().numpy(), full[:, p.edge_global], rtol=0, atol=5e-4)
    cbs = [d.read_control() for d in devs] + [single.read_control()]
    assert len({cb.it for cb in cbs}) == 1 and len({cb.status for cb in cbs}) == 1


@pytest.mark.parametrize("case", ["benchmark4", "lattice_wave", "lattice_wg", "lattice_r6", "star"])
def test_edge_major_columns_equal_incidence_major(torch_gpu, case):
    """include/gcsadmm.h edge_major_columns: the numbering of the state columns changes where a value lives, never the value --
    the same runs in both numberings are BITWISE equal (vertex programs compute the same solves, the edge kernel handles edge e
    in thread e either way, the norms are summed in the same order), for both vertex programs, the closed-form vertices, n = 6."""
    torch = torch_gpu
    from conftest import star_case
    from gcs_admm_amd.cases import load_fixture
    from gcs_admm_amd.graph import graph_from_sets
    from gcs_admm_amd.solver import DeviceSolver
    kw, dt, its = {}, "f64", 30
    if case == "benchmark4":
        g = load_fixture("benchmark4")[1]
    elif case == "lattice_wave":
        g, kw, dt = lattice_boxes(40, 30, seed=3), dict(program="wavefront"), "f32"
    elif case == "lattice_wg":
        g, kw = lattice_boxes(12, 9, seed=3), dict(program="workgroup")
    elif case == "lattice_r6":
        g, dt, its = lattice_boxes(7, 6, n=6, seed=2), "f32", 8
    else:
        As, bs, n = star_case(24)
        g, its = graph_from_sets(As, bs, n), 10
    a = DeviceSolver(g, dt, device=0, **kw)
    b = DeviceSolver(g, dt, device=0, columns="edge", **kw)
    assert not a.edge_major and b.edge_major and sorted(b.col_of.tolist()) == list(range(2 * g.num_edges))
    for d in (a, b):
        d.reset(max_it=its + 5)
        d.enqueue(its)
    torch.cuda.synchronize()
    perm = torch.from_numpy(b.col_of).to("cuda")
    assert torch.equal(a.copy, b.copy[:, perm]) and torch.equal(a.mu, b.mu[:, perm]) and torch.equal(a.zedge, b.zedge)
    assert torch.equal(a.xv, b.xv) and torch.equal(a.zv, b.zv) and torch.equal(a.yv, b.yv)
    assert torch.equal(a.trace[:its], b.trace[:its]) and a.read_control().it == b.read_control().it == its + 1
    assert float(a.copy.abs().max()) > 0.0


@pytest.mark.parametrize("n", [3, 6])
def test_workgroup_box_instantiation_against_generic(torch_gpu, n):
    """the BOX instantiation of the workgroup program (chosen at create when every vertex is a canonical axis-aligned box, n > 2)
    against the generic one (forced by wave_generic_rows = 1) on the same lattice.  Cold solves: the same arithmetic up to rounding,
    iterates equal to 1e-9 over a run.  Warm solves: each vertex's far-warm threshold learns from its own iteration counts
    (warm_start.h ws_learn), and a count that differs by one at the stop test's edge can flip a later warm / cold decision; the
    minimiser does not depend on the start, so the iterates then agree to the inner tolerance instead."""
    torch = torch_gpu
    from gcs_admm_amd.solver import DeviceSolver
    g = lattice_boxes(9, 8, n=n, seed=4)
    a = DeviceSolver(g, "f64", device=0)
    b = DeviceSolver(g, "f64", device=0, wave_generic_rows=1)
    for cold, atol, rtol in ((True, 1e-9, 1e-9), (False, 2e-6, 1e-5)):
        for d in (a, b):
            d.reset(max_it=40, cold_start=cold)
            d.enqueue(25)
        torch.cuda.synchronize()
        ca, cb = a.read_control(), b.read_control()
        assert ca.it == cb.it == 26 and ca.inner_failures == cb.inner_failures == 0
        assert float((a.copy - b.copy).abs().max()) <= atol and float((a.mu - b.mu).abs().max()) <= atol
        assert torch.allclose(a.trace[:25, 1:3], b.trace[:25, 1:3], rtol=rtol, atol=1e-3 * atol)
    assert float(a.copy.abs().max()) > 0.0


@pytest.mark.parametrize("case", ["benchmark4", "lattice_n3", "lattice_n6", "lattice_n4", "lattice_n8"])
def test_workgroup_512_threads_against_256(torch_gpu, case):
    """launches of at most one workgroup per CU run the 512-thread build of the workgroup program (second object of vertex_wg.hip,
    chosen at create); vertex_program = 3 keeps 256.  Same tasks on more threads: cold runs agree to rounding (the reductions sum in
    another order), warm runs to the inner tolerance (test_workgroup_box_instantiation_against_generic has the reason)."""
    torch = torch_gpu
    from gcs_admm_amd.cases import load_fixture
    from gcs_admm_amd.solver import DeviceSolver
    g, dt = {"benchmark4": (lambda: load_fixture("benchmark4")[1], "f64"), "lattice_n3": (lambda: lattice_boxes(9, 8, n=3, seed=4), "f64"),
             "lattice_n6": (lambda: lattice_boxes(7, 6, n=6, seed=2), "f32"),
             "lattice_n4": (lambda: lattice_boxes(8, 7, n=4, seed=5), "f64"),     # (n = 1, 4, 5, 7, 8: the second object of vertex_wg_dims.hip)
             "lattice_n8": (lambda: lattice_boxes(6, 5, n=8, seed=6), "f64")}[case]
    g = g()
    a = DeviceSolver(g, dt, device=0, program="workgroup")
    b = DeviceSolver(g, dt, device=0, program="workgroup256")
    scale = 1.0 if dt == "f64" else 1e3           # (f32 state: the copies are rounded to 1e-7 relative)
    for cold, atol in ((True, 1e-9 * scale * 1e1), (False, 2e-6 * (1.0 if dt == "f64" else 5.0))):
        for d in (a, b):
            d.reset(max_it=40, cold_start=cold)
            d.enqueue(25)
        torch.cuda.synchronize()
        ca, cb = a.read_control(), b.read_control()
        assert ca.it == cb.it == 26 and ca.inner_failures == cb.inner_failures == 0
        assert float((a.copy.double() - b.copy.double()).abs().max()) <= atol, (cold, float((a.copy.double() - b.copy.double()).abs().max()))
        assert float((a.mu.double() - b.mu.double()).abs().max()) <= atol
    assert float(a.copy.abs().max()) > 0.0


def test_unit_iterations_diagnostics(torch_gpu):
    """gcsadmm_unit_iterations: Newton iterations of the last vertex step per dispatch unit, for handles that keep them (>= 512 units);
    they add up to what the control block counts (wavefront program: a wavefront reports its slowest vertex, so the sum bounds it)"""
    torch = torch_gpu
    from gcs_admm_amd.solver import DeviceSolver
    g = lattice_boxes(60, 60, seed=1)                       # 3 602 vertices: wavefront program, ~600 wavefronts
    d = DeviceSolver(g, "f32", device=0, columns="edge")
    d.reset(max_it=50)
    d.enqueue(12); torch.cuda.synchronize()
    u = d.unit_iterations()
    q = d.query()
    assert len(u) == q["num_waves"] >= 512 and (u >= 2).all() and (u <= 60).all()
    cb = d.read_control()
    n_generic = g.num_vertices - q["num_special"]
    assert cb.inner_iters <= int(u.sum()) * 7 and cb.inner_iters >= int(u.max()) and cb.inner_iters / n_generic <= u.max()
    small = DeviceSolver(lattice_boxes(8, 8, seed=1), "f64", device=0)
    small.reset(max_it=10); small.enqueue(3); torch.cuda.synchronize()
    assert len(small.unit_iterations()) == 0              # too few units to keep (no slowest-first dispatch)
