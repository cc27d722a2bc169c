"""Graph construction at scale: the oracle's LP decisions against the committed
edge lists (CPU), the sort-and-sweep broad phase against brute force (CPU), and the device LPs
(csrc/polytope_lp.hip through the C ABI) against the oracle and the fixtures (GPU)."""
from gcs_admm_amd import IPM_TOL
import os
import sys

import numpy as np
import pytest

from conftest import BENCHMARKS, SMALL
from gcs_admm_amd.cases import fixture_sets, load_fixture

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import polytope_oracle as PO   # noqa: E402


def random_polytope(rng, n, m, centre, scale):
    """m random half-spaces at distance ~scale around `centre` plus a bounding box (always bounded)."""
    A = rng.normal(size=(m, n)); A /= np.linalg.norm(A, axis=1)[:, None]
    b = A @ centre + scale * rng.uniform(0.3, 1.0, size=m)
    A = np.vstack([A, np.eye(n), -np.eye(n)])
    b = np.hstack([b, centre + 2 * scale, -(centre - 2 * scale)])
    return A, b


# ------------------------------------------------------------------------------------------- CPU
@pytest.mark.parametrize("name", ["test1", "test2", "test3", "test_autogen1", "benchmark1", "benchmark2"])
def test_oracle_edges_match_fixture(name):
    """LP feasibility per ordered pair in double-loop order (utils.py:31-82) gives the committed edge list"""
    case, _ = load_fixture(name)
    As, bs, n, _, _ = fixture_sets(name)
    assert [list(e) for e in PO.edges(As, bs)] == case["edges"]


def test_candidate_pairs_against_brute_force():
    from gcs_admm_amd.scene import candidate_pairs
    rng = np.random.default_rng(3)
    for n in (1, 2, 4):
        c = rng.uniform(0, 8, (250, n)); w = rng.uniform(0.05, 0.9, (250, n))
        lo, hi = c - w, c + w
        a, b = candidate_pairs(lo, hi, 0.0)
        ref = {(i, j) for i in range(250) for j in range(i + 1, 250) if np.all(lo[i] <= hi[j]) and np.all(lo[j] <= hi[i])}
        assert set(zip(a.tolist(), b.tolist())) == ref and len(a) == len(ref)
    a, b = candidate_pairs(np.zeros((1, 2)), np.ones((1, 2)))
    assert len(a) == 0


# ------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("name", SMALL + BENCHMARKS)
def test_device_graph_matches_fixture(name):
    """edges (order included) and CSR of the device-built graph equal the committed fixture; the interior
    points are the Chebyshev centres (same radius as the oracle's LP)"""
    from gcs_admm_amd.scene import PolytopeScene, build_graph_device, graph_from_sets_device
    case, gref = load_fixture(name)
    As, bs, n, _, _ = fixture_sets(name)
    V, E, I_in, I_out, cen = build_graph_device(As, bs)
    assert V == case["keys"] and [list(e) for e in E] == case["edges"]
    g = graph_from_sets_device(As, bs, n)
    for f in ("inc_ptr", "inc_edge", "inc_out", "edge_inc_tail", "edge_inc_head", "poly_ptr"):
        assert np.array_equal(getattr(g, f), getattr(gref, f)), f
    scene = PolytopeScene([(As[k], bs[k]) for k in V])
    cen, rad, st = scene.centers()
    lo, hi, _ = scene.bounds(cen)
    for i, k in enumerate(V):
        xo, ro = PO.chebyshev(As[k], bs[k])
        assert abs(rad[i] - ro) <= 1e-8 * max(1.0, abs(ro)), (k, rad[i], ro)
        slack = (bs[k] - As[k] @ cen[i]) / np.linalg.norm(As[k], axis=1)
        assert slack.min() >= rad[i] - 1e-7                      # the returned point has that ball around it
        lo_o, hi_o = PO.bounding_box(As[k], bs[k])
        assert np.allclose(lo[i], lo_o, atol=1e-6) and np.allclose(hi[i], hi_o, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [2, 3, 6, 8])
def test_device_overlaps_random_polytopes(n):
    """pairwise decisions on random polytopes equal the oracle's wherever the decision has a margin"""
    from gcs_admm_amd.scene import PolytopeScene
    rng = np.random.default_rng(10 + n)
    P = 60
    polys = [random_polytope(rng, n, 4 + n, rng.uniform(0, 3.0, n), rng.uniform(0.4, 1.2)) for _ in range(P)]
    scene = PolytopeScene(polys)
    cen, rad, _ = scene.centers()
    assert np.all(rad > 0)
    pa, pb = np.triu_indices(P, 1)
    flags, st = scene.overlaps(pa, pb, 1e-9, cen)
    checked = 0
    for t, (i, j) in enumerate(zip(pa, pb)):
        r = PO.overlap_radius(polys[i][0], polys[i][1], polys[j][0], polys[j][1])
        if abs(r) < 1e-6:
            continue
        checked += 1
        assert bool(flags[t]) == (r > 0), (i, j, r, st[t])
        assert bool(flags[t]) == PO.overlap(polys[i][0], polys[i][1], polys[j][0], polys[j][1])
    assert checked > 0.9 * len(pa) and 0 < flags.sum() < len(pa)


@pytest.mark.gpu
def test_device_overlaps_degenerate_cases():
    """touching boxes overlap (closed sets, as for the reference's feasibility solve), a gap does not; the
    reference's point vertices (boxes of half-width 1e-6, utils.py:12-28) are found inside their region"""
    from gcs_admm_amd.graph import convert_pt_to_polytope
    from gcs_admm_amd.scene import PolytopeScene
    box = lambda lo, hi: (np.vstack([np.eye(2), -np.eye(2)]), np.hstack([hi, -np.asarray(lo, float)]))
    polys = [box([0, 0], [1, 1]), box([1, 0], [2, 1]), box([1 + 1e-5, 0], [2, 1]), box([1, 1], [2, 2]),
             convert_pt_to_polytope(np.array([0.5, 0.5])), convert_pt_to_polytope(np.array([1.5, 3.0])),
             box([-50, -50], [50, 50])]
    scene = PolytopeScene(polys)
    pa = np.array([0, 0, 0, 0, 0, 4, 5, 1], np.int32); pb = np.array([1, 2, 3, 4, 5, 6, 6, 2], np.int32)
    flags, _ = scene.overlaps(pa, pb)
    assert flags.tolist() == [1, 0, 1, 1, 0, 1, 1, 1]
    scene._centers = None                                         # start points at the origin instead of the centres
    flags0, _ = scene.overlaps(pa, pb, 1e-9, None)
    assert flags0.tolist() == flags.tolist()


@pytest.mark.gpu
def test_device_graph_lattice_at_scale():
    """10k-box lattice: the device pipeline (centres, boxes, sweep, LPs) finds exactly the edges the exact
    interval test finds (graph.lattice_boxes)"""
    from gcs_admm_amd.graph import lattice_boxes
    from gcs_admm_amd.scene import graph_from_sets_device
    gref = lattice_boxes(100, 100, seed=0)
    As = {k: gref.poly_A[gref.poly_ptr[i]:gref.poly_ptr[i + 1]] for i, k in enumerate(gref.keys)}
    bs = {k: gref.poly_b[gref.poly_ptr[i]:gref.poly_ptr[i + 1]] for i, k in enumerate(gref.keys)}
    g = graph_from_sets_device(As, bs, 2)
    assert g.num_edges == gref.num_edges
    assert np.array_equal(g.edge_tail, gref.edge_tail) and np.array_equal(g.edge_head, gref.edge_head)
    # the interior points are Chebyshev centres: not unique for a rectangle (any point of a segment), so compare
    # the radius of the ball around them with the one around the box midpoints of the generator
    def ball(gg):
        r = np.empty(gg.num_vertices)
        for v in range(gg.num_vertices):
            A = gg.poly_A[gg.poly_ptr[v]:gg.poly_ptr[v + 1]]; b = gg.poly_b[gg.poly_ptr[v]:gg.poly_ptr[v + 1]]
            r[v] = ((b - A @ gg.interior[v]) / np.linalg.norm(A, axis=1)).min()
        return r
    assert np.allclose(ball(g), ball(gref), atol=1e-8)


@pytest.mark.gpu
def test_device_lp_errors():
    from gcs_admm_amd.scene import PolytopeScene
    from gcs_admm_amd.solver import GcsAdmmError
    A = np.vstack([np.eye(9), -np.eye(9)]); b = np.ones(18)
    with pytest.raises(GcsAdmmError, match="n = 1..8"):
        PolytopeScene([(A, b)]).centers()
    A2 = np.vstack([np.eye(2), -np.eye(2)]); b2 = np.ones(4)
    with pytest.raises(GcsAdmmError, match="out of range"):
        PolytopeScene([(A2, b2)]).overlaps([0], [3])


# ------------------------------------------------------------------------------------------- CPU: the LP core
@pytest.fixture(scope="module")
def lp_emu():
    import ctypes as C
    import subprocess
    src = os.path.join(ROOT, "tests", "hostemu", "lp_emu.cpp")
    out = os.path.join(ROOT, "tests", "hostemu", "liblpemu.so")
    hdr = os.path.join(ROOT, "gcs_admm_amd", "csrc", "polytope_lp_core.h")
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-I" + os.path.dirname(hdr), src, "-o", out])
    return C.CDLL(out)


def _emu_ball(lib, polys, p, q, x0=None, early=0, tol=1e-9):
    import ctypes as C
    n = polys[0][0].shape[1]
    ptr = np.zeros(len(polys) + 1, np.int32); ptr[1:] = np.cumsum([len(b) for _, b in polys])
    A = np.ascontiguousarray(np.vstack([a for a, _ in polys])); b = np.ascontiguousarray(np.hstack([bb for _, bb in polys]))
    nrm = np.ascontiguousarray(np.linalg.norm(A, axis=1))
    w = np.zeros(n + 1); it = C.c_int(0)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    st = lib.lp_emu_ball(n, len(polys), vp(ptr), vp(A), vp(b), vp(nrm), p, q, vp(np.ascontiguousarray(x0)) if x0 is not None else None,
                         early, C.c_double(tol), vp(w), C.byref(it))
    return st, w, it.value


@pytest.mark.parametrize("n", [2, 3, 6, 7])
def test_lp_core_against_oracle_on_the_host(lp_emu, n):
    """the per-lane interior-point LP (host build of polytope_lp_core.h): Chebyshev radii and pairwise decisions
    equal the oracle's HiGHS LPs, also for regions hundreds of units from the origin and for the reference's
    1e-6 point boxes (utils.py:12-28), in a handful of Newton steps"""
    from gcs_admm_amd.graph import convert_pt_to_polytope
    rng = np.random.default_rng(n)
    polys = [random_polytope(rng, n, 4 + n, rng.uniform(-1, 1, n) * s, rng.uniform(0.4, 1.2)) for s in (1, 1, 30, 30, 300, 300)]
    polys += [random_polytope(rng, n, 3 + n, polys[k][1][-2 * n:-n] - 2 * 0.8 + rng.uniform(-0.5, 0.5, n), 0.8) for k in (0, 2, 4)]
    polys += [convert_pt_to_polytope(rng.uniform(-1, 1, n) * 300)]
    for p, (A, b) in enumerate(polys):
        st, w, it = _emu_ball(lp_emu, polys, p, -1)
        xo, ro = PO.chebyshev(A, b)
        assert st == 0 and it <= 25, (p, st, it)
        assert abs(w[n] - ro) <= 1e-9 * max(1.0, abs(ro)) + 1e-11, (p, w[n], ro)
        assert ((b - A @ w[:n]) / np.linalg.norm(A, axis=1)).min() >= w[n] - 1e-9
    cen = [PO.chebyshev(A, b)[0] for A, b in polys]
    checked = 0
    for p in range(len(polys)):
        for q in range(p + 1, len(polys)):
            r = PO.overlap_radius(polys[p][0], polys[p][1], polys[q][0], polys[q][1])
            if abs(r) < 1e-7:
                continue
            st, w, it = _emu_ball(lp_emu, polys, p, q, x0=cen[p], early=1)
            decided = (st == 1) or (st != 2 and w[n] >= -1e-9)
            assert decided == (r > 0), (p, q, r, st, w[n])
            checked += 1
    assert checked >= 20


@pytest.mark.gpu
def test_device_scene_far_from_origin_end_to_end(oracle_lib):
    """hexagons 300 units from the origin: device graph = oracle graph (every ordered pair by LP), all centre LPs
    converge, and the ADMM loop on that graph follows the CPU oracle"""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from scale_demo import polygon_scene
    from gcs_admm_amd.scene import PolytopeScene, graph_from_sets_device
    from gcs_admm_amd.solver import DeviceSolver
    As, bs = polygon_scene(8, seed=3)
    shift = np.array([300.0, -250.0])
    bs = {k: bs[k] + As[k] @ shift for k in As}
    g = graph_from_sets_device(As, bs, 2)
    assert [list(e) for e in PO.edges(As, bs)] == [[g.keys[t], g.keys[h]] for t, h in zip(g.edge_tail, g.edge_head)]
    _, rad, st = PolytopeScene([(As[k], bs[k]) for k in As]).centers()
    assert np.all(st == 0) and np.all(rad > 0)
    o = oracle_lib.Oracle(g, ipm_tol=IPM_TOL)
    d = DeviceSolver(g, "f64", device=0)
    d.reset()
    for it in range(10):
        d.zedge.copy_(torch.from_numpy(o.zedge)); d.mu.copy_(torch.from_numpy(o.mu))
        d.vertex_step()
        assert o.vertex_step(1.0, 1.0) == 0
        assert np.abs(d.copy.cpu().numpy() - o.copy).max() <= 1e-5
        o.edge_step(1.0)
