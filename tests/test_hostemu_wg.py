"""The workgroup-cooperative vertex program (gcs_admm_amd/csrc/vertex_wg.h) compiled for the HOST (tests/hostemu/wg_emu.cpp):
every parallel region runs its tasks serially, which equals the GPU execution as long as the tasks of a region are
independent.  Checked here without a GPU: (1) the algorithm against the CPU oracle (same interior-point method), for n = 2, 3, 6,
general polygons and boxes; (2) task-order independence: ascending and descending task order agree to round-off (the only
order-dependent operations are the floating-point sums of the workgroup reductions); (3) no read of unwritten LDS
(the emulation poisons it with NaN).  The GPU parity tests proper are test_gpu_parity.py / test_gpu_configs.py."""
from gcs_admm_amd import IPM_TOL
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from gcs_admm_amd.cases import load_fixture
from gcs_admm_amd.graph import lattice_boxes

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(ROOT, "gcs_admm_amd", "csrc")


@pytest.fixture(scope="module")
def libs():
    src = os.path.join(HERE, "hostemu", "wg_emu.cpp")
    deps = [src] + [os.path.join(CSRC, f) for f in ("vertex_wg.h", "gcs_math.h", "warm_start.h")]      # (rebuilt when the program changes)
    out = []
    for name, flags in (("libwgemu.so", []), ("libwgemu_rev.so", ["-DGCS_WG_REVERSE"])):
        so = os.path.join(HERE, "hostemu", name)
        if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(d) for d in deps):
            subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-I" + CSRC] + flags + [src, "-o", so])
        out.append(C.CDLL(so))
    return out


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class WarmRecords:
    """the warm-start workspace a handle owns on the device (warm_start.h), for the host build"""

    def __init__(self, lib, g):
        lib.wg_emu_warm_doubles.restype = C.c_longlong
        deg = np.diff(g.inc_ptr); m = np.diff(g.poly_ptr)
        size = [lib.wg_emu_warm_doubles(g.n, int(m[v]), int(deg[v])) for v in range(g.num_vertices)]
        self.ptr = np.concatenate([[0], np.cumsum(size)]).astype(np.int64)
        self.buf = np.zeros(int(self.ptr[-1]))


def wg_step(lib, fn, g, zedge, mu, rho=1.0, mu_scale=1.0, max_iter=60, warm=None):
    setter = getattr(lib, "wg_emu_set_warm_rev" if fn.endswith("_rev") else "wg_emu_set_warm")
    setter(_p(warm.buf) if warm else None, _p(warm.ptr) if warm else None)
    c, NI, V = g.c, 2 * g.num_edges, g.num_vertices
    copy = np.zeros((c, NI)); xv = np.zeros((V, 2 * g.n)); zv = np.zeros_like(xv); yv = np.zeros(V)
    cnt = np.zeros(2, dtype=np.int32); gen = np.zeros(V, dtype=np.int32)
    st = np.zeros(V, dtype=np.int32); it = np.zeros(V, dtype=np.int32)
    r = getattr(lib, fn)(g.n, V, g.num_edges, NI, _p(g.inc_ptr), _p(g.inc_edge), _p(g.inc_out), _p(g.poly_ptr), _p(g.poly_A),
                         _p(g.poly_b), _p(g.interior), g.src, g.dst, _p(zedge), _p(mu), C.c_double(rho), C.c_double(mu_scale),
                         C.c_double(1e-4), C.c_double(IPM_TOL), max_iter, _p(copy), _p(xv), _p(zv), _p(yv), _p(cnt), _p(gen),
                         _p(st), _p(it))
    assert r == 0
    return copy, xv, zv, yv, cnt, gen == 1, st, it


CASES = [("benchmark1", None, 12, 2e-3), ("benchmark4", None, 12, 2e-3), ("test_autogen2", None, 8, 2e-3),
         ("lattice n=2", (5, 4, 2), 8, 1e-6), ("lattice n=3", (4, 3, 3), 6, 1e-6), ("lattice n=6", (4, 3, 6), 5, 1e-6),
         # the program is dimension-generic (admm_solver_v3.py:363-377 takes any n): the instantiations outside BASELINE's configs
         ("intervals n=1", "chain", 8, 1e-6), ("lattice n=4", (4, 3, 4), 5, 1e-6), ("lattice n=5", (4, 3, 5), 5, 1e-6),
         ("lattice n=7", (4, 3, 7), 4, 1e-6), ("lattice n=8", (4, 3, 8), 4, 1e-6)]


@pytest.mark.parametrize("name,lat,steps,tol", CASES)
def test_workgroup_program_matches_oracle_and_is_order_independent(libs, oracle_lib, name, lat, steps, tol):
    fwd, rev = libs
    if lat == "chain":
        from conftest import interval_chain
        from gcs_admm_amd.graph import graph_from_sets
        g = graph_from_sets(*interval_chain(6))
    else:
        g = lattice_boxes(lat[0], lat[1], n=lat[2], seed=1) if lat else load_fixture(name)[1]
    o = oracle_lib.Oracle(g, ipm_tol=IPM_TOL)
    wa, wb = WarmRecords(fwd, g), WarmRecords(fwd, g)       # oracle and both builds restart from their own records (same rule)
    diffs = []
    for it in range(steps):
        z0, m0 = o.zedge.copy(), o.mu.copy()
        a = wg_step(fwd, "wg_emu_vertex_step", g, z0, m0, warm=wa)
        b = wg_step(rev, "wg_emu_vertex_step_rev", g, z0, m0, warm=wb)
        assert o.vertex_step(1.0, 1.0) == 0
        gen = a[5]
        assert a[4][0] == 0 and (a[6][gen] == 0).all()
        mask = np.zeros(2 * g.num_edges, bool)
        for v in np.nonzero(gen)[0]:
            mask[g.inc_ptr[v]:g.inc_ptr[v + 1]] = True
        assert np.isfinite(a[0][:, mask]).all()            # a read of unwritten (NaN-poisoned) LDS would surface here
        diffs.append(np.abs(a[0][:, mask] - o.copy[:, mask]).max())
        assert np.abs(a[3][gen] - o.yv[gen]).max() <= 5e-4
        # same tasks in the opposite order: identical up to the summation order of the reductions
        assert np.abs(a[0] - b[0]).max() <= max(10 * diffs[-1], 1e-9)
        o.edge_step(1.0)
    diffs = np.array(diffs)
    assert diffs.max() <= tol and np.median(diffs) <= max(1e-5, tol * 1e-2)
    # the per-vertex far-warm threshold and the cold iteration count it is judged against (record header [2], [3]: warm_start.h
    # ws_learn, oracle WS_*) follow the same rule in both: equal wherever the iteration counts were (a count that differs by one at
    # the tolerance's edge may move a threshold, never a result)
    gv = np.nonzero(gen)[0]
    hd = np.array([[wa.buf[wa.ptr[v] + k] for k in (2, 3)] for v in gv])
    ho = np.array([[o._warm[o._warm_ptr[v] + k] for k in (2, 3)] for v in gv])
    assert (hd[:, 1] > 0).all() and ((hd[:, 0] == 0) | ((hd[:, 0] >= 0.1) & (hd[:, 0] <= 10.0))).all()
    assert (np.abs(hd - ho).max(axis=1) == 0).mean() >= 0.8


@pytest.mark.parametrize("lat", [(5, 4, 2), (4, 3, 3), (4, 3, 6)])
def test_box_instantiation_equals_generic(libs, oracle_lib, lat):
    """the BOX instantiation of the workgroup program (canonical axis-aligned boxes: one-term facet rows, diagonal K_h and X_e)
    against the generic one on the same lattice: the terms it drops are exact zeros, so the two agree to rounding; task-order
    independence of the BOX regions (ascending / descending)"""
    fwd, rev = libs
    g = lattice_boxes(lat[0], lat[1], n=lat[2], seed=1)
    o = oracle_lib.Oracle(g, ipm_tol=IPM_TOL)
    wa, wb, wc = (WarmRecords(fwd, g) for _ in range(3))      # every build restarts from its own records (warm_start.h)
    for it in range(5):
        z0, m0 = o.zedge.copy(), o.mu.copy()
        a = wg_step(fwd, "wg_emu_vertex_step", g, z0, m0, warm=wa)
        fwd.wg_emu_set_box(1); rev.wg_emu_set_box_rev(1)
        try:
            b = wg_step(fwd, "wg_emu_vertex_step", g, z0, m0, warm=wb)
            c = wg_step(rev, "wg_emu_vertex_step_rev", g, z0, m0, warm=wc)
        finally:
            fwd.wg_emu_set_box(0); rev.wg_emu_set_box_rev(0)
        gen = a[5]
        assert b[4][0] == 0 and (b[6][gen] == 0).all() and np.array_equal(a[7], b[7])       # same Newton iteration counts
        mask = np.zeros(2 * g.num_edges, bool)
        for v in np.nonzero(gen)[0]:
            mask[g.inc_ptr[v]:g.inc_ptr[v + 1]] = True
        assert np.isfinite(b[0][:, mask]).all()
        assert np.abs(a[0][:, mask] - b[0][:, mask]).max() <= 1e-9 and np.abs(a[3][gen] - b[3][gen]).max() <= 1e-9
        assert np.abs(b[0][:, mask] - c[0][:, mask]).max() <= 1e-9
        assert o.vertex_step(1.0, 1.0) == 0
        assert np.abs(b[0][:, mask] - o.copy[:, mask]).max() <= 1e-6
        o.edge_step(1.0)


def test_inner_failure_keeps_previous_outputs(libs):
    """an inner solve that runs out of iterations writes nothing (admm_solver_v3.py:524-538 intent) and is counted"""
    fwd, _ = libs
    g = lattice_boxes(4, 3, seed=2)
    z = np.zeros((g.c, g.num_edges)); mu = np.zeros((g.c, 2 * g.num_edges))
    w = WarmRecords(fwd, g)
    copy, xv, zv, yv, cnt, gen, st, it = wg_step(fwd, "wg_emu_vertex_step", g, z, mu, max_iter=2, warm=w)
    assert cnt[0] == gen.sum() and (st[gen] == -1).all() and (it[gen] == 2).all()
    assert not copy.any() and not yv.any()
    assert not w.buf[w.ptr[:-1]].any()        # a failed solve leaves no record to restart from


def test_failed_warm_solve_is_repeated_cold(libs, oracle_lib):
    """a record that cannot be continued (here: its row duals zeroed, so the first factorisation breaks down) costs a cold
    solve in the same call, not an inner failure; the result is the cold solve's"""
    fwd, _ = libs
    g = load_fixture("benchmark1")[1]
    o = oracle_lib.Oracle(g, ipm_tol=IPM_TOL, warm_start=False)
    w = WarmRecords(fwd, g)
    for _ in range(3):
        wg_step(fwd, "wg_emu_vertex_step", g, o.zedge, o.mu, warm=w)
        o.vertex_step(1.0, 1.0); o.edge_step(1.0)
    assert w.buf[w.ptr[:-1]].sum() >= 2           # records exist
    cold = wg_step(fwd, "wg_emu_vertex_step", g, o.zedge, o.mu)
    gen = cold[5]
    n, NW = g.n, 2 * g.n + 1
    for v in np.nonzero(gen)[0]:                  # spoil the row duals of every unit (layout: warm_start.h): negative
        m, d = g.poly_ptr[v + 1] - g.poly_ptr[v], g.inc_ptr[v + 1] - g.inc_ptr[v]
        units, stride = (4 + 2 * n + 2 * NW + 1) & ~1, 2 * NW + 2 + 2 * m      # (4m row duals as f32: warm_start.h)
        assert w.ptr[v + 1] - w.ptr[v] == units + (d + 1) * stride
        for u in range(d + 1):
            w.buf[w.ptr[v] + units + u * stride + 2 * NW + 2:w.ptr[v] + units + (u + 1) * stride].view(np.float32)[:] = -1.0
    a = wg_step(fwd, "wg_emu_vertex_step", g, o.zedge, o.mu, warm=w)
    assert a[4][0] == 0 and (a[6][gen] == 0).all()
    assert (a[7][gen] >= cold[7][gen]).all()      # (the failed attempt's iterations are counted on top of the cold solve's:
                                                  #  here it stops at its first complementarity test, after none)
    assert (w.buf[w.ptr[:-1]][gen] == 1.0).all()  # and the cold solve left a fresh record,
    theta, n_cold = w.buf[w.ptr[:-1] + 2][gen], w.buf[w.ptr[:-1] + 3][gen]
    assert (n_cold == cold[7][gen]).all()         # its iteration count, and a far-warm threshold below this step's dT (ws_learn)
    assert ((theta >= 0.1) & (theta < 1.0)).all()
    assert np.abs(a[0] - cold[0]).max() <= 1e-12


def test_lds_budget_of_the_named_configs(libs):
    """LDS per workgroup: BASELINE config 5 (R^6, 8 incident edges, 12 facets) must leave room for TWO workgroups per CU
    (160 KB); benchmark4's largest vertex and a degree-80 box vertex must fit one."""
    fwd, _ = libs
    assert 8 * fwd.wg_emu_lds_doubles(6, 9, 12) <= 80 * 1024
    fwd.wg_emu_set_box(1)
    try:      # the BOX instantiation's structured unit layout: THREE workgroups of config 5 per CU
        assert 3 * 8 * fwd.wg_emu_lds_doubles(6, 9, 12) <= 160 * 1024
    finally:
        fwd.wg_emu_set_box(0)
    assert 8 * fwd.wg_emu_lds_doubles(2, 11, 7) <= 32 * 1024
    assert 8 * fwd.wg_emu_lds_doubles(2, 81, 4) <= 160 * 1024
