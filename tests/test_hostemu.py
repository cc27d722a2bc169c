"""The wavefront program of the vertex-step kernel (gcs_admm_amd/csrc/vertex_program.h) executed as
a lock-step host emulation (tests/hostemu/emu.cpp) against the CPU oracle.  This checks the lane
algorithm and its LDS protocol (unwritten LDS is poisoned with NaN) without a GPU; the GPU parity
tests proper are in test_gpu_parity.py.

Tolerance: both sides run the same interior-point iteration to barrier parameter 1e-9, at which
the sub-problem minimiser itself is resolved to ~1e-4 in weakly determined components; two
implementations that differ in operation order agree to ~1e-8 typically and to that 1e-4 scale in
the worst case."""
from gcs_admm_amd import IPM_TOL
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from gcs_admm_amd.cases import load_fixture

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.fixture(scope="module")
def emu():
    src = os.path.join(HERE, "hostemu", "emu.cpp")
    out = os.path.join(HERE, "hostemu", "libemu.so")
    hdr = os.path.join(ROOT, "gcs_admm_amd", "csrc", "vertex_program.h")
    deps = [src, hdr, os.path.join(HERE, "hostemu", "emu_body.inc")] + [os.path.join(os.path.dirname(hdr), f) for f in ("vertex_program.inc", "warm_start.h")]
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-I" + os.path.dirname(hdr), src, "-o", out])
    return C.CDLL(out)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class WarmRecords:
    """the warm-start workspace a handle owns on the device (csrc/warm_start.h: one record per vertex), for the host build"""

    def __init__(self, g):
        n, NW = g.n, 2 * g.n + 1
        deg = np.diff(g.inc_ptr).astype(np.int64); m = np.diff(g.poly_ptr).astype(np.int64)
        size = ((4 + 2 * n + 2 * NW + 1) & ~1) + (deg + 1) * (2 * NW + 2 + 2 * m)
        self.ptr = np.concatenate([[0], np.cumsum(size)]).astype(np.int64)
        self.buf = np.zeros(int(self.ptr[-1]))


def emu_step(lib, g, zedge, mu, rho, mu_scale, fn="emu_vertex_step", warm=None):
    getattr(lib, fn + "_set_warm")(_p(warm.buf) if warm else None, _p(warm.ptr) if warm else None)
    c, NI, V = g.c, 2 * g.num_edges, g.num_vertices
    copy = np.zeros((c, NI)); xv = np.zeros((V, 2 * g.n)); zv = np.zeros_like(xv); yv = np.zeros(V)
    cnt = np.zeros(2, dtype=np.int32); gen = np.zeros(V, dtype=np.int32)
    r = getattr(lib, fn)(g.n, V, g.num_edges, NI, _p(g.inc_ptr), _p(g.inc_edge), _p(g.inc_out), _p(g.poly_ptr),
                            _p(g.poly_A), _p(g.poly_b), _p(g.interior), g.src, g.dst, _p(zedge), _p(mu),
                            C.c_double(rho), C.c_double(mu_scale), C.c_double(1e-4), C.c_double(IPM_TOL), 60,
                            _p(copy), _p(xv), _p(zv), _p(yv), _p(cnt), _p(gen))
    assert r == 0
    return copy, xv, zv, yv, cnt, gen


@pytest.mark.parametrize("name,steps", [("benchmark1", 20), ("benchmark4", 25), ("test_autogen2", 15)])
def test_emulated_wave_program_matches_oracle(emu, oracle_lib, name, steps):
    case, g = load_fixture(name)
    o = oracle_lib.Oracle(g, ipm_tol=IPM_TOL)
    w = WarmRecords(g)          # both sides restart from their own records, by the same rule (warm_start.h)
    diffs = []
    for it in range(steps):
        copy, xv, zv, yv, cnt, gen = emu_step(emu, g, o.zedge.copy(), o.mu.copy(), 1.0, 1.0, warm=w)
        assert cnt[0] == 0
        assert o.vertex_step(1.0, 1.0) == 0
        mask = np.zeros(2 * g.num_edges, bool)
        for v in np.nonzero(gen)[0]:
            mask[g.inc_ptr[v]:g.inc_ptr[v + 1]] = True
        assert np.isfinite(copy[:, mask]).all()
        diffs.append(np.abs(copy[:, mask] - o.copy[:, mask]).max())
        assert np.abs(yv[gen == 1] - o.yv[gen == 1]).max() <= 5e-4
        o.edge_step(1.0)
    diffs = np.array(diffs)
    assert diffs.max() <= 2e-3 and np.median(diffs) <= 1e-5
    # record header [2] far-warm threshold, [3] iterations of the last cold solve: the same learning rule on both sides
    gv = np.nonzero(gen == 1)[0]
    hd = np.array([[w.buf[w.ptr[v] + k] for k in (2, 3)] for v in gv])
    ho = np.array([[o._warm[o._warm_ptr[v] + k] for k in (2, 3)] for v in gv])
    assert (hd[:, 1] > 0).all() and ((hd[:, 0] == 0) | ((hd[:, 0] >= 0.1) & (hd[:, 0] <= 10.0))).all()
    assert (np.abs(hd - ho).max(axis=1) == 0).mean() >= 0.8


def test_fixed_facet_variant_equals_generic(emu, oracle_lib):
    """vertex_program.h instantiates the program twice (any facet count / canonical axis-aligned boxes: 4 facets, half of the
    row duals in registers, compile-time facet normals); on a lattice of boxes both must produce the same numbers."""
    from gcs_admm_amd.graph import lattice_boxes
    g = lattice_boxes(7, 6, seed=2)
    o = oracle_lib.Oracle(g, ipm_tol=IPM_TOL)
    wa, wb = WarmRecords(g), WarmRecords(g)
    for it in range(12):
        z0, m0 = o.zedge.copy(), o.mu.copy()
        a = emu_step(emu, g, z0, m0, 1.0, 1.0, warm=wa)
        b = emu_step(emu, g, z0, m0, 1.0, 1.0, fn="emu_vertex_step_box", warm=wb)
        gen = a[5] == 1
        mask = np.zeros(2 * g.num_edges, bool)
        for v in np.nonzero(gen)[0]:
            mask[g.inc_ptr[v]:g.inc_ptr[v + 1]] = True
        assert np.abs(a[0][:, mask] - b[0][:, mask]).max() <= 1e-12
        assert np.abs(a[3][gen] - b[3][gen]).max() <= 1e-12
        o.vertex_step(1.0, 1.0); o.edge_step(1.0)


@pytest.mark.parametrize("n", [3, 6])
def test_emulated_wave_program_other_dimensions(emu, oracle_lib, n):
    from gcs_admm_amd.graph import lattice_boxes
    g = lattice_boxes(5, 4, n=n, seed=1)
    o = oracle_lib.Oracle(g, ipm_tol=IPM_TOL)
    w = WarmRecords(g)
    for it in range(6):
        copy, xv, zv, yv, cnt, gen = emu_step(emu, g, o.zedge.copy(), o.mu.copy(), 1.0, 1.0, warm=w)
        assert cnt[0] == 0 and o.vertex_step(1.0, 1.0) == 0
        mask = np.zeros(2 * g.num_edges, bool)
        for v in np.nonzero(gen)[0]:
            mask[g.inc_ptr[v]:g.inc_ptr[v + 1]] = True
        assert np.abs(copy[:, mask] - o.copy[:, mask]).max() <= 2e-3
        o.edge_step(1.0)


def test_emulated_wave_program_high_degree(emu, oracle_lib):
    from conftest import star_case
    from gcs_admm_amd.graph import graph_from_sets
    As, bs, n = star_case(24)
    g = graph_from_sets(As, bs, n)
    assert np.diff(g.inc_ptr).max() >= 40
    o = oracle_lib.Oracle(g, ipm_tol=IPM_TOL)
    w = WarmRecords(g)
    for it in range(10):
        copy, xv, zv, yv, cnt, gen = emu_step(emu, g, o.zedge.copy(), o.mu.copy(), 1.0, 1.0, warm=w)
        assert cnt[0] == 0 and o.vertex_step(1.0, 1.0) == 0
        mask = np.zeros(2 * g.num_edges, bool)
        for v in np.nonzero(gen)[0]:
            mask[g.inc_ptr[v]:g.inc_ptr[v + 1]] = True
        assert np.abs(copy[:, mask] - o.copy[:, mask]).max() <= 2e-3
        o.edge_step(1.0)


def test_group_placement_dense_equals_aligned(emu, monkeypatch):
    """group_base(): the dense and the row-aligned placement of the vertex groups in a wavefront are two
    schedules of the same computation -- identical results lane for lane (the emulation reduces in lane order)."""
    from gcs_admm_amd.graph import lattice_boxes
    g = lattice_boxes(9, 7, seed=4)
    rng = np.random.default_rng(0)
    zedge = 0.1 * rng.normal(size=(g.c, g.num_edges)); mu = 0.05 * rng.normal(size=(g.c, 2 * g.num_edges))
    res = {}
    for align in ("0", "1"):
        monkeypatch.setenv("GCS_EMU_ALIGN", align)
        res[align] = [emu_step(emu, g, zedge.copy(), mu.copy(), 1.3, 1.0, fn=f) for f in ("emu_vertex_step", "emu_vertex_step_box")]
    for k in range(2):
        gen = res["0"][k][5] == 1
        assert np.array_equal(res["0"][k][5], res["1"][k][5])
        for a, b in zip(res["0"][k][:4], res["1"][k][:4]):
            assert np.array_equal(np.nan_to_num(a), np.nan_to_num(b))


def test_layout_rules(emu):
    """LDS slot stride = 2 mod 4 doubles for every facet count (lanes of different slots reading the same offset
    must not meet on one bank: a stride of 0 mod 64 dwords cost 3-7 % on the lattices), and group_base() never lets
    a side segment of an aligned group straddle a 16-lane row"""
    for n in (2, 3, 6):
        for mm in range(1, 40):
            size = emu.emu_slot_size(n, mm)
            assert size % 4 == 2, (n, mm, size)
    for d in range(2, 33):
        for d_in in range(1, d):
            if d_in > 16 or d - d_in > 16:
                continue
            cur = 0
            while True:
                base = emu.emu_group_base(cur, d, d_in, 1)
                if base < 0:
                    break
                assert base >= cur and base + d + 1 <= 64
                assert (base + 1) >> 4 == (base + d_in) >> 4 and (base + d_in + 1) >> 4 == (base + d) >> 4
                cur = base + d + 1
            assert emu.emu_group_base(0, d, d_in, 1) >= 0
            assert emu.emu_group_base(7, d, d_in, 0) == (7 if 7 + d + 1 <= 64 else -1)


def test_failed_warm_solve_is_repeated_cold_in_the_same_step(emu, oracle_lib):
    """a record the solve cannot continue from (its row duals made negative: the first complementarity test fails) costs a cold
    solve inside the same step, for that vertex alone, not an inner failure; the other vertices of the wavefront go on warm"""
    g = load_fixture("benchmark1")[1]
    o = oracle_lib.Oracle(g, ipm_tol=IPM_TOL, warm_start=False)
    w = WarmRecords(g)
    for _ in range(3):
        emu_step(emu, g, o.zedge, o.mu, 1.0, 1.0, warm=w)
        o.vertex_step(1.0, 1.0); o.edge_step(1.0)
    cold = emu_step(emu, g, o.zedge, o.mu, 1.0, 1.0)
    ref = emu_step(emu, g, o.zedge, o.mu, 1.0, 1.0, warm=WarmRecordsCopy(w))
    gen = np.nonzero(cold[5] == 1)[0]
    v = gen[1]
    n, NW = g.n, 2 * g.n + 1
    m, d = g.poly_ptr[v + 1] - g.poly_ptr[v], g.inc_ptr[v + 1] - g.inc_ptr[v]
    units, stride = (4 + 2 * n + 2 * NW + 1) & ~1, 2 * NW + 2 + 2 * m      # (4m row duals as f32: warm_start.h)
    assert w.buf[w.ptr[v]] == 1.0
    for u in range(d + 1):
        w.buf[w.ptr[v] + units + u * stride + 2 * NW + 2:w.ptr[v] + units + (u + 1) * stride].view(np.float32)[:] = -1.0
    a = emu_step(emu, g, o.zedge, o.mu, 1.0, 1.0, warm=w)
    assert a[4][0] == 0                                # no inner failure
    cols = slice(g.inc_ptr[v], g.inc_ptr[v + 1])
    assert np.abs(a[0][:, cols] - cold[0][:, cols]).max() <= 1e-12            # the spoiled vertex: the cold solve's result
    others = np.ones(2 * g.num_edges, bool); others[cols] = False
    assert np.array_equal(a[0][:, others], ref[0][:, others])                  # everyone else: exactly the warm step
    assert w.buf[w.ptr[v]] == 1.0                      # and a fresh record, whose far-warm threshold came down (ws_learn)
    assert 0.1 <= w.buf[w.ptr[v] + 2] < 1.0 and w.buf[w.ptr[v] + 3] > 0


class WarmRecordsCopy:
    def __init__(self, w):
        self.ptr, self.buf = w.ptr, w.buf.copy()
