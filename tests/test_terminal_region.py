"""Terminals that are REGIONS (the reference builds 's' / 't' as points, utils.py:12-28, but admm_solver_v3.py:415-464 constrains them
like any set, with delta_sv / delta_tv in (6) and (7)).  CPU part: the oracle's reduced form (oracle/gcs_oracle.c solve_terminal_region:
live blocks + the cone + one equality) against the sub-problem AS WRITTEN -- every variable and constraint of admm_solver_v3.py:352-466
with delta = 1, nothing reduced -- solved by scipy's SLSQP (an unrelated method: no interior needed, which the written form does not have).
The GPU parity tests of the HIP twin are in test_gpu_configs.py."""
import ctypes as C

import numpy as np
import pytest

from gcs_admm_amd import IPM_TOL
from oracle import oracle as O


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def polygon(rng, m, centre, radius):
    ang = np.sort(rng.uniform(0, 2 * np.pi, m) * 0.3 + np.arange(m) * 2 * np.pi / m)
    A = np.stack([np.cos(ang), np.sin(ang)], 1)
    b = A @ centre + radius * rng.uniform(0.7, 1.3, m)
    return A, b


def oracle_terminal(n, A, b, cen, d, d_in, is_src, T, rho, tol=IPM_TOL, warm=None):
    """warm: the vertex's warm-start record (oracle_warm_doubles doubles, zero = none yet), None = cold"""
    lib = O.lib()
    ip = O._Inner(1e-4, tol, 60, None, None)
    copy = np.zeros((2 * n + 1, d)); xv = np.zeros(2 * n); zv = np.zeros(2 * n); yv = np.zeros(1)
    r = lib.oracle_solve_vertex(n, A.shape[0], _p(A), _p(b), _p(cen), d, d_in, int(is_src), int(not is_src), _p(T), C.c_double(rho),
                                C.byref(ip), _p(copy), _p(xv), _p(zv), _p(yv), _p(warm) if warm is not None else None)
    assert r >= 0, r
    return copy, xv, zv, yv[0], r


def as_written(n, A, b, d, d_in, is_src, T, rho, start):
    """admm_solver_v3.py:352-466 for v = 's' (is_src) or 't': unknowns x_v, z_v, y_v and (P_e, Q_e, y_e) per incident edge"""
    from scipy.optimize import minimize
    m = A.shape[0]
    NV = 4 * n + 1
    ix = lambda k: k; iz = lambda k: 2 * n + k; iy = 4 * n
    P = lambda e, k: NV + e * NV + k; Q = lambda e, k: NV + e * NV + 2 * n + k; Y = lambda e: NV + e * NV + 4 * n
    N = NV * (d + 1)
    own = lambda e, k: Q(e, k) if e < d_in else P(e, k)          # O_e: the endpoint that is v itself (incoming: head, outgoing: tail)

    def f(w):
        val = np.sqrt(np.sum((w[iz(0):iz(n)] - w[iz(n):iz(2 * n)]) ** 2) + 1e-18)
        for e in range(d):
            val += 1e-4 * w[Y(e)]
            val += 0.5 * rho * (np.sum((w[P(e, 0):P(e, n)] - T[0:n, e]) ** 2) + np.sum((w[Q(e, 0):Q(e, n)] - T[n:2 * n, e]) ** 2) + (w[Y(e)] - T[2 * n, e]) ** 2)
        return val

    def grad(w):
        g = np.zeros(N)
        dz = w[iz(0):iz(n)] - w[iz(n):iz(2 * n)]
        nz = np.sqrt(np.sum(dz ** 2) + 1e-18)
        g[iz(0):iz(n)] = dz / nz; g[iz(n):iz(2 * n)] = -dz / nz
        for e in range(d):
            g[Y(e)] = 1e-4 + rho * (w[Y(e)] - T[2 * n, e])
            g[P(e, 0):P(e, n)] = rho * (w[P(e, 0):P(e, n)] - T[0:n, e])
            g[Q(e, 0):Q(e, n)] = rho * (w[Q(e, 0):Q(e, n)] - T[n:2 * n, e])
        return g
    G, h, E, fe = [], [], [], []          # G w <= h, E w = fe

    def row():
        return np.zeros(N)
    for i in range(2):
        for j in range(m):
            r = row(); r[iz(i * n):iz(i * n + n)] = A[j]; r[iy] = -b[j]; G.append(r); h.append(0.0)                                   # 1
            r = row(); r[ix(i * n):ix(i * n + n)] = A[j]; r[iz(i * n):iz(i * n + n)] = -A[j]; r[iy] = b[j]; G.append(r); h.append(b[j])   # 2
            for e in range(d):
                r = row(); r[own(e, i * n):own(e, i * n) + n] = A[j]; r[Y(e)] = -b[j]; G.append(r); h.append(0.0)                     # 3
                r = row(); r[ix(i * n):ix(i * n + n)] = A[j]; r[own(e, i * n):own(e, i * n) + n] = -A[j]; r[Y(e)] = b[j]; G.append(r); h.append(b[j])   # 4
    for e in range(d):
        for k in range(n):
            r = row(); r[P(e, n + k)] = 1; r[Q(e, k)] = -1; E.append(r); fe.append(0.0)                                               # 5
    dsv, dtv = (1.0, 0.0) if is_src else (0.0, 1.0)
    r = row(); r[iy] = 1
    for e in range(d_in): r[Y(e)] = -1
    E.append(r); fe.append(dsv)                                                                                                      # 6
    r = row(); r[iy] = 1
    for e in range(d_in, d): r[Y(e)] = -1
    E.append(r); fe.append(dtv)
    for k in range(2 * n):                                                                                                           # 7
        r = row(); r[iz(k)] = 1; r[ix(k)] = -dsv
        for e in range(d_in): r[own(e, k)] = -1
        E.append(r); fe.append(0.0)
        r = row(); r[iz(k)] = 1; r[ix(k)] = -dtv
        for e in range(d_in, d): r[own(e, k)] = -1
        E.append(r); fe.append(0.0)
    G, h, E, fe = np.array(G), np.array(h), np.array(E), np.array(fe)
    bounds = [(None, None)] * N
    bounds[iy] = (0, 1)
    for e in range(d): bounds[Y(e)] = (0, 1)
    cons = [{"type": "ineq", "fun": lambda w: h - G @ w, "jac": lambda w: -G}, {"type": "eq", "fun": lambda w: E @ w - fe, "jac": lambda w: E}]
    res = minimize(f, start, jac=grad, method="SLSQP", bounds=bounds, constraints=cons, options={"maxiter": 2000, "ftol": 1e-15})
    w = res.x
    copy = np.zeros((2 * n + 1, d))
    for e in range(d):
        copy[0:n, e] = w[P(e, 0):P(e, n)]; copy[n:2 * n, e] = w[Q(e, 0):Q(e, n)]; copy[2 * n, e] = w[Y(e)]
    return copy, w[ix(0):ix(2 * n)], w[iz(0):iz(2 * n)], w[iy], f(w), float(np.abs(E @ w - fe).max()), float((G @ w - h).max())


def start_from(n, d, d_in, copy, xv):
    """a point of the written form assembled from the reduced solution (SLSQP then has to confirm or improve it) -- perturbed"""
    NV = 4 * n + 1
    w = np.zeros(NV * (d + 1))
    w[0:2 * n] = xv; w[2 * n:4 * n] = xv; w[4 * n] = 1.0
    for e in range(d):
        base = NV + e * NV
        w[base:base + n] = copy[0:n, e]; w[base + 2 * n:base + 3 * n] = copy[n:2 * n, e]; w[base + 4 * n] = copy[2 * n, e]
    return w


@pytest.mark.parametrize("seed", range(8))
def test_reduced_terminal_form_solves_the_sub_problem_as_written(seed):
    rng = np.random.default_rng(seed)
    n, is_src = 2, seed % 2 == 0
    m = 3 + seed % 4
    cen = rng.uniform(-1, 1, n)
    A, b = polygon(rng, m, cen, 0.6)
    d_in, d_out = (1 + seed % 2, 2 + seed % 3) if is_src else (2 + seed % 3, 1 + seed % 2)
    d = d_in + d_out
    rho = [0.5, 1.0, 4.0][seed % 3]
    T = np.zeros((2 * n + 1, d))
    T[0:2 * n] = np.tile(cen, 2)[:, None] * 0.4 + 0.35 * rng.normal(size=(2 * n, d))
    T[2 * n] = rng.uniform(-0.1, 0.8, d)
    copy, xv, zv, yv, its = oracle_terminal(n, A, b, cen, d, d_in, is_src, T, rho)
    lo, hi = (d_in, d) if is_src else (0, d_in)
    # the reduced solution is feasible for the written form: live y sum to one, the dead side is off, x = sum of the live blocks, every O in y X
    assert abs(copy[2 * n, lo:hi].sum() - 1) < 1e-9 and np.all(copy[2 * n, :lo] == 0) and np.all(copy[2 * n, hi:] == 0) and yv == 1.0
    assert np.allclose(xv, zv)
    # SLSQP on the written form, from two different starts: a neutral one and the reduced solution with noise
    neutral = start_from(n, d, d_in, np.vstack([np.tile(cen, 2)[:, None] * np.ones((1, d)) / max(hi - lo, 1), np.where((np.arange(d) >= lo) & (np.arange(d) < hi), 1.0 / (hi - lo), 0.0)[None]]),
                         np.tile(cen, 2))
    best = None
    for start in (neutral, start_from(n, d, d_in, copy, xv) + 1e-3 * rng.normal(size=neutral.shape)):
        got = as_written(n, A, b, d, d_in, is_src, T, rho, start)
        if got[5] < 1e-8 and got[6] < 1e-8 and (best is None or got[4] < best[4]):
            best = got
    assert best is not None
    cw, xw, zw, yw, fw = best[:5]
    # same objective value (the oracle's, evaluated as written) and the same coupled words (unique: SURVEY A.4)
    f_or = np.linalg.norm(xv[:n] - xv[n:]) + sum(1e-4 * copy[2 * n, e] + 0.5 * rho * (np.sum((copy[0:n, e] - T[0:n, e]) ** 2) + np.sum((copy[n:2 * n, e] - T[n:2 * n, e]) ** 2)
                                                                                     + (copy[2 * n, e] - T[2 * n, e]) ** 2) for e in range(d))
    assert f_or <= fw + 1e-7, (f_or, fw)
    assert abs(f_or - fw) < 2e-6 * max(1, abs(fw)), (f_or, fw)
    assert np.abs(cw - copy).max() < 2e-5, np.abs(cw - copy).max()
    assert abs(yw - 1) < 1e-7


def test_a_point_terminal_is_the_limit_of_a_small_region():
    """a box of half-width 2e-5 (just above the 1e-5 rule: the region solve) against the closed form of the point"""
    rng = np.random.default_rng(5)
    n, d_in, d_out, rho = 2, 2, 3, 1.0
    cen = np.array([0.3, -0.2])
    A = np.vstack([np.eye(n), -np.eye(n)])
    T = np.zeros((2 * n + 1, d_in + d_out)); T[:2 * n] = 0.3 * rng.normal(size=(2 * n, 5)); T[2 * n] = rng.uniform(0, 0.6, 5)
    small = oracle_terminal(n, A, np.hstack([cen + 2e-5, -cen + 2e-5]), cen, 5, d_in, True, T, rho)
    point = oracle_terminal(n, A, np.hstack([cen + 1e-6, -cen + 1e-6]), cen, 5, d_in, True, T, rho)
    assert small[4] > 0 and point[4] == 0          # (iterations: the region solve ran / the closed form did)
    assert np.abs(small[0] - point[0]).max() < 1e-4


# ---- the HIP library's solve (gcs_admm_amd/csrc/terminal_region.h) compiled for the host, against the oracle ----
@pytest.fixture(scope="module")
def term_emu():
    import os
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    csrc = os.path.join(os.path.dirname(here), "gcs_admm_amd", "csrc")
    src, out = os.path.join(here, "hostemu", "term_emu.cpp"), os.path.join(here, "hostemu", "libtermemu.so")
    deps = [src, os.path.join(csrc, "terminal_region.h"), os.path.join(csrc, "gcs_math.h")]
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-I" + csrc, src, "-o", out])
    return C.CDLL(out)


def emu_terminal(lib, n, A, b, cen, d, d_in, is_src, T, rho, tol=IPM_TOL, warm=None):
    copy = np.full((2 * n + 1, d), np.nan); xv = np.zeros(2 * n); zv = np.zeros(2 * n); yv = np.zeros(1)
    r = lib.term_emu_solve(n, A.shape[0], _p(A), _p(b), _p(cen), d, d_in, int(is_src), _p(T), C.c_double(rho), C.c_double(1e-4),
                           C.c_double(tol), 60, _p(copy), _p(xv), _p(zv), _p(yv), _p(warm) if warm is not None else None)
    assert r >= 0, r
    return copy, xv, zv, yv[0], r


def box_in(rng, n, cen, half):
    A = np.vstack([np.eye(n), -np.eye(n)])
    h = half * rng.uniform(0.7, 1.3, 2 * n)
    return A, np.hstack([cen + h[:n], -cen + h[n:]])


@pytest.mark.parametrize("n,seed", [(2, 0), (2, 1), (2, 2), (2, 3), (1, 4), (3, 5), (3, 6), (6, 7), (6, 8), (8, 9)])
def test_device_body_of_the_terminal_solve_matches_the_oracle(term_emu, n, seed):
    """same method, two implementations (block-parallel phases there, plain loops here): same iteration count (+-1), words to 2e-5
    (typically 1e-10; the last iterations are ill-conditioned and amplify the order of the sums, as in the generic vertex programs)"""
    rng = np.random.default_rng(100 + seed)
    is_src = seed % 2 == 0
    cen = rng.uniform(-1, 1, n)
    A, b = polygon(rng, 3 + seed % 4, cen, 0.6) if n == 2 else box_in(rng, n, cen, 0.5)
    d_in, d_out = 1 + seed % 3, 2 + seed % 4
    d = d_in + d_out
    rho = [0.25, 1.0, 4.0][seed % 3]
    T = np.zeros((2 * n + 1, d))
    T[0:2 * n] = np.tile(cen, 2)[:, None] * 0.4 + 0.35 * rng.normal(size=(2 * n, d))
    T[2 * n] = rng.uniform(-0.1, 0.8, d)
    a = oracle_terminal(n, A, b, cen, d, d_in, is_src, T, rho)
    e = emu_terminal(term_emu, n, A, b, cen, d, d_in, is_src, T, rho)
    assert abs(a[4] - e[4]) <= 1, (a[4], e[4])          # (a barrier parameter that lands on the tolerance may stop one of them an iteration later)
    assert not np.isnan(e[0]).any()
    assert np.abs(a[0] - e[0]).max() < 2e-5 and np.abs(a[1] - e[1]).max() < 2e-5 and e[3] == 1.0
    assert np.array_equal(e[1], e[2])


@pytest.mark.parametrize("n,seed", [(2, 0), (2, 1), (3, 2), (6, 3)])
def test_warm_started_terminal_solves(term_emu, n, seed):
    """a sequence of solves whose targets drift (as along an ADMM run), each restarted from the record the previous one left: oracle and
    device body agree with each other, and both with a cold solve of the same sub-problem; the warm solves need fewer iterations"""
    rng = np.random.default_rng(200 + seed)
    is_src = seed % 2 == 0
    cen = rng.uniform(-1, 1, n)
    A, b = polygon(rng, 4 + seed % 3, cen, 0.6) if n == 2 else box_in(rng, n, cen, 0.5)
    d_in, d_out = 2, 3
    d, L = d_in + d_out, (d_out if is_src else d_in)
    rho = [0.5, 2.0][seed % 2]
    T = np.zeros((2 * n + 1, d))
    T[0:2 * n] = np.tile(cen, 2)[:, None] * 0.4 + 0.35 * rng.normal(size=(2 * n, d))
    T[2 * n] = rng.uniform(-0.1, 0.8, d)
    O.lib().oracle_warm_doubles.restype = C.c_longlong
    term_emu.term_emu_record_doubles.restype = C.c_longlong
    wo = np.zeros(O.lib().oracle_warm_doubles(n, A.shape[0], d))
    we = np.zeros(term_emu.term_emu_record_doubles(n, A.shape[0], L))
    assert 4 + L * (2 * (2 * n + 1) + 2 * A.shape[0]) == len(we) <= len(wo)
    cold_its, warm_its = [], []
    for step in range(12):
        if step:
            T = T + (0.02 if step < 8 else 0.3) * rng.normal(size=T.shape) * (np.arange(2 * n + 1)[:, None] >= 0)
        c = oracle_terminal(n, A, b, cen, d, d_in, is_src, T, rho)
        a = oracle_terminal(n, A, b, cen, d, d_in, is_src, T, rho, warm=wo)
        e = emu_terminal(term_emu, n, A, b, cen, d, d_in, is_src, T, rho, warm=we)
        assert np.abs(a[0] - e[0]).max() < 2e-5, (step, np.abs(a[0] - e[0]).max())
        assert step >= 8 or abs(a[4] - e[4]) <= 1, (step, a[4], e[4])      # (large moves sit at the warm / cold threshold: the two may decide differently)
        assert np.abs(a[0] - c[0]).max() < 2e-3, (step, np.abs(a[0] - c[0]).max())        # the fixtures' bound between two solves of one sub-problem
        assert wo[0] == 1.0 and we[0] == 1.0 and np.abs(wo[4:4 + len(we) - 4] - we[4:]).max() < 1e-4      # same record, same layout
        cold_its.append(c[4]); warm_its.append(a[4])
    assert warm_its[0] == cold_its[0]                                   # no record yet: cold
    assert np.mean(warm_its[1:8]) <= np.mean(cold_its[1:8]) - 2         # small moves: the restart pays
