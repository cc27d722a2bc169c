"""ctypes binding of the CPU oracle (oracle/gcs_oracle.c).  Test infrastructure:
imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libgcs_oracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(HERE, "gcs_oracle.c")
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", HERE, "-s"] + (["-B"] if force else []))
    return LIB


class _Graph(C.Structure):
    _fields_ = [("n", C.c_int), ("V", C.c_int), ("E", C.c_int),
                ("edge_tail", C.c_void_p), ("edge_head", C.c_void_p),
                ("inc_ptr", C.c_void_p), ("inc_edge", C.c_void_p), ("inc_out", C.c_void_p),
                ("edge_inc_tail", C.c_void_p), ("edge_inc_head", C.c_void_p),
                ("poly_ptr", C.c_void_p), ("poly_A", C.c_void_p), ("poly_b", C.c_void_p),
                ("center", C.c_void_p), ("src", C.c_int), ("dst", C.c_int),
                ("NI", C.c_int), ("inc_counted", C.c_void_p), ("edge_counted", C.c_void_p),
                ("nx_global", C.c_double), ("nmu_global", C.c_double)]


class _Inner(C.Structure):
    _fields_ = [("eps_edge", C.c_double), ("ipm_tol", C.c_double), ("ipm_max_iter", C.c_int),
                ("warm", C.c_void_p), ("warm_ptr", C.c_void_p)]


class _Admm(C.Structure):
    _fields_ = [("rho", C.c_double), ("tau_incr", C.c_double), ("tau_decr", C.c_double), ("nu", C.c_double),
                ("it_rho_limit", C.c_int), ("eps_abs", C.c_double), ("eps_rel", C.c_double), ("max_it", C.c_int)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB)
        _lib.oracle_compute_cost.restype = C.c_double
        _lib.oracle_warm_doubles.restype = C.c_longlong
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Oracle:
    """One GCS instance on the CPU oracle.  State arrays are numpy float64 in the
    shared layout: copy/mu [c, 2E], zedge [c, E], xv/zv [V, 2n], yv [V]."""

    def __init__(self, g, ipm_tol=1e-11, ipm_max_iter=60, eps_edge=1e-4, num_incidences=None, inc_counted=None,
                 edge_counted=None, nx_global=0.0, nmu_global=0.0, warm_start=True):
        self.g = g
        self._keep = [np.ascontiguousarray(a) for a in (
            g.edge_tail, g.edge_head, g.inc_ptr, g.inc_edge, g.inc_out, g.edge_inc_tail, g.edge_inc_head,
            g.poly_ptr, g.poly_A, g.poly_b, g.interior)]
        k = self._keep
        self._ic = np.ascontiguousarray(inc_counted, dtype=np.uint8) if inc_counted is not None else None
        self._ec = np.ascontiguousarray(edge_counted, dtype=np.uint8) if edge_counted is not None else None
        self.NI = int(num_incidences) if num_incidences else 2 * g.num_edges
        self.G = _Graph(g.n, g.num_vertices, g.num_edges, _p(k[0]), _p(k[1]), _p(k[2]), _p(k[3]), _p(k[4]),
                        _p(k[5]), _p(k[6]), _p(k[7]), _p(k[8]), _p(k[9]), _p(k[10]), g.src, g.dst,
                        int(num_incidences or 0),
                        _p(self._ic) if self._ic is not None else None, _p(self._ec) if self._ec is not None else None,
                        float(nx_global), float(nmu_global))
        self.inner = _Inner(eps_edge, ipm_tol, ipm_max_iter, None, None)
        if warm_start:      # one record per vertex (layout: oracle_warm_doubles), zero = no record yet
            deg = np.diff(g.inc_ptr).astype(np.int64); m = np.diff(g.poly_ptr).astype(np.int64)
            n_ = g.n
            size = 4 + (4 * n_ + 2) + 2 * (2 * n_ + 1) + 2 + (n_ + 1) + 4 * m + deg * (2 * n_ + 3 + 4 * m + 2 * n_ + 1)
            assert size[0] == lib().oracle_warm_doubles(n_, int(m[0]), int(deg[0]))
            self._warm_ptr = np.concatenate([[0], np.cumsum(size)]).astype(np.int64)
            self._warm = np.zeros(int(self._warm_ptr[-1]))
            self.inner.warm = _p(self._warm); self.inner.warm_ptr = _p(self._warm_ptr)
        c, E, V, n = g.c, g.num_edges, g.num_vertices, g.n
        self.zedge = np.zeros((c, E)); self.mu = np.zeros((c, self.NI)); self.copy = np.zeros((c, self.NI))
        self.xv = np.zeros((V, 2 * n)); self.zv = np.zeros((V, 2 * n)); self.yv = np.zeros(V)
        self.ipm_iters = C.c_long(0)

    def vertex_step(self, rho=1.0, mu_scale=1.0, nthreads=0):
        return lib().oracle_vertex_step(C.byref(self.G), _p(self.zedge), _p(self.mu), C.c_double(mu_scale),
                                        C.c_double(rho), C.byref(self.inner), _p(self.copy), _p(self.xv),
                                        _p(self.zv), _p(self.yv), C.byref(self.ipm_iters), nthreads)

    def edge_step(self, mu_scale=1.0):
        s = np.zeros(5)
        lib().oracle_edge_step(C.byref(self.G), _p(self.copy), _p(self.zedge), _p(self.mu), C.c_double(mu_scale), _p(s))
        return s

    def run(self, max_it=1000, rho=1.0, eps_abs=1e-4, eps_rel=1e-3, tau=2.0, nu=10.0, it_rho_limit=100, nthreads=0):
        ap = _Admm(rho, tau, tau, nu, it_rho_limit, eps_abs, eps_rel, max_it)
        trace = np.zeros((max_it, 6)); status = C.c_int(0)
        it = lib().oracle_admm_run(C.byref(self.G), C.byref(ap), C.byref(self.inner), _p(self.zedge), _p(self.mu),
                                   _p(self.copy), _p(self.xv), _p(self.zv), _p(self.yv), _p(trace), C.byref(status),
                                   C.byref(self.ipm_iters), nthreads)
        k = min(it, max_it)
        return dict(iterations=it, status=status.value, trace=trace[:k],
                    rho_seq=np.concatenate([[rho], trace[:k, 0]]),
                    pri_res_seq=np.concatenate([[0.0], trace[:k, 1]]),
                    dual_res_seq=np.concatenate([[0.0], trace[:k, 2]]),
                    inner_failures=int(trace[:k, 5].sum()), cost=self.cost())

    def run_from(self, it_start, max_it, rho=1.0, eps_abs=1e-4, eps_rel=1e-3, tau=2.0, nu=10.0, it_rho_limit=100, nthreads=0):
        """The loop continued from iteration ``it_start`` (``rho`` = the penalty in force there) up to and including iteration
        ``max_it``: iterations of the reference's loop (admm_solver_v3.py:655-733) with its own numbering, so that the rho
        adaptation (``it < it_rho_limit``) sees the same counter as an uninterrupted run.  Returns (it, status, rho, trace rows)."""
        ap = _Admm(rho, tau, tau, nu, it_rho_limit, eps_abs, eps_rel, max_it)
        rows = max(max_it - it_start + 1, 1)
        trace = np.zeros((rows, 6)); status = C.c_int(0); rho_out = C.c_double(rho)
        it = lib().oracle_admm_run_from(C.byref(self.G), C.byref(ap), C.byref(self.inner), _p(self.zedge), _p(self.mu),
                                        _p(self.copy), _p(self.xv), _p(self.zv), _p(self.yv), _p(trace), C.byref(status),
                                        C.byref(self.ipm_iters), nthreads, int(it_start), C.byref(rho_out))
        return it, status.value, rho_out.value, trace[:max(min(it, max_it) - it_start + 1, 0)]

    _STATE = ("zedge", "mu", "copy", "xv", "zv", "yv")

    def snapshot(self):
        """copy of everything a continued run depends on: the ADMM state and the warm-start records of the vertex solves"""
        snap = {k: getattr(self, k).copy() for k in self._STATE}
        if getattr(self, "_warm", None) is not None:
            snap["_warm"] = self._warm.copy()
        return snap

    def restore(self, snap):
        for k, v in snap.items():
            getattr(self, k)[...] = v      # in place: the C side holds pointers into these arrays

    def cost(self):
        return lib().oracle_compute_cost(C.byref(self.G), _p(self.zv), _p(self.zedge), C.c_double(self.inner.eps_edge))

    def control(self, ap, sums, state, fails=0.0, trace_row=None):
        """loop control on (globally reduced) sums; state = np.array([rho, mu_scale, it, status])"""
        lib().oracle_control(C.byref(self.G), C.byref(ap), _p(np.ascontiguousarray(sums, dtype=np.float64)), _p(state),
                             C.c_double(fails), _p(trace_row) if trace_row is not None else None)


def admm_params(rho=1.0, tau=2.0, nu=10.0, it_rho_limit=100, eps_abs=1e-4, eps_rel=1e-3, max_it=1000):
    return _Admm(rho, tau, tau, nu, it_rho_limit, eps_abs, eps_rel, max_it)
