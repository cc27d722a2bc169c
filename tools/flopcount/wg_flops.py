"""Counted (not modelled) f64 operations of one vertex step: ctypes front end of tools/flopcount/wg_flops.cpp, the workgroup
vertex program's own source compiled for the host with a counting scalar type.  Measurement tooling for bench.py's
``roofline_fp``; never part of the product path."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "gcs_admm_amd", "csrc")
LIB = os.path.join(HERE, "libwgflops.so")


def build(force=False):
    src = os.path.join(HERE, "wg_flops.cpp")
    deps = [src, os.path.join(CSRC, "vertex_wg.h"), os.path.join(CSRC, "gcs_math.h")]
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-I" + CSRC, src, "-o", LIB])
    return LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def count_vertex_step(g, zedge=None, mu=None, rho=1.0, ipm_tol=1e-9, max_vertices=4000):
    """{"flops", "div", "sqrt", "newton_iterations_per_vertex", "vertices"} of one vertex step of graph ``g`` (from the zero
    state unless zedge / mu are given).  Large graphs are sampled: the count of the first ``max_vertices`` generic vertices
    is scaled to all of them (the lattices are homogeneous; the sample size is reported)."""
    lib = C.CDLL(build())
    c, E, V = g.c, g.num_edges, g.num_vertices
    NI = 2 * E
    zedge = np.zeros((c, E)) if zedge is None else np.ascontiguousarray(zedge, dtype=np.float64)
    mu = np.zeros((c, NI)) if mu is None else np.ascontiguousarray(mu, dtype=np.float64)
    deg = np.diff(g.inc_ptr)
    src, dst = g.src, g.dst
    din_all = np.zeros(V, dtype=np.int64)
    np.add.at(din_all, np.repeat(np.arange(V), deg), 1 - g.inc_out)
    generic = (din_all > 0) & (deg - din_all > 0)
    generic[[src, dst]] = False
    total = int(generic.sum())
    counts = np.zeros(5, dtype=np.int64)
    # sampling: count on a prefix of the vertex order (for the lattice generators: whole rows) and scale
    Vs = V if total <= max_vertices else max_vertices + 2
    inc_ptr = np.ascontiguousarray(g.inc_ptr[:Vs + 1].astype(np.int32))
    r = lib.wg_count_vertex_step(g.n, Vs, E, NI, _p(inc_ptr), _p(g.inc_edge), _p(g.inc_out), _p(g.poly_ptr), _p(g.poly_A),
                                 _p(g.poly_b), _p(g.interior), src, dst, _p(zedge), _p(mu), C.c_double(rho), C.c_double(1.0),
                                 C.c_double(1e-4), C.c_double(ipm_tol), 60, _p(counts))
    if r != 0:
        raise RuntimeError("wg_count_vertex_step failed")
    sampled = int(counts[4])
    scale = total / max(sampled, 1)
    return {"flops": float(counts[0]) * scale, "div": float(counts[1]) * scale, "sqrt": float(counts[2]) * scale,
            "newton_iterations_per_vertex": float(counts[3]) / max(sampled, 1), "vertices": total, "vertices_counted": sampled}


if __name__ == "__main__":
    import sys
    sys.path.insert(0, ROOT)
    from gcs_admm_amd.cases import load_fixture
    from gcs_admm_amd.graph import lattice_boxes
    for name, g in (("benchmark4", load_fixture("benchmark4")[1]), ("s10k", lattice_boxes(100, 100, seed=0)),
                    ("lattice n=6 20x20", lattice_boxes(20, 20, n=6, seed=0))):
        r = count_vertex_step(g)
        per = r["flops"] / r["vertices"] / r["newton_iterations_per_vertex"]
        print(name, {k: (round(v, 1) if isinstance(v, float) else v) for k, v in r.items()}, "flops per vertex per Newton iteration: %.0f" % per)
