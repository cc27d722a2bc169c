"""Driver of tools/sanitize_cpu.sh (expects /tmp/libemu_asan.so and /tmp/libo_asan.so)."""
import sys, ctypes as C, numpy as np
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from gcs_admm_amd import IPM_TOL  # noqa: E402
from gcs_admm_amd.cases import load_fixture
from gcs_admm_amd.graph import lattice_boxes, graph_from_sets
from conftest import star_case
from oracle.oracle import Oracle
lib=C.CDLL('/tmp/libemu_asan.so')
def p(a): return a.ctypes.data_as(C.c_void_p)
def step(fn,g,z,m):
    c=g.c; NI=2*g.num_edges
    copy=np.zeros((c,NI)); xv=np.zeros((g.num_vertices,2*g.n)); zv=np.zeros_like(xv); yv=np.zeros(g.num_vertices)
    cnt=np.zeros(2,dtype=np.int32); gen=np.zeros(g.num_vertices,dtype=np.int32)
    r=getattr(lib,fn)(g.n,g.num_vertices,g.num_edges,NI,p(g.inc_ptr),p(g.inc_edge),p(g.inc_out),p(g.poly_ptr),p(g.poly_A),p(g.poly_b),p(g.interior),g.src,g.dst,p(z),p(m),C.c_double(1.0),C.c_double(1.0),C.c_double(1e-4),C.c_double(IPM_TOL),60,p(copy),p(xv),p(zv),p(yv),p(cnt),p(gen))
    assert r==0
graphs=[('benchmark4',load_fixture('benchmark4')[1],'emu_vertex_step'),('benchmark3',load_fixture('benchmark3')[1],'emu_vertex_step'),
        ('lattice boxes',lattice_boxes(9,7,seed=1),'emu_vertex_step_box'),('lattice n6',lattice_boxes(5,4,n=6,seed=1),'emu_vertex_step'),
        ('star',graph_from_sets(*star_case(24)),'emu_vertex_step')]
for name,g,fn in graphs:
    o=Oracle(g,ipm_tol=IPM_TOL)
    for it in range(6):
        step(fn,g,o.zedge.copy(),o.mu.copy()); o.vertex_step(1.0,1.0,1); o.edge_step(1.0)
    print(name,'ok')

# ---- the workgroup program's host build under the sanitizers
wg = C.CDLL('/tmp/libwgemu_asan.so')
def wg_step(g, z, m):
    c = g.c; NI = 2 * g.num_edges; V = g.num_vertices
    copy = np.zeros((c, NI)); xv = np.zeros((V, 2 * g.n)); zv = np.zeros_like(xv); yv = np.zeros(V)
    cnt = np.zeros(2, dtype=np.int32); gen = np.zeros(V, dtype=np.int32); st = np.zeros(V, dtype=np.int32); it = np.zeros(V, dtype=np.int32)
    r = wg.wg_emu_vertex_step(g.n, V, g.num_edges, NI, p(g.inc_ptr), p(g.inc_edge), p(g.inc_out), p(g.poly_ptr), p(g.poly_A), p(g.poly_b),
                              p(g.interior), g.src, g.dst, p(z), p(m), C.c_double(1.0), C.c_double(1.0), C.c_double(1e-4), C.c_double(IPM_TOL), 60,
                              p(copy), p(xv), p(zv), p(yv), p(cnt), p(gen), p(st), p(it))
    assert r == 0 and cnt[0] == 0
wg_graphs = [('benchmark4', load_fixture('benchmark4')[1], 0), ('lattice n2', lattice_boxes(6, 5, seed=1), 0), ('lattice n3 box', lattice_boxes(4, 3, n=3, seed=1), 1),
             ('lattice n6', lattice_boxes(4, 3, n=6, seed=1), 0), ('lattice n6 box', lattice_boxes(4, 3, n=6, seed=1), 1), ('star', graph_from_sets(*star_case(24)), 0)]
for name, g, box in wg_graphs:
    o = Oracle(g, ipm_tol=IPM_TOL)
    wg.wg_emu_set_box(box)
    for it in range(4):
        wg_step(g, o.zedge.copy(), o.mu.copy()); o.vertex_step(1.0, 1.0, 1); o.edge_step(1.0)
    wg.wg_emu_set_box(0)
    print('workgroup program,', name, 'ok')

# ---- oracle under the sanitizers
import oracle.oracle as O
O._lib=C.CDLL('/tmp/libo_asan.so'); O._lib.oracle_compute_cost.restype=C.c_double; O._lib.oracle_warm_doubles.restype=C.c_longlong
from gcs_admm_amd.cases import load_fixture
from gcs_admm_amd.graph import lattice_boxes
for name in ('benchmark1','benchmark4'):
    case,g=load_fixture(name); r=O.Oracle(g,ipm_tol=IPM_TOL).run(nthreads=2); print(name,r['iterations'])
g=lattice_boxes(5,4,n=6,seed=1); r=O.Oracle(g,ipm_tol=IPM_TOL).run(max_it=5,eps_abs=0,eps_rel=0,nthreads=2); print('n6',r['iterations'])

# ---- terminals that are regions: the device body's host build and the oracle's restatement (both under the sanitizers)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_terminal_region as TR
term = C.CDLL('/tmp/libtermemu_asan.so')
for n, seed in [(2, 0), (2, 1), (1, 4), (3, 5), (6, 7)]:
    rng = np.random.default_rng(100 + seed)
    cen = rng.uniform(-1, 1, n)
    A, b = TR.polygon(rng, 3 + seed % 4, cen, 0.6) if n == 2 else TR.box_in(rng, n, cen, 0.5)
    d_in, d_out = 1 + seed % 3, 2 + seed % 4
    T = np.zeros((2 * n + 1, d_in + d_out)); T[:2 * n] = 0.35 * rng.normal(size=(2 * n, d_in + d_out)); T[2 * n] = rng.uniform(-0.1, 0.8, d_in + d_out)
    e = TR.emu_terminal(term, n, A, b, cen, d_in + d_out, d_in, seed % 2 == 0, T, 1.0)
    lib_o = O._lib
    ip = O._Inner(1e-4, IPM_TOL, 60, None, None)
    copy = np.zeros((2 * n + 1, d_in + d_out)); xv = np.zeros(2 * n); zv = np.zeros(2 * n); yv = np.zeros(1)
    r = lib_o.oracle_solve_vertex(n, A.shape[0], p(A), p(b), p(cen), d_in + d_out, d_in, int(seed % 2 == 0), int(seed % 2 != 0), p(T), C.c_double(1.0),
                                  C.byref(ip), p(copy), p(xv), p(zv), p(yv), None)
    assert r >= 0 and np.abs(copy - e[0]).max() < 2e-5
    print('region terminal n =', n, 'ok')
