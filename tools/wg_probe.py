"""GPU probe of the two vertex programs (wavefront / workgroup): parity of one vertex step against the oracle and
iteration rates on a few graphs.  Development tool (results go to gpurun_out/)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from gcs_admm_amd import IPM_TOL  # noqa: E402
import torch
from gcs_admm_amd.cases import load_fixture
from gcs_admm_amd.graph import lattice_boxes, graph_from_sets
from gcs_admm_amd import solver as _solver_mod
if os.environ.get('GCSADMM_PROBE_LIB'):
    _solver_mod.LIB_PATH = os.path.abspath(os.environ['GCSADMM_PROBE_LIB'])   # development: probe an experimental build
from gcs_admm_amd.solver import DeviceSolver
from oracle.oracle import Oracle

out = {}

def parity(name, g, program, steps=6, dtype="f64"):
    o = Oracle(g, ipm_tol=IPM_TOL)
    d = DeviceSolver(g, dtype, device=0, program=program)
    d.reset()
    worst = 0.0
    for it in range(steps):
        d.zedge.copy_(torch.from_numpy(o.zedge)); d.mu.copy_(torch.from_numpy(o.mu))
        d.vertex_step()
        assert o.vertex_step(1.0, 1.0) == 0
        copy = d.copy.cpu().numpy()
        worst = max(worst, float(np.abs(copy - o.copy).max()))
        o.edge_step(1.0)
    d.control()
    cb = d.read_control()
    out[f"parity_{name}_{program}"] = dict(worst=worst, inner_failures=cb.inner_failures, q=d.query())
    print(name, program, "worst", worst, d.query(), flush=True)

def rate(name, g, program, dtype, steps=100, warm=10, **kw):
    d = DeviceSolver(g, dtype, device=0, program=program, **kw)
    d.reset(max_it=steps + warm + 1, eps_abs=0.0, eps_rel=0.0)
    d.enqueue(warm); torch.cuda.synchronize()
    t0 = time.perf_counter(); d.enqueue(steps); torch.cuda.synchronize()
    el = time.perf_counter() - t0
    tm = d.enqueue_timed(min(steps, 50)) if steps + warm + 50 < 10**9 else None
    out[f"rate_{name}_{program}_{dtype}"] = dict(its=steps / el, ms=1e3 * el / steps, q=d.query())
    print(name, program, dtype, f"{steps / el:.1f} it/s  {1e3 * el / steps:.4f} ms/it", d.query(), flush=True)

which = sys.argv[1:] or ["parity", "rate"]
if "parity" in which:
    for nm in ("benchmark1", "benchmark4", "benchmark3"):
        case, g = load_fixture(nm)
        for prog in ("workgroup", "wavefront"):
            parity(nm, g, prog)
    parity("lat2", lattice_boxes(14, 11, seed=7), "workgroup")
    parity("lat3", lattice_boxes(6, 5, n=3, seed=1), "workgroup")
    parity("lat6", lattice_boxes(6, 5, n=6, seed=1), "workgroup")
    from conftest import star_case
    As, bs, n = star_case(40)
    parity("star40", graph_from_sets(As, bs, n), "auto", steps=4)
if "rate" in which:
    case, g4 = load_fixture("benchmark4")
    for prog in ("workgroup", "wavefront"):
        rate("benchmark4", g4, prog, "f64", steps=200, warm=20)
    for (nx, ny) in ((20, 20), (32, 32), (45, 45), (64, 64), (100, 100)):
        g = lattice_boxes(nx, ny, seed=0)
        for prog in ("workgroup", "wavefront"):
            rate(f"lat{nx}x{ny}", g, prog, "f32", steps=60, warm=5)
if "box" in which:      # box specialisation of the wavefront program against its 4-facet variant
    for (nx, ny) in ((45, 45), (100, 100), (317, 316)):
        g = lattice_boxes(nx, ny, seed=0)
        for dt in ("f32", "f64"):
            rate(f"lat{nx}x{ny}_m4", g, "wavefront", dt, steps=60, warm=5, wave_generic_rows=2)
            rate(f"lat{nx}x{ny}_box", g, "wavefront", dt, steps=60, warm=5)
if "cross" in which:    # crossover of the two programs on n = 2 box lattices (WG_AUTO_MAX in gcsadmm_create)
    for (nx, ny) in ((24, 24), (28, 28), (32, 32), (38, 38), (45, 45), (64, 64)):
        g = lattice_boxes(nx, ny, seed=0)
        for prog in ("workgroup", "wavefront"):
            rate(f"lat{nx}x{ny}", g, prog, "f32", steps=100, warm=10)
if "small" in which:
    case, g4 = load_fixture("benchmark4")
    rate("benchmark4", g4, "workgroup", "f64", steps=300, warm=20)
    for nm in ("benchmark3", "benchmark1"):
        rate(nm, load_fixture(nm)[1], "workgroup", "f64", steps=300, warm=20)
    for (nx, ny) in ((16, 16), (24, 24), (32, 32)):
        rate(f"lat{nx}x{ny}", lattice_boxes(nx, ny, seed=0), "workgroup", "f32", steps=100, warm=5)
if "s6d" in which:
    g = lattice_boxes(223, 224, n=6, seed=0)
    rate("s6d", g, "workgroup", "f32", steps=10, warm=2)
    g = lattice_boxes(60, 60, n=6, seed=0)
    rate("lat6_60x60", g, "workgroup", "f32", steps=20, warm=2)
    g = lattice_boxes(60, 60, n=3, seed=0)
    rate("lat3_60x60", g, "workgroup", "f32", steps=20, warm=2)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "wg_probe_" + "_".join(which) + os.environ.get("GCSADMM_PROBE_TAG", "") + ".json"), "w"), indent=1)
