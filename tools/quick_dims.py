"""Quick timing of the loop on lattices of other space dimensions / sizes (diagnostic)."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from gcs_admm_amd.graph import lattice_boxes
from gcs_admm_amd.solver import DeviceSolver
for (nx, ny, n, dt, steps) in ((50, 50, 6, "f32", 10), (223, 224, 6, "f32", 5), (316, 317, 2, "f32", 30), (100, 100, 3, "f32", 30)):
    t0 = time.time(); g = lattice_boxes(nx, ny, n=n, seed=0); tg = time.time() - t0
    d = DeviceSolver(g, dt, device=0)
    d.reset(max_it=1000, eps_abs=0.0, eps_rel=0.0)
    d.enqueue(2); torch.cuda.synchronize()
    t0 = time.time(); d.enqueue(steps); torch.cuda.synchronize(); el = time.time() - t0
    cb = d.read_control(); q = d.query()
    print(f"n={n} {nx}x{ny} V={g.num_vertices} E={g.num_edges} graph build {tg:.1f}s: {1e3*el/steps:.2f} ms/iteration, {steps/el:.1f} it/s, waves {q['num_waves']} lds {q['lds_bytes']} inner fails {cb.inner_failures} inner iters/vertex {cb.inner_iters/max(g.num_vertices-2,1):.1f}", flush=True)
