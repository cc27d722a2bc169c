import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gcs_admm_amd import solver
if len(sys.argv) > 1: solver.LIB_PATH = os.path.abspath(sys.argv[1])
from gcs_admm_amd.graph import lattice_boxes
for (nx, ny, n) in ((16, 16, 6), (6, 5, 6), (16, 16, 2)):
    g = lattice_boxes(nx, ny, n=n, seed=0)
    d = solver.DeviceSolver(g, "f32", device=0, program="workgroup")
    d.reset(max_it=1000, eps_abs=0.0, eps_rel=0.0)
    prev = 0
    for it in range(12):
        d.enqueue(1); torch.cuda.synchronize()
        cb = d.read_control()
        cnt = d.counters.cpu().numpy() if hasattr(d, "counters") else None
        print(nx, ny, n, "it", it, "inner_failures", cb.inner_failures, "inner_iterations", cb.inner_iters, flush=True)
