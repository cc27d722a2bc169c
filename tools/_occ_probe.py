import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gcs_admm_amd import solver
from gcs_admm_amd.graph import lattice_boxes
for lib in sys.argv[1:]:
    solver.LIB_PATH = os.path.abspath(lib); solver._lib = None
    for n, nx in ((3, 60), (2, 40)):
        g = lattice_boxes(nx, nx, n=n, seed=0)
        d = solver.DeviceSolver(g, "f32", device=0, program="workgroup")
        d.reset(max_it=10**6, eps_abs=0.0, eps_rel=0.0); d.enqueue(3); torch.cuda.synchronize()
        t0 = time.perf_counter(); d.enqueue(20); torch.cuda.synchronize(); el = time.perf_counter() - t0
        print(os.path.basename(lib), f"n={n} {nx}x{nx}: {20 / el:8.1f} it/s", d.query()["workgroup_lds_bytes"], flush=True)
