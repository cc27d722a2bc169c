import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gcs_admm_amd.cases import load_fixture
from gcs_admm_amd.solver import DeviceSolver
g = load_fixture("benchmark4")[1]
d = DeviceSolver(g, "f64", device=0)
for steps, chunk in ((200, 200), (1000, 1000), (1000, 250), (1000, 100), (2000, 2000), (2000, 250)):
    d.reset(max_it=10**6, eps_abs=0.0, eps_rel=0.0)
    d.enqueue(20); torch.cuda.synchronize()
    t0 = time.perf_counter()
    left = steps
    while left > 0:
        k = min(chunk, left); d.enqueue(k); left -= k
        if left > 0: torch.cuda.synchronize()
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(f"steps {steps} chunk {chunk}: {steps / el:8.1f} it/s  ms/step {1e3 * el / steps:.4f}  (enqueue returned after {1e3 * t_enq:.1f} ms of {1e3 * el:.1f})", flush=True)
