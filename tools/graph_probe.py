"""Does a captured HIP graph of the loop's launches shorten the gaps between dependent kernels?  (development probe)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gcs_admm_amd.cases import load_fixture
from gcs_admm_amd.solver import DeviceSolver
g = load_fixture("benchmark4")[1]
d = DeviceSolver(g, "f64", device=0)
K, REP = 50, 8
def run_plain():
    d.reset(max_it=10**6, eps_abs=0.0, eps_rel=0.0); d.enqueue(20); torch.cuda.synchronize()
    t0 = time.perf_counter(); d.enqueue(K * REP); torch.cuda.synchronize(); return time.perf_counter() - t0
def run_graph():
    d.reset(max_it=10**6, eps_abs=0.0, eps_rel=0.0); d.enqueue(20); torch.cuda.synchronize()
    s = torch.cuda.Stream()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        d.enqueue(2); torch.cuda.synchronize()
        with torch.cuda.graph(gr, stream=s):
            d.enqueue(K)
    torch.cuda.synchronize()
    d.reset(max_it=10**6, eps_abs=0.0, eps_rel=0.0); d.enqueue(20); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(REP): gr.replay()
    torch.cuda.synchronize(); return time.perf_counter() - t0
for name, fn in (("plain", run_plain), ("graph", run_graph), ("plain", run_plain), ("graph", run_graph)):
    el = fn(); cb = d.read_control()
    print(f"{name}: {K * REP / el:8.1f} it/s  {1e3 * el / (K * REP):.4f} ms/it   it={cb.it}", flush=True)
