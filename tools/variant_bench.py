"""A/B timing of a variant build of the library: python tools/variant_bench.py <path to .so> <workload> [steps] -- the timed window of bench.py on that library."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gcs_admm_amd import solver
solver.LIB_PATH = os.path.abspath(sys.argv[1])
import bench
wl = sys.argv[2]; steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
g, dtype, _ = bench.make_workload(wl)
columns = "edge" if g.num_edges >= 20000 else "incidence"
dev = solver.DeviceSolver(g, dtype, device=0, columns=columns)
first = bench.window_start(wl, 3, steps)
el = min(bench.time_window(dev, first, 3, steps) for _ in range(2))
cb = dev.read_control()
print(json.dumps({"lib": os.path.basename(sys.argv[1]), "workload": wl, "iterations_per_sec": steps / el, "inner_iters": cb.inner_iters, "inner_failures": cb.inner_failures}))
