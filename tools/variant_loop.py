"""The timed loop of a workload on a given build of the library, for rocprofv3: python3 tools/variant_loop.py <lib.so> <workload> [steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gcs_admm_amd import solver
solver.LIB_PATH = os.path.abspath(sys.argv[1])
import bench
wl = sys.argv[2]; steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
g, dtype, _ = bench.make_workload(wl)
dev = solver.DeviceSolver(g, dtype, device=0, columns="edge" if g.num_edges >= 20000 else "incidence")
print(steps / bench.time_window(dev, bench.window_start(wl, 3, steps), 3, steps))
