"""20 iterations of the overlapped partitioned loop (one rank, forced split) on a strip of config 4's lattice, for `rocprofv3 --kernel-trace`:
the trace shows the boundary and the interior launch of an iteration running at the same time.  python3 tools/overlap_trace.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from gcs_admm_amd.graph import lattice_boxes
from gcs_admm_amd.solver import DeviceSolver
g = lattice_boxes(40, 317, seed=0)
d = DeviceSolver(g, "f32", device=0, program="wavefront", columns="edge")
d.attach_comm(0, 1, d.unique_id(), {}, {})
print("boundary wavefronts", d.set_overlap(1), "of", d.query()["num_waves"])
d.reset(max_it=200, eps_abs=0.0, eps_rel=0.0)
d.enqueue_partitioned(80); torch.cuda.synchronize()
d.enqueue_partitioned(20); torch.cuda.synchronize()
