"""Long run of the overlapped partitioned loop against the serial one (one rank, forced split, 2 000 iterations entered in odd chunks): the
same bits.  python tools/overlap_soak.py"""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from gcs_admm_amd.graph import lattice_boxes
from gcs_admm_amd.solver import DeviceSolver
g = lattice_boxes(40, 317, seed=0)
res = {}
for mode in (2, 1):
    d = DeviceSolver(g, "f32", device=0, program="wavefront", columns="edge")
    d.attach_comm(0, 1, d.unique_id(), {}, {})
    nb = d.set_overlap(mode)
    d.reset(max_it=3000, eps_abs=0.0, eps_rel=0.0)
    for chunk in (1, 7, 100, 392, 1000, 500):      # odd chunk sizes: the loop is entered and left many times
        d.enqueue_partitioned(chunk)
        cb = d.read_control()
    assert cb.it == 2001 and cb.inner_failures == 0, (cb.it, cb.status)
    res[mode] = [t.cpu().numpy().copy() for t in (d.trace[:2000], d.copy, d.mu, d.zedge, d.yv)]
    print("mode", mode, "boundary", nb, "it", cb.it)
    d.close()
print("bitwise equal:", all(np.array_equal(a, b) for a, b in zip(res[2], res[1])))
