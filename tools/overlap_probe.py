"""Cost of the overlapped schedule's SPLIT on one GPU (the gain -- the exchange hidden behind the interior launch -- needs neighbours): a strip
of config 4's lattice as at N = 8 (12.6 k vertices) as a one-rank partition, gcsadmm_run_partitioned in its serial form and with the split
forced (first quarter of the wavefronts as boundary, second stream, events; nothing to exchange).  it/s over iterations 61-160."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from gcs_admm_amd.graph import lattice_boxes
from gcs_admm_amd.solver import DeviceSolver
out = {}
for name, g in (("strip of N = 8 (40 x 317)", lattice_boxes(40, 317, seed=0)), ("strip of N = 2 (158 x 317)", lattice_boxes(158, 317, seed=0))):
    row = {"V": g.num_vertices}
    for label, mode in (("serial", 2), ("split forced", 1), ("serial again", 2)):
        d = DeviceSolver(g, "f32", device=0, program="wavefront", columns="edge")
        d.attach_comm(0, 1, d.unique_id(), {}, {})
        nb = d.set_overlap(mode)
        el = min(bench.time_window(d, 60, 5, 100, enqueue=d.enqueue_partitioned) for _ in range(2))
        row[label] = {"boundary_wavefronts": nb, "wavefronts": d.query()["num_waves"], "iterations_per_sec": 100 / el}
        d.close()
    out[name] = row
print(json.dumps(out))
