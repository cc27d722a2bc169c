"""How much of a wavefront's run is waiting for its slowest vertex?  Wavefront program, lattices: Newton iterations per wavefront (= of its
slowest vertex) against Newton iterations per vertex, over 10 single iterations in the body of a run.   python3 tools/pack_probe.py"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from gcs_admm_amd.graph import lattice_boxes
from gcs_admm_amd.solver import DeviceSolver
out = {}
for name, (r, c), start in (("s10k", (100, 100), 150), ("s100k", (316, 317), 60)):
    g = lattice_boxes(r, c, seed=0)
    d = DeviceSolver(g, "f32", device=0, program="wavefront", columns="edge")
    d.reset(max_it=10000, eps_abs=0.0, eps_rel=0.0)
    d.enqueue(start); torch.cuda.synchronize()
    nw = d.query()["num_waves"]; ngen = g.num_vertices - d.query()["num_special"]
    per_wave, per_vtx, wave_max, hist = [], [], [], np.zeros(64)
    for _ in range(10):
        d.enqueue(1); torch.cuda.synchronize()
        u = d.unit_iterations(); cb = d.read_control()
        per_wave.append(float(u.mean())); wave_max.append(int(u.max())); per_vtx.append(cb.inner_iters / ngen)
        hist += np.bincount(np.minimum(u, 63), minlength=64)
    out[name] = dict(wavefronts=nw, vertices=ngen, newton_per_vertex=float(np.mean(per_vtx)), newton_per_wavefront=float(np.mean(per_wave)),
                     slowest_wavefront=float(np.mean(wave_max)), wavefront_histogram={int(k): int(v) for k, v in enumerate(hist) if v})
    print(name, json.dumps(out[name]), flush=True)
    d.close()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "pack_probe.json"), "w"))
