"""What a terminal that is a region costs per iteration: the same polygon scene with point terminals and with box terminals, loop rate over
200 iterations in the body of the run (workgroup program, f64).  Under rocprofv3 --kernel-trace --stats the summary also gives the duration
of terminal_region_kernel itself.   python3 tools/region_terminal_timing.py"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from scale_demo import polygon_scene
from gcs_admm_amd.graph import graph_from_sets
from gcs_admm_amd.solver import DeviceSolver
out = {}
for side in (4, 7):
    for kind in ("points", "regions"):
        As, bs = polygon_scene(side, seed=1, m=5)
        if kind == "regions":
            A = np.vstack([np.eye(2), -np.eye(2)])
            for key in ("s", "t"):
                pt = 0.5 * (bs[key][:2] - bs[key][2:])
                As[key], bs[key] = A, np.hstack([pt + 0.35, -pt + 0.35])
        g = graph_from_sets(As, bs, 2)
        d = DeviceSolver(g, "f64", device=0)
        d.reset(max_it=100000, eps_abs=0.0, eps_rel=0.0)
        d.enqueue(100); torch.cuda.synchronize()
        t0 = time.perf_counter(); d.enqueue(200); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        deg = np.diff(g.inc_ptr)
        out[f"{side}x{side} {kind}"] = dict(vertices=g.num_vertices, edges=g.num_edges, terminal_degrees=[int(deg[g.src]), int(deg[g.dst])],
                                           iterations_per_sec=200 / dt, us_per_iteration=1e6 * dt / 200, inner_failures=int(d.read_control().inner_failures))
        print(side, kind, json.dumps(out[f"{side}x{side} {kind}"]), flush=True)
        d.close()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "region_terminal_timing.json"), "w"), indent=1)
