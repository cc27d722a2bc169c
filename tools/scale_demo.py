"""End-to-end at scale with general polytopes (not boxes): random convex polygons on a jittered grid ->
device graph construction (gcs_admm_amd/scene.py) -> ADMM loop on the GPU (generic program) -> rounding.

  python tools/scale_demo.py [--side 100] [--iters 300]
"""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
from gcs_admm_amd.graph import convert_pt_to_polytope
from gcs_admm_amd.scene import graph_from_sets_device
from gcs_admm_amd.solver import DeviceSolver


def polygon_scene(side: int, seed: int = 0, m: int = 6):
    """side x side convex m-gons (regular, randomly rotated and scaled, radius ~0.75 cell) on a jittered unit grid:
    each meets its 4-8 neighbours.  's' / 't' are points inside the first / last polygon."""
    rng = np.random.default_rng(seed)
    As, bs = {}, {}
    cen = {}
    k = 0
    for j in range(side):
        for i in range(side):
            c = np.array([i, j], float) + rng.uniform(-0.1, 0.1, 2)
            r = rng.uniform(0.62, 0.8)
            th = rng.uniform(0, 2 * np.pi) + 2 * np.pi * np.arange(m) / m
            A = np.stack([np.cos(th), np.sin(th)], axis=1)
            b = A @ c + r * np.cos(np.pi / m)
            As[k] = A; bs[k] = b; cen[k] = c
            k += 1
    out_A = {'s': None, 't': None}; out_b = {}
    out_A['s'], out_b['s'] = convert_pt_to_polytope(cen[0])
    out_A['t'], out_b['t'] = convert_pt_to_polytope(cen[k - 1])
    out_A.update(As); out_b.update(bs)
    return out_A, out_b


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--side", type=int, default=100)
    ap.add_argument("--iters", type=int, default=300)
    args = ap.parse_args()
    As, bs = polygon_scene(args.side)
    t0 = time.perf_counter()
    g = graph_from_sets_device(As, bs, 2)
    t_graph = time.perf_counter() - t0
    dev = DeviceSolver(g, "f32", device=0)
    dev.reset(max_it=args.iters + 1, eps_abs=0.0, eps_rel=0.0)
    dev.enqueue(10); torch.cuda.synchronize()
    t0 = time.perf_counter()
    dev.enqueue(args.iters - 10); torch.cuda.synchronize()
    t_loop = time.perf_counter() - t0
    cb = dev.read_control()
    q = dev.query()
    print(json.dumps({"regions": g.num_vertices, "edges": g.num_edges, "facets_per_region": 6,
                      "graph_build_s": t_graph, "admm_iterations_per_sec": (args.iters - 10) / t_loop,
                      "waves": q["num_waves"], "lds_bytes_per_wave": q["lds_bytes"], "inner_failures": int(cb.inner_failures),
                      "degree_histogram": {int(a): int(b) for a, b in zip(*np.unique(np.diff(g.inc_ptr), return_counts=True))}}))


if __name__ == "__main__":
    main()
