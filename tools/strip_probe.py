"""What one rank of the sharded loop does at N = 1, 2, 4, 8, measured on ONE GPU: BASELINE config 4's lattice (316 x 317) is split into
N row strips, every strip gets its own handle, and the handles run the partitioned iteration in lock step (vertex steps, halo columns
copied between the handles on the device, edge steps, the five sums added up, control) -- the emulation of
tests/test_gpu_configs.py::test_lattice_100k_eight_partitions_match_single.  Handles run one after the other, so each strip's kernels have
the chip to themselves, as they would on a GPU of their own; every stage of every strip is bracketed by events.

Reported per N: the slowest strip's vertex step, edge step, halo pack / unpack kernels and control kernel (us per iteration, window
iterations 61-80), i.e. everything a rank does between two collectives.  What a single GPU cannot measure -- the RCCL send/recv of
~25 KB per boundary and the 48-byte all-reduce over xGMI -- enters the model at the bottom as a stated assumption.

  python tools/strip_probe.py > gpurun_out/strip_probe.json
"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from gcs_admm_amd.graph import lattice_boxes
from gcs_admm_amd.partition import build_partition, strip_owner
from gcs_admm_amd.solver import DeviceSolver

FIRST, STEPS = 60, 20
g = lattice_boxes(316, 317, seed=0)
out = {"workload": "s100k (316 x 317 lattice, f32 state, edge-major columns)", "V": g.num_vertices, "E": g.num_edges,
       "window": {"first_iteration": FIRST + 1, "last_iteration": FIRST + STEPS}, "strips": {}}


def timed(fn):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); fn(); b.record()
    return a, b


for world in (1, 2, 4, 8):
    owner = strip_owner(g, world)
    parts = [build_partition(g, owner, r, world) for r in range(world)]
    devs = [DeviceSolver(p.graph, "f32", device=0, num_incidences=p.num_incidences, inc_counted=p.inc_counted, edge_counted=p.edge_counted,
                         nx_global=p.nx_global, nmu_global=p.nmu_global, columns="edge") for p in parts]
    for r, d in enumerate(devs):
        d.reset(max_it=FIRST + STEPS + 5, eps_abs=0.0, eps_rel=0.0)
        d.attach_comm(r, world, None, parts[r].send_idx, parts[r].recv_idx)      # halo lists without a communicator: the pack / unpack kernels
    # ghost columns <- the neighbour's copies (device to device, in state-column numbering)
    idx = {(r, o): (torch.as_tensor(devs[r].col_of[parts[r].recv_idx[o]], device="cuda"), torch.as_tensor(devs[o].col_of[parts[o].send_idx[r]], device="cuda"))
           for r in range(world) for o in parts[r].recv_idx}
    ev = {k: [[] for _ in range(world)] for k in ("vertex", "pack", "unpack", "edge", "control")}
    for it in range(FIRST + STEPS):
        rec = it >= FIRST
        for r, d in enumerate(devs):
            e = timed(d.vertex_step)
            p = timed(d.halo_pack)
            if rec:
                ev["vertex"][r].append(e); ev["pack"][r].append(p)
        for (r, o), (rix, six) in idx.items():
            devs[r].copy.index_copy_(1, rix, devs[o].copy.index_select(1, six))
        tot = torch.zeros(5, dtype=torch.float64, device="cuda")
        for r, d in enumerate(devs):
            u = timed(d.halo_unpack)      # (its input buffer is stale here: the kernel's time is what is measured; the ghosts were set above)
            for (rr, o), (rix, six) in idx.items():
                if rr == r:
                    d.copy.index_copy_(1, rix, devs[o].copy.index_select(1, six))
            box = {}
            e = timed(lambda: box.setdefault("s", d.edge_step()))
            tot += box["s"]
            if rec:
                ev["unpack"][r].append(u); ev["edge"][r].append(e)
        for r, d in enumerate(devs):
            c = timed(lambda: d.control(tot))
            if rec:
                ev["control"][r].append(c)
    torch.cuda.synchronize()
    us = {k: np.array([[1e3 * a.elapsed_time(b) for a, b in lst] for lst in v]) for k, v in ev.items()}      # [rank][iteration]
    cb = [d.read_control() for d in devs]
    assert len({c.it for c in cb}) == 1 and all(c.inner_failures == 0 for c in cb)
    row = {"ranks": world, "vertices_per_strip": [p.graph.num_vertices for p in parts],
           "wavefronts_per_strip": [d.query()["num_waves"] for d in devs],
           "cut_columns_per_strip": [int(sum(len(v) for v in p.send_idx.values())) for p in parts],
           "newton_iterations_per_vertex": float(np.mean([c.inner_iters / max(p.graph.num_vertices, 1) for c, p in zip(cb, parts)]))}
    for k, a in us.items():
        row[k + "_us"] = {"slowest_strip_mean": float(a.mean(1).max()), "mean_over_strips": float(a.mean()),
                          "per_iteration_max_over_strips_mean": float(a.max(0).mean())}
    # what a rank does between collectives, per iteration: the slowest strip decides
    row["compute_us_per_iteration"] = float((us["vertex"] + us["pack"]).max(0).mean() + (us["unpack"] + us["edge"]).max(0).mean() + us["control"].max(0).mean())
    out["strips"][str(world)] = row
    for d in devs:
        d.close()
    del devs
    torch.cuda.empty_cache()
# model of the iteration at N ranks: measured compute + assumed communication
COMM = {"halo_send_recv_us": 20.0, "all_reduce_us": 20.0, "launch_gaps_us": 15.0,
        "note": "ASSUMED (one GPU per box here): an RCCL grouped send/recv of ~25 KB and a 48-byte all-reduce over xGMI are latency-bound, "
                "about 20 us each; five launches + two collectives per iteration leave ~15 us of gaps"}
base = out["strips"]["1"]["compute_us_per_iteration"]
out["model"] = {"assumptions": COMM, "rows": {}}
for w, row in out["strips"].items():
    extra = 0.0 if w == "1" else COMM["halo_send_recv_us"] + COMM["all_reduce_us"]
    t = row["compute_us_per_iteration"] + COMM["launch_gaps_us"] + extra
    out["model"]["rows"][w] = {"iteration_us": t, "iterations_per_sec": 1e6 / t, "speedup_vs_1": (base + COMM["launch_gaps_us"]) / t}
print(json.dumps(out))
