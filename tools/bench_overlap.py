"""Measurement for graph construction at scale: LPs per second of the device path
(gcs_admm_amd/scene.py -> csrc/polytope_lp.hip) on a scene of random 2-D / 3-D / 6-D polytopes, next to the CPU
restatement (oracle/polytope_oracle.py: one HiGHS LP per pair, as the reference does with MOSEK) on a bounded
sample.  One JSON line per dimension.

  python tools/bench_overlap.py [--regions 20000] [--cpu-pairs 300]
"""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (HIP runtime first, see solver.load_library)
from gcs_admm_amd.scene import PolytopeScene, candidate_pairs
from oracle import polytope_oracle as PO


def scene_polys(rng, n, P, m_extra):
    side = P ** (1.0 / min(n, 2))          # regions spread over a 2-D sheet so that each meets a handful of others
    polys = []
    for _ in range(P):
        c = np.zeros(n); c[:2] = rng.uniform(0, side, min(n, 2))
        A = rng.normal(size=(m_extra, n)); A /= np.linalg.norm(A, axis=1)[:, None]
        b = A @ c + rng.uniform(0.5, 1.1, size=m_extra)
        A = np.vstack([A, np.eye(n), -np.eye(n)]); b = np.hstack([b, c + 1.5, -(c - 1.5)])
        polys.append((A, b))
    return polys


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--regions", type=int, default=20000)
    ap.add_argument("--cpu-pairs", type=int, default=300)
    args = ap.parse_args()
    for n in (2, 3, 6):
        rng = np.random.default_rng(n)
        polys = scene_polys(rng, n, args.regions, 3 + n)
        scene = PolytopeScene(polys)
        scene.centers()                                       # warm-up (module load, allocations)
        t0 = time.perf_counter(); cen, rad, _ = scene.centers(); t_c = time.perf_counter() - t0
        t0 = time.perf_counter(); lo, hi, _ = scene.bounds(cen); t_b = time.perf_counter() - t0
        t0 = time.perf_counter(); pa, pb = candidate_pairs(lo, hi); t_s = time.perf_counter() - t0
        t0 = time.perf_counter(); flags, st = scene.overlaps(pa, pb, 1e-9, cen); t_o = time.perf_counter() - t0
        k = min(args.cpu_pairs, len(pa))
        sel = rng.choice(len(pa), k, replace=False)
        t0 = time.perf_counter()
        ref = [PO.overlap(polys[pa[t]][0], polys[pa[t]][1], polys[pb[t]][0], polys[pb[t]][1]) for t in sel]
        t_cpu = time.perf_counter() - t0
        agree = int(sum(bool(flags[t]) == r for t, r in zip(sel, ref)))
        print(json.dumps({
            "metric": "overlap_LPs_per_sec", "n": n, "regions": args.regions, "rows_per_region": int(polys[0][0].shape[0]),
            "candidate_pairs": int(len(pa)), "overlapping": int(flags.sum()),
            "device_overlap_LPs_per_sec": len(pa) / t_o, "device_centre_LPs_per_sec": args.regions / t_c,
            "device_bound_LPs_per_sec": args.regions * 2 * n / t_b, "host_sweep_s": t_s,
            "early_exit_share": float(np.mean(st > 0)),
            "note": "device times include the host<->device copies of the scene (set-up API with host pointers)",
            "cpu_baseline": {"value": k / t_cpu, "unit": "LPs/s", "cores": 1, "kind": "port",
                             "sample": f"{k} of the candidate pairs, scipy/HiGHS feasibility LP each (oracle/polytope_oracle.py)",
                             "agreement_on_sample": f"{agree}/{k}"}}), flush=True)


if __name__ == "__main__":
    main()
