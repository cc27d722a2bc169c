"""Diagnostic: cycles per parallel region of the workgroup vertex program (vertex_wg.h), workgroup 0 (the heaviest vertex),
from the -DGCS_WG_TIMING build (python -m gcs_admm_amd.build --timing).  Read shares and per-visit cycles."""
import ctypes as C, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from gcs_admm_amd import solver
solver.LIB_PATH = os.path.abspath(os.environ.get("GCSADMM_TIMING_LIB", os.path.join(ROOT, "gcs_admm_amd", "libgcsadmm_timing.so")))
from gcs_admm_amd.cases import load_fixture
from gcs_admm_amd.graph import lattice_boxes

NAMES = {0: "load+init", 1: "rows A + reduce", 2: "assembly + cone + g0", 3: "block chol", 4: "block inverse", 5: "B X", 6: "side sums",
         7: "side chol", 9: "side inverse + Y", 10: "M assembly", 11: "M chol", 12: "M inverse", 13: "affine solve (sum of 30-39)",
         14: "pass B + reduce", 15: "kappa + cone", 16: "G'kappa", 17: "G sums", 18: "corrector solve (sum of 30-39)",
         19: "pass D + reduce", 20: "alpha", 21: "update", 22: "exit",
         30: " s: t_e", 31: " s: side sums", 32: " s: v", 33: " s: rhs", 34: " s: Minv rhs", 35: " s: unpack", 36: " s: w", 37: " s: dnu", 38: " s: r_e", 39: " s: dw"}
res = {}
for wl in sys.argv[1:] or ["benchmark4", "lat6"]:
    if wl == "benchmark4":
        g, dt = load_fixture("benchmark4")[1], "f64"
    elif wl == "lat6":
        g, dt = lattice_boxes(16, 16, n=6, seed=0), "f32"
    elif wl == "lat6big":       # many workgroups per CU: the regions' times under contention (issue-bound, profiles/r03)
        g, dt = lattice_boxes(100, 100, n=6, seed=0), "f32"
    elif wl == "lat2":
        g, dt = lattice_boxes(16, 16, seed=0), "f32"
    elif wl == "lat2small":     # 146 vertices: the 512-thread object (the stamped one)
        g, dt = lattice_boxes(12, 12, seed=0), "f64"
    elif wl.startswith("benchmark"):
        g, dt = load_fixture(wl)[1], "f64"
    d = solver.DeviceSolver(g, dt, device=0, program="workgroup")
    d.reset(max_it=1000, eps_abs=0.0, eps_rel=0.0)
    d.enqueue(10); torch.cuda.synchronize()
    cyc = (C.c_ulonglong * 64)(); cnt = (C.c_ulonglong * 64)()
    d.lib.gcsadmm_debug_wg_cycles(cyc, cnt)
    c0, n0 = np.array(list(cyc), float), np.array(list(cnt), float)
    steps = 20
    d.enqueue(steps); torch.cuda.synchronize()
    d.lib.gcsadmm_debug_wg_cycles(cyc, cnt)
    c1, n1 = np.array(list(cyc), float) - c0, np.array(list(cnt), float) - n0
    main = [k for k in NAMES if k < 30]
    tot = sum(c1[k] for k in main if k not in (13, 18)) + c1[30:40].sum()
    iters = n1[21] / steps
    print(f"{wl}: workgroup 0, {iters:.1f} Newton iterations per solve, {tot / steps:.0f} s_memtime ticks per solve")
    for k in sorted(NAMES):
        if n1[k] > 0:
            print(f"  {k:2d} {NAMES[k]:32s} {100 * c1[k] / tot:6.2f} %   {c1[k] / n1[k]:9.1f} ticks/visit  x{n1[k] / steps:.1f}")
    res[wl] = dict(cycles=c1.tolist(), counts=n1.tolist(), steps=steps)
    if hasattr(d.lib, "gcsadmm_debug_wg_wave_cycles"):      # per wavefront: arrival at the closing barrier of a few regions, ticks since the region's start
        wc = (C.c_ulonglong * 512)(); d.lib.gcsadmm_debug_wg_wave_cycles(wc)
        wv = np.array(list(wc), float).reshape(64, 8)
        for k in np.nonzero(wv.sum(1))[0]:
            visits = max(n0[k] + n1[k], 1)
            print(f"  region {k:2d} ({NAMES.get(int(k), '?').strip()}): arrival of wavefronts 0..7 at its barrier, ticks/visit: " + " ".join(f"{x / visits:7.0f}" for x in wv[k]))
    if hasattr(d.lib, "gcsadmm_debug_wg_blocks"):      # whole-solve ticks per workgroup: which vertex ends the launch?
        bt = (C.c_ulonglong * 64)(); bi = (C.c_ulonglong * 64)()
        d.lib.gcsadmm_debug_wg_blocks(bt, bi)
        nb = min(64, d.query()["num_workgroup_vertices"])
        tot_steps = steps + 10
        bt, bi = np.array(list(bt), float)[:nb] / tot_steps, np.array(list(bi), float)[:nb] / tot_steps
        deg = np.diff(g.inc_ptr); fac = np.diff(g.poly_ptr)
        order = np.argsort(-bt)
        print(f"  whole solve, ticks per launch (mean over {tot_steps} launches since the handle was created): max {bt.max():.0f}, median {np.median(bt):.0f}, workgroup 0 {bt[0]:.0f}")
        for b in order[:6]:
            print(f"    workgroup {b:2d}: {bt[b]:9.0f} ticks  {bi[b]:5.1f} Newton iterations  ({bt[b] / max(bi[b], 1):7.0f} ticks each)")
        res[wl]["block_ticks"] = bt.tolist(); res[wl]["block_iters"] = bi.tolist()
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "wg_phase_timing.json"), "w"))
