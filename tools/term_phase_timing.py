"""Diagnostic: s_memtime ticks (shader clock, ~2.4 GHz on this part: 35 k ticks = 15 us) per phase of the region-terminal solve (csrc/terminal_region.h), workgroup 0, from the timing build
(python -m gcs_admm_amd.build --timing).   python3 tools/term_phase_timing.py"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from gcs_admm_amd import solver
solver.LIB_PATH = os.path.join(ROOT, "gcs_admm_amd", "libgcsadmm_timing.so")
from scale_demo import polygon_scene
from gcs_admm_amd.graph import graph_from_sets
NAMES = {0: "slacks + reduction", 1: "stop test + cone scaling (thread 0)", 2: "block Hessians", 3: "Cholesky", 4: "X, S", 5: "right-hand sides", 6: "X' rhs, H^-1 rhs",
         7: "z + small system (thread 0)", 8: "dp, ds, du, dt", 10: "rows of the predictor", 11: "sigma + cone multipliers (thread 0)", 12: "kappa + rows of the corrector",
         13: "step length (thread 0) + update"}
As, bs = polygon_scene(4, seed=1, m=5)
A = np.vstack([np.eye(2), -np.eye(2)])
for key in ("s", "t"):
    pt = 0.5 * (bs[key][:2] - bs[key][2:])
    As[key], bs[key] = A, np.hstack([pt + 0.35, -pt + 0.35])
g = graph_from_sets(As, bs, 2)
d = solver.DeviceSolver(g, "f64", device=0)
d.reset(max_it=100000, eps_abs=0.0, eps_rel=0.0)
steps = 200
d.enqueue(steps); torch.cuda.synchronize()
cyc = (C.c_ulonglong * 32)(); cnt = (C.c_ulonglong * 32)()
assert d.lib.gcsadmm_debug_term_cycles(cyc, cnt) == 0
c, n = np.array(list(cyc), float), np.array(list(cnt), float)
tot = c.sum()
print(f"{steps} solves of terminal 0: {tot / steps:.0f} ticks per solve, {n[0] / steps:.1f} Newton iterations + the final stop test")
for k in sorted(NAMES):
    if n[k] > 0:
        print(f"  {k:2d} {NAMES[k]:40s} {100 * c[k] / tot:5.1f} %   {c[k] / n[k]:7.1f} ticks/visit  x{n[k] / steps:.1f} per solve")
