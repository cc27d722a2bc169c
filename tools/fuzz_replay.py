"""Host replay of steps of tools/fuzz_soak.py: the oracle's (warm) vertex step, a cold tight solve of the same sub-problems as the truth, and the
host builds of both device programs (tests/hostemu), per step: worst entry of each against the truth.  No GPU.
python3 tools/fuzz_replay.py seed [seed ...]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from gcs_admm_amd import IPM_TOL
from gcs_admm_amd.graph import graph_from_sets
from oracle.oracle import Oracle
from scale_demo import polygon_scene
import test_hostemu as TW, test_hostemu_wg as TG

emu = TW.emu.__wrapped__() if hasattr(TW.emu, "__wrapped__") else None
wg = C.CDLL(os.path.join(ROOT, "tests", "hostemu", "libwgemu.so"))
if emu is None:
    emu = C.CDLL(os.path.join(ROOT, "tests", "hostemu", "libemu.so"))
for seed in map(int, sys.argv[1:]):
    rng = np.random.default_rng(100 + seed)
    As, bs = polygon_scene(5 + seed % 3, seed=seed, m=3 + seed % 5)
    g = graph_from_sets(As, bs, 2)
    o = Oracle(g, ipm_tol=IPM_TOL)
    rho = float([0.25, 1.0, 4.0][seed % 3])
    ww, wgw = TW.WarmRecords(g), TG.WarmRecords(wg, g)
    for it in range(8):
        if it >= 4:
            o.zedge += 0.05 * rng.normal(size=o.zedge.shape); o.mu += 0.02 * rng.normal(size=o.mu.shape)
        truth = Oracle(g, ipm_tol=1e-11, warm_start=False)
        truth.zedge[:] = o.zedge; truth.mu[:] = o.mu
        truth.vertex_step(rho, 1.0)
        cold = Oracle(g, ipm_tol=IPM_TOL, warm_start=False)
        cold.zedge[:] = o.zedge; cold.mu[:] = o.mu
        cold.vertex_step(rho, 1.0)
        ra = TW.emu_step(emu, g, o.zedge.copy(), o.mu.copy(), rho, 1.0, warm=ww)
        rb = TG.wg_step(wg, "wg_emu_vertex_step", g, o.zedge.copy(), o.mu.copy(), rho=rho, warm=wgw)
        o.vertex_step(rho, 1.0)
        # the host builds run the generic vertices only (s, t and no-flow vertices are closed-form on the device): compare those columns
        cols = np.repeat(np.asarray(rb[5], bool), np.diff(g.inc_ptr))
        a, b = np.where(cols, ra[0], o.copy), np.where(cols, rb[0], o.copy)
        e = lambda x: float(np.abs(x - truth.copy)[:, cols].max())
        print(f"seed {seed} step {it}: against a cold solve to 1e-11 -- oracle warm {e(o.copy):.2e}  oracle cold at ipm_tol {e(cold.copy):.2e}  wavefront emu {e(a):.2e}  workgroup emu {e(b):.2e}"
              f"   | oracle-wavefront {np.abs(o.copy - a).max():.2e}  oracle-workgroup {np.abs(o.copy - b).max():.2e}")
        o.edge_step(1.0)
