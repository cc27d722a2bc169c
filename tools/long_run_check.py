"""Diagnostic: long runs of the synthetic lattices -- inner failures, finiteness, trace vs the CPU oracle."""
import sys, os, time
sys.path.insert(0, os.getcwd())
from gcs_admm_amd import IPM_TOL  # noqa: E402
import numpy as np, torch
from gcs_admm_amd.graph import lattice_boxes
from gcs_admm_amd.solver import DeviceSolver
from oracle.oracle import Oracle
g = lattice_boxes(100, 100, seed=0)
for dt in ("f64", "f32"):
    d = DeviceSolver(g, dt, device=0)
    t0 = time.time(); res = d.solve(max_it=1000, chunk=100); el = time.time() - t0
    print(dt, "s10k defaults: stop at", res["iterations"], res["status"], "cost %.6f" % res["cost"], "inner failures", res["inner_failures"],
          "wall %.2fs" % el, "pri/dual last", res["pri_res_seq"][-1], res["dual_res_seq"][-1], flush=True)
    if dt == "f64":
        ref = res
o = Oracle(g, ipm_tol=IPM_TOL)
t0 = time.time(); r = o.run(max_it=120, eps_abs=0.0, eps_rel=0.0, nthreads=32); print("oracle 120 its %.1fs" % (time.time() - t0))
k = 121
print("max rel diff pri (f64 GPU vs oracle, 120 its): %.3e" % np.max(np.abs(ref["pri_res_seq"][:k] - r["pri_res_seq"][:k]) / (1e-6 + r["pri_res_seq"][:k])))
print("max rel diff dual: %.3e" % np.max(np.abs(ref["dual_res_seq"][:k] - r["dual_res_seq"][:k]) / (1e-6 + r["dual_res_seq"][:k])))
g2 = lattice_boxes(316, 317, seed=0)
d = DeviceSolver(g2, "f32", device=0)
t0 = time.time(); res = d.solve(max_it=300, chunk=100, eps_abs=0.0, eps_rel=0.0); el = time.time() - t0
print("s100k f32 300 its: inner failures", res["inner_failures"], "wall %.2fs" % el, "finite", np.isfinite(res["pri_res_seq"]).all(), "pri last", res["pri_res_seq"][-1])
