"""Diagnostic: vertex-kernel time of library variants (e.g. builds with -DGCS_DUP=k, which execute one part of the
border factorisation twice; the time added is that part's cost in place).
usage: python tools/variant_time.py lib1.so [lib2.so ...]   (one subprocess per library)"""
import sys, os, subprocess, json
if len(sys.argv) > 2 or (len(sys.argv) == 2 and sys.argv[1] == "--all"):
    libs = sys.argv[1:] if sys.argv[1] != "--all" else sorted(p for p in os.listdir("gcs_admm_amd") if p.startswith("libgcsadmm_dup"))
    for lib in libs:
        r = subprocess.run([sys.executable, __file__, lib if os.path.sep in lib else os.path.join("gcs_admm_amd", lib)], capture_output=True, text=True)
        print(r.stdout.strip() or r.stderr.strip()[-300:], flush=True)
    sys.exit(0)
sys.path.insert(0, os.getcwd())
import torch
from gcs_admm_amd import solver
solver.LIB_PATH = os.path.abspath(sys.argv[1])
from gcs_admm_amd.graph import lattice_boxes
from gcs_admm_amd.cases import load_fixture
out = {}
for wl in ("s10k", "benchmark4"):
    g = lattice_boxes(100, 100, seed=0) if wl == "s10k" else load_fixture("benchmark4")[1]
    d = solver.DeviceSolver(g, "f32" if wl == "s10k" else "f64", device=0)
    d.reset(max_it=100000, eps_abs=0.0, eps_rel=0.0)
    d.enqueue(10)
    tm = d.enqueue_timed(100)
    out[wl] = round(tm["vertex_ms"] / tm["vertex_launches"], 4)
print(os.path.basename(sys.argv[1]), json.dumps(out))
