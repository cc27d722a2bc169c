"""Build the in-tree HIP library (gfx950 only):  python -m gcs_admm_amd.build"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
MAIN = os.path.join(CSRC, "gcsadmm.hip")
DIMS = os.path.join(CSRC, "gcsadmm_dims.hip")
# (source, object name, extra flags): the n = 3 / 6 kernels are one object per (dimension, state type)
UNITS = [(MAIN, "gcsadmm.o", [])] + [(DIMS, f"gcsadmm_n{n}_f{32 if f else 64}.o", [f"-DGCS_DIM={n}", f"-DGCS_F32={f}"])
                                     for n in (6, 3) for f in (0, 1)]
LP = os.path.join(CSRC, "polytope_lp.hip")      # batched tiny LPs for graph construction (own object, own dependencies)
UNITS.append((LP, "polytope_lp.o", []))
HDR = os.path.join(ROOT, "include", "gcsadmm.h")
DEPS = [MAIN, DIMS] + [os.path.join(CSRC, f) for f in ("vertex_program.h", "vertex_program.inc", "vertex_kernel.h")] + [HDR]
UNIT_DEPS = {LP: [LP, HDR, os.path.join(CSRC, "polytope_lp_core.h")]}
OUT = os.path.join(HERE, "libgcsadmm.so")


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def build(force: bool = False, verbose: bool = False) -> str:
    """Six objects compiled concurrently (the n = 6 instantiations take over a minute each), then linked."""
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in DEPS + UNIT_DEPS[LP]):
        return OUT
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
    if verbose:
        flags.insert(0, "-Rpass-analysis=kernel-resource-usage")
    objs, procs = [], []
    for src, name, extra in UNITS:
        obj = os.path.join(HERE, name)
        objs.append(obj)
        if force or not os.path.exists(obj) or any(os.path.getmtime(obj) < os.path.getmtime(d) for d in UNIT_DEPS.get(src, DEPS)):
            procs.append((name, subprocess.Popen([hipcc()] + flags + extra + ["-c", src, "-o", obj])))
    for name, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {name}")
    subprocess.check_call([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", OUT])
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
