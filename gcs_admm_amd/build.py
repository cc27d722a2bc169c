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
WG = os.path.join(CSRC, "vertex_wg.hip")        # workgroup-cooperative vertex program (n = 2, 3, 6; own object)
WGD = os.path.join(CSRC, "vertex_wg_dims.hip")  # the same program for n = 1, 4, 5 (own object: the builds run in parallel)
LP = os.path.join(CSRC, "polytope_lp.hip")      # batched tiny LPs for graph construction (own object, own dependencies)
TERM = os.path.join(CSRC, "terminal_region.hip")  # x-update of terminals that are regions (own object)
# (source, object name, extra flags)
# vertex_wg.hip is built twice: 256 threads per workgroup, and 512 for launches of at most one workgroup per CU (own namespace and entry points)
T512 = ["-DGCS_WG_THREADS=512", "-Dgcs_wg=gcs_wg_t512", "-DGCS_WG_SYM(name)=name##_t512"]
UNITS = [(MAIN, "gcsadmm.o", []), (WG, "vertex_wg.o", []), (WG, "vertex_wg_t512.o", T512), (WGD, "vertex_wg_dims.o", []), (WGD, "vertex_wg_dims_t512.o", T512),
         (LP, "polytope_lp.o", []), (TERM, "terminal_region.o", [])]
HDR = os.path.join(ROOT, "include", "gcsadmm.h")
_c = lambda *names: [os.path.join(CSRC, f) for f in names]
DEPS = [MAIN, HDR] + _c("vertex_program.h", "vertex_program.inc", "vertex_kernel.h", "special_vertex.h", "vertex_wg_launch.h", "canonical_box.h", "warm_start.h", "terminal_launch.h")
UNIT_DEPS = {LP: [LP, HDR] + _c("polytope_lp_core.h"),
             WG: [WG, HDR] + _c("vertex_wg.h", "vertex_wg_kernel.h", "vertex_wg_launch.h", "special_vertex.h", "gcs_math.h", "warm_start.h")}
UNIT_DEPS[WGD] = [WGD] + UNIT_DEPS[WG][1:]
UNIT_DEPS[TERM] = [TERM, HDR] + _c("terminal_region.h", "terminal_launch.h", "gcs_math.h")
OUT = os.path.join(HERE, "libgcsadmm.so")


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def build(force: bool = False, verbose: bool = False) -> str:
    """The objects are compiled concurrently, then linked."""
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in DEPS + UNIT_DEPS[LP] + UNIT_DEPS[WG] + UNIT_DEPS[WGD] + UNIT_DEPS[TERM]):
        return OUT
    # the compiler's per-kernel resource remarks (registers, scratch, LDS) are kept next to each object: kernel_resources()
    flags = ["-Rpass-analysis=kernel-resource-usage", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
             "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
    objs, procs = [], []
    for src, name, extra in UNITS:
        obj = os.path.join(HERE, name)
        objs.append(obj)
        if force or not os.path.exists(obj) or not os.path.exists(obj + ".resources.txt") or \
                any(os.path.getmtime(obj) < os.path.getmtime(d) for d in UNIT_DEPS.get(src, DEPS)):
            log = open(obj + ".resources.txt", "w")
            procs.append((name, subprocess.Popen([hipcc()] + flags + extra + ["-c", src, "-o", obj], stderr=log), log))
    for name, p, log in procs:
        rc = p.wait()
        log.close()
        text = open(os.path.join(HERE, name) + ".resources.txt").read()
        if verbose or rc != 0:
            sys.stderr.write(text)
        if rc != 0:
            os.remove(os.path.join(HERE, name) + ".resources.txt")
            raise RuntimeError(f"hipcc failed on {name}")
    subprocess.check_call([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", OUT])
    return OUT


def kernel_resources() -> dict:
    """{kernel name: {"vgprs": .., "agprs": .., "scratch": bytes per lane, "lds": static bytes}} of the last build, from the
    compiler's resource remarks."""
    import re
    build()
    out = {}
    for _, name, _ in UNITS:
        cur = None
        for line in open(os.path.join(HERE, name) + ".resources.txt"):
            m = re.search(r"Function Name: (\S+)", line)
            if m:
                cur = out.setdefault(m.group(1), {})
                continue
            if cur is None:
                continue
            for key, pat in (("vgprs", r" VGPRs: (\d+)"), ("agprs", r"AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                             ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
                m = re.search(pat, line)
                if m:
                    cur[key] = int(m.group(1))
    return out


def build_timing() -> str:
    """Diagnostic library (tools/wg_phase_timing.py): the workgroup program with its region stamps compiled in -- the 512-thread object
    (the one small graphs such as benchmark4 run) and, under its own symbol names, nothing else: the 256-thread object stays the product's."""
    out = os.path.join(HERE, "libgcsadmm_timing.so")
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
    build()
    obj = os.path.join(HERE, "vertex_wg_t512_timing.o")
    subprocess.check_call([hipcc()] + flags + T512 + ["-DGCS_WG_TIMING", "-c", WG, "-o", obj])
    tobj = os.path.join(HERE, "terminal_region_timing.o")       # the region-terminal solve with its phase stamps (tools/term_phase_timing.py)
    subprocess.check_call([hipcc()] + flags + ["-DGCS_TERM_TIMING", "-c", TERM, "-o", tobj])
    objs = [os.path.join(HERE, n) for n in ("gcsadmm.o", "polytope_lp.o", "vertex_wg.o", "vertex_wg_dims.o", "vertex_wg_dims_t512.o")] + [obj, tobj]
    subprocess.check_call([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out])
    return out


if __name__ == "__main__":
    if "--timing" in sys.argv:
        print(build_timing())
        sys.exit(0)
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
