"""Properties of the compiled kernels that the design depends on, read from the compiler's resource remarks of the in-tree build
(no GPU needed: hipcc cross-compiles gfx950)."""


def test_no_kernel_uses_scratch():
    """Every kernel of the library fits the register file: no scratch (spills).  Beyond the cost (round 1: 760 B / lane made the
    vertex kernel HBM-bound on its own spills), this compiler places a spill next to a divergent branch under a partial EXEC mask
    (seen in the diagnostic timing build of the n = 6 workgroup program: every solve failed), so scratch is a correctness risk."""
    from gcs_admm_amd import build
    res = build.kernel_resources()
    names = " ".join(res)
    for must in ("vertex_wg_kernelILi2E", "vertex_wg_kernelILi3E", "vertex_wg_kernelILi6E", "vertex_prox_kernelILi6E", "vertex_kernel",
                 "edge_kernel", "halo_pack_kernel", "ball_kernel"):
        assert must in names, must
    bad = {k: v for k, v in res.items() if v.get("scratch", 0) != 0}
    assert not bad, bad


def test_workgroup_program_occupancy():
    """registers of the workgroup program allow the residency DESIGN.md section 4 states (and the auto rule of gcsadmm_create
    relies on: 4 workgroups per CU at n = 2): n = 2, 3 four wavefronts per SIMD or more (<= 128 VGPRs), n = 6 two (<= 256), the BOX
    instantiation at n = 6 three (<= 168: its 47 KB of LDS fit a CU three times, BASELINE config 5 runs on it)"""
    from gcs_admm_amd import build
    res = build.kernel_resources()
    assert any("gcs_wg_t512" in k for k in res)        # the 512-thread build (one workgroup per CU: two wavefronts per SIMD, <= 256)
    for k, v in res.items():
        if "gcs_wg_t512" in k:
            assert v["vgprs"] + v["agprs"] <= 256, (k, v)
            continue
        if "vertex_wg_kernelILi2E" in k or "vertex_wg_kernelILi3E" in k:
            assert v["vgprs"] + v["agprs"] <= 128, (k, v)
        if "vertex_wg_kernelILi6E" in k:
            assert v["vgprs"] + v["agprs"] <= (168 if "Lb1E" in k else 256), (k, v)


def _device_disassembly(tmp_path, obj_name):
    """gfx950 disassembly of one object of the in-tree build (llvm-objdump extracts the offload bundle next to its input: a copy)"""
    import os
    import shutil
    import subprocess
    from gcs_admm_amd import build
    build.build()
    llvm = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(llvm):
        import pytest
        pytest.skip("llvm-objdump not found")
    obj = tmp_path / obj_name
    shutil.copy(os.path.join(build.HERE, obj_name), obj)
    subprocess.check_call([llvm, "--offloading", str(obj)], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    dev = [f for f in os.listdir(tmp_path) if "amdgcn" in f and "gfx950" in f]
    assert len(dev) == 1, dev
    return subprocess.check_output([llvm, "-d", str(tmp_path / dev[0])], text=True)


def test_edge_kernel_hands_its_partials_over_with_sc1(tmp_path):
    """The last-workgroup hand-off of edge_kernel MODE 2 / 3 (csrc/gcsadmm.hip) is relaxed agent-scope atomics + `s_waitcnt
    vmcnt(0)` -- it relies on the partials being WRITE-THROUGH stores (sc1) that are drained before the ticket add, and on the last
    workgroup reading them with sc1 loads (visible across the XCDs' L2s).  That is a property of the generated code, so it is pinned
    here: in every MODE >= 2 instantiation, in program order: an sc1 store, a vmcnt(0) drain, the ticket's atomic add, sc1 loads."""
    import re
    asm = _device_disassembly(tmp_path, "gcsadmm.o")
    funcs = re.split(r"\n(?=[0-9a-f]{16} <)", asm)
    seen = 0
    for f in funcs:
        m = re.match(r"[0-9a-f]{16} <(_ZN12_GLOBAL__N_111edge_kernelI[df]Li([0-3])ELi\d+ELi\d+EE[^>]*)>:", f)
        if not m:
            continue
        mode = int(m.group(2))
        lines = f.splitlines()
        idx = lambda pat, start=0: next((i for i in range(start, len(lines)) if re.search(pat, lines[i])), -1)
        if mode < 2:
            assert idx(r"global_atomic_add") < 0, m.group(1)        # no ticket outside the single-launch modes
            continue
        seen += 1
        st = idx(r"global_store_dwordx2 .* sc1")
        assert st >= 0, (m.group(1), "partials are not sc1 stores")
        wt = idx(r"s_waitcnt vmcnt\(0\)", st)
        at = idx(r"global_atomic_add", st)
        assert 0 <= wt < at, (m.group(1), "no drain between the partial stores and the ticket")
        ld = idx(r"global_load_dwordx2 .* sc1", at)
        assert ld > at, (m.group(1), "the last workgroup does not read the partials with sc1 loads")
    assert seen >= 12, seen        # {f64, f32} x {MODE 2, 3} x c in {3, 5, 7, ...}


def test_every_launching_entry_point_selects_the_handles_device():
    """Entry points work on the handle's device whatever the caller's current device is (a process may hold handles on several
    GPUs): every extern "C" function of csrc/gcsadmm.hip that touches the HIP runtime or launches a kernel does so under USE_DEVICE /
    DeviceGuard, directly or by delegating at once to entry points that do."""
    import os
    import re
    from gcs_admm_amd import build
    src = open(os.path.join(build.CSRC, "gcsadmm.hip")).read()
    ext = src[src.index('extern "C" {'):]
    bodies = {}
    for m in re.finditer(r"\n(?:gcsadmm_status|void|const char \*|int) ?(gcsadmm_[a-z_0-9]+)\(([^)]*)\)\s*\n\{", ext):
        start = m.end()
        depth, i = 1, start
        while depth:
            c = ext[i]
            depth += (c == "{") - (c == "}")
            i += 1
        bodies[m.group(1)] = ext[start:i]
    assert len(bodies) >= 20, sorted(bodies)
    touches = re.compile(r"\bhip[A-Z]\w+\(|hipLaunchKernelGGL|launch_vertex<|launch_edge<|halo_pack<|halo_unpack<|halo_transfer\(|halo_upload\(|rccl\(\)\.(?!ok\b|err\b|GetErrorString\b)\w+\(|run_partitioned_loop\(|gcsadmm_wg_launch")
    guarded = re.compile(r"USE_DEVICE\(h\)|DeviceGuard device_guard_")
    exempt = {"gcsadmm_comm_unique_id", "gcsadmm_debug_sub_cycles", "gcsadmm_debug_phase_cycles"}        # no handle: ncclGetUniqueId; symbols of the diagnostic build
    for name, body in bodies.items():
        if name in exempt or not touches.search(body):
            continue
        g = guarded.search(body)
        assert g, f"{name} touches the device without selecting the handle's device"
        first = touches.search(body)
        # the guard comes before the first device call (argument checks may precede it); hipGetDeviceCount in create is the one query allowed earlier
        pre = body[:g.start()]
        early = [t.group(0) for t in touches.finditer(pre) if t.group(0) not in ("hipGetDeviceCount(", "hipGetErrorString(")]
        assert not early, (name, early)
