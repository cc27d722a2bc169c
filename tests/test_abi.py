"""The C-ABI library loads and exports every symbol include/gcsadmm.h declares (no compute calls:
this runs without a GPU), and the Python host refuses to run without the GPU."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "gcsadmm.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gcsadmm_[a-z_]+)\s*\(", text)))


def test_library_exports_header():
    from gcs_admm_amd import build, solver
    build.build()
    lib = solver.load_library()
    syms = declared_symbols()
    assert len(syms) >= 12
    for s in syms:
        assert hasattr(lib, s), s
    assert sorted(solver.EXPORTS) == syms


def test_struct_layouts_match_header(tmp_path):
    """sizes and field offsets of the ctypes mirrors equal what a C compiler derives from include/gcsadmm.h"""
    import subprocess
    from gcs_admm_amd import solver
    pairs = [("gcsadmm_graph_desc", solver.GraphDesc), ("gcsadmm_params", solver.Params), ("gcsadmm_state", solver.State),
             ("gcsadmm_control_block", solver.ControlBlock), ("gcsadmm_halo_desc", solver.HaloDesc)]
    lines = []
    for cname, ct in pairs:
        lines.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in ct._fields_:
            lines.append(f'printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "gcsadmm.h"\nint main(void){' + "".join(lines) + "return 0;}\n")
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I" + os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = dict(l.split() for l in subprocess.check_output([str(exe)], text=True).splitlines())
    for cname, ct in pairs:
        assert int(got[cname]) == ctypes.sizeof(ct), cname
        for fname, _ in ct._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(ct, fname).offset, (cname, fname)


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from gcs_admm_amd.cases import load_fixture
    from gcs_admm_amd.solver import DeviceSolver
    _, g = load_fixture("test1")
    with pytest.raises(RuntimeError, match="no HIP device"):
        DeviceSolver(g)


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "gcs_admm_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "libgcs_oracle" not in txt, f
