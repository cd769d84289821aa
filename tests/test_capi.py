"""CPU-side checks of the C-ABI boundary: the library builds for gfx950, loads, and exports exactly the
entry points include/vgx.h declares (no compute calls here: there is no GPU in the CPU suite)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "vgx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vgx_[a-z_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from vgsim_amd import _capi
    if not os.path.exists(_capi.LIB_PATH):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "vgsim_amd", "csrc")])
    lib = _capi.load_library()
    declared = header_functions()
    assert declared, "no functions parsed from include/vgx.h"
    assert sorted(_capi.SIGNATURES) == declared
    for name in declared:
        assert hasattr(lib, name), name
    out = subprocess.check_output(["nm", "-D", "--defined-only", _capi.LIB_PATH]).decode()
    exported = set(re.findall(r" T (vgx_[a-z_]+)", out))
    assert exported == set(declared)


def test_struct_layouts_match_header():
    """Field order of the ctypes mirrors == field order in include/vgx.h."""
    from vgsim_amd import _capi
    src = open(os.path.join(ROOT, "include", "vgx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    for cname, ctype in (("vgx_dims", _capi.VgxDims), ("vgx_params", _capi.VgxParams), ("vgx_state", _capi.VgxState),
                         ("vgx_run_opts", _capi.VgxRunOpts), ("vgx_counters", _capi.VgxCounters)):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (cname, cname), src, flags=re.S).group(1)
        names = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            decl = re.sub(r"^(const\s+)?(double|int64_t)\s*", "", decl)
            for part in decl.split(","):
                names.append(re.sub(r"[\s\*]|\[.*\]", "", part))
        assert names == [f[0] for f in ctype._fields_], cname


def test_no_gpu_fails_loudly():
    """Without a HIP device the engine refuses to run instead of falling back to a CPU path."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from vgsim_amd import _capi
    with pytest.raises(_capi.VgxError, match="no HIP device|no CPU fallback"):
        _capi.HipEngine(0, 1, 1, 1)


def test_product_never_imports_oracle():
    """oracle/ is test infrastructure: nothing under vgsim_amd/ may import, load or link it."""
    pat = re.compile(r"(^\s*(from|import)\s+oracle\b)|libvgx_oracle|vgx_oracle\.|oracle/", re.M)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "vgsim_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                text = open(os.path.join(dirpath, f)).read()
                assert not pat.search(text), (dirpath, f)
