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


def test_isa_hazard_scan_finds_a_dpp_read_right_after_its_write():
    """tools/isa_hazard_scan.py (run by build() on the shipped code objects) on synthetic listings: a VALU write of a DPP source one
    wait state before the read is reported, through a branch edge too; with the s_nop the chains carry it is clean."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("isa_hazard_scan", os.path.join(os.path.dirname(__file__), "..", "tools", "isa_hazard_scan.py"))
    hz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(hz)
    def listing(lines):
        out = ["0000000000001000 <k>:"]
        for n, l in enumerate(lines):
            out.append("\t%s // %012X: 00000000" % (l, 0x1000 + 4 * n))
        return "\n".join(out)
    dpp = "v_fmac_f64_dpp v[6:7], v[10:11], v[2:3] row_newbcast:0 row_mask:0xf bank_mask:0xf"
    bad = hz.parse(listing(["v_accvgpr_read_b32 v10, a3", dpp]))
    assert hz.scan(bad["k"])[1], "write right before the DPP read"
    bad = hz.parse(listing(["v_mov_b32_e32 v11, v4", "s_nop 0", dpp]))
    assert hz.scan(bad["k"])[1], "one wait state is not enough"
    ok = hz.parse(listing(["v_mov_b32_e32 v11, v4", "s_nop 1", dpp, dpp]))
    assert hz.scan(ok["k"]) == (2, [])
    ok = hz.parse(listing(["v_mov_b32_e32 v12, v4", dpp]))
    assert hz.scan(ok["k"]) == (1, [])
    # through a branch: the writer sits right before a branch to the DPP instruction (branch = 1 wait state)
    edge = hz.parse(listing(["v_mov_b32_e32 v10, v4", "s_branch 2", "s_nop 1", "s_nop 1", dpp]))
    assert hz.scan(edge["k"])[1]
    ex = hz.parse(listing(["v_cmpx_lt_f64_e32 vcc, v[0:1], v[2:3]", "s_nop 2", dpp]))
    assert hz.scan(ex["k"])[1], "EXEC written by a VALU instruction within five wait states"
