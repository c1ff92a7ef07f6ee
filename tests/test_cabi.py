"""The C-ABI library loads and exports every symbol include/facodec_hip.h declares (no compute)."""
import ctypes
import os
import re

from facodec_amd import _lib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    hdr = open(os.path.join(REPO, "include", "facodec_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return set(re.findall(r"\b(fac_[a-z0-9_]+)\s*\(", hdr)) - {"fac_pad32", "fac_cin_pad"}


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    syms = _header_symbols()
    assert len(syms) >= 25
    for s in sorted(syms):
        assert hasattr(lib, s), f"{s} declared in the header but not exported"


def test_binding_table_matches_header():
    assert set(_lib.SIGNATURES) == _header_symbols()


def test_version_and_error_string():
    lib = _lib.load()
    assert lib.fac_version() >= 1
    assert isinstance(lib.fac_last_error(), bytes)


def test_argument_validation_fails_loudly_without_gpu():
    """Bad descriptors are rejected on the host before any launch: error code + message."""
    lib = _lib.load()
    d = _lib.ConvDesc()
    rc = lib.fac_conv1d_fwd(ctypes.byref(d), None)
    assert rc == -1
    assert b"null pointer" in lib.fac_last_error()
    rc = lib.fac_lstm_layer_fwd(ctypes.c_void_p(8), ctypes.c_void_p(8), ctypes.c_void_p(8), ctypes.c_void_p(8), 4, 100, 32, None)
    assert rc == -1 and b"multiple of 64" in lib.fac_last_error()


def test_desc_struct_layouts_match_the_header(tmp_path):
    """ctypes mirrors of the descriptors have the size / field offsets gcc gives the header's structs."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        import pytest
        pytest.skip("gcc not available")
    src = tmp_path / "sz.c"
    src.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "facodec_hip.h"\n'
        'int main(){printf("%zu %zu %zu %zu %zu %zu\\n", sizeof(fac_conv_desc), offsetof(fac_conv_desc, B),'
        ' offsetof(fac_conv_desc, w_bs), sizeof(fac_vq_desc), offsetof(fac_vq_desc, codes_bs), offsetof(fac_vq_desc, Kc));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", os.path.join(REPO, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    c, v = _lib.ConvDesc, _lib.VqDesc
    assert [int(x) for x in out] == [ctypes.sizeof(c), c.B.offset, c.w_bs.offset, ctypes.sizeof(v), v.codes_bs.offset, v.Kc.offset]


def test_product_refuses_cpu_tensors():
    import pytest
    import torch
    from facodec_amd import ops
    with pytest.raises(_lib.FacodecHipError):
        ops.conv1d(torch.zeros(1, 2, 8), torch.zeros(2, 1, 32), 4, 1)
