"""The C-ABI library loads and exports every symbol include/facodec_hip.h declares (no compute)."""
import ctypes
import os
import re

from facodec_amd import _lib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    hdr = open(os.path.join(REPO, "include", "facodec_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return set(re.findall(r"\b(fac_[a-z0-9_]+)\s*\(", hdr)) - {"fac_pad32", "fac_cin_pad", "fac_convtr_rows"}   # static inline helpers


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


def _desc(B, c_in, c_out, T_in, T_out, K, stride=1, dil=1, pad_left=0, n_phase=1, res=False, y2=False, split=False, ws=True,
          alpha_out=False):
    d = _lib.ConvDesc()
    fake = ctypes.c_void_p(0x10000)          # never dereferenced: fac_conv1d_variant only reads the descriptor
    d.x, d.w, d.y = fake, fake, fake
    d.bias = fake
    d.res = fake if res else None
    d.y2 = fake if y2 else None
    d.alpha_y2 = fake if y2 else None
    d.alpha_out = fake if alpha_out else None
    d.w_split = fake if split else None
    d.ws, d.ws_bytes = (fake, 32 << 20) if ws else (None, 0)
    d.x_bs, d.x_cs, d.y_bs, d.y_cs = c_in * T_in, T_in, c_out * T_out * (n_phase if n_phase > 1 else 1), T_out * (n_phase if n_phase > 1 else 1)
    d.B, d.C_in, d.T_in, d.C_out, d.C_out_pad, d.T_out = B, c_in, T_in, c_out, (c_out + 31) // 32 * 32, T_out
    d.K, d.stride, d.dilation, d.pad_left, d.pad_mode = K, stride, dil, pad_left, 0
    d.n_phase, d.y_tstride, d.phase_shift, d.act, d.w_batched, d.w_bs = n_phase, n_phase, 0, 0, 0, 0
    return d


def test_kernel_selection_for_the_benchmark_shapes():
    """Which kernel fac_conv1d_fwd takes for the layer shapes of configs[1] (B = 32 x 2 s) and for their small-batch
    counterparts -- host logic only (fac_conv1d_variant reads the descriptor, nothing is launched)."""
    lib = _lib.load()

    def variant(d):
        buf = ctypes.create_string_buffer(96)
        return lib.fac_conv1d_variant(ctypes.byref(d), buf, 96), buf.value.decode()

    B = 32
    # k = 7 ResidualUnit convs with pre-split weights: the split-bf16 kernel, for every channel count of the codec
    for c, t in ((64, 48000), (96, 48000), (128, 24000), (192, 24000), (384, 4800), (768, 960)):
        for dil in (1, 3, 9):
            vid, name = variant(_desc(B, c, c, t, t, 7, dil=dil, pad_left=6 * dil, split=True, alpha_out=True))
            assert vid == 11, (c, dil, name)
    # k = 1 tails: streaming kernel up to 384 channels at benchmark size, tiled kernel beyond and for small batches
    for c, t, want in ((64, 48000, 14), (96, 48000, 14), (128, 24000, 14), (192, 24000, 14), (256, 4800, 14), (384, 4800, 14),
                       (512, 960, 5), (768, 960, 5)):
        vid, name = variant(_desc(B, c, c, t, t, 1, res=True, y2=True))
        assert vid == want, (c, name)
    vid, name = variant(_desc(2, 192, 192, 24000, 24000, 1, res=True, y2=True))
    assert vid == 6, name                                             # two clips: not enough column blocks to stream
    # edge convs
    assert variant(_desc(B, 1, 64, 48000, 48000, 7, pad_left=6, y2=True))[0] == 12
    assert variant(_desc(B, 96, 1, 48000, 48000, 7, pad_left=6))[0] == 9
    assert variant(_desc(1, 1024, 1, 9856, 9856, 3, pad_left=1))[0] == 13      # MPD conv_post over one row-concatenated signal
    # LSTM-sized problems: few columns -> split reduction; the input projection GEMM -> wide k = 1 tile
    assert variant(_desc(1, 6144, 1536, 32, 32, 1))[0] == 10
    assert variant(_desc(1, 1536, 6144, 5120, 5120, 1))[0] == 5
    # transposed conv (polyphase, n_phase = stride) stays on the fp32 tiles
    vid, name = variant(_desc(B, 384, 192, 4800, 4800, 2, pad_left=1, n_phase=5, y2=True))
    assert vid in (3, 4), name
