"""Weight-gradient GEMM on the bf16 matrix pipe with fp32-grade operand splitting (facodec_amd/csrc/conv1d_wgrad_split.hip)
against an fp64 restatement of torch's conv1d weight gradient, next to the fp32-MFMA kernel it replaces."""
import pytest
import torch
import torch.nn.functional as F

from facodec_amd import ops

pytestmark = pytest.mark.gpu


def _pad(x, left, right, mode):
    if mode == ops.PAD_ZERO:
        return F.pad(x, (left, right))
    # pad1d of dac/model/encodec.py:96-113 (reflect; signals shorter than the pad are zero-extended first)
    length = x.shape[-1]
    max_pad = max(left, right)
    extra = 0
    if length <= max_pad:
        extra = max_pad - length + 1
        x = F.pad(x, (0, extra))
    y = F.pad(x, (left, right), mode="reflect")
    return y[..., : y.shape[-1] - extra] if extra else y


def _ref_dw(x, dy, k, stride, dil, pad_left, mode):
    """fp64 dW of y = conv1d(pad(x), w, stride, dilation) for the given upstream gradient."""
    xd = x.double()
    t_out = dy.shape[-1]
    need = (t_out - 1) * stride + (k - 1) * dil + 1
    right = max(0, need - pad_left - x.shape[-1])
    xp = _pad(xd, pad_left, right, mode)
    w = torch.zeros(dy.shape[1], x.shape[1], k, dtype=torch.float64, requires_grad=True)
    y = F.conv1d(xp, w, stride=stride, dilation=dil)[..., :t_out]
    (y * dy.double()).sum().backward()
    return w.grad


CASES = [
    # B, C_in, C_out, T_in, K, stride, dil, pad_left, mode
    (2, 64, 64, 700, 7, 1, 1, 6, ops.PAD_REFLECT),
    (2, 96, 96, 1000, 7, 1, 3, 18, ops.PAD_REFLECT),
    (1, 128, 160, 517, 7, 1, 9, 54, ops.PAD_REFLECT),
    (2, 24, 200, 333, 1, 1, 1, 0, ops.PAD_ZERO),
    (3, 300, 40, 97, 1, 1, 1, 0, ops.PAD_ZERO),
    (2, 64, 128, 640, 4, 2, 1, 2, ops.PAD_REFLECT),
    (2, 128, 256, 650, 10, 5, 1, 5, ops.PAD_REFLECT),
    (2, 96, 192, 612, 12, 6, 1, 6, ops.PAD_REFLECT),
    (1, 32, 128, 3001, 5, 3, 1, 2, ops.PAD_ZERO),
    (1, 128, 96, 900, 5, 1, 1, 2, ops.PAD_ZERO),
    (40, 96, 32, 65, 9, 2, 1, 4, ops.PAD_ZERO),
    (40, 6, 32, 129, 9, 1, 1, 4, ops.PAD_ZERO),
    (8, 96, 32, 33, 3, 1, 1, 1, ops.PAD_ZERO),
    (2, 1, 64, 2400, 7, 1, 1, 6, ops.PAD_REFLECT),
    (2, 96, 1, 2400, 7, 1, 1, 6, ops.PAD_REFLECT),
    (2, 64, 64, 5, 7, 1, 1, 6, ops.PAD_REFLECT),          # signal shorter than the pad
    (2, 20, 256, 40, 1, 1, 1, 0, ops.PAD_ZERO),
    (1, 1536, 1536, 160, 7, 1, 1, 6, ops.PAD_REFLECT),    # largest layer shape (decoder input conv)
    # k-major kernel edges: channel counts that are not multiples of its 32-channel blocks, the smallest count it takes (16),
    # more blocks than one column tile row (K = 7 x 2 groups), odd shifts (unaligned 16-byte plane loads), 130 output rows
    (2, 48, 64, 300, 5, 1, 1, 2, ops.PAD_ZERO),
    (2, 16, 32, 200, 7, 1, 1, 6, ops.PAD_REFLECT),
    (3, 80, 130, 257, 7, 1, 3, 18, ops.PAD_REFLECT),
    (2, 40, 96, 301, 3, 1, 1, 1, ops.PAD_ZERO),
    # round 6: row tiles with 33 .. 96 real output channels run column-split wave layouts (k-split kernel: k = 7; 64 x 64 kernel: the rest)
    (2, 192, 192, 500, 7, 1, 9, 54, ops.PAD_REFLECT),
    (2, 192, 192, 500, 1, 1, 1, 0, ops.PAD_ZERO),
    (2, 64, 64, 900, 1, 1, 1, 0, ops.PAD_ZERO),
    (2, 96, 96, 900, 1, 1, 1, 0, ops.PAD_ZERO),
    (2, 64, 224, 640, 4, 2, 1, 2, ops.PAD_REFLECT),
    # round 6: k = 1, 64 .. 192 channels, T >= 4096: straight from the fp32 tensors on the fp32 matrix pipe (conv1d_wgrad_k1.hip);
    # window tails (T % 32 != 0), 2 / 3 / 4 / 6 roles per workgroup, unequal channel counts
    (2, 64, 64, 4800, 1, 1, 1, 0, ops.PAD_ZERO),
    (2, 96, 96, 4100, 1, 1, 1, 0, ops.PAD_ZERO),
    (1, 192, 192, 8192, 1, 1, 1, 0, ops.PAD_ZERO),
    (2, 128, 128, 5004, 1, 1, 1, 0, ops.PAD_ZERO),
    (3, 64, 192, 4096, 1, 1, 1, 0, ops.PAD_ZERO),
    (2, 192, 96, 4444, 1, 1, 1, 0, ops.PAD_ZERO),
    (5, 160, 64, 4112, 1, 1, 1, 0, ops.PAD_ZERO),
    # first layers on the same kernel with virtual rows (fac_conv1d_bwd_weight_taps): reflect / zero padding, window tails
    (2, 1, 64, 4800, 7, 1, 1, 6, ops.PAD_REFLECT),
    (3, 1, 64, 4099, 7, 1, 1, 6, ops.PAD_REFLECT),
    (2, 2, 32, 5000, 9, 1, 1, 4, ops.PAD_ZERO),
    (1, 6, 32, 8192, 9, 1, 1, 4, ops.PAD_ZERO),
    (2, 1, 32, 4500, 5, 1, 2, 4, ops.PAD_ZERO),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "B%d_%dto%d_T%d_k%d_s%d_d%d" % c[:7])
def test_wgrad_split_against_fp64(case, cuda):
    B, c_in, c_out, t_in, k, stride, dil, pad_left, mode = case
    g = torch.Generator().manual_seed(hash(case[:7]) % (1 << 31))
    x = torch.randn(B, c_in, t_in, generator=g)
    t_out = (t_in + pad_left - (k - 1) * dil - 1) // stride + 1 if mode == ops.PAD_ZERO else -(-t_in // stride)
    if mode == ops.PAD_ZERO:
        t_out = (t_in + 2 * pad_left - (k - 1) * dil - 1) // stride + 1
    dy = torch.randn(B, c_out, t_out, generator=g)
    ref = _ref_dw(x, dy, k, stride, dil, pad_left, mode)
    scale = float(ref.abs().max())
    lib_bytes = ops._lib.load().fac_conv1d_bwd_weight_split_ws_bytes(B, c_in, t_in, c_out, t_out, k, stride, dil, 0, 0)
    assert lib_bytes > 0, "shape must qualify for the split kernel"
    prev = ops.BF16_SPLIT
    try:
        ops.BF16_SPLIT = True
        got = ops.conv1d_bwd_weight(x.to(cuda), dy.to(cuda), k, stride=stride, dilation=dil, pad_mode=mode, pad_left=pad_left)
        ops.BF16_SPLIT = False
        base = ops.conv1d_bwd_weight(x.to(cuda), dy.to(cuda), k, stride=stride, dilation=dil, pad_mode=mode, pad_left=pad_left)
    finally:
        ops.BF16_SPLIT = prev
    e_split = float((got.cpu().double() - ref).abs().max()) / scale
    e_fp32 = float((base.cpu().double() - ref).abs().max()) / scale
    assert e_split < 1e-5, (e_split, e_fp32)
    assert e_split <= 2.0 * e_fp32 + 2e-7, (e_split, e_fp32)
    # deterministic (fixed slice order)
    ops.BF16_SPLIT = True
    try:
        again = ops.conv1d_bwd_weight(x.to(cuda), dy.to(cuda), k, stride=stride, dilation=dil, pad_mode=mode, pad_left=pad_left)
    finally:
        ops.BF16_SPLIT = prev
    assert torch.equal(again, got)
    # the bias gradient folded into the dy split pass (k-major shapes) or from fac_bias_grad (the others): same dW, db = sum dy
    ops.BF16_SPLIT = True
    try:
        dw2, db = ops.conv1d_bwd_weight(x.to(cuda), dy.to(cuda), k, stride=stride, dilation=dil, pad_mode=mode, pad_left=pad_left, want_db=True)
    finally:
        ops.BF16_SPLIT = prev
    assert torch.equal(dw2, got)
    ref_db = dy.double().sum((0, 2))
    assert float((db.cpu().double() - ref_db).abs().max()) <= 1e-5 * max(1.0, float(ref_db.abs().max()))


def test_convtr_wgrad_on_split_kernel(cuda):
    """SConvTranspose1d weight gradient (roles of input and output swapped, stride = K / 2) through the split kernel."""
    g = torch.Generator().manual_seed(3)
    B, c_in, c_out, t_in, s = 2, 96, 48, 300, 5
    x = torch.randn(B, c_in, t_in, generator=g)
    v = torch.randn(c_in, c_out, 2 * s, generator=g) * 0.1
    dy = torch.randn(B, c_out, t_in * s, generator=g)
    w = v.double().clone().requires_grad_()
    y = F.conv_transpose1d(x.double(), w, stride=s)[..., : t_in * s]
    (y * dy.double()).sum().backward()
    prev = ops.BF16_SPLIT
    try:
        ops.BF16_SPLIT = True
        _, dw = ops.conv_transpose1d_bwd(x.to(cuda), dy.to(cuda), v.to(cuda), None, s)
    finally:
        ops.BF16_SPLIT = prev
    assert float((dw.cpu().double() - w.grad).abs().max() / w.grad.abs().max()) < 1e-5


@pytest.mark.parametrize("sf,kf,F0,P,T", [(1, 9, 25, 32, 7), (2, 9, 51, 64, 7), (2, 9, 26, 32, 7), (1, 3, 13, 16, 7),
                                           # long enough for interior tiles (LDS-DMA staging of the virtual channels), a row
                                           # pitch that is not a multiple of 4 (register staging), the model's 272 / 136 pitches
                                           (1, 9, 25, 32, 40), (1, 9, 26, 34, 40), (2, 9, 51, 68, 40), (1, 3, 30, 34, 40),
                                           (1, 9, 256, 272, 9), (2, 9, 128, 136, 9),
                                           # enough columns for the 32-row split kernel (conv1d_bsplit2.hip): (9, s1), (9, s2), (3, s1)
                                           (1, 9, 25, 32, 100), (2, 9, 51, 64, 100), (1, 3, 13, 16, 300), (2, 9, 256, 272, 30)])
def test_two_level_conv_is_conv2d(sf, kf, F0, P, T, cuda):
    """(3, kf) Conv2d with stride (1, sf), padding (1, kf // 2) (dac/model/discriminator.py:110-120) as ONE 1-D conv with
    two-level taps over the row-concatenated (frame, frequency) signal: forward, data gradient and weight gradient (split
    kernel and fp32 kernel) against torch's conv2d in fp64."""
    from facodec_amd import autograd_disc as AD
    g = torch.Generator().manual_seed(11 + kf + sf)
    B, ci, co = 2, 6, 32
    pf = kf // 2
    x4 = torch.randn(B, ci, T, F0, generator=g)
    w = torch.randn(co, ci, 3, kf, generator=g) * 0.2
    bias = torch.randn(co, generator=g)
    xr = x4.double().requires_grad_()
    wr = w.double().requires_grad_()
    y_ref = F.conv2d(xr, wr, bias.double(), stride=(1, sf), padding=(1, pf))
    F1 = y_ref.shape[-1]
    r = torch.randn(*y_ref.shape, generator=g).double()
    (y_ref * r).sum().backward()
    P_out = P // sf
    assert P - F0 >= pf and P_out >= F1
    # row-concatenated layouts: (1, C, B*(T+1)*P), zero gaps and one zero separator row per clip
    cat = torch.zeros(ci, B, T + 1, P)
    cat[:, :, :T, :F0] = x4.permute(1, 0, 2, 3)
    rc = torch.zeros(co, B, T + 1, P_out)
    rc[:, :, :T, :F1] = r.permute(1, 0, 2, 3).float()
    for split in (True, False):
        prev = ops.BF16_SPLIT
        ops.BF16_SPLIT = split
        try:
            xc = cat.reshape(1, ci, -1).to(cuda).requires_grad_()
            wc = w.reshape(co, ci, 3 * kf).to(cuda).requires_grad_()
            bc = bias.to(cuda).requires_grad_()
            prof = ops.ConvLaunchProfile()
            ops.set_conv_profile(prof)
            try:
                y = AD.PlainConv.apply(xc, wc, None, bc, 3 * kf, sf, P + pf, (kf, P))
                torch.cuda.synchronize()
            finally:
                ops.set_conv_profile(None)
            assert y.shape[-1] == B * (T + 1) * P_out
            if split and ops.split2_ok(co, 3 * kf, kf, sf, y.shape[-1]):
                assert any("bsplit2" in k for k in prof.summary()), prof.summary().keys()
            yv = y.detach().cpu().reshape(co, B, T + 1, P_out)[:, :, :T, :F1].permute(1, 0, 2, 3)
            assert float((yv.double() - y_ref.detach()).abs().max() / y_ref.detach().abs().max()) < 1e-5
            (y * rc.reshape(1, co, -1).to(cuda)).sum().backward()
        finally:
            ops.BF16_SPLIT = prev
        dx = xc.grad.cpu().reshape(ci, B, T + 1, P)[:, :, :T, :F0].permute(1, 0, 2, 3)
        assert float((dx.double() - xr.grad).abs().max() / xr.grad.abs().max()) < 1e-5, split
        dw = wc.grad.cpu().reshape(co, ci, 3, kf)
        assert float((dw.double() - wr.grad).abs().max() / wr.grad.abs().max()) < 1e-5, split
        assert float((bc.grad.cpu().double() - r.sum((0, 2, 3))).abs().max()) < 1e-3


@pytest.mark.parametrize("ci,co,kf,sf,F0,P,T", [(2, 32, 9, 1, 129, 144, 480), (32, 1, 3, 1, 129, 144, 480), (2, 32, 9, 2, 129, 144, 480)])
def test_two_level_convs_with_one_or_two_live_channels_run_on_the_narrow_kernel(ci, co, kf, sf, F0, P, T, cuda):
    """Round 6: the multi-resolution discriminator's 32 -> 1 output conv (forward) and the 2-channel data gradient of its first
    layer (dac/model/discriminator.py:110-121) are two-level-tap convs with one or two output channels: VALU kernel over
    3 x C_in virtual channels (conv1d_narrow.hip) instead of a 32-row MFMA tile.  Forward and data gradient against conv2d in fp64."""
    from facodec_amd import autograd_disc as AD
    g = torch.Generator().manual_seed(5 + ci + kf)
    B, pf = 2, kf // 2
    x4 = torch.randn(B, ci, T, F0, generator=g)
    w = torch.randn(co, ci, 3, kf, generator=g) * 0.2
    bias = torch.randn(co, generator=g)
    xr = x4.double().requires_grad_()
    y_ref = F.conv2d(xr, w.double(), bias.double(), stride=(1, sf), padding=(1, pf))
    F1 = y_ref.shape[-1]
    r = torch.randn(*y_ref.shape, generator=g).double()
    (y_ref * r).sum().backward()
    P_out = P // sf
    cat = torch.zeros(ci, B, T + 1, P)
    cat[:, :, :T, :F0] = x4.permute(1, 0, 2, 3)
    rc = torch.zeros(co, B, T + 1, P_out)
    rc[:, :, :T, :F1] = r.permute(1, 0, 2, 3).float()
    xc = cat.reshape(1, ci, -1).to(cuda).requires_grad_()
    wc = w.reshape(co, ci, 3 * kf).to(cuda).requires_grad_()
    bc = bias.to(cuda).requires_grad_()
    prof = ops.ConvLaunchProfile()
    ops.set_conv_profile(prof)
    try:
        y = AD.PlainConv.apply(xc, wc, None, bc, 3 * kf, sf, P + pf, (kf, P))
        (y * rc.reshape(1, co, -1).to(cuda)).sum().backward()
        torch.cuda.synchronize()
    finally:
        ops.set_conv_profile(None)
    if sf == 1:          # (the strided first layers' data gradient is a zero-inserted stride-1 conv as well; forward with stride stays on its tile)
        assert any("narrow" in k and "two-level" in k for k in prof.summary()), prof.summary().keys()
    yv = y.detach().cpu().reshape(co, B, T + 1, P_out)[:, :, :T, :F1].permute(1, 0, 2, 3)
    assert float((yv.double() - y_ref.detach()).abs().max() / y_ref.detach().abs().max()) < 1e-5
    dx = xc.grad.cpu().reshape(ci, B, T + 1, P)[:, :, :T, :F0].permute(1, 0, 2, 3)
    assert float((dx.double() - xr.grad).abs().max() / xr.grad.abs().max()) < 1e-5


def test_column_split_wave_layouts_are_bit_identical_to_the_row_split():
    """Row tiles with 33 .. 96 real output channels (every C = 64 / 96 layer, the second row tile of the C = 192 layers): the waves split
    over the columns and multiply only real rows -- the same sums in the same order as the row-split layout that multiplies clamped
    duplicates (FAC_WGRAD_NARROW=0).  Separate processes: the switch is read once."""
    import json
    import os
    import subprocess
    import sys
    script = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "tune", "wgrad_narrow_check.py")
    outs = []
    for val in ("0", "1"):
        r = subprocess.run([sys.executable, script], env=dict(os.environ, FAC_WGRAD_NARROW=val), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("WGN ")][-1][4:]))
    assert outs[0] == outs[1] and len(outs[0]) == 9, outs


def test_k1_tails_with_few_channels_run_on_the_fp32_streaming_kernel(cuda):
    """The shapes above really take conv1d_wgrad_k1.hip (the query says so), the kernel is deterministic, and FAC_WGRAD_K1_STREAM=0
    (module switch) gives the split kernel's result within the fp32 grade."""
    lib = ops._lib.load()
    assert lib.fac_conv1d_bwd_weight_k1_ws_bytes(16, 96, 96, 48000) > 0 and lib.fac_conv1d_bwd_weight_k1_ws_bytes(16, 192, 192, 24000) > 0
    assert lib.fac_conv1d_bwd_weight_k1_ws_bytes(16, 64, 64, 48000) > 0
    assert lib.fac_conv1d_bwd_weight_k1_ws_bytes(16, 256, 256, 4800) < 0 and lib.fac_conv1d_bwd_weight_k1_ws_bytes(16, 96, 96, 960) < 0
    assert lib.fac_conv1d_bwd_weight_k1_ws_bytes(2, 96, 96, 4101) < 0 and lib.fac_conv1d_bwd_weight_k1_ws_bytes(2, 80, 96, 8192) < 0
    g = torch.Generator().manual_seed(5)
    x = torch.randn(4, 96, 12000, generator=g).to(cuda)
    dy = torch.randn(4, 96, 12000, generator=g).to(cuda)
    prev = ops.WGRAD_K1_STREAM
    try:
        ops.WGRAD_K1_STREAM = True
        a = ops.conv1d_bwd_weight(x, dy, 1, pad_mode=ops.PAD_ZERO, pad_left=0)
        b = ops.conv1d_bwd_weight(x, dy, 1, pad_mode=ops.PAD_ZERO, pad_left=0)
        ops.WGRAD_K1_STREAM = False
        c = ops.conv1d_bwd_weight(x, dy, 1, pad_mode=ops.PAD_ZERO, pad_left=0)
    finally:
        ops.WGRAD_K1_STREAM = prev
    assert torch.equal(a, b)
    ref = torch.einsum("bot,bit->oi", dy.double(), x.double()).unsqueeze(-1)
    scale = float(ref.abs().max())
    ea, ec = float((a.double() - ref).abs().max()) / scale, float((c.double() - ref).abs().max()) / scale
    assert ea < 2e-6 and ec < 2e-6, (ea, ec)
