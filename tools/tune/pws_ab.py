"""k = 1 ResidualUnit tails at the forward's shapes: fp32 streaming kernel (conv1d_pw.hip) against the bf16 x 3 streaming kernel
(conv1d_pw_split.hip), same process, alternating; error of both against fp64 on one clip."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from facodec_amd import ops, _lib
_lib.load()
dev = torch.device("cuda:0")
B = int(os.environ.get("B", "32"))


def timed(fn, n=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def rel(a, b):
    return float((a.double() - b).abs().max() / b.abs().max())


shapes = ((64, 48000), (96, 48000), (128, 24000), (192, 24000), (256, 4800), (384, 4800))
for (C, T) in shapes:
    g = torch.Generator().manual_seed(C)
    x = torch.randn(B, C, T, generator=g).to(dev)
    r = torch.randn(B, C, T, generator=g).to(dev)
    w = (torch.randn(C, C, 1, generator=g) / C ** 0.5).to(dev)
    wp = ops.pack_conv_weight(w)
    a2 = torch.ones(C, device=dev)
    bias = torch.randn(C, generator=g).to(dev) * 0.1
    res = {}
    for rep in range(2):
        for split in (False, True):
            ops.PW_SPLIT = split
            t = timed(lambda: ops.conv1d(x, wp, C, 1, bias=bias, res=r, alpha_y2=a2))
            res.setdefault(split, []).append(t)
    y64 = torch.einsum("oc,ct->ot", w[:, :, 0].double(), x[0].double()) + bias.double().view(-1, 1) + r[0].double()
    errs = {}
    for split in (False, True):
        ops.PW_SPLIT = split
        yy, _ = ops.conv1d(x, wp, C, 1, bias=bias, res=r, alpha_y2=a2)
        errs[split] = rel(yy[0], y64)
    gb = 4 * 4.0 * B * C * T / 1e9
    print(f"C={C} T={T} B={B}: fp32 {min(res[False]):.3f} ms ({gb / min(res[False]):.2f} TB/s) err {errs[False]:.2e} | "
          f"split {min(res[True]):.3f} ms ({gb / min(res[True]):.2f} TB/s) err {errs[True]:.2e} | D={os.environ.get('FAC_PWS_D', 'default')}", flush=True)

# ---- stride-2 layers with few channels: streaming kernel with taps against the tiled split GEMM kernel (the layers' own routing)
from facodec_amd import layers
with torch.no_grad():
    for kind, ci, co, T in (("convtr", 192, 96, 24000), ("conv", 64, 128, 48000), ("convtr", 128, 64, 24000), ("conv", 96, 192, 48000)):
        torch.manual_seed(ci)
        if kind == "convtr":
            m = layers.SConvTranspose1d(ci, co, 4, stride=2, causal=True, norm="weight_norm").to(dev)
        else:
            m = layers.SConv1d(ci, co, 4, stride=2, causal=True, norm="weight_norm").to(dev)
        m.w.freeze_packed = True
        x = torch.randn(B, ci, T, device=dev)
        a2 = torch.ones(co, device=dev)
        res, outs = {}, {}
        for rep in range(2):
            for taps in (False, True):
                ops.PW_TAPS = taps
                res.setdefault(taps, []).append(timed(lambda: m.run(x, alpha_y2=a2)))
                outs[taps] = m.run(x, alpha_y2=a2)[0]
        err = float((outs[True] - outs[False]).abs().max() / outs[False].abs().max())
        t_out = T * 2 if kind == "convtr" else T // 2
        gb = 4.0 * B * (ci * T + 2 * co * t_out) / 1e9
        fl = 2.0 * B * co * ci * (2 if kind == "convtr" else 4) * t_out
        print(f"{kind} {ci}->{co} s2 T_in={T} B={B}: tiled {min(res[False]):.3f} ms ({fl / min(res[False]) / 1e9:.0f} TF-eq) | "
              f"streaming taps {min(res[True]):.3f} ms ({fl / min(res[True]) / 1e9:.0f} TF-eq, {gb / min(res[True]):.2f} TB/s) | max diff {err:.1e}", flush=True)
