"""Would the ResidualUnit tails at C = 128 .. 384 be faster on the split GEMM kernel fed with P8 planes (as a k = 7 epilogue could
emit them) than on the fp32 streaming kernel?  B = 32, residual + pre-activated second output, as in the forward."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from facodec_amd import ops, _lib
_lib.load()
dev = torch.device("cuda:0")


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


for (C, T) in ((128, 24000), (192, 24000), (256, 4800), (384, 4800), (96, 48000)):
    B = 32
    x = torch.randn(B, C, T, device=dev)
    r = torch.randn(B, C, T, device=dev)
    w = torch.randn(C, C, 1, device=dev) * 0.05
    wp, ws = ops.pack_conv_weight(w), ops.pack_gemm_weight_split(w)
    a2 = torch.ones(C, device=dev)
    bias = torch.zeros(C, device=dev)
    p8 = ops.to_p8(x)
    t_pw = timed(lambda: ops.conv1d(x, wp, C, 1, bias=bias, res=r, alpha_y2=a2))
    try:
        t_gs = timed(lambda: ops.conv1d(x, None, C, 1, bias=bias, res=r, alpha_y2=a2, w_split=ws))
    except Exception as e:
        t_gs = float("nan"); print("fp32-in split:", str(e)[:120])
    try:
        t_p8 = timed(lambda: ops.conv1d(p8, None, C, 1, bias=bias, res=r, alpha_y2=a2, w_split=ws))
    except Exception as e:
        t_p8 = float("nan"); print("p8-in split:", str(e)[:120])
    t_to = timed(lambda: ops.to_p8(x))
    fl = 2.0 * B * C * C * T
    print(f"C={C} T={T}: streaming fp32 {t_pw:.3f} ms | split GEMM fp32 in {t_gs:.3f} ms ({fl / t_gs / 1e9:.0f} TF) | split GEMM P8 in {t_p8:.3f} ms ({fl / t_p8 / 1e9:.0f} TF) | to_p8 alone {t_to:.3f} ms", flush=True)
