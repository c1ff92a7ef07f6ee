"""Split GEMM kernel (1- / 2-tap convs): fp32 input (in-kernel operand split) against P8 input (both operands by LDS-DMA)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from facodec_amd import ops, _lib
_lib.load()
dev = torch.device("cuda:0")


def timed(fn, n=20):
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


cases = [("1x1 512->512 T=960", 32, 512, 512, 960, 1, 1), ("1x1 768->768 T=960", 32, 768, 768, 960, 1, 1),
         ("LSTM proj 1536->6144 T=160", 32, 1536, 6144, 160, 1, 1), ("strided 128->256 k10 s5 T=24000", 32, 128, 256, 24000, 10, 5),
         ("strided 64->128 k4 s2 T=48000", 32, 64, 128, 48000, 4, 2), ("MPD-like 128->512 k5 s3 T=90000", 16, 128, 512, 90000, 5, 3)]
for name, B, ci, co, T, k, s in cases:
    x = torch.randn(B, ci, T, device=dev)
    w = torch.randn(co, ci, k, device=dev) * 0.02
    bias = torch.zeros(co, device=dev)
    ws = ops.pack_gemm_weight_split(w, in_stride=s)
    if s == 1:
        kw = dict(bias=bias, pad_left=0, pad_mode=ops.PAD_ZERO, t_out=T)
        t_out = T
    else:
        kw = dict(bias=bias, stride=s)
        t_out = -(-T // s)
    fl = 2.0 * B * co * t_out * ci * k
    ms = timed(lambda: ops.conv1d(x, None, co, k, w_split=ws, **kw))
    p8 = ops.to_p8(x)
    ms8 = timed(lambda: ops.conv1d(p8, None, co, k, w_split=ws, **kw))
    msp = timed(lambda: ops.to_p8(x))
    print(f"{name:36s} fp32 in {ms:.3f} ms {fl / ms / 1e9:6.1f} TF | P8 in {ms8:.3f} ms {fl / ms8 / 1e9:6.1f} TF | to_p8 pass {msp:.3f} ms "
          f"({x.numel() * 10 / msp / 1e9:.2f} TB/s)")
