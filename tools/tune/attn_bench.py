"""StyleEncoder attention at the benchmark's sizes: inference kernel (B = 32) and the training Function forward + backward (B = 16)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from facodec_amd import ops, _lib
from facodec_amd.autograd_quant import _Attention
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


for B, T in ((32, 188), (16, 188)):
    H, dk = 2, 256
    q, k, v = (torch.randn(B, H * dk, T, device=dev) for _ in range(3))
    mask = torch.ones(B, T, device=dev)
    print(f"B={B} T={T}: inference attention {timed(lambda: ops.attention(q, k, v, mask, H)) * 1e3:.0f} us", flush=True)
    qc, kc, vc = (t.clone().requires_grad_(True) for t in (q, k, v))
    w = torch.randn_like(q)

    def fb():
        o = _Attention.apply(qc, kc, vc, mask, H, None, 1.0)
        o.backward(w)
    print(f"B={B} T={T}: training attention forward + backward {timed(fb) * 1e3:.0f} us", flush=True)
