"""Timing of the split k = 7 kernel on three layer shapes (average of 20 launches, HIP events) -- run against ablation builds
(FAC_BUILD_TAG / FAC_EXTRA_FLAGS=-DFAC_ABL_..., selected with FAC_LIB_PATH) to see what each part of a stage costs."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from facodec_amd import ops, _lib
_lib.load()
dev = torch.device("cuda:0")
out = []
for (C, T, d) in ((96, 48000, 1), (192, 24000, 3), (768, 960, 1)):
    B = 32
    x = torch.randn(B, C, T, device=dev)
    w = torch.randn(C, C, 7, device=dev) * 0.01
    ws = ops.pack_conv_weight_split(w)
    al = torch.ones(C, device=dev)
    bias = torch.zeros(C, device=dev)
    for _ in range(3):
        ops.conv1d(x, None, C, 7, dilation=d, bias=bias, alpha_out=al, w_split=ws)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        ops.conv1d(x, None, C, 7, dilation=d, bias=bias, alpha_out=al, w_split=ws)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    txt = f"C={C}: {ms:.3f} ms {2.0 * B * C * C * 7 * T / ms / 1e9:6.1f} TF"
    if hasattr(ops, "to_p8") and os.environ.get("FAC_ABL_P8", "1") != "0":
        p8 = ops.to_p8(x)
        for _ in range(3):
            ops.conv1d(p8, None, C, 7, dilation=d, bias=bias, alpha_out=al, w_split=ws)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(20):
            ops.conv1d(p8, None, C, 7, dilation=d, bias=bias, alpha_out=al, w_split=ws)
        e1.record()
        torch.cuda.synchronize()
        ms8 = e0.elapsed_time(e1) / 20
        txt += f" | P8 input {ms8:.3f} ms {2.0 * B * C * C * 7 * T / ms8 / 1e9:6.1f} TF"
    out.append(txt)
print(os.path.basename(os.environ.get("FAC_LIB_PATH", "default")), " | ".join(out))
