#!/usr/bin/env python
"""Per-layer timing of the weight-gradient GEMM at the training step's shapes (B = 16 x 2 s): split-bf16 kernel
(conv1d_wgrad_split.hip) next to the fp32-MFMA kernel it replaces.  TFLOP/s = 2*B*C_out*C_in*K*T_out / time."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from facodec_amd import ops  # noqa: E402

B = 16
# name, batch, C_in, C_out, T_in, K, stride, dil, pad_left, mode
SHAPES = [
    ("dec RU 768 k7 T960", B, 768, 768, 960, 7, 1, 1, 6, ops.PAD_REFLECT),
    ("dec RU 768 k1 T960", B, 768, 768, 960, 1, 1, 1, 0, ops.PAD_ZERO),
    ("dec RU 384 k7 d3 T4800", B, 384, 384, 4800, 7, 1, 3, 18, ops.PAD_REFLECT),
    ("dec RU 192 k7 d9 T24000", B, 192, 192, 24000, 7, 1, 9, 54, ops.PAD_REFLECT),
    ("dec RU 192 k1 T24000", B, 192, 192, 24000, 1, 1, 1, 0, ops.PAD_ZERO),
    ("dec RU 96 k7 T48000", B, 96, 96, 48000, 7, 1, 1, 6, ops.PAD_REFLECT),
    ("enc RU 64 k7 T48000", B, 64, 64, 48000, 7, 1, 1, 6, ops.PAD_REFLECT),
    ("enc RU 64 k1 T48000", B, 64, 64, 48000, 1, 1, 1, 0, ops.PAD_ZERO),
    ("dec RU 96 k1 T48000", B, 96, 96, 48000, 1, 1, 1, 0, ops.PAD_ZERO),
    ("enc RU 128 k7 T24000", B, 128, 128, 24000, 7, 1, 1, 6, ops.PAD_REFLECT),
    ("enc RU 512 k7 T960", B, 512, 512, 960, 7, 1, 1, 6, ops.PAD_REFLECT),
    ("dec in 1024->1536 k7 T160", B, 1024, 1536, 160, 7, 1, 1, 6, ops.PAD_REFLECT),
    ("enc down 256->512 k10 s5", B, 256, 512, 4800, 10, 5, 1, 5, ops.PAD_REFLECT),
    ("convtr 768->384 k10 s5 (as conv)", B, 384, 768, 4800, 10, 5, 1, 0, ops.PAD_ZERO),
    ("lstm W_ih 1536 (T*BP)", 1, 1536, 6144, 160 * 32, 1, 1, 1, 0, ops.PAD_ZERO),
    ("MPD p2 128->512 k5 s3", 1, 128, 512, B * 2 * 2680, 5, 3, 1, 2, ops.PAD_ZERO),
    ("MPD p2 1024->1024 k5 s1", 1, 1024, 1024, B * 2 * 300, 5, 1, 1, 2, ops.PAD_ZERO),
    ("MRD 1024 band2 96->32 k9 s2 F128", B * 188, 96, 32, 128, 9, 2, 1, 4, ops.PAD_ZERO),
    ("MRD 512 band0 96->32 k9 s2 F13", B * 376, 96, 32, 13, 9, 2, 1, 4, ops.PAD_ZERO),
]


def main():
    dev = torch.device("cuda:0")
    out = []
    for name, b, ci, co, t_in, k, s, d, pl, mode in SHAPES:
        x = torch.randn(b, ci, t_in, device=dev)
        t_out = (t_in + (2 * pl if mode == ops.PAD_ZERO else pl) - (k - 1) * d - 1) // s + 1
        if mode == ops.PAD_REFLECT:
            t_out = -(-t_in // s)
        dy = torch.randn(b, co, t_out, device=dev)
        flops = 2.0 * b * co * ci * k * t_out
        row = {"layer": name, "gflop": round(flops / 1e9, 1)}
        for label, flag in (("split", True), ("fp32", False)):
            ops.BF16_SPLIT = flag
            for _ in range(2):
                ops.conv1d_bwd_weight(x, dy, k, stride=s, dilation=d, pad_mode=mode, pad_left=pl)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            n = 5
            for _ in range(n):
                ops.conv1d_bwd_weight(x, dy, k, stride=s, dilation=d, pad_mode=mode, pad_left=pl)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / n
            row[label + "_ms"] = round(ms, 3)
            row[label + "_tflops"] = round(flops / ms / 1e9, 1)
        out.append(row)
        print(json.dumps(row), flush=True)
    ops.BF16_SPLIT = True


if __name__ == "__main__":
    main()
