"""Debug aid: split-kernel conv against the fp32 tile for backward-style launches (zero pad, full correlation)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from facodec_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for (B, ci, co, T, k, d) in [(4, 64, 64, 6000, 7, 1), (4, 64, 64, 6000, 7, 9), (4, 96, 96, 6000, 7, 3), (4, 128, 128, 3000, 7, 9),
                             (1, 128, 32, 8016, 5, 1), (1, 128, 512, 2672, 5, 1), (2, 144, 80, 1000, 7, 1), (4, 128, 128, 600, 7, 3)]:
    x = torch.randn(B, ci, T, generator=g).to(dev)
    w = (torch.randn(co, ci, k, generator=g) / (ci * k) ** 0.5).to(dev)
    tp = T + (k - 1) * d
    ws = ops.pack_conv_weight_split(w)
    y_s = ops.conv1d(x, None, co, k, dilation=d, pad_left=(k - 1) * d, pad_mode=ops.PAD_ZERO, t_out=tp, w_split=ws)
    y_f = ops.conv1d(x, ops.pack_conv_weight(w), co, k, dilation=d, pad_left=(k - 1) * d, pad_mode=ops.PAD_ZERO, t_out=tp)
    y64 = torch.nn.functional.conv1d(torch.nn.functional.pad(x.double().cpu(), ((k - 1) * d, (k - 1) * d)), w.double().cpu(), dilation=d)
    e_s = float((y_s.cpu().double() - y64).abs().max() / y64.abs().max())
    e_f = float((y_f.cpu().double() - y64).abs().max() / y64.abs().max())
    bad = (y_s.cpu().double() - y64).abs().amax((0, 1))
    print((B, ci, co, T, k, d), "split err %.2e  fp32 err %.2e" % (e_s, e_f), "worst t:", int(bad.argmax()), "of", tp)
