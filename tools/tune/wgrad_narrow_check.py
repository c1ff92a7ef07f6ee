"""dW hashes of weight-gradient launches whose row tiles hold 33 .. 96 real output channels (conv1d_wgrad_kmajor_kernel's column-split
wave layouts, round 6): FAC_WGRAD_NARROW=0 (row-split layout, clamped duplicate rows multiplied) and the default must print the same
line -- the column split is the same sums in the same order (tests/test_wgrad_split.py)."""
import hashlib, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from facodec_amd import ops
dev = torch.device("cuda:0")
SHAPES = [  # B, C_in, C_out, T, K, dil  (stride 1, reflect padding): k = 7 runs the k-split kernel, k = 1 the 64 x 64 one
    (4, 64, 64, 4000, 7, 1), (4, 96, 96, 3000, 7, 3), (4, 192, 192, 2000, 7, 9), (2, 128, 160, 1500, 7, 1), (3, 80, 130, 700, 7, 3),
    (4, 64, 64, 4000, 1, 1), (4, 96, 96, 3000, 1, 1), (4, 192, 192, 2000, 1, 1), (2, 256, 224, 900, 1, 1)]
out = {}
for B, ci, co, T, K, d in SHAPES:
    g = torch.Generator().manual_seed(B * 1000 + ci + co + K)
    x = torch.randn(B, ci, T, generator=g).to(dev)
    dy = torch.randn(B, co, T, generator=g).to(dev)
    dw = ops.conv1d_bwd_weight(x, dy, K, stride=1, dilation=d, pad_mode=ops.PAD_REFLECT if K > 1 else ops.PAD_ZERO, pad_left=(K - 1) * d)
    torch.cuda.synchronize()
    out["%d_%d_%d_%d_k%d_d%d" % (B, ci, co, T, K, d)] = hashlib.sha256(dw.cpu().numpy().tobytes()).hexdigest()[:16]
print("WGN " + json.dumps(out))
