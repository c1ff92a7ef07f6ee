"""The three k = 1 tail weight gradients of the train step (B = 16) on conv1d_wgrad_k1.hip, a few launches each: target of the PMC passes
of tools/tune/pmc_wk1.sh."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from facodec_amd import ops
dev = torch.device("cuda:0")
for C, T in ((64, 48000), (96, 48000), (192, 24000)):
    x = torch.randn(16, C, T, device=dev)
    dy = torch.randn(16, C, T, device=dev)
    for _ in range(4):
        ops.conv1d_bwd_weight(x, dy, 1, pad_mode=ops.PAD_ZERO, pad_left=0, want_db=True)
torch.cuda.synchronize()
