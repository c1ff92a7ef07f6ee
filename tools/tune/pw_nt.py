"""conv1d_pw_kernel on the forward's ResidualUnit tails (B = 32): plain stores vs nontemporal stores (-DFAC_PW_NT_STORES build)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from facodec_amd import ops, _lib
_lib.load()
dev = torch.device("cuda:0")
out = []
for (C, T) in ((64, 48000), (96, 48000), (128, 24000), (192, 24000), (256, 4800), (384, 4800)):
    B = 32
    x = torch.randn(B, C, T, device=dev)
    r = torch.randn(B, C, T, device=dev)
    w = ops.pack_conv_weight(torch.randn(C, C, 1, device=dev) * 0.05)
    a2 = torch.ones(C, device=dev)
    bias = torch.zeros(C, device=dev)
    fn = lambda: ops.conv1d(x, w, C, 1, bias=bias, res=r, alpha_y2=a2)  # noqa: E731
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    out.append(f"C={C}: {ms:.3f} ms {4.0 * B * C * T * 4 / ms / 1e9:5.2f} TB/s")
print(os.path.basename(os.environ.get("FAC_LIB_PATH", "default")), " | ".join(out))
