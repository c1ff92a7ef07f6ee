"""Timing of conv1d_bsplit2_kernel on the multi-resolution discriminator's layer shapes (row-concatenated two-level-tap convs,
32 -> 32 channels) -- run against ablation builds (FAC_BUILD_TAG / FAC_EXTRA_FLAGS=-DFAC_ABL2_NOSPLIT | -DFAC_ABL2_NOSTAGE_X, selected
with FAC_LIB_PATH) to see what the fp32 -> bf16 x 3 split in the staging waves costs (= what plane inputs could buy)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from facodec_amd import ops, _lib
_lib.load()
dev = torch.device("cuda:0")
out = []
# (columns out, k1, K2, pitch, stride)
for (t_out, k1, k2n, pitch, stride) in ((870912, 9, 3, 144, 1), (435456, 9, 3, 144, 2), (962560, 9, 3, 80, 1), (120320, 3, 3, 10, 1)):
    k = k1 * k2n
    t_in = (t_out - 1) * stride + (k2n - 1) * pitch + k1
    x = torch.randn(1, 32, t_in, device=dev)
    w = torch.randn(32, 32, k, device=dev) * 0.05
    ws = ops.pack_conv_weight_split2(w, None, k1)
    bias = torch.zeros(32, device=dev)
    fn = lambda: ops.conv1d(x, None, 32, k, bias=bias, stride=stride, pad_left=0, pad_mode=ops.PAD_ZERO, t_out=t_out, w_split=ws, k1=k1, dilation2=pitch)  # noqa: E731
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
    out.append(f"k1={k1} s={stride} P={pitch} T={t_out}: {ms:.3f} ms {2.0 * 32 * 32 * k * t_out / ms / 1e9:6.1f} TF")
print(os.path.basename(os.environ.get("FAC_LIB_PATH", "default")), " | ".join(out))
