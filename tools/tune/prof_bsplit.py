"""Wall-clock phases of the split conv kernel's workgroups (FAC_PROF build; tuning aid)."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from facodec_amd import ops, _lib
lib = _lib.load()
dev = torch.device("cuda:0")
for (C, T) in ((96, 48000), (192, 24000), (768, 960)):
    B = 32
    x = torch.randn(B, C, T, device=dev)
    w = torch.randn(C, C, 7, device=dev) * 0.01
    ws = ops.pack_conv_weight_split(w)
    al = torch.ones(C, device=dev)
    bias = torch.zeros(C, device=dev)
    dbg = torch.zeros(1 << 20, dtype=torch.int64, device=dev)
    for _ in range(3):
        ops.conv1d(x, None, C, 7, bias=bias, alpha_out=al, w_split=ws)
    torch.cuda.synchronize()
    lib.fac_debug_set_buffer(ctypes.c_void_p(dbg.data_ptr()))
    ops.conv1d(x, None, C, 7, bias=bias, alpha_out=al, w_split=ws)
    torch.cuda.synchronize()
    lib.fac_debug_set_buffer(ctypes.c_void_p(0))
    d = dbg.cpu().numpy().reshape(-1, 8)
    d = d[d[:, 3] > 0]
    t0, t1, t2, t3 = (d[:, i].astype(np.float64) for i in range(4))
    us = 1e6 / 100e6     # wall_clock64: 100 MHz
    print(f"C={C} T={T}: WGs {len(d)}  kernel span {(t3.max() - t0.min()) * us:.0f} us | per WG: prologue {np.mean(t1 - t0) * us:.1f}  loop {np.mean(t2 - t1) * us:.1f}  epilogue {np.mean(t3 - t2) * us:.1f}  total {np.mean(t3 - t0) * us:.1f} us")
    continue
    # gaps between consecutive workgroups on the same CU (hw id + xcc)
    key = d[:, 4] * 16 + d[:, 5]
    gaps = []
    for k in np.unique(key):
        sel = d[key == k]
        sel = sel[np.argsort(sel[:, 0])]
        if len(sel) > 1:
            gaps.append((sel[1:, 0] - sel[:-1, 3]).astype(np.float64))
    g = np.concatenate(gaps) * us
    print(f"    distinct (hw_id, xcc) {len(np.unique(key))}; gap between a WG's end and the next start on the same id: mean {g.mean():.1f} us, median {np.median(g):.1f}, p90 {np.percentile(g, 90):.1f}")
