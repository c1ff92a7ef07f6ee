"""Where a stage of the split k = 7 kernel spends its time (FAC_PROF2 build: s_memtime stamps in staging wave 0 and MFMA wave 0 of
every workgroup, steady-state stages only).  Prints averages per stage in shader-clock cycles."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from facodec_amd import ops, _lib
lib = _lib.load()
dev = torch.device("cuda:0")
for (C, T, d) in ((192, 24000, 3), (768, 960, 1)):
    B = 32
    x = torch.randn(B, C, T, device=dev)
    w = torch.randn(C, C, 7, device=dev) * 0.01
    ws = ops.pack_conv_weight_split(w)
    al = torch.ones(C, device=dev)
    bias = torch.zeros(C, device=dev)
    dbg = torch.zeros(1 << 21, dtype=torch.int64, device=dev)
    for _ in range(40):          # sustained load: the clock settles
        ops.conv1d(x, None, C, 7, dilation=d, bias=bias, alpha_out=al, w_split=ws)
    torch.cuda.synchronize()
    lib.fac_debug_set_buffer(ctypes.c_void_p(dbg.data_ptr()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.conv1d(x, None, C, 7, dilation=d, bias=bias, alpha_out=al, w_split=ws)
    e1.record()
    torch.cuda.synchronize()
    lib.fac_debug_set_buffer(ctypes.c_void_p(0))
    a = dbg.cpu().numpy().reshape(-1, 16).astype(np.float64)
    a = a[a[:, 4] > 0]
    ghz = a[:, 3].sum() / a[:, 4].sum() * 0.1
    stages = a[:, 5].sum()
    print(f"   shader clock over the main loops: {ghz:.3f} GHz (s_memtime / s_memrealtime); {a[:, 3].sum() / stages:.0f} shader cycles per stage "
          f"(168 MFMAs x 32 cycles = 5376)")
    a = a[a[:, 2] > 0] if (a[:, 2] > 0).any() else a
    m = a[:, :3].sum(0)
    s = a[:, 8:16].sum(0)
    n_m, n_s = max(m[2], 1), max(s[7], 1)
    names = ["wait inputs landed", "take (24 v_cndmask)", "issue 24 loads", "split + 9 ds_write (+lgkm)", "wait weights landed", "11 ds_write W + lgkm", "barrier wait"]
    print(f"C={C} T={T} d={d}: kernel {e0.elapsed_time(e1):.3f} ms, {len(a)} WGs")
    print(f"   MFMA wave 0: compute {m[0] / n_m:8.0f} cyc/stage, barrier wait {m[1] / n_m:8.0f}")
    print("   staging wave 0: " + " | ".join(f"{nm} {s[i] / n_s:.0f}" for i, nm in enumerate(names)) + f" | total {s[:7].sum() / n_s:.0f}")
