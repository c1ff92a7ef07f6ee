"""Cycle breakdown inside the conv kernel's MFMA waves (FAC_PROF build only; tuning aid)."""
import ctypes, os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from facodec_amd import ops, _lib
lib = _lib.load()
dev = torch.device("cuda:0")
for (name, C, T, K, d) in (("C=768 T=960 k7", 768, 960, 7, 1), ("C=128 T=24000 k7", 128, 24000, 7, 1), ("C=192 T=24000 k1", 192, 24000, 1, 1)):
    B = 32
    x = torch.randn(B, C, T, device=dev)
    w = torch.randn(C, C, K, device=dev) * 0.01
    wp = ops.pack_conv_weight(w)
    al = torch.ones(C, device=dev)
    dbg = torch.zeros(1 << 22, dtype=torch.int64, device=dev)
    lib.fac_debug_set_buffer(ctypes.c_void_p(dbg.data_ptr()))
    for _ in range(2):
        y = ops.conv1d(x, wp, C, K, dilation=d, alpha_in=al if K == 7 else None, alpha_out=al if K == 7 else None)
    torch.cuda.synchronize()
    n_wg = ((T + 127) // 128) * ((C + 127) // 128 if C % 96 else (C // 96)) * B
    v = dbg[: n_wg * 16].reshape(n_wg, 4, 4).double()
    print(name, "WGs", n_wg, "cycles (mean over MFMA waves): first-barrier %.0f  barrier-total %.0f  loop %.0f  total %.0f" % tuple(v.mean((0, 1)).tolist()),
          " min/max total %.0f/%.0f" % (float(v[..., 3].min()), float(v[..., 3].max())))
    n_wg = int((dbg[:1 << 21].reshape(-1, 16)[:, 3] > 0).sum())
    v = dbg[: n_wg * 16].reshape(n_wg, 4, 4).double()
    l = dbg[(1 << 21): (1 << 21) + n_wg * 16].reshape(n_wg, 4, 4).double()
    print("   real WGs", n_wg, " MFMA waves: first %.0f barrier %.0f loop %.0f total %.0f" % tuple(v.mean((0, 1)).tolist()),
          "| staging waves: issue %.0f wait %.0f snake+store %.0f barrier %.0f" % tuple(l.mean((0, 1)).tolist()))
