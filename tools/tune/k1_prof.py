import ctypes, os, sys
import torch
sys.path.insert(0, "/root/repo")
from facodec_amd import ops, _lib
lib = _lib.load()
dev = torch.device("cuda:0")
for (name, C, T, K) in (("C=192 T=24000 k1", 192, 24000, 1), ("C=384 T=4800 k1", 384, 4800, 1), ("C=96 T=48000 k1", 96, 48000, 1)):
    B = 32
    x = torch.randn(B, C, T, device=dev)
    res = torch.randn(B, C, T, device=dev)
    w = torch.randn(C, C, K, device=dev) * 0.01
    wp = ops.pack_conv_weight(w)
    al = torch.ones(C, device=dev)
    dbg = torch.zeros(1 << 22, dtype=torch.int64, device=dev)
    lib.fac_debug_set_buffer(ctypes.c_void_p(dbg.data_ptr()))
    for _ in range(2):
        y = ops.conv1d(x, wp, C, K, res=res, alpha_y2=al)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); y = ops.conv1d(x, wp, C, K, res=res, alpha_y2=al); e1.record(); torch.cuda.synchronize()
    n_wg = int((dbg[:1 << 21].reshape(-1, 16)[:, 3] > 0).sum())
    v = dbg[: n_wg * 16].reshape(n_wg, 4, 4).double()
    l = dbg[(1 << 21): (1 << 21) + n_wg * 16].reshape(n_wg, 4, 4).double()
    print(name, "%.3f ms" % e0.elapsed_time(e1), " WGs", n_wg, " MFMA waves: first %.0f barrier %.0f loop-end %.0f total %.0f" % tuple(v.mean((0, 1)).tolist()),
          "| staging waves: issue %.0f wait %.0f store %.0f barrier %.0f" % tuple(l.mean((0, 1)).tolist()))
