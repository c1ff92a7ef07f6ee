"""Per-layer timing of the MFMA conv kernel on the model's real layer shapes at B=32 (tuning aid)."""
import sys, os, json
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from facodec_amd import ops

dev = torch.device("cuda:0")
B = int(os.environ.get("B", "32"))
# (name, C_in, C_out, T_in, K, stride, dil, snake_in, snake_out, res, transposed)
L = []
for C, T in ((64, 48000), (128, 24000), (256, 4800), (512, 960)):
    for d in (1, 3, 9):
        L.append((f"enc RU k7 C={C} T={T} d={d}", C, C, T, 7, 1, d, 0, 1, 0, 0))
    L.append((f"enc RU k1 C={C} T={T}", C, C, T, 1, 1, 1, 0, 0, 1, 0))
for ci, co, T, s in ((64, 128, 48000, 2), (128, 256, 24000, 5), (256, 512, 4800, 5), (512, 1024, 960, 6)):
    L.append((f"enc down {ci}->{co} s={s}", ci, co, T, 2 * s, s, 1, 0, 0, 0, 0))
L.append(("enc in 1->64 k7", 1, 64, 48000, 7, 1, 1, 0, 0, 0, 0))
L.append(("enc out 1024->1024 k3", 1024, 1024, 160, 3, 1, 1, 1, 0, 0, 0))
L.append(("dec in 1024->1536 k7", 1024, 1536, 160, 7, 1, 1, 0, 0, 0, 0))
for ci, co, T, s in ((1536, 768, 160, 6), (768, 384, 960, 5), (384, 192, 4800, 5), (192, 96, 24000, 2)):
    L.append((f"dec up {ci}->{co} s={s}", ci, co, T, 2 * s, s, 1, 0, 0, 0, 1))
for C, T in ((768, 960), (384, 4800), (192, 24000), (96, 48000)):
    for d in (1, 9):
        L.append((f"dec RU k7 C={C} T={T} d={d}", C, C, T, 7, 1, d, 0, 1, 0, 0))
    L.append((f"dec RU k1 C={C} T={T}", C, C, T, 1, 1, 1, 0, 0, 1, 0))
L.append(("dec out 96->1 k7 tanh", 96, 1, 48000, 7, 1, 1, 1, 0, 0, 0))
L.append(("lstm proj H=1024 (T=32 tile)", 1024, 4096, 32, 1, 1, 1, 0, 0, 0, 0))
L.append(("lstm proj H=1536 (T=32 tile)", 1536, 6144, 32, 1, 1, 1, 0, 0, 0, 0))

for C, T in ((64, 48000), (128, 24000), (96, 48000)):
    L.append((f"FUSED RU C={C} T={T} d=3", C, C, T, 7, 1, 3, 0, 1, 1, 2))
sel = os.environ.get("SEL")
rows = []
for (name, ci, co, T, K, s, d, sin, sout, res, tr) in L:
    if sel and sel not in name:
        continue
    Bx = 160 if "lstm" in name else B
    x = torch.randn(Bx, ci, T, device=dev)
    ai = torch.ones(ci, device=dev) if sin else None
    ao = torch.ones(co, device=dev) if sout else None
    bias = torch.zeros(co, device=dev)
    if tr == 2:
        w = torch.randn(co, ci, K, device=dev) * 0.01
        wp = ops.pack_conv_weight(w)
        w1p = ops.pack_conv_weight(torch.randn(co, ci, 1, device=dev) * 0.05)
        r = torch.randn(Bx, co, T, device=dev)
        a2 = torch.ones(co, device=dev)
        fn = lambda: ops.conv1d(x, wp, co, K, bias=bias, dilation=d, alpha_out=ao, res=r, alpha_y2=a2, w_k1=w1p, bias_k1=bias)
        flops = 2.0 * Bx * co * T * ci * (K + 1)
    elif tr:
        w = torch.randn(ci, co, K, device=dev) * 0.01
        wp = ops.pack_convtr_weight(w, None, s)
        fn = lambda: ops.conv_transpose1d(x, wp, co, s, bias=bias, alpha_in=ai)
        flops = 2.0 * Bx * co * (T * s) * ci * 2
    else:
        w = torch.randn(co, ci, K, device=dev) * 0.01
        wp = ops.pack_conv_weight(w)
        wsp = None
        if os.environ.get("SPLIT") == "1" and K == 7 and s == 1 and not sin and ci % 16 == 0 and co > 2:
            wsp, wp = ops.pack_conv_weight_split(w), None      # SPLIT=1: k7 rows on the bf16x3 kernel
            name += " [split]"
        y0 = ops.conv1d(x, wp, co, K, bias=bias, stride=s, dilation=d, alpha_in=ai, alpha_out=ao, w_split=wsp)
        r = torch.randn_like(y0) if res else None
        a2 = torch.ones(co, device=dev) if res else None   # k1 convs also emit the pre-activated copy
        fn = lambda: ops.conv1d(x, wp, co, K, bias=bias, stride=s, dilation=d, alpha_in=ai, alpha_out=ao, res=r, alpha_y2=a2, w_split=wsp)
        flops = 2.0 * Bx * co * y0.shape[-1] * ci * K
    for _ in range(int(os.environ.get('WARM', '1'))):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = int(os.environ.get('REP', '3'))
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    rows.append((name, ms, flops / ms / 1e9))
    print(f"{name:36s} {ms:9.3f} ms  {flops / ms / 1e9:7.1f} TF   {flops/1e9:8.1f} GF", flush=True)
    del x
tot = sum(r[1] for r in rows)
print("sum ms", tot)
