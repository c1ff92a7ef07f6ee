"""Runs every HIP op against the CPU oracle and prints a table (keeps going after a failure) --
a single gpurun call then tells us everything that is wrong.  Not part of the product."""
import json
import os
import sys
import time
import traceback

import numpy as np
import torch
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from facodec_amd import ops, synth  # noqa: E402
from oracle import facodec_oracle as O  # noqa: E402

dev = torch.device("cuda:0")
rows = []


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def run(name, fn):
    t = time.time()
    try:
        r = fn()
        rows.append((name, r, time.time() - t))
        print(f"{name:48s} {r}", flush=True)
    except Exception as e:  # noqa: BLE001
        rows.append((name, "EXC " + repr(e)[:200], time.time() - t))
        print(f"{name:48s} EXC {e!r}", flush=True)
        traceback.print_exc()


g = torch.Generator().manual_seed(0)


def rnd(*s):
    return torch.randn(*s, generator=g)


def conv_case(B, ci, co, T, k, s=1, d=1, snake_in=False, snake_out=False, res=False, act=0, pad_mode="reflect"):
    def f():
        x = rnd(B, ci, T)
        w = rnd(co, ci, k) / (ci * k) ** 0.5
        b = rnd(co) * 0.1
        ai = 1 + 0.2 * torch.rand(ci, generator=g) if snake_in else None
        ao = 1 + 0.2 * torch.rand(co, generator=g) if snake_out else None
        xi = O.snake(x, ai.view(1, -1, 1)) if snake_in else x
        y = O.sconv1d(xi, w, b, stride=s, dilation=d, causal=True, pad_mode=pad_mode)
        if snake_out:
            y = O.snake(y, ao.view(1, -1, 1))
        if act == 1:
            y = torch.tanh(y)
        r = rnd(*y.shape) if res else None
        if res:
            y = y + r
        wp = ops.pack_conv_weight(w.to(dev))
        yg = ops.conv1d(x.to(dev), wp, co, k, bias=b.to(dev), stride=s, dilation=d,
                        pad_mode=ops.PAD_REFLECT if pad_mode == "reflect" else ops.PAD_ZERO,
                        alpha_in=ai.to(dev) if snake_in else None, alpha_out=ao.to(dev) if snake_out else None,
                        res=r.to(dev) if res else None, act=act)
        torch.cuda.synchronize()
        assert yg.shape == y.shape, (yg.shape, y.shape)
        return rel(yg, y)
    return f


run("conv k7 64->64 T=1000", conv_case(2, 64, 64, 1000, 7))
run("conv k7 d3 128->128 snake in/out", conv_case(2, 128, 128, 777, 7, d=3, snake_in=True, snake_out=True))
run("conv k7 d9 96->96 T=500", conv_case(1, 96, 96, 500, 7, d=9, snake_in=True))
run("conv k1 192->192 res", conv_case(2, 192, 192, 300, 1, res=True))
run("conv k1 1->64 k7 first", conv_case(2, 1, 64, 2400, 7))
run("conv k4 s2 64->128", conv_case(2, 64, 128, 2400, 4, s=2, snake_in=True))
run("conv k10 s5 128->256", conv_case(2, 128, 256, 1200, 10, s=5, snake_in=True))
run("conv k12 s6 512->1024", conv_case(1, 512, 1024, 960, 12, s=6))
run("conv k3 1024->1024 T=160", conv_case(1, 1024, 1024, 160, 3, snake_in=True))
run("conv k7 96->1 tanh", conv_case(2, 96, 1, 3000, 7, snake_in=True, act=1))
run("conv k7 short T=5 (pad>len)", conv_case(1, 8, 16, 5, 7))
run("conv k5 zero-pad none 20->256 T=33", conv_case(3, 20, 256, 33, 1))
run("conv k7 1024->1536 T=160", conv_case(1, 1024, 1536, 160, 7))
run("conv k1 T=32 (narrow tile) 256->512", conv_case(5, 256, 512, 32, 1))


def convtr_case(B, ci, co, T, s):
    def f():
        x = rnd(B, ci, T)
        v = rnd(ci, co, 2 * s) / (ci * 2) ** 0.5
        gg = torch.rand(ci, 1, 1, generator=g) + 0.5
        b = rnd(co) * 0.1
        al = 1 + 0.2 * torch.rand(ci, generator=g)
        w = O.weight_norm_weight(v, gg)
        y = O.sconvtr1d(O.snake(x, al.view(1, -1, 1)), w, b, s, causal=True)
        wp = ops.pack_convtr_weight(v.to(dev), gg.to(dev), s)
        yg = ops.conv_transpose1d(x.to(dev), wp, co, s, bias=b.to(dev), alpha_in=al.to(dev))
        torch.cuda.synchronize()
        assert yg.shape == y.shape, (yg.shape, y.shape)
        return rel(yg, y)
    return f


run("convtr s6 256->128 T=160", convtr_case(2, 256, 128, 160, 6))
run("convtr s5 128->64 T=333", convtr_case(2, 128, 64, 333, 5))
run("convtr s2 192->96 T=1000", convtr_case(1, 192, 96, 1000, 2))


def wn_case():
    v = rnd(96, 48, 7)
    gg = torch.rand(96, 1, 1, generator=g) + 0.5
    w = O.weight_norm_weight(v, gg)
    wp = ops.pack_conv_weight(v.to(dev), gg.to(dev))
    torch.cuda.synchronize()
    return rel(wp[:, :, :96].permute(2, 0, 1), w)


run("weight-norm pack", wn_case)


def lstm_case(B, H, T, L=2):
    def f():
        from facodec_amd.layers import SLSTM
        m = SLSTM(H, L)
        sd = synth.load_synthetic(m, seed=5)
        x = rnd(B, H, T)
        y = O.slstm(x, {k: v for k, v in sd.items()}, "lstm.", L)
        m = m.to(dev)
        with torch.no_grad():
            yg = m(x.to(dev))
        torch.cuda.synchronize()
        return rel(yg, y)
    return f


run("slstm B=3 H=128 T=20", lstm_case(3, 128, 20))
run("slstm B=32 H=256 T=40", lstm_case(32, 256, 40))
run("slstm B=2 H=1024 T=16", lstm_case(2, 1024, 16))


def vq_search_case():
    d = np.load(os.path.join(REPO, "tests/golden/vq_kat.npz"))
    cb = torch.from_numpy(d["codebook"])
    lat = torch.from_numpy(d["latents"])  # (4, 8, 300)
    idx = ops.vq_search(lat.permute(0, 2, 1).reshape(-1, 8).contiguous().to(dev), cb.to(dev)).cpu().reshape(4, 300)
    mism = int((idx != torch.from_numpy(d["indices"].astype(np.int64))).sum())
    gk = np.random.Generator(np.random.Philox(key=1234))
    gk.standard_normal((1024, 8)); gk.standard_normal((4, 8, 300))
    big = torch.from_numpy(gk.standard_normal((1, 8, 1 << 18)).astype(np.float32))
    idx2 = ops.vq_search(big[0].t().contiguous().to(dev), cb.to(dev)).cpu()
    mism2 = int((idx2 != torch.from_numpy(d["sweep_indices"].astype(np.int64))).sum())
    return f"kat mismatches {mism}, sweep mismatches {mism2} / {1 << 18}"


run("vq search KAT + 262k sweep", vq_search_case)


def vq_fwd_case():
    from facodec_amd.quantize import ResidualVectorQuantize
    m = ResidualVectorQuantize(256, 3, 1024, 8).eval()
    sd = synth.load_synthetic(m, seed=2)
    z = rnd(3, 256, 150)
    zq, codes, lat, cm, cb = O.rvq_forward(z, sd, "", 3, 3)
    m = m.to(dev)
    with torch.no_grad():
        zq_g, codes_g, lat_g, cm_g, cb_g = m(z.to(dev), 3)
    torch.cuda.synchronize()
    mism = int((codes_g.cpu() != codes).sum())
    return f"codes mism {mism}/{codes.numel()}, zq rel {rel(zq_g, zq):.2e}, lat rel {rel(lat_g, lat):.2e}, commit rel {abs(float(cm_g) - float(cm)) / float(cm):.2e}"


run("rvq forward 3 codebooks", vq_fwd_case)


def small_model_case():
    from facodec_amd.dac_model import Encoder, Decoder
    d = np.load(os.path.join(REPO, "tests/golden/small_layers.npz"))
    enc = Encoder(d_model=8, strides=[2, 5, 5, 6], d_latent=64, causal=True, lstm=2)
    dec = Decoder(input_channel=64, channels=128, rates=[6, 5, 5, 2], causal=True, lstm=2)
    synth.load_synthetic(enc, seed=1, prefix="encoder.")
    synth.load_synthetic(dec, seed=1, prefix="decoder.")
    enc, dec = enc.to(dev), dec.to(dev)
    x = torch.from_numpy(d["x"]).to(dev)
    with torch.no_grad():
        z = enc(x)
        y = dec(torch.from_numpy(d["z"]).to(dev))
    torch.cuda.synchronize()
    return f"enc rel {rel(z, torch.from_numpy(d['z'])):.2e}  dec rel {rel(y, torch.from_numpy(d['y'])):.2e}"


run("small encoder/decoder vs reference golden", small_model_case)


def logmel_case():
    from facodec_amd.quantize import LogMelFrontend
    fe = LogMelFrontend().to(dev)
    w = synth.synth_clips(2, 48000, seed=0)
    ref = O.logmel_frontend(w, 80)
    with torch.no_grad():
        out = fe(w.to(dev))
    torch.cuda.synchronize()
    return f"rel {rel(out, ref):.2e} absmax {float((out.cpu() - ref).abs().max()):.2e}"


run("log-mel front-end", logmel_case)


def e2e_case():
    from facodec_amd.commons import build_model, default_model_params
    d = np.load(os.path.join(REPO, "tests/golden/codec_e2e.npz"))
    model = build_model(default_model_params())
    for k in model:
        synth.load_synthetic(model[k], seed=0, prefix=k + ".")
        model[k].eval().to(dev)
    wave = synth.synth_clips(2, 48000, seed=0).to(dev)
    with torch.no_grad():
        t0 = time.time()
        z = model.encoder(wave)
        outs, quantized, commit, cbl, timbre, codes = model.quantizer(z, wave, n_c=2, return_codes=True)
        y = model.decoder(outs)
        torch.cuda.synchronize()
        dt = time.time() - t0
    res = {}
    res["z"] = rel(z[:, ::8], torch.from_numpy(d["z_probe"]))
    res["outs"] = rel(outs[:, ::8], torch.from_numpy(d["outs_probe"]))
    res["timbre"] = rel(timbre, torch.from_numpy(d["timbre"]))
    res["wave"] = rel(y[:, 0, torch.from_numpy(d["probe_t"])], torch.from_numpy(d["wave_probe"]))
    for nm, c in zip(("codes_p", "codes_c", "codes_r"), codes):
        res[nm + "_mism"] = int((c.cpu() != torch.from_numpy(d[nm].astype(np.int64))).sum())
    res["commit"] = abs(float(commit) - float(d["commitment"])) / float(d["commitment"])
    res["secs"] = round(dt, 3)
    return json.dumps(res)


run("end-to-end real config vs reference golden", e2e_case)

os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
json.dump([(n, str(r), t) for n, r, t in rows], open(os.path.join(REPO, "gpurun_out", "gpu_check.json"), "w"), indent=1)
