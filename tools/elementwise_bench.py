#!/usr/bin/env python
"""Bytes / time table of the HBM-bound kernels of the train step (K7 vq, K8 log-mel front-end, K11 / K12 loss
reductions, K13 anti-aliased SnakeBeta, and the elementwise backward set) at the configs[2] shapes (B = 16 x 2 s): each kernel
alone, HIP events around back-to-back launches, ALGORITHMIC bytes (every tensor the operation must read or write once, 4 bytes per
fp32 element) / time, as a fraction of 8 TB/s; with `--pmc profiles/rNN_pmc_train.json` the per-launch counter bytes of the same
kernel inside the real train step (tools/tune/pmc_train.sh: 2 x FETCH_SIZE + WRITE_SIZE, MI355X_MICROARCH.md HBM section) are
listed next to it.   python tools/elementwise_bench.py [--pmc file] [--out profiles/r06_hbm_kernels.json]"""
import argparse
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from facodec_amd import _lib, ops  # noqa: E402

B = 16


def timed(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n          # microseconds per call


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pmc", default=None)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.load()
    st = ops._stream
    p = ops._ptr
    rows = []

    def add(name, kernel, shape, nbytes, fn, note=""):
        us = timed(fn)
        rows.append({"op": name, "kernel": kernel, "shape": list(shape), "algorithmic_MB": round(nbytes / 1e6, 2), "us": round(us, 1),
                     "TBps": round(nbytes / us / 1e6, 3), "frac_of_8TBps": round(nbytes / us / 1e6 / 8.0, 3), "note": note})
        print(json.dumps(rows[-1]), flush=True)

    # ---- ResidualUnit backward (decoder block at C = 192, T = 24 000: the largest activations of the step)
    Cc, T = 192, 24000
    n = B * Cc * T
    x, dy, skip = (torch.randn(B, Cc, T, device=dev) for _ in range(3))
    alpha = torch.rand(Cc, device=dev) + 0.5
    add("snake backward + skip add + bias gradient", "snake_bwd_fused_kernel", (B, Cc, T), 16 * n,
        lambda: ops.snake_bwd_fused(x, alpha, dy, add=skip, want_bias=True), "reads x, dy, add; writes dx")
    Tp = T + 54
    dxpad = torch.randn(B, Cc, Tp, device=dev)
    add("snake backward reading a window of padded rows", "snake_bwd_fused_kernel", (B, Cc, T), 16 * n,
        lambda: ops.snake_bwd_fused(x, alpha, dxpad[:, :, 54:], add=skip, want_bias=True), "dy = rows of stride T + 54 (round 6)")
    dx = torch.empty(B, Cc, T, device=dev)
    add("un-padding copy of a data gradient (reflect fold)", "pad_fold_bwd_kernel", (B, Cc, T), 8 * n,
        lambda: _lib.check(lib.fac_pad_fold_bwd(p(dxpad), p(dx), B, Cc, T, Tp, 54, ops.PAD_REFLECT, st()), "fold"),
        "round 6: replaced by the in-place edge fold below wherever the consumer takes a row stride")
    add("in-place edge fold", "pad_fold_edges_kernel", (B, Cc, T), 12 * B * Cc * 54,
        lambda: _lib.check(lib.fac_pad_fold_edges(p(dxpad), B, Cc, T, Tp, 54, st()), "edges"), "54 mirrored samples per row")
    add("Snake forward", "snake_kernel", (B, Cc, T), 8 * n, lambda: ops.snake(x, alpha))
    add("bias gradient", "bias_grad_kernel", (B, Cc, T), 4 * n, lambda: ops.bias_grad(dy))
    add("fp32 -> three bf16 planes", "to_p8_kernel", (B, Cc, T), 10 * n, lambda: ops.to_p8(x), "4 B read, 6 B written per element")
    del x, dy, skip, dxpad, dx
    # ---- predictor heads: anti-aliased SnakeBeta at the frame rate (K13)
    Cc, T = 1024, 160
    n = B * Cc * T
    from facodec_amd import dsp
    x, dy = torch.randn(B, Cc, T, device=dev), torch.randn(B, Cc, T, device=dev)
    al, be = torch.zeros(Cc, device=dev), torch.zeros(Cc, device=dev)
    filt = dsp.kaiser_sinc_filter1d(0.25, 0.3, 12).reshape(-1).to(dev).float()        # alias_free_torch/filter.py:27-58, ratio 2
    add("anti-aliased SnakeBeta forward (K13)", "aa_snakebeta_kernel", (B, Cc, T), 8 * n, lambda: ops.aa_snakebeta(x, al, be, filt))
    dxx, da, db = torch.empty_like(x), torch.empty(Cc, device=dev), torch.empty(Cc, device=dev)
    scratch = torch.empty(2 * B * Cc * 4, device=dev)
    add("anti-aliased SnakeBeta backward (K13)", "aa_snakebeta_bwd_kernel", (B, Cc, T), 12 * n,
        lambda: _lib.check(lib.fac_aa_snakebeta_bwd(p(x), p(al), p(be), p(filt), p(dy), p(dxx), p(da), p(db), p(scratch), B, Cc, T, st()), "aa bwd"),
        "31 MB per launch: launch-latency-sized; round 5 kernel 110 us")
    del x, dy
    # ---- discriminators: masked LeakyReLU on a row-concatenated MRD band (pitch 144 of which 129 valid, 189 rows per clip of which 188)
    T = 870912
    x, dy = torch.randn(1, 32, T, device=dev), torch.randn(1, 32, T, device=dev)
    out = torch.empty_like(x)
    n = 32 * T
    for pitch, valid, label in ((144, 129, "pitch 144 (16-byte rows)"), (10, 9, "pitch 10"), (0, 0, "no mask")):
        rpg, vr = (189, 188) if pitch else (0, 0)
        Tm = T - T % (pitch * rpg) if pitch else T
        xm, dm, om = x[..., :Tm].contiguous(), dy[..., :Tm].contiguous(), out[..., :Tm].contiguous()
        nm = 32 * Tm
        add(f"masked LeakyReLU forward, {label}", "leaky_*_kernel", (1, 32, Tm), 8 * nm,
            lambda: _lib.check(lib.fac_leaky_relu(p(xm), p(None), p(om), nm, C.c_float(0.1), Tm, pitch, valid, rpg, vr, st()), "leaky"))
        add(f"masked LeakyReLU backward, {label}", "leaky_*_kernel", (1, 32, Tm), 12 * nm,
            lambda: _lib.check(lib.fac_leaky_relu(p(xm), p(dm), p(om), nm, C.c_float(0.1), Tm, pitch, valid, rpg, vr, st()), "leaky"))
    del x, dy, out
    # ---- quantizer (K7) and loss front-ends (K8 / K11 / K12)
    from facodec_amd import synth
    from facodec_amd.quantize import VectorQuantize
    D, F = 1024, 160
    q = VectorQuantize(D, 1024, 8).eval()
    synth.load_synthetic(q, seed=5)
    q = q.to(dev)
    z = torch.randn(B, D, F, device=dev)
    w_in, w_out, sc = q._weights()
    codes = torch.empty(B, F, device=dev, dtype=torch.int64)
    z_e = torch.empty(B, 8, F, device=dev)
    res, acc = torch.empty_like(z), torch.zeros_like(z)
    lp = torch.empty(B, ops.vq_loss_tiles(F), device=dev)
    add("vector-quantizer stage: in-proj, search, embed, out-proj, residual (K7)", "vq_fwd_kernel", (B, D, F), 4 * B * D * F * 4,
        lambda: ops.vq_step(z, w_in, q.in_proj.bias.detach(), q.codebook.weight.detach(), w_out, sc, q.out_proj.bias.detach(), codes,
                            residual=res, zq_acc=acc, z_e=z_e, loss_part=lp), "10.5 MB per launch: latency-sized")
    wave = torch.randn(B, 48000, device=dev)
    nf = 48000 // 128 + 1
    add("STFT framing, window 512 hop 128 (K8 / K11)", "stft_frames_kernel", (B, 512, nf), 4 * B * 48000 + 4 * B * 512 * nf,
        lambda: ops.stft_frames(wave, 512, nf, 128, 256, 0))
    spec = torch.randn(B, 2 * 257, nf, device=dev)
    add("|STFT|^p (K11)", "spec_power_kernel", (B, 257, nf), 12 * B * 257 * nf, lambda: ops.spec_power(spec, 1))
    a_, b_ = torch.rand(B, 80, nf, device=dev) + 0.1, torch.rand(B, 80, nf, device=dev) + 0.1
    outs, scr = torch.zeros(1, device=dev), torch.empty(4096, device=dev)
    add("L1 of log10 mel (K11 reduction)", "reduce_pair_stage1", (B, 80, nf), 8 * B * 80 * nf,
        lambda: ops.reduce_pair(a_, b_, outs, scr, 1, eps=1e-5, scale=1.0), "0.5 MB per launch: latency-sized")
    add("log-difference RMS (K12)", "logdiff_rms_kernel", (B, 80, nf), 8 * B * 80 * nf, lambda: ops.logdiff_rms(a_, b_, outs, scr, 1e-5))
    pmc = json.load(open(a.pmc)) if a.pmc else {}
    for r in rows:
        for k, v in pmc.items():
            if r["kernel"].replace("_*", "") .split("_kernel")[0] in k and "hbm_bytes" in v:
                r.setdefault("in_train_step", []).append({"kernel": k, "launches": v["launches"], "avg_us": v["avg_us"],
                                                          "counter_MB_per_launch": round(v["hbm_bytes"] / 1e6, 2), "counter_TBps": v.get("hbm_TBps")})
    if a.out:
        json.dump({"batch": B, "peak": "8 TB/s HBM3E (MI355X_MICROARCH.md); ~6.3 TB/s achievable (float4 copy), ~5.1 TB/s for 2 read + 2 write streams",
                   "rows": rows}, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
