#!/usr/bin/env python
"""Every weight-gradient launch of one configs[2] training step (B = 16, predictor heads, every chain serial) grouped by shape:
launches, total time (HIP events around operand split + GEMM + reduction), TFLOP/s, bytes of the two fp32 operands per second.
   python tools/train_wgrads.py [batch]"""
import os
import sys
from collections import OrderedDict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from facodec_amd import autograd_pred, discriminator, losses, ops, quantize, synth  # noqa: E402
from facodec_amd.commons import build_model, default_model_params  # noqa: E402
from facodec_amd.train import TrainStep  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    dev = torch.device("cuda:0")
    model = build_model(default_model_params())
    for k in ("encoder", "quantizer", "decoder", "discriminator", "fa_predictors"):
        synth.load_synthetic(model[k], seed=0, prefix=k + ".")
        model[k].to(dev)
    step = TrainStep(model, with_predictors=True)
    wave = synth.synth_clips(B, 48000, seed=0).to(dev)
    targets = bench.synthetic_predictor_targets(B, 160, dev)
    discriminator.N_STREAMS = autograd_pred.PRED_STREAMS = quantize.QUANT_STREAMS = losses.MEL_STREAMS = 1
    for _ in range(3):
        step(wave, targets=targets)
    torch.cuda.synchronize()
    recs = []
    orig = ops._bwd_weight_launch_inner

    def spy(x, dy, dw, B_, c_in, t_in, c_out, t_out, k, stride, dilation, pad_left, pad_mode, k1=0, dilation2=0, db=None):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = orig(x, dy, dw, B_, c_in, t_in, c_out, t_out, k, stride, dilation, pad_left, pad_mode, k1, dilation2, db)
        e1.record()
        recs.append(((B_, c_in, c_out, t_in, t_out, k, stride, dilation, k1), e0, e1))
        return r

    ops._bwd_weight_launch_inner = spy
    step(wave, targets=targets)
    torch.cuda.synchronize()
    ops._bwd_weight_launch_inner = orig
    groups = OrderedDict()
    for key, e0, e1 in recs:
        g = groups.setdefault(key, [0, 0.0])
        g[0] += 1
        g[1] += e0.elapsed_time(e1)
    tot = sum(g[1] for g in groups.values())
    print(f"{len(recs)} weight-gradient launches, {tot:.1f} ms (every chain serial)")
    print("    B  C_in C_out    T_in   T_out  K  s  d k1 | n   total ms   per launch   TFLOP/s   operand GB/s")
    for (B_, ci, co, ti, to, k, s, d, k1), (n, ms) in sorted(groups.items(), key=lambda kv: -kv[1][1]):
        flops = 2.0 * B_ * co * ci * k * to
        byts = 4.0 * B_ * (ci * ti + co * to)
        print(f"{B_:5d} {ci:5d} {co:5d} {ti:7d} {to:7d} {k:2d} {s:2d} {d:2d} {k1:2d} | {n:2d} {ms:9.3f} {ms / n:10.3f} {n * flops / ms / 1e9:10.1f} {n * byts / ms / 1e6:10.0f}")


if __name__ == "__main__":
    main()
