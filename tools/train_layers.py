#!/usr/bin/env python
"""Forward-conv launches (forward passes, data gradients) of one configs[2] training step grouped by shape: launches, total
time, TFLOP/s.  Weight-gradient GEMMs go through their own entry point and are listed by tools/wgrad_bench.py.
   python tools/train_layers.py [batch]"""
import os
import sys
from collections import OrderedDict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from facodec_amd import ops, synth  # noqa: E402
from facodec_amd.commons import build_model, default_model_params  # noqa: E402
from facodec_amd.train import TrainStep  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    dev = torch.device("cuda:0")
    model = build_model(default_model_params())
    for k in ("encoder", "quantizer", "decoder", "discriminator"):
        synth.load_synthetic(model[k], seed=0, prefix=k + ".")
        model[k].to(dev)
    step = TrainStep(model)
    wave = synth.synth_clips(B, 48000, seed=0).to(dev)
    step(wave)
    recs = []
    orig = ops._launch_conv

    def spy(d, what):
        buf = ops.C.create_string_buffer(96)
        ops._lib.load().fac_conv1d_variant(ops.C.byref(d), buf, 96)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        orig(d, what)
        e1.record()
        key = (d.B, d.C_in, d.C_out, d.T_out, d.K, d.K1, d.stride, d.dilation, d.dilation2, d.n_phase, d.row_phases, buf.value.decode()[:44])
        recs.append((key, 2.0 * d.B * d.n_phase * max(1, d.row_phases) * d.C_out * d.T_out * d.C_in * d.K, e0, e1))

    ops._launch_conv = spy
    step(wave)
    torch.cuda.synchronize()
    ops._launch_conv = orig
    groups = OrderedDict()
    for key, fl, e0, e1 in recs:
        g = groups.setdefault(key, [0, 0.0, 0.0])
        g[0] += 1
        g[1] += e0.elapsed_time(e1)
        g[2] += fl
    tot = sum(g[1] for g in groups.values())
    print("B C_in C_out T_out K K1 s d d2 ph rp kernel | launches  ms  TFLOP/s      (sum %.1f ms)" % tot)
    for key, (n, ms, fl) in sorted(groups.items(), key=lambda kv: -kv[1][1])[:int(os.environ.get("FAC_LAYERS_TOP", "70"))]:
        print("%3d %5d %5d %8d %2d %2d %d %d %5d %d %d %-44s | %3d %8.3f %7.1f" % (*key, n, ms, fl / ms / 1e9))


if __name__ == "__main__":
    main()
