#!/usr/bin/env python
"""Where do the small launches of one train step (BASELINE.json configs[2]: 16 clips x 2 s, predictor heads on) come from?
One step under torch.profiler (CPU side, Python stacks): every ATen op that launches a kernel (copy_, fill_, zero_, add, cat, ...)
and every C-ABI call of the small pack / scale / reduce kernels is attributed to the first facodec_amd/ source line on its stack.

    python tools/train_glue_profile.py [--top 60]
"""
import argparse
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from facodec_amd import synth  # noqa: E402
from facodec_amd.commons import build_model, default_model_params  # noqa: E402
from facodec_amd.train import TrainStep  # noqa: E402
from bench import synthetic_predictor_targets  # noqa: E402

ATEN = ("aten::copy_", "aten::fill_", "aten::zero_", "aten::add", "aten::add_", "aten::cat", "aten::mul", "aten::clone",
        "aten::contiguous", "aten::sum", "aten::mean", "aten::_foreach_copy_", "aten::index", "aten::index_put_", "aten::neg",
        "aten::sub", "aten::div", "aten::zeros", "aten::zeros_like", "aten::ones", "aten::where", "aten::stack")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--top", type=int, default=60)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    model = build_model(default_model_params())
    for k in ("encoder", "quantizer", "decoder", "discriminator", "fa_predictors"):
        synth.load_synthetic(model[k], seed=0, prefix=k + ".")
        model[k].to(dev)
    step = TrainStep(model, with_predictors=True)
    wave = synth.synth_clips(16, 48000, seed=1).to(dev)
    targets = synthetic_predictor_targets(16, 160, dev, seed=3)
    for _ in range(2):
        step(wave, targets=targets)
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        step(wave, targets=targets)
        torch.cuda.synchronize()
    by_site = collections.Counter()
    by_op = collections.Counter()
    kernels = collections.Counter()
    for e in prof.events():
        if e.device_type == torch.autograd.DeviceType.CUDA:
            kernels[e.name[:80]] += 1
            continue
        if e.name not in ATEN:
            continue
        if not e.kernels:
            continue
        site = "?"
        for fr in e.stack or []:
            if "facodec_amd/" in fr and "ops.py" not in fr.split("facodec_amd/")[1][:8]:
                site = fr.split("facodec_amd/")[1]
                break
        else:
            for fr in e.stack or []:
                if "facodec_amd/" in fr:
                    site = fr.split("facodec_amd/")[1]
                    break
        by_site[(e.name, site)] += max(1, len(e.kernels))
        by_op[e.name] += max(1, len(e.kernels))
    print("device kernels in the step:", sum(kernels.values()))
    print("ATen ops that launch, by op:", dict(by_op.most_common()))
    print(f"top {a.top} (op, first facodec_amd frame):")
    for (op, site), n in by_site.most_common(a.top):
        print(f"{n:6d}  {op:24s} {site}")


if __name__ == "__main__":
    main()
