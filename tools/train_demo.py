#!/usr/bin/env python
"""A few generator steps on a fixed synthetic batch (smoke run of forward, backward, clipping and the fused AdamW; the
gradients themselves are checked against autograd in tests/test_gpu_parity.py).  With formula-generated weights and
hard VQ assignments the loss of the first steps is noisy -- this prints the trajectory, it asserts only finiteness.
python tools/train_demo.py [--steps 30] [--batch 4] [--lr 1e-4]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from facodec_amd import synth  # noqa: E402
from facodec_amd.commons import build_model, default_model_params  # noqa: E402
from facodec_amd.train import GeneratorStep  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--lr", type=float, default=1e-4)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = build_model(default_model_params())
    for k in ("encoder", "quantizer", "decoder"):
        synth.load_synthetic(model[k], seed=0, prefix=k + ".")
        model[k].to(dev)
    step = GeneratorStep(model, lr=a.lr)
    wave = synth.synth_clips(a.batch, 24000, seed=3).to(dev)
    B = a.batch
    masks = dict(p=torch.ones(1, B), c=torch.ones(2, B), r=torch.ones(3, B), res=torch.ones(B), dropout=False)
    hist = []
    for i in range(a.steps):
        out = step(wave, masks)
        hist.append((float(out["loss"]), float(out["mel"]), float(out["commitment"])))
        if i % 5 == 0 or i == a.steps - 1:
            print(f"step {i:3d}  loss {hist[-1][0]:9.4f}  mel {hist[-1][1]:7.4f}  commitment {hist[-1][2]:8.4f}  "
                  f"|g| enc {float(out['grad_norm']['encoder']):8.2f} dec {float(out['grad_norm']['decoder']):8.2f}", flush=True)
    first, last = sum(h[0] for h in hist[:3]) / 3, sum(h[0] for h in hist[-3:]) / 3
    print(f"mean loss first 3 steps {first:.4f} -> last 3 steps {last:.4f}")
    assert all(h[0] == h[0] and abs(h[0]) < 1e6 for h in hist), "loss diverged"


if __name__ == "__main__":
    main()
