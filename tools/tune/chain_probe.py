"""Probe of cross-stream autograd fan-in / fan-out around A.rvq (tuning aid): a ResidualVectorQuantize in training mode runs on a
side stream (ops.run_chains) or on the caller's stream; its output feeds three consumer chains on side streams (like the
predictor heads) and a main-stream consumer; gradients must be identical."""
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from facodec_amd import autograd as A  # noqa: E402
from facodec_amd import ops, synth  # noqa: E402
from facodec_amd.layers import SConv1d  # noqa: E402
from facodec_amd.quantize import ResidualVectorQuantize  # noqa: E402

dev = torch.device("cuda:0")
m = ResidualVectorQuantize(1024, 2, 1024, 8, quantizer_dropout=0.5).train()
synth.load_synthetic(m, seed=2)
m.to(dev)
heads = [SConv1d(1024, 256, 1, causal=True, norm="weight_norm") for _ in range(4)]
for i, h in enumerate(heads):
    synth.load_synthetic(h, seed=10 + i)
    h.to(dev)
B, T = 16, 160
x0 = torch.randn(B, 1024, T, device=dev)
a = torch.randn_like(x0) * 0.1
b = torch.randn_like(x0) * 0.1
mask = torch.ones(2, B, device=dev)
w = torch.randn_like(x0)
params = list(m.parameters()) + [p for h in heads for p in h.parameters()]


def run(side_rvq, side_heads):
    for p in params:
        p.grad = None
    x = x0.clone().requires_grad_()
    h = x * 1.0
    if side_rvq:
        _, _, (zq, codes, cm, cb) = ops.run_chains([lambda: None, lambda: None, lambda: A.rvq(m, h, mask)], dev, 3)       # third chain = side stream 2, as in FAquantizer
    else:
        zq, codes, cm, cb = A.rvq(m, h, mask)
    r = A.sub_detached(h, a, b)
    chains = [lambda hd=hd: A.conv(hd, zq).pow(2).mean() for hd in heads]
    outs = ops.run_chains(chains, dev, 3) if side_heads else [c() for c in chains]
    loss = (zq * w).sum() * 1e-3 + 3.0 * cm + 2.0 * cb + (r * w).sum() * 0.5e-3 + sum(outs)
    loss.backward()
    torch.cuda.synchronize()
    return x.grad.clone(), [p.grad.clone() for p in params]


g0, p0 = run(False, False)
for sr, sh in ((True, False), (False, True), (True, True), (True, True)):
    g1, p1 = run(sr, sh)
    print("rvq on side:", sr, "heads on side:", sh, "dx rel err", float((g1 - g0).abs().max() / g0.abs().max()),
          "param max rel", max(float((u - v).abs().max() / (v.abs().max() + 1e-30)) for u, v in zip(p1, p0)))
