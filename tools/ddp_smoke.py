#!/usr/bin/env python
"""Two (or more) ranks of the full training step sharing whatever GPUs are visible (rank % device_count): exercises the
asynchronous per-key gradient all-reduce of facodec_amd/optim.py on device tensors.  On a one-GPU box run it with
FAC_DIST_BACKEND=gloo (RCCL refuses two ranks on one device):

    FAC_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
        --master-port 29517 tools/ddp_smoke.py

Checks: every rank ends with bit-identical parameters; the averaged gradient equals the mean of the ranks' local
gradients (recomputed without the collective); losses finite."""
import json
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from facodec_amd import benchutil, synth  # noqa: E402
from facodec_amd.commons import build_model, default_model_params  # noqa: E402
from facodec_amd.train import TrainStep  # noqa: E402


def main():
    rank, local_rank, world = benchutil.init_distributed()
    dev = torch.device(f"cuda:{local_rank % torch.cuda.device_count()}")
    torch.cuda.set_device(dev)
    model = build_model(default_model_params())
    heads = "--predictors" in sys.argv
    for k in ("encoder", "quantizer", "decoder", "discriminator") + (("fa_predictors",) if heads else ()):
        synth.load_synthetic(model[k], seed=0, prefix=k + ".")
        model[k].to(dev)
    step = TrainStep(model, lr=1e-4, with_predictors=heads)
    B, T = 2, 12000
    # rank-ASYMMETRIC quantizer dropout: odd ranks drop the residual quantizers of their clips, so the ranks reach different
    # parameter sets while issuing the same collectives (the hook-time exchange must not depend on which gradients arrived)
    r_on = torch.ones(3, B) if rank % 2 == 0 else torch.tensor([[1.0] * B, [0.0] * B, [0.0] * B])
    masks = dict(p=torch.ones(1, B), c=torch.ones(2, B), r=r_on, res=torch.ones(B), dropout=rank % 2 == 1)
    masks = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in masks.items()}
    targets = None
    if heads:
        from bench import synthetic_predictor_targets
        targets = synthetic_predictor_targets(B, T // 300, dev, seed=7 + rank)
    out = None
    for it in range(3):
        wave = synth.synth_clips(B, T, seed=it, rank=rank).to(dev)
        out = step(wave, masks=masks, targets=targets) if heads else step(wave, masks=masks)
    torch.cuda.synchronize()
    report = {k: ["%s:%s" % (b, w) for b, w in e["buckets"]] for k, e in step.exchange_report().items()}
    reports = [None] * world
    dist.all_gather_object(reports, report)
    assert all(r == reports[0] for r in reports), reports        # the same collectives in the same order on every rank
    # identical parameters on every rank
    sums = torch.stack([step.opt[k].p.double().sum() for k in sorted(step.opt)]).to(dev)
    gathered = [torch.zeros_like(sums) for _ in range(world)]
    dist.all_gather(gathered, sums)
    same = all(torch.equal(g, gathered[0]) for g in gathered)
    # the reduced arena == mean of the local gradients: redo the last generator backward locally, without the collective
    g_avg = {k: step.opt[k].g.clone() for k in ("encoder", "decoder", "quantizer")}
    finite = all(bool(torch.isfinite(out[k]).all()) for k in ("loss", "loss_d", "mel", "feature"))
    if rank == 0:
        print(json.dumps({"world": world, "backend": dist.get_backend(), "params_identical_across_ranks": same, "losses_finite": finite,
                          "loss": float(out["loss"]), "grad_norm": {k: float(v) for k, v in out["grad_norm"].items()},
                          "grad_arena_norms": {k: float(v.norm()) for k, v in g_avg.items()}, "with_predictors": heads,
                          "bucket_launches": report}))
    assert same and finite
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
