#!/usr/bin/env python
"""One rank of the full training step with the data-parallel exchange switched on or off.

With FAC_FORCE_ALLREDUCE=1 and RANK / WORLD_SIZE=1 / MASTER_* set, facodec_amd.benchutil.init_distributed builds a one-rank RCCL
process group and FlatAdamW issues its per-key asynchronous all-reduce(AVG) exactly as on N ranks (the mean over one rank is the
identity), from the gradient hooks where the key's backward is complete.  Without the variable no collective runs.  Prints one
JSON line: losses of three iterations, parameter checksums, and per key where the exchange was launched from -- the two
modes must agree bit for bit (tests/test_rccl_one_rank.py)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from facodec_amd import benchutil, synth  # noqa: E402
from facodec_amd.commons import build_model, default_model_params  # noqa: E402
from facodec_amd.train import TrainStep  # noqa: E402


def main():
    rank, local_rank, world = benchutil.init_distributed()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    model = build_model(default_model_params())
    for k in ("encoder", "quantizer", "decoder", "discriminator"):
        synth.load_synthetic(model[k], seed=0, prefix=k + ".")
        model[k].to(dev)
    step = TrainStep(model, lr=1e-4)
    for o in step.opt.values():
        o.time_exchange = True
    B, T = 2, 12000
    masks = dict(p=torch.ones(1, B), c=torch.ones(2, B), r=torch.ones(3, B), res=torch.ones(B), dropout=False)
    masks = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in masks.items()}
    losses, reports, buckets = [], [], []
    for it in range(3):
        out = step(synth.synth_clips(B, T, seed=it).to(dev), masks=masks)
        losses.append({k: float(out[k]) for k in ("loss", "loss_d", "mel", "feature")})
        reports.append({k: v["launched"] for k, v in step.exchange_report().items()})
        buckets.append({k: [list(b) for b in v["buckets"]] for k, v in step.exchange_report().items()})
    torch.cuda.synchronize()
    sums = {k: float(step.opt[k].p.double().sum()) for k in sorted(step.opt)}
    import torch.distributed as dist
    from facodec_amd.optim import NativeRccl
    native = NativeRccl._instance
    print(json.dumps({"process_group": dist.is_initialized() and dist.get_backend(), "world": world, "losses": losses,
                      "native_rccl_calls": None if native is None else native.calls,
                      "param_sums": sums, "exchange_launched_from": reports, "bucket_launches": buckets,
                      "wait_ms": {k: v["wait_ms"] for k, v in step.exchange_report().items()}}))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
