#!/usr/bin/env python
"""Generator training step timing (BASELINE.json configs[2] shape: batch 16 x 2 s per GPU; generator half only --
see facodec_amd/train.py for what the step covers this round).

    python tools/train_bench.py [--batch 16] [--steps 3]
    python -m torch.distributed.run --nproc-per-node N tools/train_bench.py   (one rank per GPU, RCCL all-reduce)
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from facodec_amd import benchutil, synth  # noqa: E402
from facodec_amd.commons import build_model, default_model_params  # noqa: E402
from facodec_amd.train import GeneratorStep, TrainStep  # noqa: E402


def stale_packs():
    """TIMING EXPERIMENT ONLY (FAC_EXP_STALE_PACKS=1, used through tools/train_ab.py): memoises the weight-norm scales and packed
    weight layouts ACROSS steps by monkeypatching facodec_amd.ops from here -- the weights go stale, the results are WRONG -- to
    measure what the ~1 500 small scale / pack launches of a step cost in wall time.  Lives in this tool, not in the
    product module, so that no environment variable can freeze the weights of a real training run."""
    import functools
    from facodec_amd import ops
    memo = {}

    def wrap(fn):
        @functools.wraps(fn)
        def wrapped(*args, **kw):
            key = [fn.__name__]
            for a in list(args) + [kw.get(k) for k in sorted(kw) if k not in ("out", "scale")]:
                key.append((a.data_ptr(), tuple(a.shape)) if torch.is_tensor(a) else a)
            key = tuple(key)
            if key not in memo:
                kw.pop("out", None)
                memo[key] = fn(*args, **kw)
            return memo[key]
        return wrapped

    for name in ("wn_scale", "pack_conv_weight", "pack_convtr_weight", "pack_convtr_weight_rows", "pack_gemm_weight_split",
                 "pack_conv_weight_split2", "pack_convtr_weight_rows_split", "pack_conv_weight_split"):
        setattr(ops, name, wrap(getattr(ops, name)))
    print("[train_bench] FAC_EXP_STALE_PACKS=1: weight scales / packs memoised across steps -- RESULTS ARE WRONG, timing only", file=sys.stderr)


def main():
    if os.environ.get("FAC_EXP_STALE_PACKS") == "1":
        stale_packs()
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--generator-only", action="store_true", help="skip the discriminator step and the GAN terms")
    ap.add_argument("--predictors", action="store_true", help="include the predictor heads with synthetic targets (bench.py's step)")
    a = ap.parse_args()
    rank, local_rank, world = benchutil.init_distributed()
    dev = torch.device(f"cuda:{local_rank % torch.cuda.device_count()}")
    torch.cuda.set_device(dev)
    model = build_model(default_model_params())
    for k in ("encoder", "quantizer", "decoder", "discriminator"):
        synth.load_synthetic(model[k], seed=0, prefix=k + ".")
        model[k].to(dev)
    kw = {}
    if a.predictors:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from bench import synthetic_predictor_targets
        synth.load_synthetic(model.fa_predictors, seed=0, prefix="fa_predictors.")
        model.fa_predictors.to(dev)
        kw["targets"] = synthetic_predictor_targets(a.batch, 160, dev, seed=3 + rank)
    step = GeneratorStep(model) if a.generator_only else TrainStep(model, with_predictors=a.predictors)
    wave = synth.synth_clips(a.batch, 48000, seed=0, rank=rank).to(dev)
    for _ in range(a.warmup):
        out = step(wave, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = step(wave, **kw)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    if rank == 0:
        print(json.dumps({"metric": ("generator training step (encoder + FA-quantizer + decoder fwd/bwd, mel + VQ losses, AdamW)"
                                     if a.generator_only else
                                     "training step: discriminator step + generator step (mel + feature matching + adversarial + VQ losses), 4 x AdamW"),
                          "batch_per_gpu": a.batch, "n_gpus": world, "ms_per_step": round(1e3 * dt, 1),
                          "audio_s_per_s": round(world * a.batch * 2.0 / dt, 2), "loss": float(out["loss"]),
                          "mel": float(out["mel"]), "grad_norm": {k: float(v) for k, v in out["grad_norm"].items()},
                          "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)}))
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
