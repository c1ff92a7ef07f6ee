#!/usr/bin/env python
"""bench.py -- FAcodec encode -> FA-quantizer -> decode throughput on MI355X.

Metric (BASELINE.json): 24 kHz audio-seconds encoded+decoded per wall-second.
Workload at N=1 = BASELINE.json configs[1]: batch 32 x 2 s @ 24 kHz, forward-only
encoder -> FVQ (6 codebooks, codes returned) -> decoder, synthetic clips resident in HBM,
formula-generated weights of the shipped architecture (configs/config.yml model_params),
fp32 throughout (fp32 MFMA; bf16 cannot hold bit-exact codes -- SURVEY.md 0.5).
N>1: one process per GPU (torch.distributed.run), every rank runs the same per-GPU batch on its
own clips (weak scaling, no data-path collective: clips are independent units).

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for how each field is computed).
"""
import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

from facodec_amd import benchutil, ops, synth  # noqa: E402
from facodec_amd.commons import build_model, default_model_params  # noqa: E402

CLIP_SECONDS = 2.0
SAMPLE_RATE = 24000
# Algorithmic work of the forward path per 2 s clip (SURVEY.md 8d / BASELINE.md 2, hook-counted on
# the reference): 118.44 GMAC = 236.88 GFLOP, i.e. 118.44 GFLOP per audio-second.
FLOP_PER_AUDIO_S = 118.44e9
FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
BF16_MFMA_PEAK_TFLOPS = 2516.6  # MI355X_MICROARCH.md: dense bf16 MFMA peak
# conv1d_bsplit.hip computes every fp32 product as six exact bf16 products (fp32 accumulate): its roofline is the
# bf16 pipe running 6x the algorithmic FLOPs, i.e. 2516.6 / 6 fp32-equivalent TFLOP/s.
SPLIT_PEAK_TFLOPS = BF16_MFMA_PEAK_TFLOPS / 6.0


def build(device, seed=0):
    model = build_model(default_model_params())
    for k in ("encoder", "quantizer", "decoder"):        # the forward path; predictors are train-step only
        synth.load_synthetic(model[k], seed=seed, prefix=k + ".")
        model[k].eval().to(device)
    return model


def make_step(model, wave):
    def step():
        with torch.no_grad():
            z = model.encoder(wave)
            outs, quantized, commit, cbl, timbre, codes = model.quantizer(z, wave, n_c=2, return_codes=True)
            y = model.decoder(outs)
        return y, codes
    return step


def cpu_baseline(batch=4, passes=5):
    """The CPU oracle (port of the reference algorithm, pinned to it by tests/golden) timed on this
    host's cores over a bounded sample of the same workload."""
    from oracle import facodec_oracle as O
    model = build_model(default_model_params())
    sds = {k: synth.synth_state_dict(synth.param_shapes(model[k]), 0, k + ".") for k in ("encoder", "quantizer", "decoder")}
    del model
    wave = synth.synth_clips(batch, int(CLIP_SECONDS * SAMPLE_RATE), seed=0)
    # thread sweep on the GPU box's host (tests/tools/cpu_thread_sweep.py: 8/16/32/64/128 threads ->
    # 1.91/2.14/1.79/1.21/0.48 audio-s/s): 16 threads is the oracle's best case, so that is the baseline
    prev_threads = torch.get_num_threads()
    threads = min(16, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    with torch.no_grad():
        O.codec_forward(sds, wave[:1], n_c=2)  # warm-up (page-in, oneDNN primitive cache)
        t0 = time.perf_counter()
        for _ in range(passes):
            O.codec_forward(sds, wave, n_c=2)
        dt = time.perf_counter() - t0
    torch.set_num_threads(prev_threads)
    return dict(value=round(batch * CLIP_SECONDS * passes / dt, 3), unit="audio-s/s", cores=threads, kind="port",
                sample=f"{passes} passes of {batch} clips x 2 s (oracle/facodec_oracle.py codec_forward, torch-CPU fp32, "
                       f"{threads} threads, {os.cpu_count()} logical cores on host), {dt:.1f} s of CPU work")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="clips per GPU per step (configs[1]: 32)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true", help="skip per-launch HIP-event timing of the conv kernel")
    args = ap.parse_args()

    rank, local_rank, world = benchutil.init_distributed()
    if world != args.gpus and rank == 0:
        print(f"[bench] note: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    device = torch.device(f"cuda:{local_rank % torch.cuda.device_count()}")
    torch.cuda.set_device(device)

    model = build(device)
    n_samples = int(CLIP_SECONDS * SAMPLE_RATE)
    wave = synth.synth_clips(args.batch, n_samples, seed=0, rank=rank).to(device)   # resident in HBM
    step = make_step(model, wave)
    sync = torch.cuda.synchronize

    prof = None
    if not args.no_roofline:
        # warm-up happens un-instrumented; events are recorded over the timed region only
        for _ in range(args.warmup):
            step()
        sync()
        prof = ops.ConvLaunchProfile()
        ops.set_conv_profile(prof)
        elapsed = benchutil.timed_steps(step, args.steps, 0, sync, device)
        ops.set_conv_profile(None)
    else:
        elapsed = benchutil.timed_steps(step, args.steps, args.warmup, sync, device)

    units = benchutil.aggregate_units(args.batch * CLIP_SECONDS * args.steps, device)
    value = units / elapsed

    fp32_ref = None
    if ops.BF16_SPLIT and not args.no_roofline:
        # the same step with every conv on the fp32 matrix pipe (FAC_BF16_SPLIT=0), for reference; all ranks
        ops.BF16_SPLIT = False
        try:
            n_ref = max(2, args.steps // 3)
            el = benchutil.timed_steps(step, n_ref, 1, sync, device)
            fp32_ref = {"value": round(world * args.batch * CLIP_SECONDS * n_ref / el, 2), "unit": "audio-s/s",
                        "ms_per_step": round(1e3 * el / n_ref, 3), "steps": n_ref,
                        "note": "same step with FAC_BF16_SPLIT=0 (k=7 convs on v_mfma_f32_32x32x2_f32)"}
        finally:
            ops.BF16_SPLIT = True

    if rank != 0:
        if torch.distributed.is_initialized():
            torch.distributed.destroy_process_group()
        return

    out = {
        "metric": "24kHz audio sec encoded+decoded per wall-sec",
        "value": round(value, 2), "unit": "audio-s/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "arithmetic": ("fp32 tensors and fp32 accumulation everywhere; the k=7 ResidualUnit convs form each fp32 product from "
                       "three-way exact bf16 splits of both operands on the bf16 matrix pipe (error vs fp64 <= the fp32 MFMA's, "
                       "tests/test_gpu_parity.py::test_split_bf16_conv_matches_fp32_grade)" if ops.BF16_SPLIT else "fp32 MFMA"),
        "config": {"workload": f"configs[1]: batch={args.batch}/GPU x 2 s @ 24 kHz, forward encoder->FVQ(6 codebooks)->decoder, "
                               "FAcodec configs/config.yml model (137.7 M params), weights formula-generated, "
                               "weight-norm re-materialised every step", "clips_per_gpu": args.batch,
                   "samples_per_clip": n_samples, "parallelism": f"dp{world} (independent clips, no collective)"},
    }
    if prof is not None:
        summ = prof.summary()
        name, best = max(summ.items(), key=lambda kv: kv[1]["ms"])
        tot_ms = sum(v["ms"] for v in summ.values())
        achieved = best["flops"] / (best["ms"] * 1e-3) / 1e12
        traffic = None
        pmc = os.path.join(REPO, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            traffic = json.load(open(pmc)).get(name.split(" ")[0])
        is_split = "bsplit" in name
        peak = SPLIT_PEAK_TFLOPS if is_split else FP32_MFMA_PEAK_TFLOPS
        out["roofline"] = {
            "bound": "mfma", "kernel": name, "achieved": round(achieved, 2), "peak": round(peak, 1),
            "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": traffic,
            "peak_basis": ("bf16 MFMA dense peak 2516.6 / 6 MFMAs per fp32-equivalent K step (fp32-exact operand splitting); "
                           "achieved = algorithmic fp32 FLOPs / time" if is_split else "fp32 MFMA dense peak"),
            "launches_per_step": best["launches"] // args.steps,
            "avg_launch_us": round(1e3 * best["ms"] / best["launches"], 2),
            "avg_launch_gflop": round(best["flops"] / best["launches"] / 1e9, 3),
            "kernel_share_of_step": round(best["ms"] / (1e3 * elapsed), 4),
            "all_conv_variants": {k: {"launches_per_step": v["launches"] // args.steps,
                                      "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2),
                                      "ms_per_step": round(v["ms"] / args.steps, 3)} for k, v in summ.items()},
            "conv_ms_per_step": round(tot_ms / args.steps, 3),
            "whole_step_tflops": round(value / world * FLOP_PER_AUDIO_S / 1e12, 2),
        }
    if fp32_ref is not None:
        out["fp32_mfma_only"] = fp32_ref
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline()
    print(json.dumps(out), flush=True)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
