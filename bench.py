#!/usr/bin/env python
"""bench.py -- FAcodec encode -> FA-quantizer -> decode throughput on MI355X.

Metric (BASELINE.json): 24 kHz audio-seconds encoded+decoded per wall-second.
Workload at N=1 = BASELINE.json configs[1]: batch 32 x 2 s @ 24 kHz, forward-only
encoder -> FVQ (6 codebooks, codes returned) -> decoder, synthetic clips resident in HBM,
formula-generated weights of the shipped architecture (configs/config.yml model_params),
fp32 throughout (fp32 MFMA; bf16 cannot hold bit-exact codes -- SURVEY.md 0.5).
N>1: one process per GPU, every rank runs the same per-GPU batch on its own clips (weak scaling, no
data-path collective in the forward: clips are independent units).  `python bench.py --gpus N` spawns the N
ranks itself (re-exec under torch.distributed.run on 127.0.0.1) when it is not already running under a launcher.

The same line also carries `train_step`: BASELINE.json configs[2], the full train.py iteration (discriminator
step + generator step, 16 clips per GPU) with the RCCL gradient all-reduce of every model key INSIDE the timed
region, and `codes_match`: the bench model's code indices on the golden clips against the real reference's
(tests/golden/codec_e2e.npz), checked before anything is timed.

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for how each field is computed).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import threading
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
# Train step (configs[2]): SURVEY.md 8d ESTIMATES 3 x (118.44 + 20.51 predictors) + ~10 x 34.8 discriminator GMAC per clip
# = 0.77 TFLOP per audio-second; the line reports the figure COUNTED at the C-ABI call sites of one real step instead
# (facodec_amd.ops.FlopCounter: conv forward / data-gradient / weight-gradient launches and LSTM recurrences, 2 FLOP per
# multiply-add of the mathematical definition; DFT-as-GEMM front-ends booked separately and not included).
TRAIN_FLOP_PER_AUDIO_S_SURVEY = 0.77e12
TRAIN_BATCH = 16
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


def check_codes(model, device):
    """Correctness gate of the metric ("...; code-index match"), part 1: the six code streams of the two golden clips must EQUAL
    the REAL reference's (tests/golden/codec_e2e.npz was produced by importing /root/reference); any difference aborts the run
    (the near-tie triage of facodec_amd/diagnostics.py only words the message)."""
    import numpy as np
    from facodec_amd.diagnostics import LatentCapture, classify_faquantizer_codes
    gold = np.load(os.path.join(REPO, "tests", "golden", "codec_e2e.npz"))
    wave = synth.synth_clips(2, int(CLIP_SECONDS * SAMPLE_RATE), seed=0).to(device)
    with LatentCapture(model.quantizer) as cap:
        _, codes = make_step(model, wave)()
    # FAquantizer.forward_v2 (modules/quantize.py:398-437) runs prosody, content and residual quantizers in this order
    expected = [gold[k] for k in ("codes_p", "codes_c", "codes_r")]
    if not all(torch.equal(c.cpu().long(), torch.as_tensor(e).long()) for c, e in zip(codes, expected)):
        raise SystemExit("[bench] code-index mismatch against the reference golden vectors (tests/golden/codec_e2e.npz): "
                         f"{classify_faquantizer_codes(cap, codes, expected)}")
    return True


def check_codes_b32(model, wave, rank):
    """Part 2, configs[1] at its own size: the six code streams of the TIMED batch (rank 0's 32 clips = synth.synth_clips(32,
    48000, seed=0)) against the REAL reference's runs on the same 32 clips.  tests/golden/codec_b32_decidable.npz holds the
    reference's answers in fp32 / all threads (= codec_b32.npz), fp32 / one thread and fp64 (tests/golden/make_golden_bench.py):
    every position on which the three runs agree must be equal, with ZERO allowance; a frame that holds a position on which the
    reference disagrees with itself must equal one run's whole code column (facodec_amd.diagnostics.check_codes_decidable).
    Anything else aborts the run.  Returns None when the batch is not that batch."""
    import numpy as np
    from facodec_amd.diagnostics import LatentCapture, check_codes_decidable, classify_faquantizer_codes
    path = os.path.join(REPO, "tests", "golden", "codec_b32_decidable.npz")
    if rank != 0 or tuple(wave.shape) != (32, 1, 48000) or not os.path.exists(path):
        return None
    fx = np.load(path)
    with LatentCapture(model.quantizer) as cap:
        _, codes = make_step(model, wave)()
    v = check_codes_decidable(codes, fx)
    if not v["ok"]:
        ref = fx["codes_f32_mt"]
        triage = classify_faquantizer_codes(cap, codes, [ref[:, lo:hi] for lo, hi in ((0, 1), (1, 3), (3, 6))])   # prosody | content | residual
        raise SystemExit(f"[bench] code-index mismatch on the timed 32-clip batch against the reference: {v}; triage on this run's latents: {triage}")
    rep = json.loads(str(fx["report"]))
    return {"clips": 32, "codes": v["positions"], "ok": True,
            "decidable_positions": v["decidable"], "decidable_mismatches": v["decidable_mismatches"],
            "undecidable_positions_clip_codebook_frame": v["undecidable_positions"],
            "undecidable_frames_not_a_reference_column": v["undecidable_frames_not_a_reference_column"],
            "mismatches": v["differs_from_fp32_reference"], "equals_reference_run": v["equals_run"],
            "sha256_equal": v["sha256_equal"], "sha256_equal_fp64_reference": v["sha256_equal_fp64_reference"],
            "rule": "positions on which the reference's fp32 (all threads), fp32 (1 thread) and fp64 runs agree: equal, zero allowance; a frame "
                    "holding a position on which they disagree: the whole 6-code column equals ONE of those runs' columns",
            "reference_self_disagreement": {"fp64_top2_gap_at_undecidable": rep["fp64_gap_at_undecidable"],
                                            "smallest_fp64_top2_gap_among_decidable": rep["smallest_fp64_gap_among_decidable"]},
            "fixture": "tests/golden/codec_b32_decidable.npz (real reference on this batch: fp32 all threads = codec_b32.npz, fp32 one thread, fp64)"}


def check_codes_four_batches(model, device, rank):
    """Part 3: four 32-clip batches (seeds 0..3 of synth.synth_clips; seed 0 is the timed batch) against the real
    reference's fp32 / fp32-one-thread / fp64 runs with the reference's own margin noise in the definition of decidable
    (facodec_amd.diagnostics.check_codes_decidable_noise; tests/golden/codec_b32x4_decidable.npz).  A mismatch aborts the run.
    Per batch the line reports where the product differs from the fp64 and fp32 runs and the fp64 gaps there."""
    import numpy as np
    from facodec_amd.diagnostics import check_codes_decidable_noise
    path = os.path.join(REPO, "tests", "golden", "codec_b32x4_decidable.npz")
    if rank != 0 or not os.path.exists(path):
        return None
    fx = np.load(path)
    out = []
    for bi, seed in enumerate(fx["seeds"].tolist()):
        wave = synth.synth_clips(32, int(CLIP_SECONDS * SAMPLE_RATE), seed=int(seed)).to(device)
        _, codes = make_step(model, wave)()
        v = check_codes_decidable_noise(codes, fx, bi)
        if not v["ok"]:
            raise SystemExit(f"[bench] code-index mismatch on batch seed {seed} against the reference (noise-aware rule): {v}")
        out.append({"seed": int(seed), "ok": True, "decidable": v["decidable"], "noise": v["noise"], "noise_flips": v["noise_flips"],
                    "cascade_positions": v["cascade_positions"], "differs_from_fp32": v["differs_from_fp32"],
                    "differs_from_fp64": v["differs_from_fp64"], "differs_from_fp64_at": v["differs_from_fp64_positions_and_gaps"],
                    "equals_run": v["equals_run"]})
    return out


def check_train_step_against_reference(step, device):
    """The train leg checks itself before it is timed: ONE iteration on the inputs of tests/golden/train_b16.npz (the REAL reference's
    train.py:188-374 iteration at 16 x 2 s with recorded random draws, dropout off; tests/golden/make_golden_bench.py train16) --
    the 15 loss scalars of the step at 1e-4 relative (north_star's bar; the GPU test holds 1e-5) and the five pre-clip gradient
    norms at 2e-4.  Needs the freshly initialised model and optimiser state the fixture was made with.  A miss aborts the run."""
    import numpy as np
    from facodec_amd.train import crop_segments
    path = os.path.join(REPO, "tests", "golden", "train_b16.npz")
    if not os.path.exists(path):
        return None
    d = np.load(path)
    B, seg = len(d["wave_lens"]), int(d["seg_frames"])
    waves = synth.synth_clips(B, int(d["t_full"]), seed=int(d["wave_seed"])).squeeze(1)
    for b, n in enumerate(d["wave_lens"]):
        waves[b, int(n):] = 0.0
    waves = waves.to(device)
    wav_seg, _, _ = crop_segments(waves, [int(n) // 300 for n in d["wave_lens"]], max_frame_len=seg,
                                  starts=torch.from_numpy(d["crop_start"]).to(torch.int64))
    dev = lambda t: t.to(device)   # noqa: E731
    masks = dict(p=dev(torch.from_numpy(d["mask_p"])), c=dev(torch.from_numpy(d["mask_c"])), r=dev(torch.from_numpy(d["mask_r"])),
                 res=dev(torch.from_numpy(d["mask_res"])), dropout=False)
    targets = dict(f0=dev(torch.from_numpy(d["f0_targets"])), uv=dev(torch.from_numpy(d["real_norm"])),
                   phones=dev(torch.from_numpy(d["phones"]).to(torch.int64)), speaker=dev(torch.from_numpy(d["speaker"]).to(torch.int64)))
    out = step(wav_seg, masks=masks, targets=targets, full_waves=waves, wave_lens=dev(torch.from_numpy(d["wave_lens"]).to(torch.int64)))
    got = dict(loss_d=out["loss_d"], loss_gen_all=out["loss"], mel_loss=out["mel"], loss_g=out["loss_g"], loss_feature=out["feature"],
               commitment_loss=out["commitment"], codebook_loss=out["codebook"])
    got.update({k: out[k] for k in ("f0_loss", "uv_loss", "rev_f0_loss", "rev_uv_loss", "content_loss", "rev_content_loss", "spk_loss", "x_spk_loss")})
    rel = {k: abs(float(v) - float(d[k])) / abs(float(d[k])) for k, v in got.items()}
    gn = {k: abs(float(out["grad_norm"][k]) - float(d[f"grad_norm64_{k}"])) / float(d[f"grad_norm64_{k}"]) for k in out["grad_norm"]}
    worst, worst_g = max(rel, key=rel.get), max(gn, key=gn.get)
    res = {"ok": rel[worst] < 1e-4 and gn[worst_g] < 2e-4, "losses_checked": len(rel), "max_loss_rel_err": float("%.3g" % rel[worst]),
           "worst_loss": worst, "loss_bar": 1e-4, "max_grad_norm_rel_err": float("%.3g" % gn[worst_g]), "worst_grad_norm": worst_g,
           "grad_norm_bar": 2e-4, "loss_gen_all": round(float(out["loss"]), 5), "reference_loss_gen_all": round(float(d["loss_gen_all"]), 5),
           "fixture": "tests/golden/train_b16.npz (the real reference's iteration at 16 x 2 s)"}
    if not res["ok"]:
        raise SystemExit(f"[bench] the train step does not reproduce the reference's iteration (tests/golden/train_b16.npz): {res}; all: {rel} {gn}")
    return res


def synthetic_predictor_targets(batch, frames, device, seed=3):
    """What train.py:214-262 obtains from the external pitch extractor / CTC phoneme model / speaker model, as synthetic
    tensors of the same shapes and ranges (tests/golden/make_golden_train.py uses the same recipe): normalised log-F0 with
    -10 on unvoiced frames, log-normalised mel energy, phone ids in [0, 1024), speaker ids in [0, 20000)."""
    g = torch.Generator().manual_seed(seed)
    f0 = torch.randn(batch, frames, generator=g)
    f0[torch.rand(batch, frames, generator=g) < 0.3] = -10.0
    t = dict(f0=f0, uv=torch.randn(batch, frames, generator=g) * 0.5, phones=torch.randint(0, 1024, (batch, frames), generator=g),
             speaker=torch.randint(0, 20000, (batch,), generator=g))
    return {k: v.to(device) for k, v in t.items()}


def train_leg(model, device, rank, world, steps, warmup):
    """configs[2]: discriminator step + generator step WITH the predictor heads (train.py:270,314-365) on 16 clips per GPU,
    gradient all-reduce (RCCL) inside the timing."""
    import torch.distributed as dist
    from facodec_amd.train import TrainStep
    for k in ("discriminator", "fa_predictors"):
        synth.load_synthetic(model[k], seed=0, prefix=k + ".")
        model[k].to(device)
    import numpy as np
    step = TrainStep(model, with_predictors=True)
    ref_check = check_train_step_against_reference(step, device)        # first iteration on fresh weights = the fixture's; aborts on a miss
    n_samples = int(CLIP_SECONDS * SAMPLE_RATE)
    wave = synth.synth_clips(TRAIN_BATCH, n_samples, seed=1, rank=rank).to(device)
    targets = synthetic_predictor_targets(TRAIN_BATCH, n_samples // 300, device, seed=3 + rank)
    last = {}
    calls = [0]

    def fn():
        # dropout / quantizer-dropout draws (CPU torch.randint, device bernoulli_) come from the default generators: seeded per call,
        # so the losses of the line are reproducible run to run -- same draw sequence for a given (steps, warmup)
        torch.manual_seed(7000 + 100 * rank + calls[0])
        np.random.seed(7000 + 100 * rank + calls[0])            # the residual mask comes from np.random.choice (modules/quantize.py:420-423)
        calls[0] += 1
        last.update(step(wave, targets=targets))

    torch.cuda.reset_peak_memory_stats()
    elapsed = benchutil.timed_steps(fn, steps, warmup, torch.cuda.synchronize, device)
    units = benchutil.aggregate_units(TRAIN_BATCH * CLIP_SECONDS * steps, device)
    peak_mem = torch.cuda.max_memory_allocated()
    finite = all(bool(torch.isfinite(last[k]).all()) for k in ("loss", "loss_d", "mel", "feature", "f0_loss", "content_loss", "spk_loss"))
    # one more (untimed) step with the FLOP counter and the exchange stopwatch on
    fc = ops.FlopCounter()
    ops.set_flop_counter(fc)
    for o in step.opt.values():
        o.time_exchange = True
    fn()
    torch.cuda.synchronize()
    ops.set_flop_counter(None)
    # ... and one with HIP events around every conv / weight-gradient launch (recorded on the stream each one is launched on):
    # which kernel the step spends most time in, and that kernel's own rate against the pipe it runs on
    # (the concurrent chains are run one after the other for this one step, so that a launch's event time is its own time)
    from facodec_amd import autograd_pred, discriminator, losses, quantize
    prof = ops.ConvLaunchProfile()
    ops.set_conv_profile(prof)
    saved = (discriminator.N_STREAMS, autograd_pred.PRED_STREAMS, quantize.QUANT_STREAMS, losses.MEL_STREAMS)
    discriminator.N_STREAMS = autograd_pred.PRED_STREAMS = quantize.QUANT_STREAMS = losses.MEL_STREAMS = 1     # every chain serial
    try:
        fn()
        torch.cuda.synchronize()
    finally:
        discriminator.N_STREAMS, autograd_pred.PRED_STREAMS, quantize.QUANT_STREAMS, losses.MEL_STREAMS = saved
        ops.set_conv_profile(None)
    ksum = prof.summary()
    exchange = step.exchange_report()
    for o in step.opt.values():
        o.time_exchange = False
    ar_ms, ar_bytes = None, sum(o.g.numel() * 4 for o in step.opt.values())
    if world > 1:   # the exchange alone (all five arenas back to back, blocking): what the overlap has to hide
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(3):
            for o in step.opt.values():
                dist.all_reduce(o.g, op=dist.ReduceOp.AVG if dist.get_backend() == "nccl" else dist.ReduceOp.SUM)
        torch.cuda.synchronize()
        ar_ms = 1e3 * (time.perf_counter() - t0) / 3
    value = units / elapsed
    flop_per_audio_s = fc.total / (TRAIN_BATCH * CLIP_SECONDS)
    per_gpu_tflops = value / world * flop_per_audio_s / 1e12
    return {
        "value": round(value, 2), "unit": "audio-s/s", "ms_per_step": round(1e3 * elapsed / steps, 2), "steps": steps, "warmup": warmup,
        "workload": f"configs[2]: train.py iteration, {TRAIN_BATCH} clips/GPU x 2 s: encoder -> FA-quantizer -> decoder + predictor heads "
                    "forward + backward, 5 MPD + 3 MRD discriminators (discriminator step, then generator step with adversarial + "
                    "feature-matching), 7-scale mel loss, commitment + codebook + F0 / UV / content(focal) / speaker losses with "
                    "synthetic targets, 5 x (clip + AdamW + ExponentialLR)",
        "with_predictors": True,
        "concurrent_chains": {"discriminator_streams": _disc_streams(), "predictor_streams": _pred_streams(),
                              "note": "independent sub-networks (8 discriminators, 4 predictor heads) run side by side on side streams of the "
                                      "one GPU, forward and backward; FAC_DISC_STREAMS=1 FAC_PRED_STREAMS=1 runs them one after the other"},
        "parallelism": f"dp{world}: one all-reduce(mean) per model key over RCCL, launched asynchronously under backward",
        "allreduce_bytes_per_step": ar_bytes, "allreduce_ms_standalone": None if ar_ms is None else round(ar_ms, 2),
        "allreduce_overlap": {"launched_early": [k for k, e in exchange.items() if e["launched"] == "hook"],
                              "launched_after_backward": [k for k, e in exchange.items() if e["launched"] == "end"],
                              "wait_ms": {k: (None if e["wait_ms"] is None else round(e["wait_ms"], 3)) for k, e in exchange.items()},
                              "buckets": {k: {"n": e["n_buckets"], "launches": ["%s:%s" % (b, w) for b, w in e["buckets"]]} for k, e in exchange.items()},
                              "note": "each key's gradient arena leaves in <= 64 MB buckets, end of the arena first; launched_early = the "
                                      "key's first bucket was issued from a gradient hook inside backward (bucket:where lists every launch); "
                                      "wait_ms = stall of the compute stream at the key's optimiser step (what the overlap failed to hide)"},
        "reference_check": ref_check,
        "losses_finite": finite, "loss": round(float(last["loss"]), 4), "mel": round(float(last["mel"]), 4),
        "loss_seeded": seeded_loss_check(float(last["loss"]), float(last["mel"]), steps, warmup, world),
        "peak_mem_GB": round(peak_mem / 2 ** 30, 1),
        "counted_flops": {"per_step_TFLOP": round(fc.total / 1e12, 3), "per_audio_s_TFLOP": round(flop_per_audio_s / 1e12, 4),
                          "by_kind_TFLOP": {k: round(v / 1e12, 3) for k, v in fc.flops.items()},
                          "launches": dict(fc.launches),
                          "survey_estimate_per_audio_s_TFLOP": TRAIN_FLOP_PER_AUDIO_S_SURVEY / 1e12,
                          "basis": "counted at the C-ABI call sites of one real step (facodec_amd/ops.py FlopCounter): conv fwd / "
                                   "data-grad / weight-grad launches + LSTM recurrences, batch padding excluded; 'dft' = windowed-DFT "
                                   "and mel GEMMs of the STFT front-ends, NOT part of the total"},
        "roofline": train_roofline(ksum, per_gpu_tflops),
    }


def seeded_loss_check(loss, mel, steps, warmup, world):
    """The loss / mel of the last event-timed step against the values recorded for the same (steps, warmup) on one GPU
    (tests/golden/bench_train_seeded.json, written by `python bench.py --record-train-loss` on an MI355X): seeded draws make the
    sequence reproducible, so a drift beyond 1e-4 means the training path changed what it computes.  Reported, not fatal: the
    sequence runs through ten AdamW steps of a GAN loss with arg-max codes in it, and the fatal check is the one against the
    reference (reference_check)."""
    path = os.path.join(REPO, "tests", "golden", "bench_train_seeded.json")
    rec = json.load(open(path)).get(f"steps{steps}_warmup{warmup}") if os.path.exists(path) and world == 1 else None
    if rec is None:
        return {"recorded": None}
    e_l, e_m = abs(loss - rec["loss"]) / abs(rec["loss"]), abs(mel - rec["mel"]) / abs(rec["mel"])
    return {"recorded": {"loss": rec["loss"], "mel": rec["mel"]}, "rel_err": [float("%.3g" % e_l), float("%.3g" % e_m)], "bar": 1e-4,
            "ok": bool(e_l < 1e-4 and e_m < 1e-4)}


def train_roofline(ksum, per_gpu_tflops):
    """Dominant kernel of the train step (largest summed launch time in the event-timed extra step, which runs the concurrent
    chains one after the other so that a launch's event time is its own) priced against the pipe it runs on, next to the
    whole-step figures.  A weight-gradient record is the whole launch: operand planes + GEMM + split-K reduction."""
    name, best = max(ksum.items(), key=lambda kv: kv[1]["ms"])
    is_split = any(t in name for t in ("bsplit", "gemm_split", "wgrad_split", "kmajor"))
    peak = SPLIT_PEAK_TFLOPS if is_split else FP32_MFMA_PEAK_TFLOPS
    achieved = best["flops"] / (best["ms"] * 1e-3) / 1e12
    tot_ms = sum(v["ms"] for v in ksum.values())
    return {"bound": "mfma", "kernel": name, "achieved": round(achieved, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
            "frac": round(achieved / peak, 4),
            "launches_per_step": best["launches"], "avg_launch_us": round(1e3 * best["ms"] / best["launches"], 2),
            "avg_launch_gflop": round(best["flops"] / best["launches"] / 1e9, 3),
            "kernel_share_of_gemm_launch_time": round(best["ms"] / tot_ms, 4),
            "all_variants": {k: {"launches_per_step": v["launches"], "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2),
                                 "ms_per_step": round(v["ms"], 3)}
                             for k, v in sorted(ksum.items(), key=lambda kv: -kv[1]["ms"])[:12]},
            "whole_step_tflops": round(per_gpu_tflops, 2),
            "whole_step_frac_of_fp32_mfma_peak": round(per_gpu_tflops / FP32_MFMA_PEAK_TFLOPS, 4),
            "whole_step_frac_of_split_peak": round(per_gpu_tflops / SPLIT_PEAK_TFLOPS, 4),
            "basis": "kernel / achieved / frac: HIP events around every conv and weight-gradient launch of one extra step (chains serialised), the variant with "
                     "the largest total time, algorithmic FLOPs / its time, against the pipe it runs on (bf16 dense / 6 for the "
                     "three-way-split kernels).  whole_step_*: audio-s/s x COUNTED TFLOP per audio-second, per GPU; most of those FLOPs "
                     "run on the bf16 pipe, so the fp32-peak fraction flatters and the split-peak fraction is the honest one"}


def _disc_streams():
    from facodec_amd import discriminator
    return discriminator.N_STREAMS


def _pred_streams():
    from facodec_amd import autograd_pred
    return autograd_pred.PRED_STREAMS


def streaming_leg(model, device, hops=90000, check_minutes=5.0):
    """configs[4] at its stated size: 30 min of 24 kHz audio = 90 000 hops of 480 samples through one streaming session
    (HIP-graph replay, two chains per hop), p50 / p99 / max per-hop latency and RTF; the first `check_minutes` of the stream are
    compared code by code and sample by sample with the offline causal model (facodec_amd.benchutil.streaming_soak)."""
    r = benchutil.streaming_soak(model, device, hops, check_minutes=check_minutes)
    r["workload"] = (f"configs[4]: {r['hops']} hops of {r['hop_samples']} samples ({r['audio_minutes']} min of audio), one stream, carried "
                     "conv / LSTM state, HIP-graph replay, encoder and quantizer+decoder halves of a hop as two concurrent chains")
    return r


def respawn_under_launcher(n):
    """`python bench.py --gpus N` without a launcher: re-exec as N ranks (one per GPU) under torch.distributed.run."""
    if torch.cuda.device_count() < n:
        raise SystemExit(f"[bench] --gpus {n} but only {torch.cuda.device_count()} GPU(s) visible: refusing to measure fewer")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def cpu_baseline(batch=4, passes=5, warmups=2, with_train=True, threads=None, sweep_only=False, pin=False):
    """The CPU oracle (port of the reference algorithm, pinned to it by tests/golden) timed on this host's cores over a
    bounded sample of the same workload (B = 4 clips of 2 s, median of 5 passes after 2 warm-ups).
    `value` = (A) the eval forward; `train` = (B) the train-mode iteration (forward + backward of encoder / quantizer /
    predictors / decoder, 7-scale mel loss, both discriminator passes and its AdamW step: oracle/train_iteration.py), with
    fewer passes when one pass is long (said in `sample`).
    threads: the intra-op thread count, set BEFORE any CPU work of this call.  sweep_only: one timed pass (after one warm pass) at
    8, 16, 32, 64 threads in ascending order -> {"threads_swept": {...}}.  bench.py's main() runs the sweep and the measurement in
    two FRESH processes (cpu_baseline_isolated): a process that has once run with more threads measures up to 45 % lower at the
    smaller count afterwards (round 5: 3.15 -> 1.74 audio-s/s at 8 threads after the 64-thread probe; heap blocks first touched
    across the sockets), so the count that is measured must be the largest one its process has ever used."""
    pinned = None
    if pin and threads is not None:
        pinned = pick_cpus(int(threads))
        if pinned:
            os.sched_setaffinity(0, pinned)           # before the first parallel region: the worker threads inherit the mask
    if threads is not None:
        torch.set_num_threads(max(1, min(int(threads), os.cpu_count() or 1)))
    import statistics
    from oracle import facodec_oracle as O
    model = build_model(default_model_params())
    keys = ("encoder", "quantizer", "decoder", "discriminator", "fa_predictors") if with_train else ("encoder", "quantizer", "decoder")
    sds = {}
    for k in keys:       # parameters by formula, buffers (anti-aliasing filters, mel tables) as the modules build them
        synth.load_synthetic(model[k], seed=0, prefix=k + ".")
        sds[k] = {n: v.detach().clone() for n, v in model[k].state_dict().items()}
    names = {k: {n for n, _ in model[k].named_parameters()} for k in keys}
    del model
    n_samples = int(CLIP_SECONDS * SAMPLE_RATE)
    wave = synth.synth_clips(batch, n_samples, seed=0)
    # Which thread count?  SURVEY 8d says "all host cores"; on the 256-logical-core hosts of the GPU boxes that is the port's WORST
    # case (profiles/r04_cpu_thread_sweep.log: 128 threads 0.72 audio-s/s against 3.2 at 16), so the baseline is the port's BEST
    # count from a short ascending sweep (sweep_only, its own process), stated in the line.
    prev_threads = torch.get_num_threads()
    ncpu = os.cpu_count() or 1
    with torch.no_grad():
        O.codec_forward(sds, wave[:1], n_c=2)  # page-in, oneDNN primitive cache
        if sweep_only:
            swept = {}
            for n in (8, 16, 32, 64):
                if n > ncpu:
                    break
                torch.set_num_threads(n)
                O.codec_forward(sds, wave, n_c=2)
                t0 = time.perf_counter()
                O.codec_forward(sds, wave, n_c=2)
                swept[str(n)] = round(batch * CLIP_SECONDS / (time.perf_counter() - t0), 3)
            return {"threads_swept": swept}
        threads = min(threads or 16, ncpu)
        torch.set_num_threads(threads)
        times = []
        for i in range(warmups + passes):
            t0 = time.perf_counter()
            O.codec_forward(sds, wave, n_c=2)
            times.append(time.perf_counter() - t0)
    med = statistics.median(times[warmups:])
    out = dict(value=round(batch * CLIP_SECONDS / med, 3), unit="audio-s/s", cores=threads, kind="port",
               pinned_cpus=(f"{pinned[0]}..{pinned[-1]} ({len(pinned)} physical cores of one NUMA node / socket)" if pinned else None),
               sample=f"median of {passes} passes after {warmups} warm-ups, {batch} clips x 2 s each (oracle/facodec_oracle.py "
                      f"codec_forward, torch-CPU fp32, {threads} threads, {os.cpu_count()} logical cores on host), "
                      f"{sum(times):.1f} s of CPU work")
    if with_train:
        from oracle.train_iteration import oracle_iteration
        frames = n_samples // 300
        tg = synthetic_predictor_targets(batch, frames, "cpu", seed=3)
        ones = lambda n: torch.ones(n, batch)  # noqa: E731
        t = dict(wav_seg=wave, waves=wave.squeeze(1), wave_lens=torch.full((batch,), n_samples, dtype=torch.int64),
                 targets=dict(f0=tg["f0"], uv=tg["uv"], phones=tg["phones"], speaker=tg["speaker"]),
                 masks=dict(p=ones(1), c=ones(2), r=ones(3), res=torch.ones(batch), dropout=False))
        tt = []
        t0 = time.perf_counter()
        oracle_iteration(sds, names, t)
        first = time.perf_counter() - t0
        # BASELINE.md 3 asks for the median of 5 after 2 warm-ups; one iteration of this port takes tens of seconds at B = 4, so
        # the number of timed iterations is cut to what ~100 s of CPU allow (said in `sample`): the default bench.py run must
        # still finish within a few minutes
        tp = max(1, min(5, int(100.0 / max(first, 1e-3)) - 1))
        tw = 2 if tp == 5 else 1
        for i in range(tw - 1 + tp):
            t0 = time.perf_counter()
            oracle_iteration(sds, names, t)
            tt.append(time.perf_counter() - t0)
        medt = statistics.median(tt[tw - 1:])
        out["train"] = dict(value=round(batch * CLIP_SECONDS / medt, 3), unit="audio-s/s", ms_per_step=round(1e3 * medt, 1), cores=threads,
                            kind="port",
                            sample=f"median of {tp} iterations after {tw} warm-up(s), {batch} clips x 2 s: train.py:265-374 through "
                                   f"oracle/train_iteration.py + torch autograd (encoder / quantizer / predictors / decoder forward + "
                                   f"backward, 7-scale mel loss, discriminator step + generator step), {threads} threads, "
                                   f"{first + sum(tt):.1f} s of CPU work")
    torch.set_num_threads(prev_threads)
    return out


def pick_cpus(n):
    """n CPUs for the CPU baseline, one hardware thread per physical core, all from ONE NUMA node / socket when it has that many
    cores: the first node (by id) that intersects this process's affinity mask is filled first, then its neighbours.  Returns the
    sorted list (None when /sys does not say)."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
        import glob

        def parse(txt):
            out = []
            for part in txt.strip().split(","):
                if "-" in part:
                    a, b = part.split("-")
                    out += list(range(int(a), int(b) + 1))
                elif part:
                    out.append(int(part))
            return out

        nodes = []
        for d in sorted(glob.glob("/sys/devices/system/node/node[0-9]*"), key=lambda x: int(x.rsplit("node", 1)[1])):
            cpus = [c for c in parse(open(os.path.join(d, "cpulist")).read()) if c in allowed]
            if cpus:
                nodes.append(cpus)
        if not nodes:
            nodes = [allowed]
        picked, seen_cores = [], set()
        for cpus in nodes:
            for c in cpus:
                try:
                    sib = tuple(parse(open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list").read()))
                except OSError:
                    sib = (c,)
                if sib in seen_cores:
                    continue
                seen_cores.add(sib)
                picked.append(c)
                if len(picked) == n:
                    return sorted(picked)
        return sorted(picked) if picked else None
    except (OSError, ValueError, AttributeError):
        return None


def cpu_baseline_isolated(sweep=False, threads=16):
    """cpu_baseline in a fresh process pinned (os.sched_setaffinity, before the first parallel region) to `threads` physical cores
    of one socket / NUMA node of the host (pick_cpus).  Rounds 3-5 reported 6.85 / 3.64 / 2.31 audio-s/s at the same 16 threads:
    r3 was the MEAN of 5 in-process passes on an otherwise idle host, r4 / r5 a median on other hosts with the 16 threads free to
    wander over the 256 logical cores of two sockets (the r5 sweep shows 8 / 16 / 32 threads within 5 % and a 45 % penalty once a
    process has touched its heap from the far socket): unpinned, the number is a property of where the scheduler put the threads.
    sweep=True additionally runs the one-pass thread sweep (8 .. 64, its own process, unpinned) for the record.  The workers see no GPU."""
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)

    def worker(*extra):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-worker", *extra], env=env, capture_output=True, text=True)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not lines:
            raise RuntimeError(f"cpu baseline worker failed ({r.returncode}): {r.stderr[-1500:]}")
        return json.loads(lines[-1])

    ncpu = os.cpu_count() or 1
    threads = min(threads, ncpu)
    out = worker("measure", "--cpu-threads", str(threads), "--cpu-pin")
    if sweep:
        out["threads_swept_unpinned"] = worker("sweep")["threads_swept"]
    return out


def latest_pmc_traffic():
    """The HBM-traffic counters are collected by separate rocprofv3 --pmc passes (tools/pmc_traffic.py; MI355X_MICROARCH.md HBM
    section), not inside this run: returns (per-kernel dict, source description) of the newest committed profiles/rNN_pmc_traffic.json."""
    import glob
    import re
    best = None
    for f in glob.glob(os.path.join(REPO, "profiles", "r*_pmc_traffic.json")):
        m = re.match(r"r(\d+)_pmc_traffic\.json", os.path.basename(f))
        if m and (best is None or int(m.group(1)) > best[0]):
            best = (int(m.group(1)), f)
    if best is None:
        return {}, None
    d = json.load(open(best[1]))
    src = dict(d.get("_source", {}), file=os.path.relpath(best[1], REPO),
               note="separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over bench.py, per launch; not measured inside this timed run")
    return d, src


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="clips per GPU per step (configs[1]: 32)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-worker", choices=("sweep", "measure"), default=None, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-threads", type=int, default=16, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-pin", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-sweep", action="store_true", help="also run the unpinned 8 .. 64 thread sweep of the CPU baseline (+ ~40 s)")
    ap.add_argument("--detail", default=None, help="file for the full, unabridged result (default: gpurun_out/bench_detail.json when that directory exists)")
    ap.add_argument("--record-train-loss", action="store_true", help="write this run's seeded train-leg loss / mel to tests/golden/bench_train_seeded.json")
    ap.add_argument("--no-roofline", action="store_true", help="skip per-launch HIP-event timing of the conv kernel")
    ap.add_argument("--no-train", action="store_true", help="skip the configs[2] train-step leg")
    ap.add_argument("--train-steps", type=int, default=4)
    ap.add_argument("--train-warmup", type=int, default=2)
    ap.add_argument("--no-streaming", action="store_true", help="skip the configs[4] streaming-latency leg")
    ap.add_argument("--stream-hops", type=int, default=90000, help="configs[4]: 90 000 hops of 480 samples = 30 min of audio (~110 s)")
    ap.add_argument("--stream-check-minutes", type=float, default=5.0, help="prefix of the stream compared with the offline model")
    args = ap.parse_args()

    if args.cpu_worker:          # child of cpu_baseline_isolated(): the CPU oracle only, no GPU
        r = (cpu_baseline(sweep_only=True, with_train=False) if args.cpu_worker == "sweep"
             else cpu_baseline(threads=args.cpu_threads, pin=args.cpu_pin))
        print(json.dumps(r), flush=True)
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        respawn_under_launcher(args.gpus)
    rank, local_rank, world = benchutil.init_distributed()
    if world != args.gpus:
        raise SystemExit(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}")
    if world > 1:
        backend = torch.distributed.get_backend()
        if backend != "nccl" and os.environ.get("FAC_DIST_BACKEND") != backend:
            raise SystemExit(f"[bench] {world} ranks over '{backend}': multi-GPU numbers are RCCL numbers (backend 'nccl'); "
                             "FAC_DIST_BACKEND=gloo is the explicit opt-in for the two-ranks-on-one-GPU smoke run")
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
        if torch.cuda.device_count() < local_world and os.environ.get("FAC_DIST_BACKEND") != "gloo":
            raise SystemExit(f"[bench] {local_world} local ranks but {torch.cuda.device_count()} GPU(s) visible: one process per GPU")
    device = torch.device(f"cuda:{local_rank % torch.cuda.device_count()}")
    torch.cuda.set_device(device)

    model = build(device)
    n_samples = int(CLIP_SECONDS * SAMPLE_RATE)
    codes_e2e = check_codes(model, device)
    wave = synth.synth_clips(args.batch, n_samples, seed=0, rank=rank).to(device)   # resident in HBM
    codes_b32 = check_codes_b32(model, wave, rank)
    codes_x4 = check_codes_four_batches(model, device, rank)
    # all three gates (each aborts the run on failure): 2 golden clips equal; timed batch under the strict rule; four batches
    # under the noise-aware rule
    codes_match = bool(codes_e2e and (codes_b32 is None or codes_b32["ok"]) and (codes_x4 is None or all(b["ok"] for b in codes_x4)))
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
            try:                                   # the same two code gates on this leg (a failure is recorded, the main line stands)
                fp32_ref["codes_match"] = bool(check_codes(model, device))
                fp32_ref["codes_match_timed_batch"] = check_codes_b32(model, wave, rank)
            except SystemExit as e:
                fp32_ref["codes_match"] = False
                fp32_ref["codes_error"] = str(e)[:2000]
                fp32_ref["codes_note"] = ("comparison leg only (every conv on the fp32 matrix pipe; NOT the product path, whose gates passed above). "
                                          "decidable_mismatch_fp64_gaps = the reference's own fp64 top-2 gap at the flipped positions; the "
                                          "reference's fp32-vs-fp64 margin noise on this batch reaches 4.1e-6 (tests/golden/codec_b32_decidable_report.json)")
        finally:
            ops.BF16_SPLIT = True

    streaming = None
    if rank == 0 and world == 1 and not args.no_streaming:
        streaming = streaming_leg(model, device, args.stream_hops, args.stream_check_minutes)

    out = {
        "metric": "24kHz audio sec encoded+decoded per wall-sec",
        "value": round(value, 2), "unit": "audio-s/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic", "codes_match": codes_match, "codes_match_timed_batch": codes_b32,
        "arithmetic": ("fp32 tensors and fp32 accumulation everywhere; the k=7 ResidualUnit convs form each fp32 product from "
                       "three-way exact bf16 splits of both operands on the bf16 matrix pipe, six of the nine cross products kept "
                       "(tested bars, tests/test_gpu_parity.py::test_split_bf16_conv_matches_fp32_grade: 1e-5 of the oracle like the fp32 "
                       "kernel, and error vs fp64 < 1.5 x the fp32-MFMA kernel's + 1e-7): fp32-GRADE, not bit-identical to an fp32 FMA chain. "
                       "Code indices: codes_match = both gates (2 golden clips equal; timed batch equal on every position the reference "
                       "decides, see codes_match_timed_batch.rule)" if ops.BF16_SPLIT else "fp32 MFMA"),
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
        pmc, pmc_src = latest_pmc_traffic()
        traffic = pmc.get(name.split(" ")[0])
        is_split = "bsplit" in name
        peak = SPLIT_PEAK_TFLOPS if is_split else FP32_MFMA_PEAK_TFLOPS
        out["roofline"] = {
            "bound": "mfma", "kernel": name, "achieved": round(achieved, 2), "peak": round(peak, 1),
            "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": traffic, "traffic_source": pmc_src if traffic is not None else None,
            "peak_basis": ("bf16 MFMA dense peak 2516.6 / 6 MFMAs per fp32-equivalent K step (fp32-grade operand splitting); "
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
            "whole_step_frac_of_fp32_mfma_peak": round(value / world * FLOP_PER_AUDIO_S / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
            "whole_step_frac_of_split_peak": round(value / world * FLOP_PER_AUDIO_S / 1e12 / SPLIT_PEAK_TFLOPS, 4),
            "note": "SURVEY.md 8d prices the whole forward against the fp32 MFMA peak (157.3): whole_step_frac_of_fp32_mfma_peak.  That is "
                    "flattering here, because most of the step's FLOPs run on the bf16 pipe as six exact products each: "
                    "whole_step_frac_of_split_peak prices the same step against that pipe (2516.6 / 6 = 419.4 fp32-equivalent TFLOP/s). "
                    "The bf16 pipe sustains 1913 TFLOP/s on random operands on this part (tools/microbench), so the practical ceiling of the "
                    "split kernels is ~319 fp32-equivalent TFLOP/s.",
        }
    # The forward line above is complete at this point.  The train leg runs collectives from gradient hooks; on more than one GPU
    # that path has only ever run under gloo and one-rank RCCL, so a watchdog makes sure a stall there cannot take the forward
    # measurement with it: on expiry rank 0 prints the line with the failure recorded and every rank leaves.
    train = None
    if not args.no_train:
        del step
        limit = float(os.environ.get("FAC_TRAIN_LEG_TIMEOUT", "420"))

        def expired():
            if rank == 0:
                out["train_step"] = {"error": f"train leg did not finish within {limit:.0f} s on {world} GPU(s); forward line unaffected"}
                print(json.dumps(compact_line(out)), flush=True)
            os._exit(3)                     # the line is printed, but a stalled exchange must not look like a clean run

        dog = threading.Timer(limit, expired)
        dog.daemon = True
        dog.start()
        try:
            train = train_leg(model, device, rank, world, args.train_steps, args.train_warmup)
        finally:
            dog.cancel()

    if rank != 0:
        if torch.distributed.is_initialized():
            torch.distributed.destroy_process_group()
        return
    if train is not None:
        out["train_step"] = train
    if streaming is not None:
        out["streaming"] = streaming
    if fp32_ref is not None:
        out["fp32_mfma_only"] = fp32_ref
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_isolated(sweep=args.cpu_sweep)
    out["codes_match_four_batches"] = codes_x4
    torch.cuda.synchronize()
    out["lstm_timeouts"] = ops.lstm_timeouts()
    if args.record_train_loss and train is not None:
        path = os.path.join(REPO, "tests", "golden", "bench_train_seeded.json")
        rec = json.load(open(path)) if os.path.exists(path) else {}
        rec[f"steps{args.train_steps}_warmup{args.train_warmup}"] = {"loss": train["loss"], "mel": train["mel"], "batch": TRAIN_BATCH}
        json.dump(rec, open(path, "w"), indent=1)
    detail = args.detail or (os.path.join("gpurun_out", "bench_detail.json") if os.path.isdir("gpurun_out") else None)
    if detail:
        try:
            json.dump(out, open(detail, "w"), indent=1)
        except OSError:
            detail = None
    line = compact_line(out, detail)
    print(json.dumps(line), flush=True)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()
    # a benchmark line must not look clean when what it measured is not what it claims: streaming drift against the
    # offline model, or a resident-LSTM wait that gave up somewhere in this run
    if streaming is not None and not streaming.get("ok", True):
        raise SystemExit(f"[bench] configs[4]: the streaming session drifted from the offline model: {streaming.get('drift_check')}")
    if out["lstm_timeouts"]:
        raise SystemExit(f"[bench] {out['lstm_timeouts']} resident-LSTM wait(s) timed out during this run: its numbers are void")


def compact_line(out, detail=None):
    """The ONE line the driver stores, <= 4 KB: numbers only (the prose that used to ride along is DESIGN.md section 4 / 11; the
    full record goes to --detail).  What a truncated tail must still show comes LAST: codes_match, the train step's time, the
    roofline fractions."""
    def pick(d, *keys):
        return {k: d[k] for k in keys if d is not None and k in d}

    line = pick(out, "metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data")
    line["config"] = out["config"]
    rf = out.get("roofline")
    if rf:
        line["roofline"] = pick(rf, "bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "launches_per_step", "avg_launch_us",
                                "avg_launch_gflop", "kernel_share_of_step", "conv_ms_per_step", "whole_step_tflops",
                                "whole_step_frac_of_fp32_mfma_peak", "whole_step_frac_of_split_peak")
    cb = out.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = pick(cb, "value", "unit", "cores", "kind", "pinned_cpus")
        line["cpu_baseline"]["sample"] = cb["sample"][:96]
        if "train" in cb:
            line["cpu_baseline"]["train"] = pick(cb["train"], "value", "unit", "ms_per_step", "cores", "kind")
    st = out.get("streaming")
    if st:
        line["streaming"] = pick(st, "hops", "audio_minutes", "p50_ms", "p99_ms", "p99.9_ms", "max_ms", "device_ms_p50", "rtf", "ok")
        dc = st.get("drift_check") or {}
        line["streaming"].update(pick(dc, "checked_minutes", "code_mismatches_vs_offline", "codes_compared", "wave_rel_err_vs_offline"))
    fr = out.get("fp32_mfma_only")
    if fr:
        line["fp32_mfma_only"] = pick(fr, "value", "ms_per_step", "codes_match")
    cm = out.get("codes_match_timed_batch")
    line["codes"] = {"golden_2_clips_equal": True,
                     "timed_batch_strict_rule": pick(cm, "ok", "decidable_positions", "decidable_mismatches", "mismatches", "equals_reference_run"),
                     "four_batches_noise_rule": [pick(b, "seed", "ok", "noise_flips", "cascade_positions", "differs_from_fp32", "differs_from_fp64")
                                                 for b in (out.get("codes_match_four_batches") or [])]}
    tr = out.get("train_step")
    if tr and "error" in tr:
        line["train_step"] = tr
    elif tr:
        t = pick(tr, "value", "unit", "ms_per_step", "steps", "warmup", "with_predictors", "reference_check", "losses_finite", "loss", "mel",
                 "loss_seeded", "allreduce_bytes_per_step", "allreduce_ms_standalone", "peak_mem_GB")
        if "reference_check" in t and t["reference_check"]:
            t["reference_check"] = pick(t["reference_check"], "ok", "losses_checked", "max_loss_rel_err", "loss_bar", "max_grad_norm_rel_err",
                                        "grad_norm_bar")
        t["counted_TFLOP_per_step"] = tr["counted_flops"]["per_step_TFLOP"]
        r = tr["roofline"]
        t["roofline"] = pick(r, "bound", "kernel", "achieved", "peak", "unit", "frac", "launches_per_step", "avg_launch_us", "whole_step_tflops",
                             "whole_step_frac_of_split_peak")
        t["roofline"]["kernel"] = r["kernel"][:48]
        t["roofline"]["top"] = {k[:30]: [v["launches_per_step"], v["tflops"], v["ms_per_step"]] for k, v in list(r["all_variants"].items())[:6]}
        line["train_step"] = t
    line["lstm_timeouts"] = out.get("lstm_timeouts", 0)
    line["detail"] = detail
    line["tail"] = {"codes_match": out["codes_match"], "train_ms_per_step": (tr or {}).get("ms_per_step"),
                    "roofline_frac": (rf or {}).get("frac"), "train_roofline_frac": ((tr or {}).get("roofline") or {}).get("frac"),
                    "train_whole_step_frac_of_split_peak": ((tr or {}).get("roofline") or {}).get("whole_step_frac_of_split_peak")}
    line["codes_match"] = out["codes_match"]
    return line


if __name__ == "__main__":
    main()
