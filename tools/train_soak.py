#!/usr/bin/env python
"""Soak of the whole train step (BASELINE.json configs[2]: 16 clips x 2 s per GPU, predictor heads, discriminator + generator
halves; train.py:188-374) over a few hundred iterations with FRESH synthetic batches, quantizer-dropout draws and dropout masks
every iteration: losses stay finite, the allocator's peak stops growing after the first iterations (no leak), the step time does
not drift, no resident-LSTM wait ever times out.  The loss TRAJECTORY is reported per term, not judged: formula-generated weights,
Gaussian-noise "audio" and random predictor targets are not a learnable task (the step's arithmetic is pinned to the reference's by
tests/test_train_golden.py; what this run adds is stability of the machinery over hundreds of optimiser steps).

    python tools/train_soak.py [--steps 200] [--batch 16]      -> one JSON line
"""
import argparse
import json
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from facodec_amd import _lib, synth  # noqa: E402
from facodec_amd.commons import build_model, default_model_params  # noqa: E402
from facodec_amd.train import TrainStep  # noqa: E402
from bench import synthetic_predictor_targets  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--batch", type=int, default=16)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = build_model(default_model_params())
    for k in ("encoder", "quantizer", "decoder", "discriminator", "fa_predictors"):
        synth.load_synthetic(model[k], seed=0, prefix=k + ".")
        model[k].to(dev)
    step = TrainStep(model, with_predictors=True)
    # a small pool of clips cycled with different crops would be the reference's epoch; here: 8 distinct batches, revisited
    pool = [synth.synth_clips(a.batch, 48000, seed=100 + i).to(dev) for i in range(8)]
    targets = [synthetic_predictor_targets(a.batch, 160, dev, seed=200 + i) for i in range(8)]
    hist, times, peaks = [], [], []
    torch.cuda.reset_peak_memory_stats()
    for it in range(a.steps):
        t0 = time.perf_counter()
        out = step(pool[it % 8], targets=targets[it % 8])          # masks=None: the reference's own random draws, dropout on
        vals = {k: float(v) for k, v in out.items() if torch.is_tensor(v) and v.numel() == 1}     # float() synchronises
        times.append(time.perf_counter() - t0)
        peaks.append(torch.cuda.max_memory_allocated())
        hist.append(vals)
        if it % 25 == 0 or it == a.steps - 1:
            print(f"[soak] it {it:4d}  loss {vals['loss']:10.3f}  mel {vals['mel']:7.4f}  loss_d {vals['loss_d']:7.4f}  feat {vals['feature']:8.4f}  "
                  f"{1e3 * times[-1]:7.1f} ms  peak {peaks[-1] / 2 ** 30:6.2f} GiB", file=sys.stderr, flush=True)
    finite = all(all(v == v and abs(v) < 1e9 for v in h.values()) for h in hist)
    n = a.steps
    mel_first, mel_last = statistics.mean(h["mel"] for h in hist[:10]), statistics.mean(h["mel"] for h in hist[-10:])
    rep = {"steps": n, "batch": a.batch, "losses_finite": finite,
           "mel_first10": round(mel_first, 4), "mel_last10": round(mel_last, 4),
           "terms_median_first10": {k: round(statistics.median(h[k] for h in hist[:10]), 4) for k in hist[0]},
           "terms_median_last10": {k: round(statistics.median(h[k] for h in hist[-10:]), 4) for k in hist[0]},
           "ms_per_step_first_quarter": round(1e3 * statistics.median(times[5:n // 4]), 1),
           "ms_per_step_last_quarter": round(1e3 * statistics.median(times[-n // 4:]), 1),
           "peak_GiB_after_10": round(peaks[min(9, n - 1)] / 2 ** 30, 2), "peak_GiB_at_end": round(peaks[-1] / 2 ** 30, 2),
           "resident_lstm_timeouts": int(_lib.load().fac_lstm_persist_timeouts()),
           "lr_end": {k: o.get_last_lr()[0] for k, o in step.opt.items()}}
    print(json.dumps(rep))
    assert finite and rep["resident_lstm_timeouts"] == 0
    assert rep["peak_GiB_at_end"] <= rep["peak_GiB_after_10"] * 1.05 + 0.5, "allocator peak keeps growing"


if __name__ == "__main__":
    main()
