#!/usr/bin/env python
"""Streaming-hop A/B inside ONE process: two (or more) sessions built under different switches, fed the same audio in alternating
blocks of 300 hops -- box-to-box and minute-to-minute drift of the per-hop latency (+-0.08 ms between two 30 s runs of
tools/stream_bench.py on one box) cancels.  Variants are "name:KEY=VAL,KEY=VAL" with keys
  gemv (FAC_GEMV), fold (ops.STREAM_FOLD), lstm_real (streaming.LSTM_REAL_COLUMNS), two (StreamingCodec.two_streams).
   python tools/tune/stream_ab_inproc.py base: "nofold:fold=0" ...
(Round 6 used it with two more switches that were then removed: the quantizer as its own third chain, and a smaller workgroup budget
for the encoder chain's split-reduction launches -- both exactly neutral, profiles/r06_streaming_inprocess_ab.log.)"""
import json
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from facodec_amd import ops, streaming, synth  # noqa: E402
from facodec_amd.commons import build_model, default_model_params  # noqa: E402
from facodec_amd.streaming import HOP, StreamingCodec  # noqa: E402


def apply(cfg):
    os.environ["FAC_GEMV"] = str(cfg.get("gemv", 1))
    ops.STREAM_FOLD = bool(int(cfg.get("fold", 1)))
    streaming.LSTM_REAL_COLUMNS = bool(int(cfg.get("lstm_real", 1)))


def main():
    variants = []
    for a in sys.argv[1:]:
        name, _, rest = a.partition(":")
        variants.append((name, dict(kv.split("=") for kv in rest.split(",") if kv)))
    dev = torch.device("cuda:0")
    model = build_model(default_model_params())
    for k in ("encoder", "quantizer", "decoder"):
        synth.load_synthetic(model[k], seed=0, prefix=k + ".")
        model[k].eval().to(dev)
    blocks, per = 8, 300
    wave = synth.synth_clips(1, 4800 + (blocks * per + 20) * HOP, seed=0).to(dev)
    sessions = []
    with torch.no_grad():
        timbre = model.quantizer(model.encoder(wave[:, :, :48000]), wave[:, :, :48000], n_c=2)[4]
        for name, cfg in variants:
            apply(cfg)
            s = StreamingCodec(model, timbre, n_c=2)
            s.two_streams = bool(int(cfg.get("two", 1)))
            s.prime(wave[:, :, :4800])
            for h in range(15):                       # eager period, capture period, first replays -- under this variant's switches
                s.push(wave[:, :, 4800 + h * HOP:4800 + (h + 1) * HOP])
            sessions.append((name, cfg, s, []))
        torch.cuda.synchronize()
        pos = 15
        for b in range(blocks):
            for name, cfg, s, lat in sessions:
                apply(cfg)
                for h in range(per):
                    hop = wave[:, :, 4800 + (pos + h) * HOP:4800 + (pos + h + 1) * HOP]
                    t0 = time.perf_counter()
                    s.push(hop)
                    torch.cuda.synchronize()
                    lat.append(time.perf_counter() - t0)
            pos += per
    out = {}
    for name, cfg, s, lat in sessions:
        srt = sorted(lat)
        per_block = [round(1e3 * statistics.median(lat[i * per:(i + 1) * per]), 4) for i in range(blocks)]
        out[name] = {"cfg": cfg, "p50_ms": round(1e3 * srt[len(srt) // 2], 4), "p90_ms": round(1e3 * srt[int(0.9 * len(srt))], 4),
                     "mean_ms": round(1e3 * sum(lat) / len(lat), 4), "per_block_p50": per_block}
        print(name, json.dumps(out[name]), flush=True)


if __name__ == "__main__":
    main()
