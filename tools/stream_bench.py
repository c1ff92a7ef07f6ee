#!/usr/bin/env python
"""Streaming latency / real-time factor (BASELINE.json configs[4]): 24 kHz audio in 480-sample hops through
encoder -> FA-quantizer -> decoder with carried state on one MI355X.

    python tools/stream_bench.py --minutes 30 [--streams 1] [--no-graphs]

Prints one JSON line: p50 / p90 / p99 per-hop latency (host wall clock around push() incl. the device sync),
RTF = processing time / audio time, for synthetic audio resident in HBM.
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from facodec_amd import synth  # noqa: E402
from facodec_amd.commons import build_model, default_model_params  # noqa: E402
from facodec_amd.streaming import HOP, StreamingCodec  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--minutes", type=float, default=30.0)
    ap.add_argument("--streams", type=int, default=1)
    ap.add_argument("--no-graphs", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    model = build_model(default_model_params())
    for k in ("encoder", "quantizer", "decoder"):
        synth.load_synthetic(model[k], seed=0, prefix=k + ".")
        model[k].eval().to(dev)
    n_hops = int(a.minutes * 60 * 24000 // HOP)
    n_hops -= n_hops % 5
    # one minute of distinct synthetic audio, looped (the stream state never repeats; HBM holds 30 min easily
    # but generating it on the host is slow)
    loop = synth.synth_clips(a.streams, 24000 * 60, seed=0).to(dev)
    with torch.no_grad():
        enrol = loop[:, :, :48000]
        timbre = model.quantizer(model.encoder(enrol), enrol, n_c=2)[4]
        sess = StreamingCodec(model, timbre, n_c=2, use_graphs=not a.no_graphs)
        sess.prime(loop[:, :, :4800])
        torch.cuda.synchronize()
        lat = []
        pos = 4800
        t_all = time.perf_counter()
        for h in range(n_hops):
            if pos + HOP > loop.shape[-1]:
                pos = 0
            hop = loop[:, :, pos:pos + HOP]
            pos += HOP
            t0 = time.perf_counter()
            out = sess.push(hop)
            torch.cuda.synchronize()
            lat.append(time.perf_counter() - t0)
        wall = time.perf_counter() - t_all
    steady = sorted(lat[10:])
    q = lambda p: round(1e3 * steady[min(len(steady) - 1, int(p * len(steady)))], 4)  # noqa: E731
    audio_s = n_hops * HOP / 24000.0
    print(json.dumps({"metric": "streaming per-hop latency / RTF", "hop_samples": HOP, "streams": a.streams,
                      "hops": n_hops, "audio_minutes": round(audio_s / 60, 2), "graphs": not a.no_graphs,
                      "p50_ms": q(0.5), "p90_ms": q(0.9), "p99_ms": q(0.99), "max_ms": round(1e3 * steady[-1], 3),
                      "rtf": round(wall / audio_s, 5), "wall_s": round(wall, 2),
                      "frames_emitted_last_hop": None if out["codes"] is None else int(out["codes"][0].shape[-1])}))


if __name__ == "__main__":
    main()
