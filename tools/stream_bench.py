#!/usr/bin/env python
"""Streaming latency / real-time factor (BASELINE.json configs[4]): 24 kHz audio in 480-sample hops through
encoder -> FA-quantizer -> decoder with carried state on one MI355X.

    python tools/stream_bench.py --minutes 30 [--streams 1] [--no-graphs] [--check-minutes 5]

Prints one JSON line: p50 / p90 / p99 per-hop latency (host wall clock around push() incl. the device sync),
RTF = processing time / audio time, for synthetic audio resident in HBM.

--check-minutes M (soak check): every code and every output sample of the first M minutes of the stream
(M x 3 000 hops, graph replay, both chains) is kept and compared at the end with the OFFLINE causal model run once over the
same M minutes (+ 1 s, so that the offline STFT's right-edge reflection stays outside the compared frames): codes must be
bit-exact and the waveform within 1e-4 -- any drift of the hop / frame phase bookkeeping, the ring buffers or the carried
LSTM state over tens of thousands of hops shows up there.  The remaining minutes continue on the same session (finite
outputs and the emitted frame count are checked to the end).
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from facodec_amd import benchutil, synth  # noqa: E402
from facodec_amd.commons import build_model, default_model_params  # noqa: E402
from facodec_amd.streaming import HOP  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--minutes", type=float, default=30.0)
    ap.add_argument("--streams", type=int, default=1)
    ap.add_argument("--no-graphs", action="store_true")
    ap.add_argument("--check-minutes", type=float, default=0.0)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    model = build_model(default_model_params())
    for k in ("encoder", "quantizer", "decoder"):
        synth.load_synthetic(model[k], seed=0, prefix=k + ".")
        model[k].eval().to(dev)
    n_hops = int(a.minutes * 60 * 24000 // HOP)
    r = benchutil.streaming_soak(model, dev, n_hops, check_minutes=a.check_minutes, streams=a.streams, use_graphs=not a.no_graphs)
    print(json.dumps(dict({"metric": "streaming per-hop latency / RTF"}, **r)))
    if not r["ok"]:
        raise SystemExit(f"streaming session drifted from the offline model or produced non-finite output: {r['drift_check']}")


if __name__ == "__main__":
    main()
