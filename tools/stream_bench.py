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
    ap.add_argument("--check-minutes", type=float, default=0.0)
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
    check_frames = int(a.check_minutes * 60 * 80)                      # 80 frames of 300 samples per second
    loop_len = 24000 * 60 if not check_frames else check_frames * 300 + 24000
    loop = synth.synth_clips(a.streams, loop_len, seed=0).to(dev)
    keep_codes = torch.zeros(a.streams, 6, check_frames, dtype=torch.int64, device=dev) if check_frames else None
    keep_wave = torch.zeros(a.streams, 1, check_frames * 300, device=dev) if check_frames else None
    frames_seen, finite = 0, True

    def keep(o):
        nonlocal frames_seen, finite
        if o["codes"] is None:
            return
        f0, n = o["frame0"], o["codes"][0].shape[-1]
        assert f0 == frames_seen, (f0, frames_seen)
        frames_seen += n
        if f0 < check_frames:
            m = min(n, check_frames - f0)
            keep_codes[:, :, f0:f0 + m] = torch.cat(o["codes"], 1)[:, :, :m]
            keep_wave[:, :, 300 * f0:300 * (f0 + m)] = o["wave"][:, :, :300 * m]

    with torch.no_grad():
        if check_frames:        # the offline pass comes first: the session is conditioned on ITS timbre vector (as in the parity test)
            # timbre from a 2 s enrolment clip handed in as the "full utterance" (modules/quantize.py:378-383): the style encoder's
            # attention over all frames of a 5-minute signal is outside its kernel's tile, and a session is enrolled this way anyway
            enrol = loop[:, 0, :48000].contiguous()
            lens = torch.full((a.streams,), 48000, dtype=torch.int64, device=dev)
            z = model.encoder(loop)
            outs, _, _, _, timbre, codes = model.quantizer(z, loop, n_c=2, return_codes=True, full_waves=enrol, wave_lens=lens)
            ref_codes = torch.cat(codes, 1)[:, :, :check_frames].clone()
            ref_y = model.decoder(outs)[:, :, :check_frames * 300].clone()
            del z, outs, codes
            torch.cuda.empty_cache()
        else:
            enrol = loop[:, :, :48000]
            timbre = model.quantizer(model.encoder(enrol), enrol, n_c=2)[4]
        sess = StreamingCodec(model, timbre, n_c=2, use_graphs=not a.no_graphs)
        first = sess.prime(loop[:, :, :4800])
        if check_frames:
            keep(first)
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
            if check_frames:
                keep(out)                                    # device-side copies, outside the timed bracket
                if h % 1000 == 999 and out["wave"] is not None:
                    finite = finite and bool(torch.isfinite(out["wave"]).all())
        wall = time.perf_counter() - t_all
        drift = None
        if check_frames:
            torch.cuda.synchronize()                         # the first pass over the buffer IS the checked prefix (+ 1 s)
            mism = int((ref_codes != keep_codes).sum())
            wave_rel = float((keep_wave - ref_y).abs().max() / ref_y.abs().max())
            drift = {"checked_minutes": a.check_minutes, "checked_frames": check_frames, "checked_hops": int(check_frames * 300 // HOP),
                     "code_mismatches_vs_offline": mism, "codes_compared": int(ref_codes.numel()), "wave_rel_err_vs_offline": wave_rel,
                     "frames_emitted_total": frames_seen, "frame_numbering_contiguous": True, "outputs_finite_to_the_end": finite,
                     "reference": "offline causal model (encoder -> quantizer -> decoder) over the same first minutes + 1 s in one pass"}
    steady = sorted(lat[10:])
    q = lambda p: round(1e3 * steady[min(len(steady) - 1, int(p * len(steady)))], 4)  # noqa: E731
    audio_s = n_hops * HOP / 24000.0
    print(json.dumps({"metric": "streaming per-hop latency / RTF", "hop_samples": HOP, "streams": a.streams,
                      "hops": n_hops, "audio_minutes": round(audio_s / 60, 2), "graphs": not a.no_graphs,
                      "p50_ms": q(0.5), "p90_ms": q(0.9), "p99_ms": q(0.99), "max_ms": round(1e3 * steady[-1], 3),
                      "rtf": round(wall / audio_s, 5), "wall_s": round(wall, 2),
                      "frames_emitted_last_hop": None if out["codes"] is None else int(out["codes"][0].shape[-1]),
                      "drift_check": drift}))


if __name__ == "__main__":
    main()
