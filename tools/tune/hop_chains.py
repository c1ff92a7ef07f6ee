#!/usr/bin/env python
"""How long each chain of a streaming hop is by itself: the encoder chain and the quantizer + decoder chain of one steady-state hop
(B = 1) captured as two separate HIP graphs (same launches and buffers as the session's own graph; the counters are put back after
each capture, the numbers written are not used) and replayed alone, then both at once on two streams as the session does.
   python tools/tune/hop_chains.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from facodec_amd import synth  # noqa: E402
from facodec_amd.commons import build_model, default_model_params  # noqa: E402
from facodec_amd.streaming import HOP, StreamingCodec  # noqa: E402


def timed(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return round(1e3 * e0.elapsed_time(e1) / n, 1)


def main():
    dev = torch.device("cuda:0")
    model = build_model(default_model_params())
    for k in ("encoder", "quantizer", "decoder"):
        synth.load_synthetic(model[k], seed=0, prefix=k + ".")
        model[k].eval().to(dev)
    wave = synth.synth_clips(1, 4800 + 20 * HOP, seed=0).to(dev)
    res = {}
    with torch.no_grad():
        timbre = model.quantizer(model.encoder(wave), wave, n_c=2)[4]
        sess = StreamingCodec(model, timbre, n_c=2, use_graphs=False)
        sess.prime(wave[:, :, :4800])
        for h in range(10):
            sess.push(wave[:, :, 4800 + h * HOP:4800 + (h + 1) * HOP])
        torch.cuda.synchronize()
        for phase in range(5):
            hop = wave[:, :, 4800 + (10 + phase) * HOP:4800 + (11 + phase) * HOP].contiguous()
            base = sess._state()
            zero = [[0] * len(b) for b in base]
            side = torch.cuda.Stream()
            graphs = {}

            def chain_enc():
                return sess.enc.run(hop)

            def chain_pros():
                sess.qs.push(hop, None)
                return sess.qs.prosody(False)

            def chain_qd():
                sess.qs.push(hop, None)
                x = sess.qs.take_latents(False)
                outs, codes = sess.qs.run(2, False, x=x)
                return sess.dec.run(outs)

            def chain_dec_only():
                sess.qs.push(hop, None)
                x = sess.qs.take_latents(False)
                return sess.dec.run(torch.zeros(1, 1024, x.shape[-1], device=dev))

            for name, fn in (("encoder", chain_enc), ("prosody_branch", chain_pros), ("quantizer_decoder", chain_qd), ("decoder", chain_dec_only)):
                g = torch.cuda.CUDAGraph()
                keep = None
                with torch.cuda.graph(g):
                    keep = fn()
                sess._set_state(base, zero, 0)
                graphs[name] = (g, keep)
            row = {name: timed(g.replay) for name, (g, _) in graphs.items()}

            def both():
                main = torch.cuda.current_stream()
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    graphs["quantizer_decoder"][0].replay()
                graphs["encoder"][0].replay()
                main.wait_stream(side)

            row["both_on_two_streams (two graph launches)"] = timed(both)
            # what the hop costs as ONE eager step after these captures (state advances)
            sess._set_state(base, zero, 0)
            sess.push(hop)
            res["phase %d" % phase] = row
            print(phase, json.dumps(row), flush=True)
    if len(sys.argv) > 1:
        json.dump(res, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
