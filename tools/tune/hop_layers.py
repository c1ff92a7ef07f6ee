#!/usr/bin/env python
"""Every conv launch of one steady-state streaming hop (B = 1), by shape: captured as a HIP graph of 24 back-to-back launches of
that conv and replayed -- the time per launch the hop's graph pays for it (split-reduction kernel + its reduce kernel), without a
profiler in the way.   python tools/tune/hop_layers.py [out.json]"""
import ctypes as C
import json
import os
import sys
from collections import OrderedDict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from facodec_amd import ops, synth  # noqa: E402
from facodec_amd.commons import build_model, default_model_params  # noqa: E402
from facodec_amd.streaming import HOP, StreamingCodec  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    model = build_model(default_model_params())
    for k in ("encoder", "quantizer", "decoder"):
        synth.load_synthetic(model[k], seed=0, prefix=k + ".")
        model[k].eval().to(dev)
    wave = synth.synth_clips(1, 4800 + 12 * HOP, seed=0).to(dev)
    recs = []
    with torch.no_grad():
        enrol = wave[:, :, :48000]
        timbre = model.quantizer(model.encoder(wave), wave, n_c=2)[4]
        sess = StreamingCodec(model, timbre, n_c=2, use_graphs=False)
        sess.two_streams = False
        sess.prime(wave[:, :, :4800])
        for h in range(5):
            sess.push(wave[:, :, 4800 + h * HOP:4800 + (h + 1) * HOP])
        orig = ops._launch_conv
        keep = []

        def spy(d, what):
            buf = C.create_string_buffer(96)
            ops._lib.load().fac_conv1d_variant(C.byref(d), buf, 96)
            d2 = ops.ConvDesc()
            C.memmove(C.byref(d2), C.byref(d), C.sizeof(d))
            recs.append((d2, buf.value.decode()[:40]))
            orig(d, what)

        ops._launch_conv = spy
        outs = []
        for h in range(5, 10):          # one period; tensors are kept alive so that the recorded pointers stay valid
            outs.append(sess.push(wave[:, :, 4800 + h * HOP:4800 + (h + 1) * HOP]))
            keep.append(outs[-1])
        ops._launch_conv = orig
    torch.cuda.synchronize()
    # the recorded descriptors point at freed activations; give each its own buffers of the right size
    groups = OrderedDict()
    for d, name in recs:
        key = (d.B, d.C_in, d.C_out, d.T_in, d.T_out, d.K, d.stride, d.dilation, d.n_phase, d.act, bool(d.res), bool(d.y2), bool(d.alpha_out), name)
        groups.setdefault(key, [0, d])[0] += 1
    lib = ops._lib.load()
    rows = []
    for key, (count, d) in groups.items():
        x = torch.randn(d.B, d.C_in, max(d.T_in, 1) + 8, device=dev)
        ny = d.B * d.C_out * d.T_out * max(1, d.n_phase) + 64
        y, y2, res = torch.empty(ny, device=dev), torch.empty(ny, device=dev), torch.randn(ny, device=dev)
        d.x = x.data_ptr()
        d.x_bs, d.x_cs = x.stride(0), x.stride(1)
        d.y = y.data_ptr()
        d.y_bs, d.y_cs = d.C_out * d.T_out * max(1, d.n_phase), d.T_out * max(1, d.n_phase)
        if d.y2:
            d.y2 = y2.data_ptr()
        if d.res:
            d.res = res.data_ptr()
        if d.act in (ops.ACT_GATE, ops.ACT_WN_RES_SKIP):
            d.y_bs = (d.C_out // 2) * d.T_out
        ws = ops._conv_workspace(dev)
        d.ws, d.ws_bytes = ws.data_ptr(), ops.CONV_WS_BYTES
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            for _ in range(3):
                ops._lib.check(lib.fac_conv1d_fwd(C.byref(d), ops._stream()), "conv")
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(24):
                ops._lib.check(lib.fac_conv1d_fwd(C.byref(d), ops._stream()), "conv")
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / (20 * 24)
        rows.append(dict(B=key[0], C_in=key[1], C_out=key[2], T_in=key[3], T_out=key[4], K=key[5], stride=key[6], dil=key[7], phases=key[8],
                         act=key[9], res=key[10], y2=key[11], snake=key[12], kernel=key[13], launches_per_period=count, us=round(us, 2)))
    rows.sort(key=lambda r: -r["us"] * r["launches_per_period"])
    tot = sum(r["us"] * r["launches_per_period"] for r in rows) / 5
    print("conv launches per hop: %.1f, their graph time per hop: %.1f us" % (sum(r["launches_per_period"] for r in rows) / 5, tot))
    for r in rows:
        print("%d %5d %5d T_in %4d T_out %4d K %2d s %d d %2d ph %d act %d res %d y2 %d sn %d | x%2d %7.2f us  %s" % (
            r["B"], r["C_in"], r["C_out"], r["T_in"], r["T_out"], r["K"], r["stride"], r["dil"], r["phases"], r["act"], r["res"], r["y2"], r["snake"],
            r["launches_per_period"], r["us"], r["kernel"]))
    if len(sys.argv) > 1:
        json.dump(dict(per_hop_us=tot, rows=rows), open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
