#!/usr/bin/env python
"""Every conv launch of one configs[1] forward step (B = 32 x 2 s): shape, kernel variant, time, TFLOP/s.
   python tools/fwd_layers.py [batch]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from facodec_amd import ops  # noqa: E402


class ShapeProfile(ops.ConvLaunchProfile):
    pass


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    dev = torch.device("cuda:0")
    model = bench.build(dev) if hasattr(bench, "build") else None
    wave = bench.synth.synth_clips(B, int(bench.CLIP_SECONDS * bench.SAMPLE_RATE), seed=1).to(dev)
    step = bench.make_step(model, wave)
    for _ in range(2):
        step()
    recs = []
    orig = ops._launch_conv

    def spy(d, what):
        buf = ops.C.create_string_buffer(96)
        ops._lib.load().fac_conv1d_variant(ops.C.byref(d), buf, 96)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        orig(d, what)
        e1.record()
        recs.append((dict(B=d.B, ci=d.C_in, co=d.C_out, T=d.T_out, K=d.K, s=d.stride, dil=d.dilation, ph=d.n_phase,
                          ain=bool(d.alpha_in), aout=bool(d.alpha_out), res=bool(d.res), y2=bool(d.y2), split=bool(d.w_split),
                          fused=bool(d.w_k1)), buf.value.decode()[:44], 2.0 * d.B * d.n_phase * d.C_out * d.T_out * d.C_in * d.K, e0, e1))

    ops._launch_conv = spy
    step()
    torch.cuda.synchronize()
    ops._launch_conv = orig
    tot = 0.0
    for sh, name, fl, e0, e1 in recs:
        ms = e0.elapsed_time(e1)
        tot += ms
        flags = "".join(c for c, on in (("i", sh["ain"]), ("o", sh["aout"]), ("r", sh["res"]), ("2", sh["y2"]), ("S", sh["split"]), ("F", sh["fused"])) if on)
        print(f"{sh['ci']:5d}->{sh['co']:5d} K{sh['K']:2d} s{sh['s']} d{sh['dil']} ph{sh['ph']} B{sh['B']:3d} T{sh['T']:6d} {flags:6s} {name:44s} {ms:7.3f} ms {fl / ms / 1e9:7.1f} TF")
    print("sum of conv launches: %.2f ms" % tot)


if __name__ == "__main__":
    main()
