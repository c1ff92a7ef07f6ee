"""Which parameters of the quantizer key differ when FAquantizer's three chains run side by side (QUANT_STREAMS = 3) instead of in
turn (1)?  The B = 8 half-batch step of tests/test_train_golden.py::test_train_step_batch16_linearity_full_size, lr = 0, fixed
masks, dropout off: the serial run is the reference, then N concurrent runs; per run the parameters whose gradient differs."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from facodec_amd import quantize, synth  # noqa: E402
from facodec_amd.commons import build_model, default_model_params  # noqa: E402
from facodec_amd.train import TrainStep  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    model = build_model(default_model_params())
    for k in ("encoder", "quantizer", "decoder", "discriminator"):
        synth.load_synthetic(model[k], seed=0, prefix=k + ".")
        model[k].to(dev)
    step = TrainStep(model, lr=0.0)
    full = synth.synth_clips(16, 48000, seed=4).to(dev)
    names = [n for n, p in model.quantizer.named_parameters() if p.requires_grad]
    opt = step.opt["quantizer"]

    def run(lo, hi, streams):
        B = 16
        ones = lambda n: torch.ones(n, B)   # noqa: E731
        masks = dict(p=ones(1), c=ones(2), r=torch.cat([ones(2), (torch.arange(B) % 2).float().reshape(1, B)]),
                     res=(torch.arange(B) % 4 != 1).float(), dropout=False)
        mk = {k: (v[..., lo:hi].contiguous().to(dev) if torch.is_tensor(v) else v) for k, v in masks.items()}
        quantize.QUANT_STREAMS = streams
        step(full[lo:hi].contiguous(), masks=mk)
        torch.cuda.synchronize()
        return {k: step.opt[k].g.clone() for k in step.opt}

    # the test's own sequence first (everything concurrent from the first step on), the serial references afterwards
    run(0, 16, 3)
    trials = [(run(0, 8, 3), run(8, 16, 3)) for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4)]
    refs = (run(0, 8, 1), run(8, 16, 1))
    refs2 = (run(0, 8, 1), run(8, 16, 1))
    print("serial vs serial:", [float((refs[h]["quantizer"] - refs2[h]["quantizer"]).abs().max()) for h in (0, 1)])
    for t, pair in enumerate(trials):
        for h in (0, 1):
            ref, got = refs[h], pair[h]
            bad = []
            for i, (off, n) in enumerate(opt.slices):
                a, b = ref["quantizer"][off:off + n], got["quantizer"][off:off + n]
                d = float((a - b).abs().max())
                if d > 1e-6 * max(1.0, float(a.abs().max())):
                    bad.append((names[i], round(float(a.norm()), 4), round(float(b.norm()), 4)))
            other = {k: float((ref[k] - got[k]).abs().max()) for k in ref if k != "quantizer"}
            print(f"trial {t} half {h}: quantizer norm {float(got['quantizer'].double().norm()):.6f} (serial {float(ref['quantizer'].double().norm()):.6f}); "
                  f"{len(bad)} parameters differ: {bad[:10]}; other keys {other}", flush=True)


if __name__ == "__main__":
    main()
