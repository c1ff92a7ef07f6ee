"""Where the HOST spends its time in a configs[2] train step (Python / ctypes / torch dispatcher overhead of ~7 500 launches): cProfile
over two steps with the autograd engine on the calling thread (torch.autograd.set_multithreading_enabled(False)), sorted by own time.
The GPU is idle ~10 % of a step (tools/gpu_idle.py): whatever keeps the host from running ahead of the device shows up here.
   python tools/tune/host_profile.py [top N]"""
import cProfile
import io
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from facodec_amd import synth  # noqa: E402
from facodec_amd.commons import build_model, default_model_params  # noqa: E402
from facodec_amd.train import TrainStep  # noqa: E402


def main():
    top = int(sys.argv[1]) if len(sys.argv) > 1 else 45
    dev = torch.device("cuda:0")
    model = build_model(default_model_params())
    for k in ("encoder", "quantizer", "decoder", "discriminator", "fa_predictors"):
        synth.load_synthetic(model[k], seed=0, prefix=k + ".")
        model[k].to(dev)
    step = TrainStep(model, with_predictors=True)
    wave = synth.synth_clips(16, 48000, seed=1).to(dev)
    targets = bench.synthetic_predictor_targets(16, 160, dev, seed=3)
    for _ in range(2):
        step(wave, targets=targets)
    torch.cuda.synchronize()
    # host-only time of a step: enqueue everything, stop the clock BEFORE waiting for the device
    t0 = time.perf_counter()
    for _ in range(3):
        step(wave, targets=targets)
    t_host = (time.perf_counter() - t0) / 3
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / 3
    print(f"host enqueue time per step {1e3 * t_host:.1f} ms; wall per step incl. device {1e3 * t_all:.1f} ms", flush=True)
    # host time of the bookkeeping at the step boundary (the device is nearly idle there: tools/gpu_idle.py shows one 3.6 - 4.3 ms gap per step)
    import collections
    from facodec_amd import optim, train
    acc = collections.Counter()

    def timed(cls, name):
        orig = getattr(cls, name)

        def wrapped(*a, **k):
            t = time.perf_counter()
            try:
                return orig(*a, **k)
            finally:
                acc[cls.__name__ + "." + name] += time.perf_counter() - t
        setattr(cls, name, wrapped)
        return orig

    saved = [(c, n, timed(c, n)) for c, n in ((train.GeneratorStep, "_zero"), (optim.FlatAdamW, "_rebind"), (optim.FlatAdamW, "zero_grad"),
                                              (optim.FlatAdamW, "step"), (optim.FlatAdamW, "exchange_for_step"), (optim.FlatAdamW, "end_step"),
                                              (optim.FlatAdamW, "_upload_flags"), (train.GeneratorStep, "_report_lstm_timeouts"))]
    t0 = time.perf_counter()
    for _ in range(3):
        step(wave, targets=targets)
    t_host = (time.perf_counter() - t0) / 3
    torch.cuda.synchronize()
    print(f"host enqueue {1e3 * t_host:.1f} ms per step; of it (ms per step, nested calls counted in their parents too):",
          {k: round(1e3 * v / 3, 2) for k, v in sorted(acc.items(), key=lambda kv: -kv[1])}, flush=True)
    for c, n, o in saved:
        setattr(c, n, o)
    torch.autograd.set_multithreading_enabled(False)
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(2):
        step(wave, targets=targets)
    pr.disable()
    torch.cuda.synchronize()
    for key in ("tottime", "cumtime"):
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats(key).print_stats(top)
        print(s.getvalue()[:9000])


if __name__ == "__main__":
    main()
