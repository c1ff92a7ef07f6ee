"""Thread sweep of the CPU oracle's forward (the `cpu_baseline` leg of bench.py) on this host: which thread count is the port's
best case.  Output committed under profiles/ (rNN_cpu_thread_sweep.log)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import facodec_oracle as O  # noqa: E402
from facodec_amd import synth  # noqa: E402
from facodec_amd.commons import build_model, default_model_params  # noqa: E402

model = build_model(default_model_params())
sds = {k: synth.synth_state_dict(synth.param_shapes(model[k]), 0, k + ".") for k in ("encoder", "quantizer", "decoder")}
wave = synth.synth_clips(4, 48000, seed=0)
print("logical cores:", os.cpu_count(), flush=True)
for n in (4, 8, 16, 32, 64, 128):
    if n > (os.cpu_count() or 1):
        break
    torch.set_num_threads(n)
    with torch.no_grad():
        O.codec_forward(sds, wave[:1])
        ts = []
        for _ in range(3):
            t = time.perf_counter()
            O.codec_forward(sds, wave)
            ts.append(time.perf_counter() - t)
    print(n, "threads:", round(8 / sorted(ts)[1], 3), "audio-s/s (median of 3 passes of 4 clips x 2 s)", flush=True)
