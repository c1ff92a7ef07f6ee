import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import facodec_oracle as O
from facodec_amd import synth
from facodec_amd.commons import build_model, default_model_params
model = build_model(default_model_params())
sds = {k: synth.synth_state_dict(synth.param_shapes(model[k]), 0, k + ".") for k in model}
wave = synth.synth_clips(4, 48000, seed=0)
for n in (8, 16, 32, 64, 128):
    torch.set_num_threads(n)
    with torch.no_grad():
        O.codec_forward(sds, wave[:1])
        t = time.perf_counter(); O.codec_forward(sds, wave); dt = time.perf_counter() - t
    print(n, 'threads', round(8 / dt, 3), 'audio-s/s', flush=True)
