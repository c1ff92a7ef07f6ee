"""Is the training step host-bound?  Time to ENQUEUE one step (Python + ctypes + autograd, no device sync) against the time
until the device has finished it."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from facodec_amd import synth
from facodec_amd.commons import build_model, default_model_params
from facodec_amd.train import TrainStep
dev = torch.device("cuda:0")
model = build_model(default_model_params())
for k in ("encoder", "quantizer", "decoder", "discriminator", "fa_predictors"):
    synth.load_synthetic(model[k], seed=0, prefix=k + ".")
    model[k].to(dev)
step = TrainStep(model, with_predictors=True)
wave = synth.synth_clips(16, 48000, seed=1).to(dev)
targets = bench.synthetic_predictor_targets(16, 160, dev)
for _ in range(2):
    step(wave, targets=targets)
torch.cuda.synchronize()
enq, tot = [], []
for _ in range(4):
    t0 = time.perf_counter()
    step(wave, targets=targets)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    enq.append(1e3 * (t1 - t0)); tot.append(1e3 * (t2 - t0))
print("enqueue ms:", [round(v, 1) for v in enq], " until done ms:", [round(v, 1) for v in tot])
