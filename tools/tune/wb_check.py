"""Seeded train steps (configs[2], B = 16 x 2 s, learning rate on: the weights move every step) -> per-step losses, gradient norms and a
hash of every parameter arena after the last step; FAC_WEIGHT_BATCH=0 / 1 must print the same line (tests/test_weight_batch.py)."""
import hashlib, json, os, sys, random
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from facodec_amd import synth
from facodec_amd.commons import build_model, default_model_params
from facodec_amd.train import TrainStep
dev = torch.device("cuda:0")
torch.manual_seed(1234); np.random.seed(1234); random.seed(1234)
model = build_model(default_model_params())
for k in ("encoder", "quantizer", "decoder", "discriminator"):
    synth.load_synthetic(model[k], seed=0, prefix=k + ".")
    model[k].to(dev)
step = TrainStep(model)
wave = synth.synth_clips(16, 48000, seed=0).to(dev)
outs = []
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    torch.manual_seed(100 + i); np.random.seed(100 + i)
    o = step(wave)
    outs.append((float(o["loss"]), float(o["loss_d"]), float(o["mel"]), {k: float(v) for k, v in o["grad_norm"].items()}))
torch.cuda.synchronize()
sha = {k: hashlib.sha256(step.opt[k].p.detach().cpu().numpy().tobytes()).hexdigest()[:16] for k in step.opt}
info = {k: c.info() for k, c in step._weight_caches().items()}
print("WB " + json.dumps(dict(steps=outs, params=sha)))
print("WBINFO " + json.dumps(info))
