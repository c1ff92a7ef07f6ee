import os, sys, json, torch
sys.path.insert(0, "/root/repo")
from facodec_amd import synth
from facodec_amd.commons import build_model, default_model_params
from facodec_amd.train import TrainStep
dev = torch.device("cuda:0")
torch.manual_seed(1234)
import numpy as np, random
np.random.seed(1234); random.seed(1234)
model = build_model(default_model_params())
for k in ("encoder", "quantizer", "decoder", "discriminator"):
    synth.load_synthetic(model[k], seed=0, prefix=k + ".")
    model[k].to(dev)
step = TrainStep(model)
wave = synth.synth_clips(16, 48000, seed=0).to(dev)
outs = []
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    torch.manual_seed(100 + i); np.random.seed(100 + i)
    o = step(wave)
    outs.append((float(o["loss"]), float(o["mel"]), {k: float(v) for k, v in o["grad_norm"].items()}))
print("PW=%s" % os.environ.get("FAC_PW", "1"), json.dumps(outs))
