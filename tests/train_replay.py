"""Shared by the CPU and GPU tests of the whole training iteration: loads tests/golden/train_step.npz (made by
tests/golden/make_golden_train.py from the REAL reference in .train() mode) and replays the same iteration through the
CPU oracle with torch autograd (train.py:265-374).  Test infrastructure only."""
import json
import os

import numpy as np
import torch

KEYS = ("encoder", "quantizer", "decoder", "discriminator", "fa_predictors")
SCALARS = ("loss_d", "loss_gen_all", "mel_loss", "loss_g", "loss_feature", "commitment_loss", "codebook_loss", "f0_loss", "uv_loss",
           "rev_f0_loss", "rev_uv_loss", "content_loss", "rev_content_loss", "spk_loss", "x_spk_loss", "stft_loss", "waveform_loss")


def load_fixture(golden_dir):
    d = np.load(os.path.join(golden_dir, "train_step.npz"))
    fx = {k: d[k] for k in d.files}
    waves = torch.from_numpy(fx["waves"])
    seg = int(fx["seg_frames"])
    fx["t"] = dict(
        waves=waves, wave_lens=torch.from_numpy(fx["wave_lens"]).to(torch.int64), starts=torch.from_numpy(fx["crop_start"]).to(torch.int64),
        wav_seg=torch.stack([waves[b, int(s) * 300:(int(s) + seg) * 300] for b, s in enumerate(fx["crop_start"])]).unsqueeze(1),
        masks=dict(p=torch.from_numpy(fx["mask_p"]), c=torch.from_numpy(fx["mask_c"]), r=torch.from_numpy(fx["mask_r"]),
                   res=torch.from_numpy(fx["mask_res"]), dropout=False),
        targets=dict(f0=torch.from_numpy(fx["f0_targets"]), uv=torch.from_numpy(fx["real_norm"]),
                     phones=torch.from_numpy(fx["phones"]).to(torch.int64), speaker=torch.from_numpy(fx["speaker"]).to(torch.int64)))
    fx["no_grad"] = json.loads(str(fx["params_without_grad"]))
    fx["fmap_shapes"] = json.loads(str(fx["fmap_shapes"]))
    return fx


def probe_index(numel, n=64):
    step = max(1, numel // n)
    return np.arange(0, numel, step)[:n]


def grad_probe_names(fx, key):
    pre = f"grad.{key}."
    return sorted({k[len(pre):-len(".norm")] for k in fx if k.startswith(pre) and k.endswith(".norm")})


def compare_grads(fx, key, named_grads, tol_norm, tol_probe):
    """named_grads: {parameter name: gradient tensor}.  Returns the worst (name, norm rel err, probe err / max|probe|)."""
    worst = ("", 0.0, 0.0)
    for n in grad_probe_names(fx, key):
        g = named_grads[n].detach().cpu().reshape(-1)
        ref_norm, ref_probe = float(fx[f"grad.{key}.{n}.norm"]), fx[f"grad.{key}.{n}.probe"]
        e_norm = abs(float(g.double().norm()) - ref_norm) / max(ref_norm, 1e-30)
        got = g[probe_index(g.numel())].numpy()
        e_probe = float(np.abs(got - ref_probe).max() / max(np.abs(ref_probe).max(), 1e-30))
        if max(e_norm / tol_norm, e_probe / tol_probe) > max(worst[1] / tol_norm, worst[2] / tol_probe):
            worst = (f"{key}.{n}", e_norm, e_probe)
    return worst


def product_state(model):
    """({key: CPU state dict incl. buffers}, {key: parameter names}) of a facodec_amd model (any device)."""
    sds = {k: {n: v.detach().cpu().clone() for n, v in model[k].state_dict().items()} for k in KEYS}
    names = {k: {n for n, _ in model[k].named_parameters()} for k in KEYS}
    return sds, names


def oracle_iteration(O, sds, param_names, fx):
    """train.py:265-374 through the oracle + torch autograd (oracle/train_iteration.py)."""
    from oracle.train_iteration import oracle_iteration as it
    return it(sds, param_names, fx["t"])
