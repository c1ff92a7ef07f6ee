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
    """train.py:265-374 through the oracle + torch autograd.  sds: {key: flat state dict}; param_names: {key: names of
    the trained tensors}.  Returns (scalars, grads {key: {name: grad}}, grad norms, discriminator parameters after its
    AdamW step)."""
    import torch.nn.functional as F
    t = fx["t"]
    wav, tg = t["wav_seg"], t["targets"]
    leaves = {k: {n: (v.clone().requires_grad_() if n in param_names[k] else v) for n, v in sds[k].items()} for k in KEYS}
    z = O.encoder_forward(leaves["encoder"], wav)
    outs, quantized, cm, cb, timbre, _ = O.quantizer_forward_train(leaves["quantizer"], z, wav, t["masks"], side_branches_no_grad=False,
                                                                full_waves=t["waves"], wave_lens=t["wave_lens"])
    preds, rev = O.predictors_forward(leaves["fa_predictors"], quantized, timbre)
    pred = O.decoder_forward(leaves["decoder"], outs)
    # discriminator step
    dl = leaves["discriminator"]
    ld, _, _ = O.gan_losses(O.discriminator_forward(dl, pred.detach()), O.discriminator_forward(dl, wav))
    dparams = [v for v in dl.values() if v.requires_grad]
    dgrads = torch.autograd.grad(ld, dparams)
    grads = {"discriminator": {n: g for (n, v), g in zip(((n, v) for n, v in dl.items() if v.requires_grad), dgrads)}}
    for p, g in zip(dparams, dgrads):
        p.grad = g.clone()
    norms = {"discriminator": float(torch.sqrt(sum(g.double().pow(2).sum() for g in dgrads)))}
    torch.nn.utils.clip_grad_norm_(dparams, 10.0)
    torch.optim.AdamW(dparams, lr=1e-4, betas=(0.9, 0.98), eps=1e-9, weight_decay=0.1).step()
    d_after = {n: v.detach().clone() for n, v in dl.items()}
    # generator step
    dl2 = {n: v.detach() for n, v in dl.items()}
    _, lg, lf = O.gan_losses(O.discriminator_forward(dl2, pred), O.discriminator_forward(dl2, wav))
    mel = O.mel_spectrogram_loss(pred, wav)
    n = min(preds["f0"].shape[-2], tg["f0"].shape[-1])
    f0_t, uv_t, ph = tg["f0"][..., :n], tg["uv"][..., :n], tg["phones"][..., :n]
    sc = dict(f0_loss=F.smooth_l1_loss(f0_t, preds["f0"].squeeze(-1)[..., :n]), uv_loss=F.smooth_l1_loss(uv_t, preds["uv"].squeeze(-1)[..., :n]),
              rev_f0_loss=F.smooth_l1_loss(f0_t, rev["rev_f0"].squeeze(-1)[..., :n]),
              rev_uv_loss=F.smooth_l1_loss(uv_t, rev["rev_uv"].squeeze(-1)[..., :n]),
              content_loss=O.focal_loss(preds["content"].transpose(1, 2)[..., :n], ph),
              rev_content_loss=O.focal_loss(rev["rev_content"].transpose(1, 2)[..., :n], ph),
              spk_loss=F.cross_entropy(preds["timbre"], tg["speaker"]), x_spk_loss=F.cross_entropy(rev["x_timbre"], tg["speaker"]))
    total = 15.0 * mel + lf + lg + 0.25 * cm + cb + (sc["f0_loss"] + sc["rev_f0_loss"]) + (sc["uv_loss"] + sc["rev_uv_loss"]) \
        + 5.0 * (sc["content_loss"] + sc["rev_content_loss"]) + (sc["spk_loss"] + sc["x_spk_loss"])
    total.backward()
    for k in ("encoder", "quantizer", "decoder", "fa_predictors"):
        grads[k] = {n: v.grad for n, v in leaves[k].items() if v.requires_grad and v.grad is not None}
        norms[k] = float(torch.sqrt(sum(g.double().pow(2).sum() for g in grads[k].values())))
    with torch.no_grad():
        sc.update(loss_d=ld, loss_gen_all=total, mel_loss=mel, loss_g=lg, loss_feature=lf, commitment_loss=cm, codebook_loss=cb,
                  stft_loss=O.multiscale_stft_loss(pred, wav), waveform_loss=O.waveform_l1_loss(pred, wav))
    return {k: float(v) for k, v in sc.items()}, grads, norms, d_after
