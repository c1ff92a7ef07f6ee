"""One whole training iteration of the reference (train.py:265-374: forward, discriminator step, generator step with all
seventeen loss terms) restated through the CPU oracle + torch autograd.  TEST INFRASTRUCTURE: used by the tests (pinned to
the real reference's .train() run, tests/golden/train_step.npz) and by bench.py's `cpu_baseline.train` leg as the timed CPU
port; never by the product path."""
import torch
import torch.nn.functional as F

from . import facodec_oracle as O

KEYS = ("encoder", "quantizer", "decoder", "discriminator", "fa_predictors")


def oracle_iteration(sds, param_names, t):
    """train.py:265-374 through the oracle + torch autograd.  sds: {key: flat state dict}; param_names: {key: names of
    the trained tensors}; t: dict(wav_seg, targets, masks, waves, wave_lens) as tests/train_replay.load_fixture builds it.  Returns (scalars, grads {key: {name: grad}}, grad norms, discriminator parameters after its
    AdamW step)."""
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
