"""Train-mode golden vectors from the REAL reference (build container only; the reference never travels).

Run:  python tests/golden/make_golden_train.py            (about 3 minutes on 8 cores)

What it does: builds the reference's five networks (modules/commons.py:283-348) with the formula weights of
facodec_amd/synth.py, puts them in .train() mode and executes the loss assembly of train.py:188-212,265-374
on fixed inputs, with the reference's three random sites replaced by recorded values:

  * torch.randint          dac/nn/quantize.py:166 (quantizer dropout, one call per RVQ: prosody, content, residual)
  * np.random.choice       modules/quantize.py:420  (residual mask)
  * np.random.randint      train.py:195             (random crop start)
  * every nn.Dropout       p = 0                    (WaveNet 0.2, StyleEncoder 0.1)

The predictor targets that train.py obtains from external networks (pitch extractor, wav2vec CTC phones, speaker
model) are fixed random tensors of the right shape / range.  Stored in train_step.npz: the inputs, the recorded
random draws (as masks), every loss scalar, the five pre-clip gradient norms, and gradient probes (norm + strided
slice) of ~40 tensors of the generator step and ~10 of the discriminator step, taken at the moments train.py calls
clip_grad_norm_ (:290, :362-365).

Also written here: fvq.npz (quantize/fvq.py + quantize/rvq.py, eval and train forward), recon_misc.npz
(losses.py:65-89 reconstruction_loss, losses.py:264-276 FocalLoss, meldataset.py:42-47 preprocess -- the two mel
rows run over the torchaudio shim of make_golden.py: parity unpinned for the STFT itself).
"""
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)

import make_golden as MG  # noqa: E402
from facodec_amd import synth  # noqa: E402
from oracle import facodec_oracle as O  # noqa: E402

KEYS = ("encoder", "quantizer", "decoder", "discriminator", "fa_predictors")
B, SEG_FRAMES, T_FULL = 4, 20, 12000
WAVE_LENS = [12000, 9000, 10500, 12000]
CROP_START = [3, 0, 7, 10]                 # frames; train.py:195 `np.random.randint(0, mel_length - mel_seg_len)`
DROPOUT_DRAWS = {"p": [1, 1, 1, 1], "c": [1, 2, 2, 1], "r": [2, 1, 3, 3]}   # torch.randint(1, n + 1, (B,)) per RVQ
RES_MASK = [1, 0, 1, 1]


def probe_index(numel, n=64):
    step = max(1, numel // n)
    return np.arange(0, numel, step)[:n]


def grad_probes(model, key, names):
    out = {}
    params = dict(model[key].named_parameters())
    for n in names:
        g = params[n].grad
        flat = g.reshape(-1)
        out[f"grad.{key}.{n}.norm"] = np.float64(flat.double().norm())
        out[f"grad.{key}.{n}.probe"] = flat[probe_index(flat.numel())].numpy().copy()
    return out


def masks_from_draws(n_codebooks, draws, quantizer_dropout=0.5, B=B):
    """dac/nn/quantize.py:163-183: n_quantizers = n_codebooks + 1, first int(B * p) samples take the draw."""
    nq = torch.ones(B) * n_codebooks + 1
    nd = int(B * quantizer_dropout)
    nq[:nd] = torch.tensor(draws[:nd], dtype=torch.float32)
    return torch.stack([(torch.full((B,), float(i)) < nq).float() for i in range(n_codebooks)])


def build_reference_in_train_mode():
    build_model, recursive_munch = MG.ref_imports()
    model = build_model(recursive_munch(MG.model_params()))
    sds = {}
    for k in KEYS:
        sds[k] = synth.load_synthetic(model[k], seed=0, prefix=k + ".")
        model[k].train()
        for m in model[k].modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
    return model, sds


DEFAULT_CFG = dict(B=B, SEG_FRAMES=SEG_FRAMES, T_FULL=T_FULL, WAVE_LENS=WAVE_LENS, CROP_START=CROP_START, DROPOUT_DRAWS=DROPOUT_DRAWS,
                   RES_MASK=RES_MASK, wave_seed=23, target_seed=99, probes=True)


def iteration(model, cfg):
    """One train.py:188-374 iteration of the real reference on the inputs `cfg` describes.  Returns (out: what goes into the
    .npz, aux: live tensors for the oracle pinning in main())."""
    B, SEG_FRAMES, T_FULL, WAVE_LENS, CROP_START = cfg["B"], cfg["SEG_FRAMES"], cfg["T_FULL"], cfg["WAVE_LENS"], cfg["CROP_START"]
    DROPOUT_DRAWS, RES_MASK = cfg["DROPOUT_DRAWS"], cfg["RES_MASK"]
    out = {}
    # ------------------------------------------------------------------ inputs (train.py:176-212)
    waves = synth.synth_clips(B, T_FULL, seed=cfg["wave_seed"]).squeeze(1)   # (B, T_full) padded batch
    wave_lengths = torch.tensor(WAVE_LENS)
    for b, n in enumerate(WAVE_LENS):
        waves[b, n:] = 0.0
    g = torch.Generator().manual_seed(cfg["target_seed"])
    f0_targets = torch.randn(B, SEG_FRAMES, generator=g)
    f0_targets[torch.rand(B, SEG_FRAMES, generator=g) < 0.3] = -10.0          # unvoiced frames (train.py:241)
    real_norm = torch.randn(B, SEG_FRAMES, generator=g)
    phones = torch.randint(0, 1024, (B, SEG_FRAMES), generator=g)
    spk_labels = torch.randint(0, 20000, (B,), generator=g)
    wav_seg = torch.stack([waves[b, s * 300:(s + SEG_FRAMES) * 300] for b, s in enumerate(CROP_START)]).float().unsqueeze(1)
    out.update(waves=waves.numpy(), wave_lens=np.array(WAVE_LENS), crop_start=np.array(CROP_START), seg_frames=np.int64(SEG_FRAMES),
               f0_targets=f0_targets.numpy(), real_norm=real_norm.numpy(), phones=phones.numpy(), speaker=spk_labels.numpy())

    # ------------------------------------------------------------------ patched random sites
    draws = [DROPOUT_DRAWS["p"], DROPOUT_DRAWS["c"], DROPOUT_DRAWS["r"]]
    calls = {"randint": 0}
    real_randint, real_choice = torch.randint, np.random.choice

    def fake_randint(lo, hi, size, **kw):
        d = draws[calls["randint"] % 3]
        calls["randint"] += 1
        assert tuple(size) == (B,) and all(lo <= v < hi for v in d), (lo, hi, size, d)
        return torch.tensor(d)

    def fake_choice(a, size=None, p=None, **kw):
        assert size == B
        return np.array(RES_MASK)

    torch.randint, np.random.choice = fake_randint, fake_choice
    try:
        # -------------------------------------------------------------- forward (train.py:265-277)
        z = model.encoder(wav_seg)
        zq, quantized, commitment_loss, codebook_loss, timbre = model.quantizer(z, wav_seg, n_c=2, full_waves=waves,
                                                                              wave_lens=wave_lengths)
        preds, rev_preds = model.fa_predictors(quantized, timbre)
        pred_wave = model.decoder(zq)
    finally:
        torch.randint, np.random.choice = real_randint, real_choice
    assert calls["randint"] == 3
    wav_seg_target = wav_seg
    assert wav_seg_target.size(-1) == pred_wave.size(-1)
    masks = {k: masks_from_draws(n, DROPOUT_DRAWS[k], B=B) for k, n in (("p", 1), ("c", 2), ("r", 3))}
    out.update(mask_p=masks["p"].numpy(), mask_c=masks["c"].numpy(), mask_r=masks["r"].numpy(),
               mask_res=np.array(RES_MASK, np.float32))

    # -------------------------------------------------------------- discriminator step (train.py:279-292)
    d_fake = model.discriminator(pred_wave.detach())
    d_real = model.discriminator(wav_seg_target)
    loss_d = 0
    for x_fake, x_real in zip(d_fake, d_real):
        loss_d += torch.mean(x_fake[-1] ** 2)
        loss_d += torch.mean((1 - x_real[-1]) ** 2)
    for k in KEYS:
        model[k].zero_grad()
    loss_d.backward()
    disc_probe_names = [n for n, _ in model.discriminator.named_parameters()
                        if n.startswith(("discriminators.0.convs.0.", "discriminators.4.convs.3.0.weight_v", "discriminators.2.conv_post.",
                                         "discriminators.5.band_convs.0.0.", "discriminators.7.band_convs.4.3.0.weight_g",
                                         "discriminators.6.conv_post."))]
    if cfg["probes"]:
        out.update(grad_probes(model, "discriminator", disc_probe_names))
    out["grad_norm64_discriminator"] = np.float64(torch.sqrt(sum(p.grad.double().pow(2).sum() for p in model.discriminator.parameters())))
    gn_d = torch.nn.utils.clip_grad_norm_(model.discriminator.parameters(), 10.0)
    opt_d = torch.optim.AdamW(model.discriminator.parameters(), lr=1e-4, betas=(0.9, 0.98), eps=1e-9, weight_decay=0.1)  # optimizers.py:91-97
    opt_d.step()
    out.update(loss_d=np.float64(loss_d.detach()), grad_norm_discriminator=np.float64(gn_d))
    pd = dict(model.discriminator.named_parameters())
    for n in disc_probe_names[:4] if cfg["probes"] else []:
        flat = pd[n].detach().reshape(-1)
        out[f"param_after.discriminator.{n}.probe"] = flat[probe_index(flat.numel())].numpy().copy()

    # -------------------------------------------------------------- generator step (train.py:294-374)
    from audiotools import AudioSignal
    from dac.nn.loss import L1Loss, MelSpectrogramLoss, MultiScaleSTFTLoss
    sys.modules.setdefault("torchaudio.transforms", sys.modules["torchaudio"].transforms)
    from losses import FocalLoss, reconstruction_loss
    content_criterion = FocalLoss(gamma=2)
    mel_criterion = MelSpectrogramLoss(n_mels=[5, 10, 20, 40, 80, 160, 320], window_lengths=[32, 64, 128, 256, 512, 1024, 2048],
                                       mel_fmin=[0] * 7, mel_fmax=[None] * 7, pow=1.0, mag_weight=0.0, clamp_eps=1e-5)
    signal, recons = AudioSignal(wav_seg_target, sample_rate=24000), AudioSignal(pred_wave, sample_rate=24000)
    stft_loss = MultiScaleSTFTLoss()(recons, signal)
    mel_loss = mel_criterion(recons, signal)
    waveform_loss = L1Loss()(recons, signal)
    d_fake = model.discriminator(pred_wave)
    d_real = model.discriminator(wav_seg_target)
    loss_g = 0
    for x_fake in d_fake:
        loss_g += torch.mean((1 - x_fake[-1]) ** 2)
    loss_feature = 0
    for i in range(len(d_fake)):
        for j in range(len(d_fake[i]) - 1):
            loss_feature += F.l1_loss(d_fake[i][j], d_real[i][j].detach())
    out["fmap_shapes"] = np.array(json.dumps([[list(t.shape) for t in fm] for fm in d_fake]))
    pred_f0, pred_uv = preds["f0"], preds["uv"]
    rev_pred_f0, rev_pred_uv = rev_preds["rev_f0"], rev_preds["rev_uv"]
    n = min(pred_f0.size(-2), f0_targets.size(-1))
    f0_t, rn_t = f0_targets[..., :n], real_norm[..., :n]
    f0_loss = F.smooth_l1_loss(f0_t, pred_f0.squeeze(-1)[..., :n])
    uv_loss = F.smooth_l1_loss(rn_t, pred_uv.squeeze(-1)[..., :n])
    rev_f0_loss = F.smooth_l1_loss(f0_t, rev_pred_f0.squeeze(-1)[..., :n])
    rev_uv_loss = F.smooth_l1_loss(rn_t, rev_pred_uv.squeeze(-1)[..., :n])
    tot_f0_loss, tot_uv_loss = f0_loss + rev_f0_loss, uv_loss + rev_uv_loss
    tgt = phones[..., :n].float()
    content_loss = content_criterion(preds["content"].transpose(1, 2)[..., :n], tgt.long())
    rev_content_loss = content_criterion(rev_preds["rev_content"].transpose(1, 2)[..., :n], tgt.long())
    tot_content_loss = content_loss + rev_content_loss
    spk_loss = F.cross_entropy(preds["timbre"], spk_labels)
    x_spk_loss = F.cross_entropy(rev_preds["x_timbre"], spk_labels)
    tot_spk_loss = spk_loss + x_spk_loss
    loss_gen_all = mel_loss * 15.0 + loss_feature * 1.0 + loss_g * 1.0 + commitment_loss * 0.25 + codebook_loss * 1.0 \
        + tot_f0_loss * 1.0 + tot_uv_loss * 1.0 + tot_content_loss * 5.0 + tot_spk_loss * 1.0
    for k in KEYS:
        model[k].zero_grad()
    loss_gen_all.backward()
    gen_probe = {
        "encoder": ["block.0.conv.conv.weight_v", "block.0.conv.conv.weight_g", "block.0.conv.conv.bias", "block.1.block.0.block.0.alpha",
                    "block.2.block.1.block.1.conv.conv.weight_v", "block.3.block.4.conv.conv.weight_g", "block.5.lstm.weight_hh_l0",
                    "block.5.lstm.weight_ih_l1", "block.5.lstm.bias_hh_l1", "block.7.conv.conv.weight_v"],
        "quantizer": ["prosody_quantizer.quantizers.0.codebook.weight", "prosody_quantizer.quantizers.0.in_proj.weight_g",
                      "content_quantizer.quantizers.1.codebook.weight", "content_quantizer.quantizers.0.out_proj.weight_v",
                      "residual_quantizer.quantizers.2.in_proj.weight_v", "residual_quantizer.quantizers.0.codebook.weight",
                      "residual_quantizer.quantizers.1.out_proj.bias", "timbre_linear.weight", "timbre_linear.bias",
                      "melspec_linear.conv.conv.weight", "melspec_encoder.in_layers.3.conv.conv.weight_v",
                      "melspec_encoder.res_skip_layers.7.conv.conv.weight_g", "melspec_linear2.conv.conv.bias",
                      "timbre_encoder.spectral.0.weight", "timbre_encoder.temporal.1.conv1.weight", "timbre_encoder.slf_attn.conv_q.weight", "timbre_encoder.slf_attn.conv_o.bias", "timbre_encoder.fc.weight"],
        "decoder": ["model.0.conv.conv.weight_v", "model.1.lstm.weight_hh_l1", "model.2.block.1.convtr.convtr.weight_v",
                    "model.2.block.1.convtr.convtr.weight_g", "model.3.block.3.block.1.conv.conv.weight_v", "model.4.block.0.alpha",
                    "model.5.block.4.block.3.conv.conv.bias", "model.7.conv.conv.weight_v", "model.7.conv.conv.weight_g"],
        "fa_predictors": ["f0_predictor.heads.0.weight", "f0_predictor.model.0.block.1.weight_v", "f0_predictor.model.1.block.0.act.alpha",
                          "phone_predictor.heads.0.bias", "timbre_predictor.weight", "rev_f0_predictor.1.heads.1.weight",
                          "rev_content_predictor.1.model.2.block.3.weight_g", "rev_timbre_predictor.1.heads.0.weight",
                          "rev_timbre_predictor.1.model.0.block.2.act.beta"],
    }
    for k, names in gen_probe.items():
        have = dict(model[k].named_parameters())
        missing = [n for n in names if n not in have]
        assert not missing, (k, missing, list(have)[:40])
        if cfg["probes"]:
            out.update(grad_probes(model, k, names))
    # parameters that must receive no gradient in the reference (so the optimiser skips them)
    no_grad = {k: [n for n, p in model[k].named_parameters() if p.grad is None] for k in KEYS if k != "discriminator"}
    out["params_without_grad"] = np.array(json.dumps(no_grad))
    for k in ("encoder", "decoder", "quantizer", "fa_predictors"):   # fp64 total (torch's fp32 CPU norm of 1e8 elements is itself 1e-4 off)
        out[f"grad_norm64_{k}"] = np.float64(torch.sqrt(sum(p.grad.double().pow(2).sum() for p in model[k].parameters() if p.grad is not None)))
    gn = {k: torch.nn.utils.clip_grad_norm_(model[k].parameters(), 1000.0) for k in ("encoder", "decoder", "quantizer", "fa_predictors")}
    scal = dict(loss_gen_all=loss_gen_all, mel_loss=mel_loss, stft_loss=stft_loss, waveform_loss=waveform_loss, loss_g=loss_g,
                loss_feature=loss_feature, commitment_loss=commitment_loss, codebook_loss=codebook_loss, f0_loss=f0_loss, uv_loss=uv_loss,
                rev_f0_loss=rev_f0_loss, rev_uv_loss=rev_uv_loss, content_loss=content_loss, rev_content_loss=rev_content_loss,
                spk_loss=spk_loss, x_spk_loss=x_spk_loss)
    out.update({k: np.float64(v.detach()) for k, v in scal.items()})
    out.update({f"grad_norm_{k}": np.float64(v) for k, v in gn.items()})
    out.update(pred_wave_probe=pred_wave.detach()[:, 0, ::13].numpy(), timbre=timbre.detach().numpy(),
               zq_probe=zq.detach()[:, ::8, :].numpy(), z_probe=z.detach()[:, ::8, :].numpy())
    aux = dict(wav_seg=wav_seg, z=z, zq=zq, masks=masks, waves=waves, wave_lengths=wave_lengths, commitment_loss=commitment_loss,
               codebook_loss=codebook_loss, timbre=timbre, preds=preds, n=n, tgt=tgt, content_loss=content_loss)
    return out, aux


def main():
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count())
    model, sds = build_reference_in_train_mode()
    report = {}
    out, aux = iteration(model, DEFAULT_CFG)
    wav_seg, z, zq, masks, waves, wave_lengths = (aux[k] for k in ("wav_seg", "z", "zq", "masks", "waves", "wave_lengths"))
    commitment_loss, codebook_loss, timbre, preds, n, tgt, content_loss = (aux[k] for k in (
        "commitment_loss", "codebook_loss", "timbre", "preds", "n", "tgt", "content_loss"))
    from losses import FocalLoss, reconstruction_loss

    # -------------------------------------------------------------- oracle pinning of the train-mode restatement
    with torch.no_grad():
        oz = O.encoder_forward(sds["encoder"], wav_seg)
    report["train_encoder_oracle_rel"] = MG.rel_err(oz, z.detach())
    leaves = {n: v.clone().requires_grad_() for n, v in sds["quantizer"].items() if v.dtype.is_floating_point}
    omask = dict(p=masks["p"], c=masks["c"], r=masks["r"], res=torch.tensor(RES_MASK, dtype=torch.float32))
    zl = z.detach().clone().requires_grad_()
    o_outs, o_q, o_cm, o_cb, o_t, _ = O.quantizer_forward_train(leaves, zl, wav_seg, omask, side_branches_no_grad=False,
                                                                full_waves=waves, wave_lens=wave_lengths)
    report["train_quantizer_outs_oracle_rel"] = MG.rel_err(o_outs.detach(), zq.detach())
    report["train_commitment_oracle_rel"] = abs(float(o_cm) - float(commitment_loss)) / abs(float(commitment_loss))
    report["train_codebook_oracle_rel"] = abs(float(o_cb) - float(codebook_loss)) / abs(float(codebook_loss))
    report["train_timbre_oracle_rel"] = MG.rel_err(o_t.detach(), timbre.detach())
    report["focal_oracle_rel"] = abs(float(O.focal_loss(preds["content"].detach().transpose(1, 2)[..., :n], tgt.long())) - float(content_loss)) / float(content_loss)

    np.savez_compressed(os.path.join(HERE, "train_step.npz"), **out)

    # ------------------------------------------------------------------ fvq.py / rvq.py (dead code in the reference, API parity row a11)
    from quantize.fvq import FactorizedVectorQuantize
    from quantize.rvq import ResidualVQ
    fv = {}
    gq = torch.Generator().manual_seed(5)
    vq = FactorizedVectorQuantize(dim=64, codebook_size=1024, codebook_dim=8, commitment=0.15)
    sd_f = synth.load_synthetic(vq, seed=4, prefix="fvq.")
    zin = torch.randn(3, 64, 50, generator=gq)
    for mode in ("eval", "train"):
        vq.train(mode == "train")
        with torch.no_grad():
            zq_f, idx_f, loss_f = vq(zin)
            o_zq, o_idx, o_loss = O.fvq_forward(zin, sd_f, "", training=(mode == "train"))
        report[f"fvq_{mode}_oracle_rel"] = MG.rel_err(o_zq, zq_f)
        report[f"fvq_{mode}_oracle_code_mismatch"] = int((o_idx != idx_f).sum())
        report[f"fvq_{mode}_oracle_loss_abs"] = float((o_loss - loss_f).abs().max())
        fv.update({f"{mode}_zq": zq_f.numpy(), f"{mode}_idx": idx_f.numpy().astype(np.int16), f"{mode}_loss": loss_f.numpy()})
    rv = ResidualVQ(num_quantizers=3, codebook_size=10, dim=64, codebook_dim=8, commitment=0.15).eval()
    synth.load_synthetic(rv, seed=6, prefix="rvq.")
    with torch.no_grad():
        q_out, all_idx, all_loss, all_q = rv(zin)
    fv.update(z=zin.numpy(), rvq_out=q_out.numpy(), rvq_idx=all_idx.numpy().astype(np.int16), rvq_losses=all_loss.numpy(),
              rvq_quantized_probe=all_q[:, :, ::4, ::5].numpy())
    np.savez_compressed(os.path.join(HERE, "fvq.npz"), **fv)

    # ------------------------------------------------------------------ losses.py:65-89, :264-276; meldataset.py:42-47
    misc = {}
    with torch.no_grad():
        xa, xb = synth.synth_clips(2, 8000, seed=31).squeeze(1), synth.synth_clips(2, 8000, seed=32).squeeze(1) * 0.7
        rl = reconstruction_loss(xa, xb)
        report["reconstruction_loss_oracle_rel"] = abs(float(O.reconstruction_loss(xa, xb)) - float(rl)) / float(rl)
        lg = torch.randn(5, 33, 17, generator=gq) * 3
        lb = torch.randint(0, 33, (5, 17), generator=gq)
        fl = {g_: float(FocalLoss(gamma=g_)(lg, lb)) for g_ in (0, 2)}
        misc.update(recon_x=xa.numpy(), recon_gx=xb.numpy(), recon_loss=np.float64(rl), focal_logits=lg.numpy(), focal_labels=lb.numpy(),
                    focal_gamma0=np.float64(fl[0]), focal_gamma2=np.float64(fl[2]))
        try:
            sys.modules.setdefault("soundfile", type(sys)("soundfile"))
            sys.modules.setdefault("librosa", type(sys)("librosa"))
            import meldataset as ref_md
            mw = synth.synth_clips(1, 6000, seed=33).reshape(-1)
            mm = ref_md.preprocess(mw.numpy())
            report["meldataset_oracle_rel"] = MG.rel_err(O.meldataset_preprocess(mw), mm)
            misc.update(meldataset_wave=mw.numpy(), meldataset_mel_probe=mm[0, ::4, :].numpy(), meldataset_shape=np.array(mm.shape))
        except Exception as e:  # the module drags in dataset-side packages; say so instead of hiding it
            report["meldataset_import_error"] = repr(e)
    np.savez_compressed(os.path.join(HERE, "recon_misc.npz"), **misc)

    json.dump(report, open(os.path.join(HERE, "oracle_pinning_report_train.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps(report, indent=1, sort_keys=True))
    print({k: float(v) for k, v in out.items() if isinstance(v, np.floating)})


if __name__ == "__main__":
    main()
