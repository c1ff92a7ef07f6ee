"""Generator half of the training step (reference: train.py:265-272, 295-299, 357-374) on the HIP path:

    z = encoder(wave);  outs, _, commitment, codebook, _ = quantizer(z, wave);  wave_hat = decoder(outs)
    loss = 15 * mel_loss(wave_hat, wave) + 0.25 * commitment + 1.0 * codebook
    backward;  per model key: average gradients across ranks (one all-reduce), clip at 1000, AdamW, ExponentialLR

Not in this round: the discriminator terms (feature matching + adversarial) and the predictor heads'
losses (they need external phoneme / F0 / speaker targets).  Every parameter of encoder, quantizer and decoder that
the remaining loss reaches is trained."""
import torch

from . import losses, optim


class GeneratorStep:
    def __init__(self, model, lr=1e-4, sample_rate=24000):
        self.model = model
        for k in ("encoder", "quantizer", "decoder"):
            model[k].train()
        self.mel = losses.MelSpectrogramLoss(n_mels=[5, 10, 20, 40, 80, 160, 320], window_lengths=[32, 64, 128, 256, 512, 1024, 2048],
                                             mel_fmin=[0] * 7, mel_fmax=[None] * 7, pow=1.0, mag_weight=0.0, clamp_eps=1e-5,
                                             sample_rate=sample_rate)          # train.py:155-163
        self.opt = {"encoder": optim.FlatAdamW(model.encoder.parameters(), lr=lr),
                    "quantizer": optim.FlatAdamW(model.quantizer.parameters(), lr=lr),
                    "decoder": optim.FlatAdamW(model.decoder.parameters(), lr=lr)}

    def forward_backward(self, wave, masks=None):
        m = self.model
        z = m.encoder(wave)
        outs, _, commitment, codebook, _ = m.quantizer(z, wave, n_c=2, masks=masks)
        wave_hat = m.decoder(outs)
        mel = self.mel(wave_hat, wave)
        loss = 15.0 * mel + 0.25 * commitment + 1.0 * codebook          # train.py:357-358 without the GAN / predictor terms
        loss.backward()
        return dict(loss=loss.detach(), mel=mel.detach(), commitment=commitment.detach(), codebook=codebook.detach())

    def __call__(self, wave, masks=None):
        out = self.forward_backward(wave, masks)
        for k in ("encoder", "decoder", "quantizer"):                    # train.py:362-374
            self.opt[k].step()
        out["grad_norm"] = {k: self.opt[k].grad_norm() for k in self.opt}
        return out


class TrainStep(GeneratorStep):
    """Both halves of train.py's iteration (:265-374).  The predictor-head losses (:314-356) are included when
    `with_predictors=True` and the caller supplies their targets (the reference obtains them from a pitch extractor, a
    CTC phoneme model and a speaker model, none of which is part of this build):

        pred = decoder(quantizer(encoder(wave)))
        discriminator:  loss_d = sum_k mean(D_k(pred.detach())^2) + mean((1 - D_k(wave))^2);  clip 10;  AdamW
        generator:      15 mel + feature matching + adversarial + 0.25 commitment + codebook;  clip 1000;  AdamW x3
    """

    def __init__(self, model, lr=1e-4, sample_rate=24000, with_predictors=False):
        super().__init__(model, lr, sample_rate)
        model.discriminator.train()
        self.opt["discriminator"] = optim.FlatAdamW(model.discriminator.parameters(), lr=lr, max_norm=10.0)
        self.with_predictors = with_predictors
        if with_predictors:
            model.fa_predictors.train()
            self.opt["fa_predictors"] = optim.FlatAdamW(model.fa_predictors.parameters(), lr=lr)

    def predictor_losses(self, quantized, timbre, targets):
        """train.py:314-356 given the targets the reference takes from external models: f0 (B, F) normalised log-F0
        (-10 = unvoiced), uv (B, F) the log-normalised mel energy `real_norm`, phones (B, F) int64 in [0, 1024),
        speaker (B,) int64 in [0, 20000).  Returns 1*f0 + 1*uv + 5*content + 1*speaker (:357-358)."""
        from . import autograd_disc as AD
        preds, rev = self.model.fa_predictors(quantized, timbre)
        n = min(preds["f0"].shape[-2], targets["f0"].shape[-1])
        f0_t, uv_t = targets["f0"][..., :n].contiguous(), targets["uv"][..., :n].contiguous()
        ph_t = targets["phones"][..., :n].contiguous().reshape(-1)

        def sl1(p, t):        # F.smooth_l1_loss(target, pred.squeeze(-1)[..., :n])
            return AD.PairMean.apply(p.squeeze(-1)[..., :n].contiguous(), t, 3)

        def ce(logits_btc):   # criterion(pred.transpose(1, 2)[..., :n], target): rows (b, t), classes contiguous
            lg = logits_btc[:, :n].contiguous()
            return AD.CrossEntropy.apply(lg.reshape(-1, lg.shape[-1]), ph_t)

        tot_f0 = sl1(preds["f0"], f0_t) + (sl1(rev["rev_f0"], f0_t) if rev["rev_f0"] is not None else 0.0)
        tot_uv = sl1(preds["uv"], uv_t) + (sl1(rev["rev_uv"], uv_t) if rev["rev_uv"] is not None else 0.0)
        tot_content = ce(preds["content"]) + (ce(rev["rev_content"]) if rev["rev_content"] is not None else 0.0)
        tot_spk = AD.CrossEntropy.apply(preds["timbre"], targets["speaker"])
        if rev["x_timbre"] is not None:
            tot_spk = tot_spk + AD.CrossEntropy.apply(rev["x_timbre"], targets["speaker"])
        return 1.0 * tot_f0 + 1.0 * tot_uv + 5.0 * tot_content + 1.0 * tot_spk

    def __call__(self, wave, masks=None, targets=None):
        from .discriminator import gan_losses
        m = self.model
        z = m.encoder(wave)
        outs, quantized, commitment, codebook, timbre = m.quantizer(z, wave, n_c=2, masks=masks)
        pred = m.decoder(outs)
        # ---- discriminator step (train.py:279-292)
        d_fake, d_real = m.discriminator(pred.detach()), m.discriminator(wave)
        loss_d, _, _ = gan_losses(d_fake, d_real)
        loss_d.backward()
        self.opt["discriminator"].step()
        # ---- generator step (:295-374): the discriminator is only differentiated w.r.t. its input
        for p in self.opt["discriminator"].params:
            p.requires_grad_(False)
        try:
            d_fake = m.discriminator(pred)
            with torch.no_grad():
                d_real = m.discriminator(wave)
            _, loss_g, loss_feat = gan_losses(d_fake, d_real)
            mel = self.mel(pred, wave)
            loss = 15.0 * mel + 1.0 * loss_feat + 1.0 * loss_g + 0.25 * commitment + 1.0 * codebook
            if self.with_predictors:
                if targets is None:
                    raise ValueError("with_predictors=True needs targets (f0, uv, phones, speaker)")
                loss = loss + self.predictor_losses(quantized, timbre, targets)
            loss.backward()
        finally:
            for p in self.opt["discriminator"].params:
                p.requires_grad_(True)
        for k in ("encoder", "decoder", "quantizer") + (("fa_predictors",) if self.with_predictors else ()):
            self.opt[k].step()
        return dict(loss=loss.detach(), loss_d=loss_d.detach(), loss_g=loss_g.detach(), feature=loss_feat.detach(),
                    mel=mel.detach(), commitment=commitment.detach(), codebook=codebook.detach(),
                    grad_norm={k: self.opt[k].grad_norm() for k in self.opt})
