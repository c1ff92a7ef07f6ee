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
