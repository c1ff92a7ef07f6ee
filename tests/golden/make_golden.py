"""Generates the golden vectors under tests/golden/ by importing the REAL reference from
/root/reference (build container only -- the reference never travels to the GPU box).

Run:  python tests/golden/make_golden.py            (about 2 minutes on 8 cores)

The reference needs three third-party packages that are absent from this image; the stand-ins
below are import shims written for this script only:
  * munch.Munch                -- a dict with attribute access;
  * audiotools                 -- AudioSignal / STFTParams / ml.BaseModel with the upstream
                                  STFT + mel semantics; the arithmetic is
                                  torch.stft + a librosa-style Slaney filterbank from the oracle;
  * torchaudio                 -- transforms.MelSpectrogram + functional.create_dct restated from
                                  upstream semantics (HTK mel, power 2).
Everything that flows through the audiotools / torchaudio shims is therefore "parity unpinned"
(the shim and the oracle share the restated third-party semantics); the encoder, decoder, LSTM,
VQ, WaveNet, StyleEncoder, LayerNorm arithmetic below comes from the reference's own code.

Outputs (all small):
  state_shapes.json     parameter names + shapes of the reference's encoder/quantizer/decoder
  small_layers.npz      reduced-config encoder/decoder + per-layer probes (full tensors)
  vq_kat.npz            VQ known-answer tests (ties, zero latent, random sweep digest)
  codec_e2e.npz         real config, 2 clips of 2 s: codes, timbre, latent/wave probes, losses
  mel_frontend.npz      log-mel front-end + loss scalars (parity-unpinned rows)
"""
import hashlib
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)

from oracle import facodec_oracle as O  # noqa: E402
from facodec_amd import synth  # noqa: E402


# ------------------------------------------------------------------------------ import shims
def install_shims():
    munch = types.ModuleType("munch")

    class Munch(dict):
        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError as e:
                raise AttributeError(k) from e

        def __setattr__(self, k, v):
            self[k] = v

    munch.Munch = Munch
    sys.modules["munch"] = munch

    at = types.ModuleType("audiotools")
    ml = types.ModuleType("audiotools.ml")

    class BaseModel(torch.nn.Module):
        INTERN, EXTERN = [], []

    ml.BaseModel = BaseModel
    ml.Accelerator = object
    at.ml = ml

    class STFTParams:
        def __init__(self, window_length=None, hop_length=None, window_type=None, match_stride=None,
                     padding_type=None):
            self.window_length, self.hop_length, self.window_type = window_length, hop_length, window_type
            self.match_stride, self.padding_type = match_stride, padding_type

    class AudioSignal:
        """Only what dac/nn/loss.py touches: stft(), .magnitude, mel_spectrogram(), audio_data."""

        def __init__(self, audio, sample_rate, stft_params=None):
            self.audio_data = audio
            self.sample_rate = sample_rate
            self.stft_params = stft_params
            self.stft_data = None

        def stft(self, window_length=None, hop_length=None, window_type=None, **kw):
            b, c, t = self.audio_data.shape
            p = self.stft_params
            if window_length is None and p is not None and p.match_stride:     # MRD front-end (discriminator.py:150-151)
                s = O.stft_match_stride(self.audio_data.reshape(-1, t), p.window_length)
                self.stft_data = s.reshape(b, c, s.shape[-2], s.shape[-1])
                return self.stft_data
            s = O.stft_complex(self.audio_data.reshape(-1, t), window_length, hop_length)
            self.stft_data = s.reshape(b, c, s.shape[-2], s.shape[-1])
            return self.stft_data

        @property
        def magnitude(self):
            return self.stft_data.abs()

        def mel_spectrogram(self, n_mels, mel_fmin=0.0, mel_fmax=None, **kw):
            mag = self.stft(**kw).abs()
            nf = mag.shape[2]
            fb = O.mel_filterbank_slaney(self.sample_rate, 2 * (nf - 1), n_mels, mel_fmin, mel_fmax)
            return (mag.transpose(2, -1) @ fb.T).transpose(-1, 2)

        def __sub__(self, other):
            return AudioSignal(self.audio_data - other.audio_data, self.sample_rate)

    at.AudioSignal, at.STFTParams = AudioSignal, STFTParams
    sys.modules["audiotools"], sys.modules["audiotools.ml"] = at, ml
    ab = types.ModuleType("argbind")
    ab.bind = lambda *a, **k: (lambda f: f)
    sys.modules["argbind"] = ab

    ta = types.ModuleType("torchaudio")
    tt = types.ModuleType("torchaudio.transforms")
    tf = types.ModuleType("torchaudio.functional")

    class MelSpectrogram(torch.nn.Module):
        def __init__(self, sample_rate=16000, n_fft=400, win_length=None, hop_length=None, n_mels=128, **kw):
            super().__init__()
            self.sr, self.n_fft, self.n_mels = sample_rate, n_fft, n_mels
            self.win = win_length or n_fft
            self.hop = hop_length or self.win // 2

        def forward(self, w):
            p = O.stft_complex(w, self.n_fft, self.hop, self.win).abs().pow(2)
            fb = O.mel_filterbank_htk(self.n_fft // 2 + 1, self.n_mels, self.sr)
            return torch.matmul(p.transpose(-1, -2), fb).transpose(-1, -2)

    def create_dct(n_mfcc, n_mels, norm):
        n = torch.arange(float(n_mels))
        k = torch.arange(float(n_mfcc)).unsqueeze(1)
        dct = torch.cos(np.pi / float(n_mels) * (n + 0.5) * k)
        if norm is None:
            dct *= 2.0
        else:
            dct[0] *= 1.0 / np.sqrt(2.0)
            dct *= np.sqrt(2.0 / float(n_mels))
        return dct.t()

    tt.MelSpectrogram, tf.create_dct = MelSpectrogram, create_dct
    ta.transforms, ta.functional = tt, tf
    sys.modules["torchaudio"], sys.modules["torchaudio.transforms"], sys.modules["torchaudio.functional"] = ta, tt, tf


def ref_imports():
    install_shims()
    sys.path.insert(0, REF)
    from modules.commons import build_model, recursive_munch  # noqa
    return build_model, recursive_munch


def model_params(encoder_dim=64, decoder_dim=1536):
    return dict(fixed=True, causal=True, lstm=2, norm_f0=True, use_gr_content_f0=False,
                use_gr_prosody_phone=False, use_gr_timbre_prosody=False, separate_prosody_encoder=True,
                n_c_codebooks=2, timbre_norm=True, use_gr_content_global_f0=True, w2v="w2v-ctc",
                DAC=dict(encoder_dim=encoder_dim, encoder_rates=[2, 5, 5, 6], decoder_dim=decoder_dim,
                         decoder_rates=[6, 5, 5, 2], sr=24000))


def rel_err(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def sha(t):
    return hashlib.sha256(np.ascontiguousarray(t.numpy()).tobytes()).hexdigest()


def main():
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count())
    build_model, recursive_munch = ref_imports()
    from dac.model.dac import Encoder, Decoder
    report = {}

    # --------------------------------------------------------------- 1. state-dict contract
    with torch.no_grad():
        model = build_model(recursive_munch(model_params()))
    shapes = {k: {n: list(v.shape) for n, v in model[k].state_dict().items()}
              for k in ("encoder", "quantizer", "decoder", "fa_predictors")}
    json.dump(shapes, open(os.path.join(HERE, "state_shapes.json"), "w"), indent=0, sort_keys=True)

    # --------------------------------------------------------------- 2. reduced config, layer probes
    with torch.no_grad():
        enc_s = Encoder(d_model=8, strides=[2, 5, 5, 6], d_latent=64, causal=True, lstm=2).eval()
        dec_s = Decoder(input_channel=64, channels=128, rates=[6, 5, 5, 2], causal=True, lstm=2).eval()
        sd_e = synth.load_synthetic(enc_s, seed=1, prefix="encoder.")
        sd_d = synth.load_synthetic(dec_s, seed=1, prefix="decoder.")
        x = synth.synth_clips(2, 2400, seed=3)
        probes = {}
        hooks = []
        for name, mod in list(enc_s.named_modules()) + [("dec." + n, m) for n, m in dec_s.named_modules()]:
            if name.count(".") <= 1 and name:
                hooks.append(mod.register_forward_hook(lambda m, i, o, name=name: probes.__setitem__(name, o.detach().clone())))
        z = enc_s(x)
        y = dec_s(z)
        for h in hooks:
            h.remove()
        zo = O.encoder_forward(sd_e, x)
        yo = O.decoder_forward(sd_d, z)
        report["small_encoder_oracle_rel"] = rel_err(zo, z)
        report["small_decoder_oracle_rel"] = rel_err(yo, y)
        np.savez_compressed(os.path.join(HERE, "small_layers.npz"), x=x.numpy(), z=z.numpy(), y=y.numpy(),
                            **{"probe." + k: v.numpy() for k, v in probes.items()})

    # --------------------------------------------------------------- 3. VQ known-answer tests
    from dac.nn.quantize import VectorQuantize
    with torch.no_grad():
        vq = VectorQuantize(64, 1024, 8).eval()
        g = np.random.Generator(np.random.Philox(key=1234))
        cb = torch.from_numpy(g.standard_normal((1024, 8)).astype(np.float32))
        cb[700] = cb[3]          # exact duplicate rows: the lowest index must win
        cb[512] = cb[511] * 2.0  # same direction, different norm: normalised rows tie
        vq.codebook.weight.copy_(cb)
        lat = torch.from_numpy(g.standard_normal((4, 8, 300)).astype(np.float32))
        lat[0, :, 0] = cb[3]
        lat[0, :, 1] = cb[511] * 0.5
        lat[1, :, 5] = 0.0       # zero latent: norm clamps at 1e-12, e = 0
        _, idx = vq.decode_latents(lat)
        _, idx_o = O.vq_nearest(lat, cb)
        report["vq_kat_oracle_mismatch"] = int((idx != idx_o).sum())
        big = torch.from_numpy(g.standard_normal((1, 8, 1 << 18)).astype(np.float32))
        _, idx_big = vq.decode_latents(big)
        _, idx_big_o = O.vq_nearest(big, cb)
        report["vq_sweep_oracle_mismatch"] = int((idx_big != idx_big_o).sum())
        np.savez_compressed(os.path.join(HERE, "vq_kat.npz"), codebook=cb.numpy(), latents=lat.numpy(),
                            indices=idx.numpy().astype(np.int16), sweep_key=np.int64(1234),
                            sweep_indices=idx_big.numpy().astype(np.int16).reshape(-1),
                            zero_latent_index=np.int64(idx[1, 5]))

    # --------------------------------------------------------------- 4. real config end to end
    with torch.no_grad():
        for k in ("encoder", "quantizer", "decoder"):
            model[k].eval()
        sds = {k: synth.load_synthetic(model[k], seed=0, prefix=k + ".") for k in ("encoder", "quantizer", "decoder")}
        wave = synth.synth_clips(2, 48000, seed=0)
        z = model.encoder(wave)
        outs, quantized, commit, cbl, timbre, codes = model.quantizer(z, wave, n_c=2, return_codes=True)
        y = model.decoder(outs)
        oz = O.encoder_forward(sds["encoder"], wave)
        report["e2e_encoder_oracle_rel"] = rel_err(oz, z)
        o_outs, o_q, o_cm, o_cb, o_t, o_codes = O.quantizer_forward(sds["quantizer"], z, wave, n_c=2)
        report["e2e_quantizer_oracle_rel"] = rel_err(o_outs, outs)
        report["e2e_timbre_oracle_rel"] = rel_err(o_t, timbre)
        report["e2e_codes_oracle_mismatch"] = int(sum((a != b).sum() for a, b in zip(o_codes, codes)))
        report["e2e_commit_oracle_rel"] = abs(float(o_cm) - float(commit)) / abs(float(commit))
        oy = O.decoder_forward(sds["decoder"], outs)
        report["e2e_decoder_oracle_rel"] = rel_err(oy, y)
        # quantizer run on the ORACLE's latent (what a from-scratch pipeline sees)
        _, _, _, _, _, o_codes2 = O.quantizer_forward(sds["quantizer"], oz, wave, n_c=2)
        report["e2e_codes_oracle_pipeline_mismatch"] = int(sum((a != b).sum() for a, b in zip(o_codes2, codes)))

        probe_t_early = np.arange(0, 48000, 47)
        # discriminator (train.py:280-312): the real dac/model/discriminator.py over the audiotools shim
        from dac.model.discriminator import Discriminator
        disc = Discriminator(rates=[], periods=[2, 3, 5, 7, 11], fft_sizes=[2048, 1024, 512], sample_rate=24000).eval()
        sd_d = synth.load_synthetic(disc, seed=0, prefix="discriminator.")
        xd = synth.synth_clips(2, 24000, seed=5)
        fm_ref = disc(xd)
        fm_or = O.discriminator_forward(sd_d, xd)
        report["discriminator_oracle_rel_max"] = max(rel_err(a, b) for fr, fo in zip(fm_ref, fm_or) for a, b in zip(fo, fr))
        shapes["discriminator"] = {n: list(v.shape) for n, v in disc.state_dict().items()}
        json.dump(shapes, open(os.path.join(HERE, "state_shapes.json"), "w"), indent=0, sort_keys=True)
        np.savez_compressed(os.path.join(HERE, "discriminator.npz"),
                            **{f"logit{i}": fr[-1].numpy() for i, fr in enumerate(fm_ref)},
                            **{f"fmap{i}_mean_abs": np.array([float(f.abs().mean()) for f in fr], np.float32) for i, fr in enumerate(fm_ref)})

        # predictor heads (train.py:270), eval mode, on the quantizer's outputs
        model.fa_predictors.eval()
        sd_p = synth.load_synthetic(model.fa_predictors, seed=0, prefix="fa_predictors.")
        preds, rev_preds = model.fa_predictors(quantized, timbre)
        sd_p_full = {k: v for k, v in model.fa_predictors.state_dict().items()}
        o_preds, o_rev = O.predictors_forward(sd_p_full, quantized, timbre)
        report["predictors_oracle_rel"] = max(rel_err(o_preds[k], preds[k]) for k in preds)
        report["rev_predictors_oracle_rel"] = max(rel_err(o_rev[k], rev_preds[k]) for k in rev_preds)
        np.savez_compressed(
            os.path.join(HERE, "predictors.npz"), f0=preds["f0"].numpy(), uv=preds["uv"].numpy(),
            content_probe=preds["content"][:, ::4, ::16].numpy(), timbre_probe=preds["timbre"][:, ::50].numpy(),
            rev_f0=rev_preds["rev_f0"].numpy(), rev_uv=rev_preds["rev_uv"].numpy(),
            rev_content_probe=rev_preds["rev_content"][:, ::4, ::16].numpy(),
            x_timbre_probe=rev_preds["x_timbre"][:, ::50].numpy())

        # voice-conversion path (reconstruct_redecoder.py:110-122): stage 'redecoder', non-causal decoder
        rparams = dict(encoder_causal=True, decoder_causal=False, encoder_lstm=2, decoder_lstm=0, n_c_codebooks=2,
                       n_p_codebooks=1, timbre_norm=True, separate_prosody_encoder=True, encoder_type="wavenet",
                       wavenet_embed_dim=512, mamba_embed_dim=768, prob_random_mask_prosody=1.0,
                       prob_random_mask_content=[0.0, 1.0],
                       DAC=dict(encoder_dim=64, encoder_rates=[2, 5, 5, 6], decoder_dim=1536, decoder_rates=[6, 5, 5, 2], sr=24000))
        rmodel = build_model(recursive_munch(rparams), stage="redecoder")
        for k in ("encoder", "decoder"):
            rmodel[k].eval()
        sd_re = synth.load_synthetic(rmodel.encoder, seed=0, prefix="redecoder.encoder.")
        sd_rd = synth.load_synthetic(rmodel.decoder, seed=0, prefix="redecoder.decoder.")
        timbre_tgt = timbre.flip(0)                      # "target speaker" = the other clip's timbre
        zr = rmodel.encoder(codes[0], codes[1], timbre_tgt, use_p_code=False, n_c=1)
        yr = rmodel.decoder(zr)
        o_zr = O.redecoder_forward(sd_re, codes[0], codes[1], timbre_tgt, use_p_code=False, n_c=1)
        o_yr = O.decoder_forward(sd_rd, zr, causal=False, lstm=0)
        report["redecoder_oracle_rel"] = rel_err(o_zr, zr)
        report["redecoder_decoder_oracle_rel"] = rel_err(o_yr, yr)
        shapes["redecoder.encoder"] = {n: list(v.shape) for n, v in rmodel.encoder.state_dict().items()}
        shapes["redecoder.decoder"] = {n: list(v.shape) for n, v in rmodel.decoder.state_dict().items()}
        json.dump(shapes, open(os.path.join(HERE, "state_shapes.json"), "w"), indent=0, sort_keys=True)
        np.savez_compressed(os.path.join(HERE, "redecoder.npz"), z_probe=zr[:, ::8, :].numpy(),
                            wave_probe=yr[:, 0, probe_t_early].numpy(), probe_t=probe_t_early,
                            wave_absmax=np.float32(yr.abs().max()))

        # losses through the reference's own dac/nn/loss.py (over the audiotools shim)
        from dac.nn.loss import MelSpectrogramLoss, MultiScaleSTFTLoss, L1Loss
        from audiotools import AudioSignal
        mel_c = MelSpectrogramLoss(n_mels=[5, 10, 20, 40, 80, 160, 320], window_lengths=[32, 64, 128, 256, 512, 1024, 2048],
                                   mel_fmin=[0] * 7, mel_fmax=[None] * 7, pow=1.0, mag_weight=0.0, clamp_eps=1e-5)
        sig, rec = AudioSignal(wave, 24000), AudioSignal(y, 24000)
        mel_l = float(mel_c(rec, sig))
        stft_l = float(MultiScaleSTFTLoss()(rec, sig))
        l1_l = float(L1Loss()(rec, sig))
        report["loss_mel_oracle_rel"] = abs(float(O.mel_spectrogram_loss(y, wave)) - mel_l) / mel_l
        report["loss_stft_oracle_rel"] = abs(float(O.multiscale_stft_loss(y, wave)) - stft_l) / stft_l
        mel80 = model.quantizer.preprocess(wave, n_bins=80)
        report["logmel_oracle_rel"] = rel_err(O.logmel_frontend(wave, 80), mel80)

        probe_t = np.arange(0, 48000, 47)
        np.savez_compressed(
            os.path.join(HERE, "codec_e2e.npz"),
            codes_p=codes[0].numpy().astype(np.int16), codes_c=codes[1].numpy().astype(np.int16),
            codes_r=codes[2].numpy().astype(np.int16), timbre=timbre.numpy(),
            z_probe=z[:, ::8, :].numpy(), outs_probe=outs[:, ::8, :].numpy(),
            wave_probe=y[:, 0, probe_t].numpy(), probe_t=probe_t, wave_sha256=np.array(sha(y)),
            commitment=np.float32(commit), codebook=np.float32(cbl),
            zq_p_probe=quantized[0][:, ::16, :].numpy(), zq_c_probe=quantized[1][:, ::16, :].numpy(),
            zq_r_probe=quantized[2][:, ::16, :].numpy(),
            loss_mel=np.float32(mel_l), loss_stft=np.float32(stft_l), loss_l1=np.float32(l1_l),
            wave_absmax=np.float32(y.abs().max()), z_absmax=np.float32(z.abs().max()))
        np.savez_compressed(os.path.join(HERE, "mel_frontend.npz"), mel80_probe=mel80[:, ::4, ::4].numpy(),
                            mel80_mean=np.float32(mel80.mean()))

    json.dump(report, open(os.path.join(HERE, "oracle_pinning_report.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps(report, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
