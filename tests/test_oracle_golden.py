"""Pins the CPU oracle against vectors produced by the REAL reference (tests/golden/make_golden.py
imports /root/reference in the build container).  Tolerances: fp32 noise floor of the reference
itself is ~1e-6 relative; codes must be identical."""
import json
import os

import numpy as np
import pytest
import torch

from facodec_amd import synth
from oracle import facodec_oracle as O


def rel(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def _shapes(golden_dir, key):
    ref = json.load(open(os.path.join(golden_dir, "state_shapes.json")))
    return {k: v for k, v in ref[key].items()}


def test_pinning_report_is_green(golden_dir):
    r = json.load(open(os.path.join(golden_dir, "oracle_pinning_report.json")))
    assert r["e2e_codes_oracle_mismatch"] == 0 and r["e2e_codes_oracle_pipeline_mismatch"] == 0
    assert r["vq_kat_oracle_mismatch"] == 0 and r["vq_sweep_oracle_mismatch"] == 0
    for k in ("e2e_encoder_oracle_rel", "e2e_decoder_oracle_rel", "small_encoder_oracle_rel", "small_decoder_oracle_rel"):
        assert r[k] < 1e-5, k


def test_small_encoder_decoder_against_reference(golden_dir):
    d = np.load(os.path.join(golden_dir, "small_layers.npz"))
    from facodec_amd.dac_model import Encoder, Decoder  # only for parameter shapes (CPU, no forward)
    enc = Encoder(d_model=8, strides=[2, 5, 5, 6], d_latent=64, causal=True, lstm=2)
    dec = Decoder(input_channel=64, channels=128, rates=[6, 5, 5, 2], causal=True, lstm=2)
    sd_e = synth.synth_state_dict(synth.param_shapes(enc), 1, "encoder.")
    sd_d = synth.synth_state_dict(synth.param_shapes(dec), 1, "decoder.")
    x = torch.from_numpy(d["x"])
    assert rel(O.encoder_forward(sd_e, x), d["z"]) < 1e-5
    assert rel(O.decoder_forward(sd_d, torch.from_numpy(d["z"])), d["y"]) < 2e-5


def test_vq_known_answers(golden_dir):
    d = np.load(os.path.join(golden_dir, "vq_kat.npz"))
    cb, lat = torch.from_numpy(d["codebook"]), torch.from_numpy(d["latents"])
    _, idx = O.vq_nearest(lat, cb)
    assert torch.equal(idx, torch.from_numpy(d["indices"].astype(np.int64)))
    assert int(idx[0, 0]) == 3          # duplicate rows 3 / 700 -> lowest index
    assert int(idx[0, 1]) in (511, 512)  # same direction: normalised tie, reference's answer recorded
    assert int(idx[1, 5]) == int(d["zero_latent_index"])


def test_vq_sweep_262k(golden_dir):
    d = np.load(os.path.join(golden_dir, "vq_kat.npz"))
    g = np.random.Generator(np.random.Philox(key=int(d["sweep_key"])))
    g.standard_normal((1024, 8)); g.standard_normal((4, 8, 300))
    big = torch.from_numpy(g.standard_normal((1, 8, 1 << 18)).astype(np.float32))
    _, idx = O.vq_nearest(big, torch.from_numpy(d["codebook"]))
    assert torch.equal(idx.reshape(-1), torch.from_numpy(d["sweep_indices"].astype(np.int64)))


@pytest.mark.timeout(600)
def test_end_to_end_real_config(golden_dir):
    """configs[0]: 2 s clips on CPU, encoder -> FVQ -> decoder; codes bit-exact vs the reference."""
    d = np.load(os.path.join(golden_dir, "codec_e2e.npz"))
    sds = {k: synth.synth_state_dict({n: s for n, s in _shapes(golden_dir, k).items()}, 0, k + ".")
           for k in ("encoder", "quantizer", "decoder")}
    wave = synth.synth_clips(2, 48000, seed=0)
    with torch.no_grad():
        r = O.codec_forward(sds, wave, n_c=2)
    for nm, c in zip(("codes_p", "codes_c", "codes_r"), r["codes"]):
        assert torch.equal(c, torch.from_numpy(d[nm].astype(np.int64))), nm
    assert rel(r["z"][:, ::8], d["z_probe"]) < 1e-5
    assert rel(r["timbre"], d["timbre"]) < 1e-5
    assert rel(r["outs"][:, ::8], d["outs_probe"]) < 1e-5
    assert rel(r["wave"][:, 0, torch.from_numpy(d["probe_t"])], d["wave_probe"]) < 2e-5
    assert abs(float(r["commitment"]) - float(d["commitment"])) / float(d["commitment"]) < 1e-5
    # losses (third-party STFT/mel semantics restated: parity UNPINNED; self-consistency only)
    y = r["wave"]
    assert abs(float(O.mel_spectrogram_loss(y, wave)) - float(d["loss_mel"])) / float(d["loss_mel"]) < 1e-4
    assert abs(float(O.multiscale_stft_loss(y, wave)) - float(d["loss_stft"])) / float(d["loss_stft"]) < 1e-4
    assert abs(float(O.waveform_l1_loss(y, wave)) - float(d["loss_l1"])) / float(d["loss_l1"]) < 1e-4


def test_predictor_filter_buffer_equals_reference_formula():
    """alias_free_torch/filter.py:27-58 restated in dsp.kaiser_sinc_filter1d: beta ~ 4.663, taps sum to 1."""
    from facodec_amd import dsp
    f = dsp.kaiser_sinc_filter1d(0.25, 0.3, 12)
    assert f.shape == (1, 1, 12) and abs(float(f.sum()) - 1.0) < 1e-6
    assert torch.allclose(f.flatten(), f.flatten().flip(0), atol=1e-7)      # symmetric even-length filter


def test_filterbanks_against_transformers():
    """Cross-check of the restated third-party filterbanks."""
    from transformers.audio_utils import mel_filter_bank
    fb = O.mel_filterbank_htk(1025, 80, 24000)
    ref = mel_filter_bank(1025, 80, 0.0, 12000.0, 24000, norm=None, mel_scale="htk")
    assert np.abs(fb.numpy() - ref).max() < 2e-5
    sl = O.mel_filterbank_slaney(24000, 2048, 80)
    ref2 = mel_filter_bank(1025, 80, 0.0, 12000.0, 24000, norm="slaney", mel_scale="slaney")
    assert np.abs(sl.numpy().T - ref2).max() < 1e-6


def test_stft_mel_pipelines_against_transformers_spectrogram():
    """Whole-pipeline cross-check of the parity-UNPINNED rows (a15-a18, the log-mel front-end) that does not route through the
    oracle on the checking side: `transformers.audio_utils.spectrogram` (numpy framing + rfft, periodic Hann,
    center / reflect) with its own `mel_filter_bank` against (i) O.logmel_frontend (torchaudio MelSpectrogram semantics: power
    2, HTK mel, 1 200-sample window centred in a 2 048-point FFT), (ii) one scale of MelSpectrogramLoss (audiotools / librosa
    semantics: magnitude, Slaney mel with Slaney norm, window = FFT length, hop = window / 4), (iii) one scale of
    MultiScaleSTFTLoss.  Still not the third-party packages themselves (absent here): an independent implementation of the
    same published semantics agreeing to fp32 rounding."""
    from transformers.audio_utils import mel_filter_bank, spectrogram, window_function
    wave = synth.synth_clips(2, 12000, seed=17)
    est = (0.7 * wave + 0.05 * synth.synth_clips(2, 12000, seed=18)).contiguous()
    # (i) log-mel front-end
    win = window_function(1200, "hann", periodic=True, frame_length=2048, center=True)
    fb_htk = mel_filter_bank(1025, 80, 0.0, 12000.0, 24000, norm=None, mel_scale="htk")
    got = O.logmel_frontend(wave, 80)
    for b in range(2):
        mel = spectrogram(wave[b, 0].numpy().astype(np.float64), win, 2048, 300, fft_length=2048, power=2.0, center=True, pad_mode="reflect",
                          mel_filters=fb_htk, mel_floor=0.0, dtype=np.float64)
        ref = (np.log(1e-5 + mel) + 4.0) / 4.0
        assert np.abs(got[b].numpy() - ref[:, :got.shape[-1]]).max() < 2e-4 * np.abs(ref).max()
    # (ii) / (iii) one loss scale each, computed entirely from transformers' spectrogram
    w = 512
    win_w = window_function(w, "hann", periodic=True)
    fb_sl = mel_filter_bank(w // 2 + 1, 80, 0.0, 12000.0, 24000, norm="slaney", mel_scale="slaney")

    def mags(x, **kw):
        return np.stack([spectrogram(x[b, 0].numpy().astype(np.float64), win_w, w, w // 4, fft_length=w, power=1.0, center=True,
                                     pad_mode="reflect", dtype=np.float64, **kw) for b in range(x.shape[0])])

    xm, ym = mags(est, mel_filters=fb_sl, mel_floor=0.0), mags(wave, mel_filters=fb_sl, mel_floor=0.0)
    mel_ref = np.abs(np.log10(np.maximum(xm, 1e-5)) - np.log10(np.maximum(ym, 1e-5))).mean()
    mel_got = float(O.mel_spectrogram_loss(est, wave, n_mels=(80,), window_lengths=(w,)))
    assert abs(mel_got - mel_ref) / mel_ref < 1e-4, (mel_got, mel_ref)
    xs, ys = mags(est), mags(wave)
    stft_ref = np.abs(np.log10(np.maximum(xs, 1e-5) ** 2) - np.log10(np.maximum(ys, 1e-5) ** 2)).mean() + np.abs(xs - ys).mean()
    stft_got = float(O.multiscale_stft_loss(est, wave, window_lengths=(w,)))
    assert abs(stft_got - stft_ref) / stft_ref < 1e-4, (stft_got, stft_ref)


def test_product_dsp_tables_equal_oracle_tables():
    from facodec_amd import dsp
    assert np.array_equal(dsp.mel_fbank_htk(1025, 80, 24000), O.mel_filterbank_htk(1025, 80, 24000).numpy())
    assert np.array_equal(dsp.mel_fbank_slaney(24000, 512, 80), O.mel_filterbank_slaney(24000, 512, 80).numpy())
    assert np.array_equal(dsp.hann_periodic(1200), O.hann_periodic(1200).numpy())


def test_dft_basis_matches_torch_stft():
    from facodec_amd import dsp
    w = synth.synth_clips(1, 6000, seed=4)[0]
    basis, off = dsp.dft_basis(2048, 1200)
    ref = O.stft_complex(w, 2048, 300, 1200)[0]            # (1025, frames)
    xp = torch.nn.functional.pad(w.unsqueeze(0), (1024, 1024), mode="reflect")[0, 0]
    frames = xp.unfold(0, 2048, 300)[:, off:off + 1200].double()  # (frames, 1200)
    spec = torch.from_numpy(basis).double() @ frames.t()
    assert rel(spec[:1025], ref.real) < 1e-5 and rel(spec[1025:], ref.imag) < 1e-5


def test_reflect_short_input_matches_pad1d():
    """Inputs shorter than the pad: reference zero-extends before reflecting (encodec.py:104-111)."""
    x = torch.arange(1.0, 6.0).reshape(1, 1, 5)
    w = torch.zeros(1, 1, 7); w[0, 0, 0] = 1.0   # picks xpad[t] i.e. x[t-6] reflected
    y = O.sconv1d(x, w, None)
    assert y.shape[-1] == 5
    assert torch.allclose(y[0, 0], torch.tensor([0.0, 0.0, 5.0, 4.0, 3.0]))


def test_pinning_report_covers_redecoder_and_discriminator(golden_dir):
    r = json.load(open(os.path.join(golden_dir, "oracle_pinning_report.json")))
    assert r["redecoder_oracle_rel"] < 1e-5 and r["redecoder_decoder_oracle_rel"] < 1e-5
    assert r["discriminator_oracle_rel_max"] < 1e-5


def test_discriminator_oracle_against_reference_logits(golden_dir):
    """5 MPD + 3 MRD logits of the REAL dac/model/discriminator.py (over the audiotools STFT restatement) on two 1 s
    clips with formula weights: the oracle reproduces them from the committed fixture alone."""
    shapes = _shapes(golden_dir, "discriminator")
    sd = synth.synth_state_dict(shapes, 0, "discriminator.")
    x = synth.synth_clips(2, 24000, seed=5)
    gold = np.load(os.path.join(golden_dir, "discriminator.npz"))
    with torch.no_grad():
        fm = O.discriminator_forward(sd, x)
    assert [len(f) for f in fm] == [6] * 5 + [26] * 3
    for i, f in enumerate(fm):
        assert rel(f[-1], gold[f"logit{i}"]) < 1e-5, i
        ma = np.array([float(t.abs().mean()) for t in f], np.float32)
        assert np.allclose(ma, gold[f"fmap{i}_mean_abs"], rtol=1e-4), i


def test_quantizer_dropout_masks_follow_reference_rule():
    """dac/nn/quantize.py:163-183: the first int(B * dropout) samples draw n in [1, n_codebooks], the rest use all."""
    from facodec_amd.autograd import draw_quantizer_masks
    g = torch.Generator().manual_seed(5)
    m = draw_quantizer_masks(3, 8, 0.5, g)
    assert m.shape == (3, 8) and torch.all(m[:, 4:] == 1)            # undropped half: every quantizer
    assert torch.all(m[0] == 1)                                       # at least one quantizer for everyone
    assert torch.all(m[1:] <= m[:-1])                                 # masks are prefixes: i < n_quantizers
    g2 = torch.Generator().manual_seed(5)
    drop = torch.randint(1, 4, (8,), generator=g2)
    assert torch.equal(m[:, :4].sum(0).long(), drop[:4])


def test_gan_and_train_losses_restated_like_train_py():
    """train.py:282-285, 304-312 on toy feature maps."""
    fake = [[torch.full((2, 3), 0.5), torch.full((2, 1), 0.25)]]
    real = [[torch.full((2, 3), 1.5), torch.full((2, 1), 0.75)]]
    ld, lg, lf = O.gan_losses(fake, real)
    assert abs(float(ld) - (0.25 ** 2 + 0.25 ** 2)) < 1e-7 and abs(float(lg) - 0.75 ** 2) < 1e-7 and abs(float(lf) - 1.0) < 1e-7


# ------------------------------------------------------------------------------------------------ training iteration
def test_train_pinning_report_is_green(golden_dir):
    """tests/golden/make_golden_train.py: the oracle's train-mode restatements against the real reference in .train()."""
    r = json.load(open(os.path.join(golden_dir, "oracle_pinning_report_train.json")))
    assert r["fvq_eval_oracle_code_mismatch"] == 0 and r["fvq_train_oracle_code_mismatch"] == 0
    for k in ("train_quantizer_outs_oracle_rel", "train_commitment_oracle_rel", "train_codebook_oracle_rel", "train_timbre_oracle_rel",
              "train_encoder_oracle_rel", "focal_oracle_rel", "fvq_eval_oracle_rel", "fvq_train_oracle_rel",
              "reconstruction_loss_oracle_rel", "meldataset_oracle_rel"):
        assert r[k] < 1e-5, (k, r[k])


def test_fvq_and_rvq_against_reference(golden_dir):
    """quantize/fvq.py:36-116 and quantize/rvq.py:32-73 (row a11): outputs of the real classes, eval and train mode."""
    d = np.load(os.path.join(golden_dir, "fvq.npz"))
    from facodec_amd.fvq import FactorizedVectorQuantize, ResidualVQ   # parameter shapes only (CPU, no forward)
    vq = FactorizedVectorQuantize(dim=64, codebook_size=1024, codebook_dim=8, commitment=0.15)
    sd = synth.synth_state_dict(synth.param_shapes(vq), 4, "fvq.")
    z = torch.from_numpy(d["z"])
    for mode in ("eval", "train"):
        zq, idx, loss = O.fvq_forward(z, sd, "", training=(mode == "train"))
        assert torch.equal(idx, torch.from_numpy(d[f"{mode}_idx"].astype(np.int64)))
        assert rel(zq, d[f"{mode}_zq"]) < 1e-5
        assert float((loss - torch.from_numpy(d[f"{mode}_loss"])).abs().max()) < 1e-6
    rv = ResidualVQ(num_quantizers=3, codebook_size=10, dim=64, codebook_dim=8, commitment=0.15)
    sdr = synth.synth_state_dict(synth.param_shapes(rv), 6, "rvq.")
    residual, total = z, 0.0
    for i in range(3):
        q, idx, _ = O.fvq_forward(residual, sdr, f"layers.{i}.")
        assert torch.equal(idx, torch.from_numpy(d["rvq_idx"][i].astype(np.int64)))
        assert rel(q[:, ::4, ::5], d["rvq_quantized_probe"][i]) < 1e-5
        residual, total = residual - q, total + q
    assert rel(total, d["rvq_out"]) < 1e-5


def _fvq_upstream(shape, seed):
    """The fixed upstream-gradient weights of tests/golden/make_golden_fvq_train.py."""
    n = int(np.prod(shape))
    k = torch.arange(n, dtype=torch.float64)
    return torch.sin(0.37 * k + seed).reshape(tuple(shape)).float() / float(np.sqrt(n))


def test_fvq_and_rvq_train_gradients_against_reference(golden_dir):
    """Row a11 in TRAIN mode (quantize/fvq.py:66-78 detach placements + straight-through; quantize/rvq.py:36-64 quantizer dropout,
    'linear' and 'exp'): the oracle's values and torch-autograd gradients against those of the real classes (fvq_train.npz)."""
    d = np.load(os.path.join(golden_dir, "fvq_train.npz"))
    from facodec_amd.fvq import FactorizedVectorQuantize, ResidualVQ   # parameter shapes only (CPU, no forward)
    vq = FactorizedVectorQuantize(dim=64, codebook_size=1024, codebook_dim=8, commitment=0.15)
    sd = {k: v.requires_grad_() for k, v in synth.synth_state_dict(synth.param_shapes(vq), 4, "fvq.").items()}
    z = torch.from_numpy(d["fvq_z"]).requires_grad_()
    zq, idx, loss = O.fvq_forward(z, sd, "", training=True)
    assert torch.equal(idx, torch.from_numpy(d["fvq_idx"].astype(np.int64)))
    assert rel(zq.detach(), d["fvq_zq"]) < 1e-5 and float((loss.detach() - torch.from_numpy(d["fvq_loss"])).abs().max()) < 1e-6
    ((zq * _fvq_upstream(zq.shape, 1)).sum() + (loss * _fvq_upstream(loss.shape, 2)).sum()).backward()
    assert rel(z.grad, d["fvq_dz"]) < 1e-5
    for n, v in sd.items():
        assert rel(v.grad, d["fvq_grad." + n]) < 1e-5, n
    for kind, nq in (("linear", 3), ("exp", 4)):
        rv = ResidualVQ(num_quantizers=nq, codebook_size=10, dim=64, codebook_dim=8, commitment=0.15, quantizer_dropout=0.75, dropout_type=kind)
        sdr = {k: v.requires_grad_() for k, v in synth.synth_state_dict(synth.param_shapes(rv), 6, "rvq.").items()}
        p = f"rvq_{kind}_"
        x = torch.from_numpy(d[p + "x"]).requires_grad_()
        draw = torch.from_numpy(d[p + "draw"])
        draw = torch.pow(2, draw) if kind == "exp" else draw                    # rvq.py:44
        q_out, all_idx, all_loss, all_q = O.residual_vq_forward(x, sdr, "", nq, training=True, dropout=draw, quantizer_dropout=0.75)
        assert torch.equal(all_idx, torch.from_numpy(d[p + "idx"].astype(np.int64)))
        assert rel(q_out.detach(), d[p + "out"]) < 1e-5 and rel(all_q.detach()[:, :, ::4, ::5], d[p + "quantized_probe"]) < 1e-5
        assert float((all_loss.detach() - torch.from_numpy(d[p + "losses"])).abs().max()) < 1e-6
        ((q_out * _fvq_upstream(q_out.shape, 3)).sum() + (all_loss * _fvq_upstream(all_loss.shape, 4)).sum()
         + (all_q * _fvq_upstream(all_q.shape, 5)).sum()).backward()
        assert rel(x.grad, d[p + "dx"]) < 1e-5
        for n, v in sdr.items():
            assert rel(v.grad, d[p + "grad." + n]) < 2e-5, (kind, n)


def test_focal_recon_meldataset_against_reference(golden_dir):
    """losses.py:264-276 FocalLoss (pure torch: pinned), losses.py:65-89 and meldataset.py:42-47 (over the torchaudio
    shim: composition pinned, STFT unpinned)."""
    d = np.load(os.path.join(golden_dir, "recon_misc.npz"))
    lg, lb = torch.from_numpy(d["focal_logits"]), torch.from_numpy(d["focal_labels"])
    assert abs(float(O.focal_loss(lg, lb, 2.0)) - float(d["focal_gamma2"])) / float(d["focal_gamma2"]) < 1e-6
    assert abs(float(O.focal_loss(lg, lb, 0.0)) - float(d["focal_gamma0"])) / float(d["focal_gamma0"]) < 1e-6
    rl = O.reconstruction_loss(torch.from_numpy(d["recon_x"]), torch.from_numpy(d["recon_gx"]))
    assert abs(float(rl) - float(d["recon_loss"])) / float(d["recon_loss"]) < 1e-5
    mm = O.meldataset_preprocess(torch.from_numpy(d["meldataset_wave"]))
    assert tuple(mm.shape) == tuple(d["meldataset_shape"]) and rel(mm[0, ::4, :], d["meldataset_mel_probe"]) < 1e-5


def test_train_iteration_oracle_against_reference(golden_dir):
    """The whole iteration of train.py:265-374 replayed through the oracle + torch autograd against the real reference's
    .train() run (tests/golden/train_step.npz): 17 loss scalars, 5 gradient norms, gradient probes of ~60 tensors,
    the set of parameters that receive no gradient, and the discriminator's weights after its AdamW step."""
    from facodec_amd.commons import build_model, default_model_params
    import train_replay as TR
    fx = TR.load_fixture(golden_dir)
    model = build_model(default_model_params())            # parameter / buffer holders on the CPU (no forward here)
    for k in TR.KEYS:
        synth.load_synthetic(model[k], seed=0, prefix=k + ".")
    sds, names = TR.product_state(model)
    sc, grads, norms, d_after = TR.oracle_iteration(O, sds, names, fx)
    for k in TR.SCALARS:
        assert abs(sc[k] - float(fx[k])) / abs(float(fx[k])) < 2e-5, (k, sc[k], float(fx[k]))
    for k in TR.KEYS:
        # fp64 norm over the key; the reference's own clip_grad_norm_ value (fp32 accumulation over ~1e8 elements) is kept
        # in the fixture too and sits up to 4.5e-4 away from it
        assert abs(norms[k] - float(fx[f"grad_norm64_{k}"])) / float(fx[f"grad_norm64_{k}"]) < 1e-5, k
        assert abs(norms[k] - float(fx[f"grad_norm_{k}"])) / float(fx[f"grad_norm_{k}"]) < 1e-3, k
        worst = TR.compare_grads(fx, k, grads[k], 1e-4, 1e-4)
        assert worst[1] < 1e-4 and worst[2] < 1e-4, worst
    for k, missing in fx["no_grad"].items():
        assert sorted(n for n in names[k] if n not in grads[k]) == sorted(missing), k
    for key in fx:
        if key.startswith("param_after.discriminator."):
            n = key[len("param_after.discriminator."):-len(".probe")]
            flat = d_after[n].reshape(-1)
            assert np.abs(flat[TR.probe_index(flat.numel())].numpy() - fx[key]).max() < 1e-6, n


def test_gradient_probe_conditioning_justifies_the_gpu_bars(golden_dir):
    """Why tests/test_train_golden.py holds gradient PROBES (single entries) to 3e-3 / 5e-4 while losses sit at 1e-5: the
    loss of train.py:282-358 is kinked everywhere (L1 of log-mel magnitudes, L1 feature matching, LeakyReLU, code
    assignment), so a change of the input at rounding level moves individual gradient entries orders of magnitude more than
    it moves the losses or the key norms.  Experiment: scale the waveforms by (1 + 1e-7) -- one or two
    ulps -- and replay the whole iteration through the oracle.  Recorded in gpurun_out/gradient_conditioning.json; the
    asserts pin the regime the GPU bars rely on."""
    import json
    import os
    from facodec_amd.commons import build_model, default_model_params
    import train_replay as TR
    fx = TR.load_fixture(golden_dir)
    model = build_model(default_model_params())
    for k in TR.KEYS:
        synth.load_synthetic(model[k], seed=0, prefix=k + ".")
    sds, names = TR.product_state(model)
    sc0, g0, n0, _ = TR.oracle_iteration(O, sds, names, fx)
    fx2 = dict(fx)
    t = dict(fx["t"])
    t["wav_seg"] = t["wav_seg"] * (1.0 + 1e-7)
    t["waves"] = t["waves"] * (1.0 + 1e-7)
    fx2["t"] = t
    sc1, g1, n1, _ = TR.oracle_iteration(O, sds, names, fx2)
    loss_move = {k: abs(sc1[k] - sc0[k]) / abs(sc0[k]) for k in TR.SCALARS}
    norm_move = {k: abs(n1[k] - n0[k]) / n0[k] for k in TR.KEYS}
    probe_move, worst = {}, {}
    for k in TR.KEYS:
        w = ("", 0.0)
        for n in TR.grad_probe_names(fx, k):
            a, b = g0[k][n].reshape(-1), g1[k][n].reshape(-1)
            idx = TR.probe_index(a.numel())
            e = float((a[idx] - b[idx]).abs().max() / max(float(a[idx].abs().max()), 1e-30))
            if e > w[1]:
                w = (n, e)
        probe_move[k], worst[k] = w[1], w[0]
    report = dict(perturbation="waveforms * (1 + 1e-7)", loss_rel_move=loss_move, key_norm_rel_move=norm_move,
                  worst_probe_rel_move=probe_move, worst_probe_tensor=worst)
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    json.dump(report, open(os.path.join(out_dir, "gradient_conditioning.json"), "w"), indent=1)
    assert max(loss_move.values()) < 1e-6, loss_move                       # the losses barely notice ...
    assert max(norm_move.values()) < 1e-4, norm_move                       # ... nor do the key norms (GPU bar: 2e-4)
    # ... while single gradient entries of the discriminator move at the 1e-4 .. 1e-3 level (GPU bar 3e-3); everything the
    # generator and predictor heads own stays an order of magnitude below (GPU bar 5e-4)
    assert probe_move["discriminator"] < 3e-3, (worst["discriminator"], probe_move["discriminator"])
    for k in ("encoder", "quantizer", "decoder", "fa_predictors"):
        assert probe_move[k] < 2.5e-4, (k, worst[k], probe_move[k])
    assert probe_move["discriminator"] > 20 * max(loss_move.values())      # the conditioning gap the comment claims
