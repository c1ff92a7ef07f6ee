"""Spectral reconstruction losses of the training step, forward values on the HIP C ABI.

Reference call sites: train.py:154-164,295-299 (`MelSpectrogramLoss`, `MultiScaleSTFTLoss`, `L1Loss`
from dac/nn/loss.py, built on descript-audiotools' AudioSignal.stft / mel_spectrogram) and
losses.py:65-89 (`reconstruction_loss`, torchaudio MelSpectrogram; named by the north star, not
called by train.py).  The STFT / mel semantics of those third-party packages are restated in dsp.py
(PARITY UNPINNED: the reference vendors neither package nor holds tests for them).

Launch plan per scale: estimate and reference are stacked into one batch of 2B signals ->
`fac_stft_frames` (centre / reflect framing) -> windowed real-DFT as ONE GEMM on the MFMA conv kernel
(`fac_conv1d_fwd`, K = window) -> `fac_spec_power` (|.| or |.|^2) -> mel filterbank GEMM ->
`fac_reduce_pair` / `fac_logdiff_rms` (deterministic two-stage reductions accumulating into one scalar).
Backward (d loss / d estimate) arrives with the training path.
"""
import math
import os

import torch
from torch import nn

from . import dsp, ops


def _audio(x):
    """Accepts a (B, 1, T) / (B, T) tensor or an object exposing `.audio_data` (audiotools.AudioSignal)."""
    x = getattr(x, "audio_data", x)
    return x.reshape(x.shape[0], x.shape[-1])


MEL_STREAMS = int(os.environ.get("FAC_MEL_STREAMS", "3"))      # concurrent scales of MelSpectrogramLoss (1 = serial)


class _SpectralScale:
    """Constants of one STFT scale, packed once per device: windowed DFT basis and (optional) mel bank."""

    def __init__(self, device, n_fft, win, hop, fbank=None):
        basis, off = dsp.dft_basis(n_fft, win)
        self.n_fft, self.win, self.hop, self.off = n_fft, win, hop, off
        self.F = n_fft // 2 + 1
        basis_t = torch.from_numpy(basis).to(device).unsqueeze(-1)
        self.basis = ops.pack_conv_weight(basis_t)
        self.basis_bwd = ops.pack_conv_weight_bwd(basis_t)       # transposed GEMM of the backward pass
        # the same two matrices as split-bf16 planes for the GEMM kernel (conv1d_gemm_split.hip; fp32-grade): the windowed-DFT
        # GEMMs of the long windows are 1x1 convs with 256 .. 2048 "channels" and ran at 23 TFLOP/s on the fp32 tile
        self.basis_t = basis_t
        self._basis_split = self._basis_split_t = None
        self.n_mels = None
        if fbank is not None:            # (n_mels, F)
            fb = torch.as_tensor(fbank, dtype=torch.float32, device=device).contiguous()
            self.n_mels = fb.shape[0]
            self.fb = ops.pack_conv_weight(fb.unsqueeze(-1))
            self.fb_bwd = ops.pack_conv_weight_bwd(fb.unsqueeze(-1))

    def dft(self, frames, n_frames):
        """frames (N, win, n_frames) -> (N, 2F, n_frames) [Re | Im]: the windowed-DFT GEMM."""
        two_f = 2 * self.F
        with ops.flop_key("dft"):
            if ops.gemm_split_ok(two_f, self.win, 1, frames.shape[0] * n_frames):
                if self._basis_split is None:
                    self._basis_split = ops.pack_gemm_weight_split(self.basis_t)
                return ops.conv1d(frames, None, two_f, 1, pad_left=0, pad_mode=ops.PAD_ZERO, t_out=n_frames, w_split=self._basis_split)
            return ops.conv1d(frames, self.basis, two_f, 1, pad_left=0, pad_mode=ops.PAD_ZERO, t_out=n_frames)

    def dft_adjoint(self, dspec, n_frames):
        """dspec (N, 2F, n_frames) -> gradient of the frames (N, win, n_frames): the transposed GEMM."""
        with ops.flop_key("dft"):
            if ops.gemm_split_ok(self.win, 2 * self.F, 1, dspec.shape[0] * n_frames):
                if self._basis_split_t is None:
                    self._basis_split_t = ops.pack_gemm_weight_split_t(self.basis_t)
                return ops.conv1d(dspec, None, self.win, 1, pad_left=0, pad_mode=ops.PAD_ZERO, t_out=n_frames, w_split=self._basis_split_t)
            return ops.conv1d(dspec, self.basis_bwd, self.win, 1, pad_left=0, pad_mode=ops.PAD_ZERO, t_out=n_frames)

    def spectrum(self, waves, power):
        """waves (N, T) -> (N, F, frames) magnitude (power 1) or power (2) spectrogram."""
        N, T = waves.shape
        frames_n = 1 + T // self.hop
        fr = ops.stft_frames(waves, self.win, frames_n, self.hop, self.n_fft // 2, self.off)
        spec = self.dft(fr, frames_n)
        return ops.spec_power(spec, power)

    def mel(self, spec):
        with ops.flop_key("dft"):
            return ops.conv1d(spec, self.fb, self.n_mels, 1, pad_left=0, pad_mode=ops.PAD_ZERO, t_out=spec.shape[-1])

    def backward_to_wave(self, waves, terms, use_mel):
        """d/d waves of sum_i scale_i * pair_term(mode_i)(S(waves), target) for S = mel or magnitude spectrogram.
        terms: [(mode, eps, scale)], target already computed.  Recomputes the forward of `waves` (cheap, saves
        keeping every scale's spectra alive between forward and backward)."""
        N, T = waves.shape
        frames_n = 1 + T // self.hop
        fr = ops.stft_frames(waves, self.win, frames_n, self.hop, self.n_fft // 2, self.off)
        spec = self.dft(fr, frames_n)
        mag = ops.spec_power(spec, 1)
        feat = self.mel(mag) if use_mel else mag
        target = self._target
        dfeat = torch.empty_like(feat)
        for i, (mode, eps, scale) in enumerate(terms):
            ops.pair_bwd(feat, target, dfeat, mode, eps, scale, accumulate=i > 0)
        with ops.flop_key("dft"):
            if use_mel:
                dmag = ops.conv1d(dfeat, self.fb_bwd, self.F, 1, pad_left=0, pad_mode=ops.PAD_ZERO, t_out=frames_n)
            else:
                dmag = dfeat
            dspec = ops.spec_power_bwd(spec, dmag, 1)
        dfr = self.dft_adjoint(dspec, frames_n)
        return ops.stft_frames_bwd(dfr, T, self.hop, self.n_fft // 2, self.off)


class _SpectralLossFn(torch.autograd.Function):
    """value = module._value(x, y); backward = d value / d x through the HIP adjoint kernels."""

    @staticmethod
    def forward(ctx, x, y, module):
        ctx.module = module
        ctx.save_for_backward(x, y)
        return module._value(x.detach(), y.detach()).clone()

    @staticmethod
    def backward(ctx, g):
        x, y = ctx.saved_tensors
        dx = ctx.module._grad_x(x.detach(), y.detach())
        return (dx * g).reshape(x.shape), None, None


class _LossBase(nn.Module):
    def __init__(self):
        super().__init__()
        self._scales = {}
        self._scratch = None

    def _bufs(self, device):
        if self._scratch is None or self._scratch.device != device:
            self._scratch = torch.empty(1024, device=device, dtype=torch.float32)
        return torch.zeros(1, device=device, dtype=torch.float32), self._scratch


class MelSpectrogramLoss(_LossBase):
    """dac/nn/loss.py:231-327.  Per scale: L1 of log10(clamp(mel, eps)^pow) (x log_weight) and L1 of mel
    (x mag_weight); mel = |STFT| @ librosa-style Slaney filterbank (window = periodic Hann of the scale)."""

    def __init__(self, n_mels=(150, 80), window_lengths=(2048, 512), loss_fn=None, clamp_eps=1e-5, mag_weight=1.0,
                 log_weight=1.0, pow=2.0, weight=1.0, match_stride=False, mel_fmin=(0.0, 0.0), mel_fmax=(None, None),
                 window_type=None, sample_rate=24000):
        super().__init__()
        if match_stride:
            raise NotImplementedError("match_stride STFT (discriminator front-end) is not built yet")
        self.n_mels, self.window_lengths = list(n_mels), list(window_lengths)
        self.clamp_eps, self.mag_weight, self.log_weight, self.pow, self.weight = clamp_eps, mag_weight, log_weight, pow, weight
        self.mel_fmin, self.mel_fmax, self.sample_rate = list(mel_fmin), list(mel_fmax), sample_rate

    def forward(self, x, y):
        xa = getattr(x, "audio_data", x)
        if torch.is_tensor(xa) and xa.requires_grad:
            return _SpectralLossFn.apply(xa, getattr(y, "audio_data", y), self)
        return self._value(x, y)

    def _scale(self, device, sr, nm, fmin, fmax, w):
        key = (device, w, nm, fmin, fmax, sr)
        if key not in self._scales:
            self._scales[key] = _SpectralScale(device, w, w, w // 4, dsp.mel_fbank_slaney(sr, w, nm, fmin, fmax))
        return self._scales[key]

    def _scale_list(self):
        return list(zip(self.n_mels, self.mel_fmin, self.mel_fmax, self.window_lengths))

    def _grad_x(self, x, y):
        """d value / d x.  The scales are independent pipelines of small launches (framing -> DFT GEMM -> magnitude -> mel GEMM ->
        pair term and back): they run side by side (ops.run_chains inside this one autograd node: kernel concurrency only, the
        engine never sees the streams) and their gradients are added in scale order, as the serial loop did: same bits."""
        xs, ys = _audio(x).contiguous(), _audio(y).contiguous()
        sr = self.sample_rate

        def one(nm, fmin, fmax, w):
            sc = self._scale(xs.device, sr, nm, fmin, fmax, w)
            sc._target = sc.mel(sc.spectrum(ys, 1))
            n = sc._target.numel()
            terms = [(1, self.clamp_eps, self.log_weight * self.pow / n)]
            if self.mag_weight != 0.0:
                terms.append((0, 0.0, self.mag_weight / n))
            return sc.backward_to_wave(xs, terms, use_mel=True)

        parts = ops.run_chains([lambda a=a: one(*a) for a in self._scale_list()], xs.device, MEL_STREAMS)
        dx = parts[0]
        for d in parts[1:]:
            dx = ops.add(dx, d)
        return dx

    def _value(self, x, y):
        xs, ys = _audio(x), _audio(y)
        B = xs.shape[0]
        sr = getattr(x, "sample_rate", self.sample_rate)
        both = torch.cat([xs, ys], 0).contiguous()
        scales = self._scale_list()
        dev = both.device
        if self._scratch is None or self._scratch.device != dev or self._scratch.shape[0] != len(scales):
            self._scratch = torch.empty(len(scales), 1024, device=dev, dtype=torch.float32)
        outs = torch.zeros(len(scales), device=dev, dtype=torch.float32)         # one accumulator per scale (chains run concurrently)

        def one(i, nm, fmin, fmax, w):
            sc = self._scale(dev, sr, nm, fmin, fmax, w)
            mel = sc.mel(sc.spectrum(both, 1))                       # (2B, n_mels, frames)
            n = mel[:B].numel()
            out, scratch = outs[i:i + 1], self._scratch[i]
            # log10(clamp(m, eps)^pow) = pow * log10(max(m, eps))
            ops.reduce_pair(mel[:B], mel[B:], out, scratch, 1, self.clamp_eps, self.log_weight * self.pow / n, True)
            if self.mag_weight != 0.0:
                ops.reduce_pair(mel[:B], mel[B:], out, scratch, 0, 0.0, self.mag_weight / n, True)
            return None

        ops.run_chains([lambda i=i, a=a: one(i, *a) for i, a in enumerate(scales)], dev, MEL_STREAMS)
        total = outs[0]
        for i in range(1, len(scales)):          # the order the single accumulator of the serial loop summed in
            total = total + outs[i]
        return total


class MultiScaleSTFTLoss(_LossBase):
    """dac/nn/loss.py:142-228: per window L1 of log10(clamp(|S|, eps)^pow) + L1 of |S|."""

    def __init__(self, window_lengths=(2048, 512), loss_fn=None, clamp_eps=1e-5, mag_weight=1.0, log_weight=1.0,
                 pow=2.0, weight=1.0, match_stride=False, window_type=None):
        super().__init__()
        if match_stride:
            raise NotImplementedError("match_stride STFT (discriminator front-end) is not built yet")
        self.window_lengths = list(window_lengths)
        self.clamp_eps, self.mag_weight, self.log_weight, self.pow, self.weight = clamp_eps, mag_weight, log_weight, pow, weight

    def forward(self, x, y):
        xs, ys = _audio(x), _audio(y)
        B = xs.shape[0]
        both = torch.cat([xs, ys], 0).contiguous()
        out, scratch = self._bufs(both.device)
        for w in self.window_lengths:
            key = (both.device, w)
            if key not in self._scales:
                self._scales[key] = _SpectralScale(both.device, w, w, w // 4)
            mag = self._scales[key].spectrum(both, 1)
            n = mag[:B].numel()
            ops.reduce_pair(mag[:B], mag[B:], out, scratch, 1, self.clamp_eps, self.log_weight * self.pow / n, True)
            ops.reduce_pair(mag[:B], mag[B:], out, scratch, 0, 0.0, self.mag_weight / n, True)
        return out[0]


class L1Loss(_LossBase):
    """dac/nn/loss.py:11-48 on the waveforms."""

    def __init__(self, attribute="audio_data", weight=1.0, **kwargs):
        super().__init__()
        self.weight = weight

    def forward(self, x, y):
        xs, ys = _audio(x).contiguous(), _audio(y).contiguous()
        out, scratch = self._bufs(xs.device)
        ops.reduce_pair(xs, ys, out, scratch, 0, 0.0, 1.0 / xs.numel(), False)
        return out[0]


_RECON_CACHE = {}


def _recon_scale(dev, s):
    n_fft = max(s, 512)
    key = (dev, s)
    if key not in _RECON_CACHE:
        fb = dsp.mel_fbank_htk(n_fft // 2 + 1, 64, 16000).T.copy()          # (64, F)
        _RECON_CACHE[key] = _SpectralScale(dev, n_fft, s, s // 4, fb)
    return _RECON_CACHE[key]


def _reconstruction_value(xs, gs, eps):
    B = xs.shape[0]
    dev = xs.device
    out = torch.zeros(1, device=dev, dtype=torch.float32)
    scratch = torch.empty(1024, device=dev, dtype=torch.float32)
    ops.reduce_pair(xs, gs, out, scratch, 2, 0.0, 100.0 / xs.numel(), False)
    both = torch.cat([xs, gs], 0).contiguous()
    for i in range(6, 12):
        s = 2 ** i
        sc = _recon_scale(dev, s)
        mel = sc.mel(sc.spectrum(both, 2))                                       # (2B, 64, frames)
        n = mel[:B].numel()
        ops.reduce_pair(mel[:B], mel[B:], out, scratch, 0, 0.0, 1.0 / n, True)
        ops.logdiff_rms(mel[:B], mel[B:], out, scratch, eps, math.sqrt(s / 2) / (B * mel.shape[-1]), True)
    return out[0]


def _reconstruction_grad(xs, gs, eps):
    """d reconstruction_loss / d G_x: MSE term + per scale the adjoint chain pair / log-RMS derivative -> transposed mel GEMM
    -> |.|^2 adjoint -> transposed windowed-DFT GEMM -> framing adjoint (the kernels of the mel-loss backward)."""
    B, T = gs.shape
    dev = gs.device
    dg = torch.empty_like(gs)
    ops.pair_bwd(gs, xs, dg, 2, 0.0, 100.0 / xs.numel(), accumulate=False)      # d/dG 100 * mean (x - G)^2
    for i in range(6, 12):
        s = 2 ** i
        sc = _recon_scale(dev, s)
        frames_n = 1 + T // sc.hop
        mel_x = sc.mel(sc.spectrum(xs, 2))
        fr = ops.stft_frames(gs, sc.win, frames_n, sc.hop, sc.n_fft // 2, sc.off)
        spec = sc.dft(fr, frames_n)
        mel_g = sc.mel(ops.spec_power(spec, 2))
        dmel = torch.empty_like(mel_g)
        ops.pair_bwd(mel_g, mel_x, dmel, 0, 0.0, 1.0 / mel_g.numel(), accumulate=False)
        ops.logdiff_rms_bwd(mel_x, mel_g, dmel, eps, math.sqrt(s / 2) / (B * mel_g.shape[-1]), accumulate=True)
        with ops.flop_key("dft"):
            dpow = ops.conv1d(dmel, sc.fb_bwd, sc.F, 1, pad_left=0, pad_mode=ops.PAD_ZERO, t_out=frames_n)
            dspec = ops.spec_power_bwd(spec, dpow, 2)
        dfr = sc.dft_adjoint(dspec, frames_n)
        dg = ops.add(dg, ops.stft_frames_bwd(dfr, T, sc.hop, sc.n_fft // 2, sc.off))
    return dg


class _ReconstructionLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, G_x, eps):
        xs, gs = _audio(x).contiguous(), _audio(G_x).contiguous()
        ctx.save_for_backward(xs, gs)
        ctx.eps, ctx.shape = eps, G_x.shape
        return _reconstruction_value(xs.detach(), gs.detach(), eps).clone()

    @staticmethod
    def backward(ctx, g):
        xs, gs = ctx.saved_tensors
        if ctx.needs_input_grad[0]:
            raise NotImplementedError("reconstruction_loss is differentiated w.r.t. the estimate G_x only")
        return None, (_reconstruction_grad(xs.detach(), gs.detach(), ctx.eps) * g).reshape(ctx.shape), None


def reconstruction_loss(x, G_x, eps=1e-7):
    """losses.py:65-89: 100*MSE + sum_{s=64..2048} [ L1(mel) + sqrt(s/2) * mean_t RMS_mel(log diff) ] with
    torchaudio MelSpectrogram(sample_rate=16000, n_fft=max(s,512), win_length=s, hop=s//4, n_mels=64).  Differentiable
    w.r.t. the estimate G_x (forward and backward on the HIP kernels)."""
    if torch.is_grad_enabled() and torch.is_tensor(G_x) and G_x.requires_grad:
        return _ReconstructionLossFn.apply(x, G_x, eps)
    return _reconstruction_value(_audio(x).contiguous(), _audio(G_x).contiguous(), eps)
