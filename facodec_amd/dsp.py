"""Host-side construction (numpy, float64 -> float32) of the constant operands of the spectral
kernels: windows, mel filterbanks and the windowed real-DFT bases that turn an STFT into a GEMM
for the MFMA conv kernel.  These are computed once per module and uploaded; no signal arithmetic
happens here.

Third-party semantics being matched (not vendored in the reference):
  * torchaudio.transforms.MelSpectrogram -> periodic Hann, HTK mel, norm=None, power 2
    (modules/quantize.py:228-230, meldataset.py:37-38, losses.py:75-83);
  * audiotools AudioSignal.stft / mel_spectrogram -> scipy periodic Hann, magnitude,
    librosa Slaney-scale + Slaney-norm filterbank (dac/nn/loss.py:221-227,319-320).
"""
import math

import numpy as np


def hann_periodic(n):
    k = np.arange(n, dtype=np.float64)
    return (0.5 - 0.5 * np.cos(2.0 * math.pi * k / n)).astype(np.float32)


def mel_fbank_htk(n_freqs, n_mels, sample_rate, f_min=0.0, f_max=None):
    """(n_freqs, n_mels) triangular HTK filterbank.  Built with torch float32 tensor ops in the
    same order torchaudio.functional.melscale_fbanks uses, because construction precision alone
    moves these weights at the 1e-5 level."""
    import torch
    if f_max is None:
        f_max = float(sample_rate // 2)
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_min = 2595.0 * math.log10(1.0 + f_min / 700.0)
    m_max = 2595.0 * math.log10(1.0 + f_max / 700.0)
    m_pts = torch.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.clamp(torch.min(down, up), min=0.0).numpy()


def _hz_to_mel_slaney(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, f / f_sp)


def _mel_to_hz_slaney(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_fbank_slaney(sr, n_fft, n_mels, fmin=0.0, fmax=None):
    """(n_mels, 1 + n_fft//2) Slaney-scale, area-normalised filterbank (librosa.filters.mel defaults)."""
    if fmax is None:
        fmax = sr / 2.0
    fftfreqs = np.fft.rfftfreq(n=n_fft, d=1.0 / sr)
    mel_f = _mel_to_hz_slaney(np.linspace(_hz_to_mel_slaney(fmin), _hz_to_mel_slaney(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    weights = np.zeros((n_mels, 1 + n_fft // 2), dtype=np.float32)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2: n_mels + 2] - mel_f[:n_mels])
    weights *= enorm[:, np.newaxis]
    return weights


def dft_basis(n_fft, win_length=None, window=None):
    """Windowed one-sided real-DFT basis as a (2F, n_win) float32 matrix, F = n_fft//2 + 1:
    rows [0,F) = w[n] cos(2 pi f (n+off)/n_fft), rows [F,2F) = -w[n] sin(...), where the window of
    win_length taps sits centred in the n_fft frame (off = (n_fft - win_length)//2, torch.stft).
    Returns (basis, off).  Angles are reduced mod n_fft in integers before the float64 cos/sin."""
    win_length = win_length or n_fft
    if window is None:
        window = hann_periodic(win_length)
    off = (n_fft - win_length) // 2
    F = n_fft // 2 + 1
    f = np.arange(F, dtype=np.int64)[:, None]
    n = (np.arange(win_length, dtype=np.int64) + off)[None, :]
    ang = 2.0 * math.pi * ((f * n) % n_fft).astype(np.float64) / n_fft
    w = window.astype(np.float64)[None, :]
    return np.concatenate([np.cos(ang) * w, -np.sin(ang) * w], 0).astype(np.float32), off


def kaiser_sinc_filter1d(cutoff, half_width, kernel_size):
    """alias_free_torch/filter.py:27-58 restated with torch float32 ops in the same order (so the buffer
    equals the one a reference checkpoint carries).  Returns (1, 1, kernel_size)."""
    import torch
    even = kernel_size % 2 == 0
    half_size = kernel_size // 2
    delta_f = 4 * half_width
    A = 2.285 * (half_size - 1) * math.pi * delta_f + 7.95
    if A > 50.0:
        beta = 0.1102 * (A - 8.7)
    elif A >= 21.0:
        beta = 0.5842 * (A - 21) ** 0.4 + 0.07886 * (A - 21.0)
    else:
        beta = 0.0
    window = torch.kaiser_window(kernel_size, beta=beta, periodic=False)
    time = (torch.arange(-half_size, half_size) + 0.5) if even else (torch.arange(kernel_size) - half_size)
    if cutoff == 0:
        return torch.zeros(1, 1, kernel_size)
    filt = 2 * cutoff * window * torch.sinc(2 * cutoff * time)
    filt = filt / filt.sum()
    return filt.view(1, 1, kernel_size)
