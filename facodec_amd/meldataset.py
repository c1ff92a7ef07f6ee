"""Dataloader-side log-mel front-end and the synthetic data source of the reference (meldataset.py).

`preprocess` mirrors meldataset.py:37-47: torchaudio MelSpectrogram(n_mels=80, n_fft=2048, win 1200,
hop 300) built WITHOUT a sample rate -- i.e. with torchaudio's default 16 000 Hz filterbank applied to
24 kHz audio (a quirk of the reference, kept) -- then (log(1e-5 + mel) + 4) / 4, all centred frames.
Same kernel chain as the quantizer's front-end (framing -> DFT GEMM -> power -> mel GEMM + log epilogue).
"""
import torch

from .quantize import LogMelFrontend
from .synth import synth_clips

_FRONTENDS = {}


def preprocess(wave):
    """wave (T,) or (B, T) float32 on the GPU -> (1, 80, 1 + T // 300) (or (B, 80, frames) for a batch)."""
    w = torch.as_tensor(wave, dtype=torch.float32)
    if w.dim() == 1:
        w = w.unsqueeze(0)
    fe = _FRONTENDS.get(w.device)
    if fe is None:
        fe = _FRONTENDS[w.device] = LogMelFrontend(sample_rate=16000, n_fft=2048, win_length=1200, hop_length=300,
                                                   n_mels=80).to(w.device)
    return fe(w, all_frames=True)


class PseudoDataset(torch.utils.data.Dataset):
    """meldataset.py:50-71: Gaussian-noise "audio", peak-normalised, 1-30 s (here: seeded per index)."""

    def __init__(self, sr=24000, range=(1, 30), length=100, seed=0):
        self.sr, self.duration_range, self.length, self.seed = sr, range, length, seed

    def __len__(self):
        return self.length

    def __getitem__(self, idx):
        lo, hi = self.duration_range
        secs = lo + (idx * 7919 + self.seed) % (hi - lo + 1)
        wave = synth_clips(1, self.sr * secs, seed=self.seed, step=idx)[0, 0]
        return wave
