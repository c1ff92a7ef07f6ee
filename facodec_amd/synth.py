"""Deterministic synthetic weights and clips (there is no network for checkpoints or datasets).

Every tensor is generated from its *name* and shape through a counter-based Philox stream, so the
reference (when imported for golden-vector generation), the CPU oracle and the HIP modules all see
bit-identical parameters without a 550 MB blob ever being stored.

Scales follow PyTorch's default initialisers (U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for conv / linear /
LSTM, N(0,1) for nn.Embedding) with weight-norm gains perturbed away from ||v|| and Snake alphas
perturbed away from 1 so that those code paths are actually exercised.

Clips mirror the reference's only fake data source, PseudoDataset (meldataset.py:50-71): Gaussian
noise, peak-normalised.
"""
import zlib

import numpy as np
import torch

GAIN = 1.3  # multiplies the U(-1/sqrt(fan_in), 1/sqrt(fan_in)) bound of conv / linear weights


def _rng(name, seed):
    key = (zlib.crc32(name.encode()) << 32) | (seed & 0xFFFFFFFF)
    return np.random.Generator(np.random.Philox(key=key))


def synth_tensor(name, shape, seed=0, dtype=torch.float32):
    shape = tuple(int(s) for s in shape)
    g = _rng(name, seed)
    n = int(np.prod(shape)) if shape else 1
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "alpha" or leaf == "beta":  # Snake1d alpha (1,C,1); SnakeBeta log-scale params (C,)
        base = 1.0 if len(shape) == 3 else 0.0
        out = base + 0.2 * (g.random(n) - 0.5)
    elif leaf == "weight_g":
        out = None  # filled by synth_state_dict from the matching weight_v
    elif name.endswith("codebook.weight"):
        out = g.standard_normal(n)
    elif leaf.startswith(("weight_ih", "weight_hh", "bias_ih", "bias_hh")):
        hidden = shape[0] // 4
        b = 1.0 / np.sqrt(hidden)
        out = (g.random(n) * 2 - 1) * b
    elif len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        b = GAIN / np.sqrt(fan_in)
        out = (g.random(n) * 2 - 1) * b
    elif leaf == "bias":
        out = (g.random(n) * 2 - 1) * 0.05
    else:
        out = g.standard_normal(n) * 0.1
    if out is None:
        return None
    return torch.from_numpy(np.asarray(out, dtype=np.float64).reshape(shape)).to(dtype)


def synth_state_dict(shapes, seed=0, prefix=""):
    """shapes: {key: shape}; returns {key: tensor}.  `prefix` namespaces the RNG streams (e.g.
    'encoder.') so equal keys in different sub-models get different values."""
    sd = {}
    for k, shp in shapes.items():
        t = synth_tensor(prefix + k, shp, seed)
        if t is not None:
            sd[k] = t
    for k, shp in shapes.items():
        if k.endswith("weight_g"):
            v = sd[k[: -len("weight_g")] + "weight_v"]
            norm = v.reshape(v.shape[0], -1).norm(dim=1)
            jitter = torch.from_numpy(1.0 + 0.2 * (_rng(prefix + k, seed).random(v.shape[0]) - 0.5)).to(v.dtype)
            sd[k] = (norm * jitter).reshape(tuple(shp))
    return sd


def param_shapes(module):
    """{state-dict key: shape} of the module's parameters (buffers such as STFT windows excluded)."""
    names = {n for n, _ in module.named_parameters()}
    return {k: tuple(v.shape) for k, v in module.state_dict().items() if k in names}


def load_synthetic(module, seed=0, prefix=""):
    """Fill an nn.Module's parameters (matched by state-dict key) with synthetic values."""
    sd = synth_state_dict(param_shapes(module), seed, prefix)
    missing, unexpected = module.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    return sd


def synth_clips(batch, n_samples=48000, seed=0, rank=0, step=0):
    """(batch, 1, n_samples) fp32 clips: N(0,1) peak-normalised per clip (meldataset.py:67-68)."""
    g = _rng("clips", 1000 * rank + step + 7919 * seed)
    w = g.standard_normal((batch, n_samples))
    w = w / np.abs(w).max(axis=1, keepdims=True)
    return torch.from_numpy(w.astype(np.float32)).unsqueeze(1)
