"""CPU oracle for the FAcodec encode -> factorized-VQ -> decode (+ spectral loss) hot path.

TEST INFRASTRUCTURE ONLY.  This file is a CPU restatement (torch-CPU fp32 tensor algebra, no
nn.Module, no autograd) of the reference algorithm, used as the checker in tests/,
__graft_entry__.smoke() and as bench.py's `cpu_baseline` leg.  Nothing under facodec_amd/ may
import it; the product path is the HIP library and fails loudly without it.

Pinning: every function below is checked against outputs of the real reference (imported in the
build container from /root/reference with import shims) by tests/golden/make_golden.py; the
resulting vectors live in tests/golden/*.npz and are re-checked by tests/test_oracle_golden.py.
Encoder / decoder / LSTM / VQ / WaveNet / StyleEncoder arithmetic is fully pinned that way.
The STFT / mel pieces (`logmel_frontend`, `mel_filterbank_*`, the losses) restate un-vendored
third-party code (torchaudio, descript-audiotools, librosa -- SURVEY.md section 8c) that is absent from
the reference tree and from this image: for those rows PARITY IS UNPINNED (the reference holds no
tests or vectors for them); they are cross-checked against transformers.audio_utils only.

All functions take a flat state dict {reference key name: tensor} exactly as
`module.state_dict()` of the reference would produce it.
Citations are file:line under /root/reference.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

# ---------------------------------------------------------------------------------------------
# basic layers
# ---------------------------------------------------------------------------------------------


def weight_norm_weight(v, g):
    """Old-style torch weight_norm: w = v * (g / ||v||), norm over every dim but 0.
    dac/model/encodec.py:42-51 (apply_parametrization_norm -> torch.nn.utils.weight_norm),
    dac/nn/layers.py:9-14.  For ConvTranspose1d dim 0 is C_in."""
    norm = v.reshape(v.shape[0], -1).norm(dim=1).reshape([-1] + [1] * (v.dim() - 1))
    return v * (g / norm)


def conv_weight(sd, prefix):
    """Effective weight of a conv stored either weight-normed (weight_g/weight_v) or plain."""
    if prefix + "weight_v" in sd:
        return weight_norm_weight(sd[prefix + "weight_v"], sd[prefix + "weight_g"])
    return sd[prefix + "weight"]


def snake(x, alpha):
    """dac/nn/layers.py:18-24: x + (alpha + 1e-9)^-1 * sin(alpha x)^2, alpha (1, C, 1)."""
    return x + (alpha + 1e-9).reciprocal() * torch.sin(alpha * x).pow(2)


def _extra_padding(length, k_eff, stride, padding_total):
    """dac/model/encodec.py:71-78."""
    n_frames = (length - k_eff + padding_total) / stride + 1
    ideal = (math.ceil(n_frames) - 1) * stride + (k_eff - padding_total)
    return ideal - length


def _pad1d(x, left, right, mode):
    """dac/model/encodec.py:96-113: reflect padding that tolerates inputs shorter than the pad by
    zero-extending first and trimming afterwards."""
    if mode != "reflect":
        return F.pad(x, (left, right))
    length = x.shape[-1]
    max_pad = max(left, right)
    extra = 0
    if length <= max_pad:
        extra = max_pad - length + 1
        x = F.pad(x, (0, extra))
    y = F.pad(x, (left, right), mode="reflect")
    return y[..., : y.shape[-1] - extra]


def sconv1d(x, w, b, stride=1, dilation=1, causal=True, pad_mode="reflect"):
    """SConv1d.forward dac/model/encodec.py:212-228."""
    k_eff = (w.shape[-1] - 1) * dilation + 1
    padding_total = k_eff - stride
    extra = _extra_padding(x.shape[-1], k_eff, stride, padding_total)
    if causal:
        x = _pad1d(x, padding_total, extra, pad_mode)
    else:
        right = padding_total // 2
        x = _pad1d(x, padding_total - right, right + extra, pad_mode)
    return F.conv1d(x, w, b, stride=stride, dilation=dilation)


def sconvtr1d(x, w, b, stride, causal=True):
    """SConvTranspose1d.forward dac/model/encodec.py:248-270 (trim_right_ratio = 1)."""
    k = w.shape[-1]
    padding_total = k - stride
    y = F.conv_transpose1d(x, w, b, stride=stride)
    if causal:
        right, left = padding_total, 0
    else:
        right = padding_total // 2
        left = padding_total - right
    return y[..., left: y.shape[-1] - right]


def slstm(x, sd, prefix, num_layers):
    """SLSTM.forward dac/model/encodec.py:282-288 with nn.LSTM semantics (gates i,f,g,o; zero
    initial state; layer l consumes layer l-1's hidden sequence) and the skip connection."""
    seq = x.permute(2, 0, 1)  # (T, B, H)
    inp = seq
    for l in range(num_layers):
        w_ih, w_hh = sd[f"{prefix}weight_ih_l{l}"], sd[f"{prefix}weight_hh_l{l}"]
        b_ih, b_hh = sd[f"{prefix}bias_ih_l{l}"], sd[f"{prefix}bias_hh_l{l}"]
        H = w_hh.shape[1]
        h = torch.zeros(inp.shape[1], H, dtype=x.dtype)
        c = torch.zeros_like(h)
        pre = inp @ w_ih.t() + b_ih  # (T, B, 4H)
        outs = []
        for t in range(inp.shape[0]):
            gates = pre[t] + (h @ w_hh.t() + b_hh)
            i, f, g, o = gates.chunk(4, dim=1)
            c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
            h = torch.sigmoid(o) * torch.tanh(c)
            outs.append(h)
        inp = torch.stack(outs, 0)
    return (inp + seq).permute(1, 2, 0)


# ---------------------------------------------------------------------------------------------
# encoder / decoder  (dac/model/dac.py)
# ---------------------------------------------------------------------------------------------


def residual_unit(x, sd, p, dilation, causal):
    """ResidualUnit dac/model/dac.py:25-42: Snake -> conv k7 (dilated) -> Snake -> conv k1, + x."""
    y = snake(x, sd[p + "block.0.alpha"])
    y = sconv1d(y, conv_weight(sd, p + "block.1.conv.conv."), sd[p + "block.1.conv.conv.bias"],
                dilation=dilation, causal=causal)
    y = snake(y, sd[p + "block.2.alpha"])
    y = sconv1d(y, conv_weight(sd, p + "block.3.conv.conv."), sd[p + "block.3.conv.conv.bias"], causal=causal)
    pad = (x.shape[-1] - y.shape[-1]) // 2
    if pad > 0:
        x = x[..., pad:-pad]
    return x + y


def encoder_forward(sd, wave, rates=(2, 5, 5, 6), causal=True, lstm=2):
    """Encoder.forward dac/model/dac.py:69-104.  wave (B,1,T) -> latent (B, d_latent, ceil(T/prod(rates)))."""
    x = sconv1d(wave, conv_weight(sd, "block.0.conv.conv."), sd["block.0.conv.conv.bias"], causal=causal)
    idx = 1
    for s in rates:  # EncoderBlock :45-66
        p = f"block.{idx}."
        for j, d in enumerate((1, 3, 9)):
            x = residual_unit(x, sd, f"{p}block.{j}.", d, causal)
        x = snake(x, sd[p + "block.3.alpha"])
        x = sconv1d(x, conv_weight(sd, p + "block.4.conv.conv."), sd[p + "block.4.conv.conv.bias"],
                    stride=s, causal=causal)
        idx += 1
    if lstm:
        x = slstm(x, sd, f"block.{idx}.lstm.", lstm)
        idx += 1
    x = snake(x, sd[f"block.{idx}.alpha"])
    p = f"block.{idx + 1}.conv.conv."
    return sconv1d(x, conv_weight(sd, p), sd[p + "bias"], causal=causal)


def decoder_forward(sd, z, rates=(6, 5, 5, 2), causal=True, lstm=2):
    """Decoder.forward dac/model/dac.py:131-165.  z (B,1024,F) -> wave (B,1,F*prod(rates))."""
    x = sconv1d(z, conv_weight(sd, "model.0.conv.conv."), sd["model.0.conv.conv.bias"], causal=causal)
    idx = 1
    if lstm:
        x = slstm(x, sd, f"model.{idx}.lstm.", lstm)
        idx += 1
    for s in rates:  # DecoderBlock :107-128
        p = f"model.{idx}."
        x = snake(x, sd[p + "block.0.alpha"])
        x = sconvtr1d(x, conv_weight(sd, p + "block.1.convtr.convtr."), sd[p + "block.1.convtr.convtr.bias"],
                      s, causal=causal)
        for j, d in enumerate((1, 3, 9)):
            x = residual_unit(x, sd, f"{p}block.{j + 2}.", d, causal)
        idx += 1
    x = snake(x, sd[f"model.{idx}.alpha"])
    p = f"model.{idx + 1}.conv.conv."
    x = sconv1d(x, conv_weight(sd, p), sd[p + "bias"], causal=causal)
    return torch.tanh(x)


# ---------------------------------------------------------------------------------------------
# vector quantization  (dac/nn/quantize.py; quantize/fvq.py shares the search)
# ---------------------------------------------------------------------------------------------


def vq_nearest(latents, codebook):
    """decode_latents dac/nn/quantize.py:78-94 (== quantize/fvq.py:101-116).
    latents (B, 8, T) -> z_q (B, 8, T) from the RAW codebook, indices (B, T) int64."""
    B, D, T = latents.shape
    enc = latents.permute(0, 2, 1).reshape(B * T, D)
    e = F.normalize(enc)
    c = F.normalize(codebook)
    dist = e.pow(2).sum(1, keepdim=True) - 2 * e @ c.t() + c.pow(2).sum(1, keepdim=True).t()
    idx = (-dist).max(1)[1].reshape(B, T)
    z_q = F.embedding(idx, codebook).transpose(1, 2)
    return z_q, idx


def vq_forward(z, sd, p):
    """VectorQuantize.forward dac/nn/quantize.py:34-70 (1x1 weight-normed conv projections)."""
    z_e = F.conv1d(z, conv_weight(sd, p + "in_proj."), sd[p + "in_proj.bias"])
    z_q, idx = vq_nearest(z_e, sd[p + "codebook.weight"])
    commit = (z_e - z_q).pow(2).mean([1, 2])
    cbl = (z_q - z_e).pow(2).mean([1, 2])
    z_q = z_e + (z_q - z_e)
    out = F.conv1d(z_q, conv_weight(sd, p + "out_proj."), sd[p + "out_proj.bias"])
    return out, commit, cbl, idx, z_e


def vq_forward_train(z, sd, p):
    """VectorQuantize.forward dac/nn/quantize.py:55-67 with its detach placements (for autograd references):
    commitment = mse(z_e, z_q.detach()), codebook = mse(z_q, z_e.detach()), straight-through z_e + (z_q - z_e).detach()."""
    z_e = F.conv1d(z, conv_weight(sd, p + "in_proj."), sd[p + "in_proj.bias"])
    with torch.no_grad():
        _, idx = vq_nearest(z_e, sd[p + "codebook.weight"])
    z_q = F.embedding(idx, sd[p + "codebook.weight"]).transpose(1, 2)
    commit = (z_e - z_q.detach()).pow(2).mean([1, 2])
    cbl = (z_q - z_e.detach()).pow(2).mean([1, 2])
    z_st = z_e + (z_q - z_e).detach()
    out = F.conv1d(z_st, conv_weight(sd, p + "out_proj."), sd[p + "out_proj.bias"])
    return out, commit, cbl, idx, z_e


def rvq_forward_train(z, sd, p, n_codebooks, mask, return_latents=False):
    """ResidualVectorQuantize.forward in training mode (dac/nn/quantize.py:171-196) with the per-sample quantizer
    masks (n, B) given instead of drawn (:163-168).  return_latents: also the concatenated z_e_i of :195 (5-tuple like the
    reference's return)."""
    z_q = torch.zeros_like(z)
    residual = z
    commit = torch.zeros(())
    cbl = torch.zeros(())
    codes, latents = [], []
    for i in range(n_codebooks):
        out, c_i, cb_i, idx, z_e_i = vq_forward_train(residual, sd, f"{p}quantizers.{i}.")
        latents.append(z_e_i)
        z_q = z_q + out * mask[i][:, None, None]
        residual = residual - out
        commit = commit + (c_i * mask[i]).mean()
        cbl = cbl + (cb_i * mask[i]).mean()
        codes.append(idx)
    if return_latents:
        return z_q, torch.stack(codes, 1), torch.cat(latents, 1), commit, cbl
    return z_q, torch.stack(codes, 1), commit, cbl


def rvq_forward(z, sd, p, n_codebooks, n_quantizers=None):
    """ResidualVectorQuantize.forward dac/nn/quantize.py:127-198, eval mode (no quantizer dropout:
    the torch.randint mask of :166-171 only exists under self.training)."""
    if n_quantizers is None:
        n_quantizers = n_codebooks
    z_q = torch.zeros_like(z)
    residual = z
    commit = torch.zeros(())
    cbl = torch.zeros(())
    codes, latents = [], []
    for i in range(n_codebooks):
        if i >= n_quantizers:
            break
        out, c_i, cb_i, idx, z_e = vq_forward(residual, sd, f"{p}quantizers.{i}.")
        z_q = z_q + out
        residual = residual - out
        commit = commit + c_i.mean()
        cbl = cbl + cb_i.mean()
        codes.append(idx)
        latents.append(z_e)
    return z_q, torch.stack(codes, 1), torch.cat(latents, 1), commit, cbl


def fvq_forward(z, sd, p, training=False, commitment=0.15):
    """FactorizedVectorQuantize.forward quantize/fvq.py:36-86: weight-normed nn.Linear projections
    around the same search; returns (z_q, indices, commit_loss (B,)) with the loss zero in eval.  Differentiable with
    the reference's detach placements (:67-71 commitment against z_q.detach(), codebook loss against z_e.detach(); :76-78
    straight-through estimator), so torch autograd on this function gives the reference's gradients."""
    zt = z.transpose(1, 2)
    z_e = F.linear(zt, conv_weight(sd, p + "in_proj."), sd[p + "in_proj.bias"]).transpose(1, 2)
    z_q, idx = vq_nearest(z_e, sd[p + "_codebook.weight"])
    if training:
        loss = (z_e - z_q.detach()).pow(2).mean([1, 2]) * commitment + (z_q - z_e.detach()).pow(2).mean([1, 2])
    else:
        loss = torch.zeros(z.shape[0])
    z_q = z_e + (z_q - z_e).detach()
    out = F.linear(z_q.transpose(1, 2), conv_weight(sd, p + "out_proj."), sd[p + "out_proj.bias"]).transpose(1, 2)
    return out, idx, loss


def residual_vq_forward(x, sd, p, num_quantizers, training=False, n_quantizers=None, dropout=None, quantizer_dropout=0.0,
                        commitment=0.15):
    """ResidualVQ.forward quantize/rvq.py:27-73.  `dropout`: the (B,) draw the reference takes from torch.randint at :38-44
    (already 2 ** draw for dropout_type 'exp'); the first int(B * quantizer_dropout) samples keep only that many
    quantizers.  Returns (quantized_out, all_indices (N, B, T), all_losses (N,), all_quantized (N, B, D, T))."""
    B = x.shape[0]
    if n_quantizers is None:
        n_quantizers = num_quantizers
    if training:
        n_quantizers = torch.ones(B) * num_quantizers + 1
        n_drop = int(B * quantizer_dropout)
        n_quantizers[:n_drop] = torch.as_tensor(dropout)[:n_drop].to(n_quantizers.dtype)
    quantized_out, residual = 0.0, x
    losses, idxs, quants = [], [], []
    for i in range(num_quantizers):
        if not training and i >= n_quantizers:
            break
        q, idx, loss = fvq_forward(residual, sd, f"{p}layers.{i}.", training=training, commitment=commitment)
        mask = torch.full((B,), float(i)) < n_quantizers
        residual = residual - q
        quantized_out = quantized_out + q * mask[:, None, None]
        losses.append((loss * mask).mean())
        idxs.append(idx)
        quants.append(q)
    return quantized_out, torch.stack(idxs), torch.stack(losses), torch.stack(quants)


# ---------------------------------------------------------------------------------------------
# STFT / mel front-ends  (third-party semantics restated: PARITY UNPINNED, see module docstring)
# ---------------------------------------------------------------------------------------------


def hann_periodic(n):
    """torch.hann_window(n, periodic=True) == scipy.signal.get_window('hann', n) (fftbins=True)."""
    k = torch.arange(n, dtype=torch.float64)
    return (0.5 - 0.5 * torch.cos(2.0 * math.pi * k / n)).to(torch.float32)


def mel_filterbank_htk(n_freqs, n_mels, sample_rate, f_min=0.0, f_max=None):
    """torchaudio.functional.melscale_fbanks(norm=None, mel_scale='htk') as used by
    torchaudio.transforms.MelSpectrogram (modules/quantize.py:228-230).  fp32 construction like
    torchaudio: all_freqs = linspace(0, sr//2, n_freqs).  Returns (n_freqs, n_mels)."""
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
    return torch.clamp(torch.min(down, up), min=0.0)


def _hz_to_mel_slaney(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, mels)


def _mel_to_hz_slaney(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    freqs = f_sp * m
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), freqs)


def mel_filterbank_slaney(sr, n_fft, n_mels, fmin=0.0, fmax=None):
    """librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax) defaults (htk=False, norm='slaney',
    float32 result from float64 construction), which audiotools' AudioSignal.mel_spectrogram
    multiplies the magnitude STFT with (dac/nn/loss.py:319-320).  Returns (n_mels, 1+n_fft//2)."""
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
    return torch.from_numpy(weights.astype(np.float32))


def stft_complex(wave2d, n_fft, hop, win_length=None):
    """torch.stft(center=True, pad_mode='reflect', onesided, normalized=False) with a periodic Hann
    window of win_length centred in n_fft.  wave2d (N, T) -> complex (N, n_fft//2+1, frames)."""
    win_length = win_length or n_fft
    return torch.stft(wave2d, n_fft, hop_length=hop, win_length=win_length, window=hann_periodic(win_length),
                      center=True, pad_mode="reflect", normalized=False, onesided=True, return_complex=True)


def logmel_frontend(wave, n_bins, sample_rate=24000, n_fft=2048, win_length=1200, hop=300, n_mels=80):
    """FAquantizer.preprocess modules/quantize.py:239-242: torchaudio MelSpectrogram (power 2, HTK
    mel, no norm) -> (log(1e-5 + mel) + 4) / 4 -> first n_bins mels, first T//hop frames."""
    w = wave.squeeze(1)
    spec = stft_complex(w, n_fft, hop, win_length).abs().pow(2)
    fb = mel_filterbank_htk(n_fft // 2 + 1, n_mels, sample_rate)
    mel = torch.matmul(spec.transpose(-1, -2), fb).transpose(-1, -2)
    mel = (torch.log(1e-5 + mel) - (-4)) / 4
    return mel[:, :n_bins, : int(wave.shape[-1] / hop)]


# ---------------------------------------------------------------------------------------------
# quantizer side networks
# ---------------------------------------------------------------------------------------------


def wavenet_forward(x, sd, p, hidden, n_layers, kernel_size=5, dilation_rate=1, causal=True, g=None):
    """WN.forward modules/wavenet.py:138-166, mask of ones, dropout off (eval);
    gate = tanh((a+g_l)[:hidden]) * sigmoid((a+g_l)[hidden:]) (modules/commons.py:113-120); g (B, gin, 1)
    goes through cond_layer (a non-causal weight-normed 1x1 SConv1d, :120-121,142-143)."""
    out = torch.zeros_like(x)
    if g is not None:
        q = f"{p}cond_layer.conv.conv."
        g = sconv1d(g, conv_weight(sd, q), sd[q + "bias"], causal=False)
    for i in range(n_layers):
        q = f"{p}in_layers.{i}.conv.conv."
        a = sconv1d(x, conv_weight(sd, q), sd[q + "bias"], dilation=dilation_rate ** i, causal=causal)
        if g is not None:
            a = a + g[:, i * 2 * hidden:(i + 1) * 2 * hidden, :]
        acts = torch.tanh(a[:, :hidden]) * torch.sigmoid(a[:, hidden:])
        q = f"{p}res_skip_layers.{i}.conv.conv."
        rs = sconv1d(acts, conv_weight(sd, q), sd[q + "bias"], causal=causal)
        if i < n_layers - 1:
            x = x + rs[:, :hidden]
            out = out + rs[:, hidden:]
        else:
            out = out + rs
    return out


def mish(x):
    """modules/style_encoder.py:6-10."""
    return x * torch.tanh(F.softplus(x))


def style_attention(x, sd, p, n_heads, mask2d=None):
    """MultiHeadAttention.forward/attention modules/attentions.py:158-199 (no relative window,
    no proximal bias, dropout off)."""
    q = F.conv1d(x, sd[p + "conv_q.weight"], sd[p + "conv_q.bias"])
    k = F.conv1d(x, sd[p + "conv_k.weight"], sd[p + "conv_k.bias"])
    v = F.conv1d(x, sd[p + "conv_v.weight"], sd[p + "conv_v.bias"])
    B, C, T = q.shape
    dk = C // n_heads
    qh = q.view(B, n_heads, dk, T).transpose(2, 3)
    kh = k.view(B, n_heads, dk, T).transpose(2, 3)
    vh = v.view(B, n_heads, dk, T).transpose(2, 3)
    scores = torch.matmul(qh / math.sqrt(dk), kh.transpose(-2, -1))
    if mask2d is not None:
        scores = scores.masked_fill(mask2d == 0, -1e4)
    pa = F.softmax(scores, dim=-1)
    o = torch.matmul(pa, vh).transpose(2, 3).contiguous().view(B, C, T)
    return F.conv1d(o, sd[p + "conv_o.weight"], sd[p + "conv_o.bias"])


def style_encoder_forward(mel, sd, p, mask=None, n_heads=2):
    """StyleEncoder.forward modules/style_encoder.py:63-91 (eval: dropouts are identity).
    mel (B, 80, F), mask (B, 1, F) bool/float or None (= ones) -> timbre (B, out_dim)."""
    if mask is None:
        mask = torch.ones(mel.shape[0], 1, mel.shape[2])
    mask = mask.to(mel.dtype)
    x = mish(F.conv1d(mel, sd[p + "spectral.0.weight"], sd[p + "spectral.0.bias"]))
    x = mish(F.conv1d(x, sd[p + "spectral.3.weight"], sd[p + "spectral.3.bias"])) * mask
    for i in range(2):  # Conv1dGLU :13-31
        a = F.conv1d(x, sd[f"{p}temporal.{i}.conv1.weight"], sd[f"{p}temporal.{i}.conv1.bias"], padding=2)
        half = a.shape[1] // 2
        x = x + a[:, :half] * torch.sigmoid(a[:, half:])
    x = x * mask
    attn_mask = mask.unsqueeze(2) * mask.unsqueeze(-1)
    x = x + style_attention(x, sd, p + "slf_attn.", n_heads, attn_mask)
    x = F.conv1d(x, sd[p + "fc.weight"], sd[p + "fc.bias"])
    return x.sum(dim=2) / mask.sum(dim=2)


def sequence_mask(lengths, max_len):
    """modules/quantize.py:127-131."""
    return torch.arange(max_len).unsqueeze(0) < lengths.unsqueeze(1)


def quantizer_forward(sd, z, wave, n_c=2, full_waves=None, wave_lens=None, hop=300):
    """FAquantizer.forward_v2 modules/quantize.py:375-454, eval mode (res_mask = 1, no dropout),
    config separate_prosody_encoder=True, timbre_norm=True (configs/config.yml:27-46).
    Returns outs, [z_p, z_c, z_r], commitment, codebook, timbre, [codes_p, codes_c, codes_r]."""
    if full_waves is None:
        mel = logmel_frontend(wave, 80)
        timbre = style_encoder_forward(mel, sd, "timbre_encoder.")
    else:
        mel = logmel_frontend(full_waves.unsqueeze(1), 80)
        m = sequence_mask(wave_lens // hop, mel.shape[-1]).unsqueeze(1)
        timbre = style_encoder_forward(mel, sd, "timbre_encoder.", m)
    f0 = logmel_frontend(wave, 20)
    f0 = sconv1d(f0, sd["melspec_linear.conv.conv.weight"], sd["melspec_linear.conv.conv.bias"])
    f0 = wavenet_forward(f0, sd, "melspec_encoder.", hidden=256, n_layers=8)
    f0 = sconv1d(f0, sd["melspec_linear2.conv.conv.weight"], sd["melspec_linear2.conv.conv.bias"])
    n = min(f0.shape[2], z.shape[2])
    f0, x = f0[:, :, :n], z[:, :, :n]
    z_p, codes_p, _, cm_p, cb_p = rvq_forward(f0, sd, "prosody_quantizer.", 1, 1)
    z_c, codes_c, _, cm_c, cb_c = rvq_forward(x, sd, "content_quantizer.", _count(sd, "content_quantizer."), n_c)
    z_r, codes_r, _, cm_r, cb_r = rvq_forward(x - z_p - z_c, sd, "residual_quantizer.", 3, 3)
    outs = z_p + z_c + z_r
    style = F.linear(timbre, sd["timbre_linear.weight"], sd["timbre_linear.bias"]).unsqueeze(2)
    gamma, beta = style.chunk(2, 1)
    outs = F.layer_norm(outs.transpose(1, 2), (outs.shape[1],)).transpose(1, 2)
    outs = outs * gamma + beta
    return outs, [z_p, z_c, z_r], cm_p + cm_c + cm_r, cb_p + cb_c + cb_r, timbre, [codes_p, codes_c, codes_r]


def quantizer_forward_train(sd, z, wave, masks, side_branches_no_grad=True, full_waves=None, wave_lens=None, hop=300):
    """FAquantizer.forward_v2 in training mode (modules/quantize.py:375-454) with its detach placements (:402-411),
    the quantizer-dropout masks (dac/nn/quantize.py:163-183) and the residual mask (:419-435) given explicitly.
    side_branches_no_grad=True runs the timbre encoder and prosody WaveNet forward-only; dropouts (WaveNet 0.2,
    StyleEncoder 0.1) are not applied.  full_waves / wave_lens: the timbre encoder sees the whole utterance under
    a sequence mask (:378-383, train.py:266-269).
    Pinned to the real reference's .train() run by tests/golden/make_golden_train.py (train_step.npz)."""
    ctx = torch.no_grad() if side_branches_no_grad else torch.enable_grad()
    with ctx:
        if full_waves is None:
            mel = logmel_frontend(wave, 80)
            timbre = style_encoder_forward(mel, sd, "timbre_encoder.")
        else:
            mel_full = logmel_frontend(full_waves.unsqueeze(1), 80)
            m = sequence_mask(wave_lens // hop, mel_full.shape[-1]).unsqueeze(1)
            timbre = style_encoder_forward(mel_full, sd, "timbre_encoder.", m)
        f0 = logmel_frontend(wave, 20)
        f0 = sconv1d(f0, sd["melspec_linear.conv.conv.weight"], sd["melspec_linear.conv.conv.bias"])
        f0 = wavenet_forward(f0, sd, "melspec_encoder.", hidden=256, n_layers=8)
        f0 = sconv1d(f0, sd["melspec_linear2.conv.conv.weight"], sd["melspec_linear2.conv.conv.bias"])
    n = min(f0.shape[2], z.shape[2])
    f0, x = f0[:, :, :n], z[:, :, :n]
    z_p, codes_p, cm_p, cb_p = rvq_forward_train(f0, sd, "prosody_quantizer.", 1, masks["p"])
    z_c, codes_c, cm_c, cb_c = rvq_forward_train(x, sd, "content_quantizer.", _count(sd, "content_quantizer."), masks["c"])
    z_r, codes_r, cm_r, cb_r = rvq_forward_train(x - z_p.detach() - z_c.detach(), sd, "residual_quantizer.", 3, masks["r"])
    outs = z_p.detach() + z_c.detach() + z_r * masks["res"].to(z.dtype)[:, None, None]
    style = F.linear(timbre, sd["timbre_linear.weight"], sd["timbre_linear.bias"]).unsqueeze(2)
    gamma, beta = style.chunk(2, 1)
    outs = F.layer_norm(outs.transpose(1, 2), (outs.shape[1],)).transpose(1, 2)
    outs = outs * gamma + beta
    return outs, [z_p, z_c, z_r], cm_p + cm_c + cm_r, cb_p + cb_c + cb_r, timbre, [codes_p, codes_c, codes_r]


def _count(sd, p):
    n = 0
    while f"{p}quantizers.{n}.codebook.weight" in sd:
        n += 1
    return n


def codec_forward(sds, wave, n_c=2):
    """reconstruct.py:56-61: encoder -> quantizer -> decoder."""
    z = encoder_forward(sds["encoder"], wave)
    outs, quantized, commit, cbl, timbre, codes = quantizer_forward(sds["quantizer"], z, wave, n_c=n_c)
    y = decoder_forward(sds["decoder"], outs)
    return dict(z=z, outs=outs, quantized=quantized, commitment=commit, codebook=cbl, timbre=timbre,
                codes=codes, wave=y)


def redecoder_forward(sd, p_code, c_code, timbre, use_p_code=True, use_c_code=True, n_c=2, embed_dim=512,
                      causal=False):
    """Redecoder.forward modules/redecoder.py:35-48 (wavenet type): summed code embeddings -> WN(16 layers,
    g = timbre) -> 1x1 conv."""
    B, _, T = p_code.shape
    x = torch.zeros(B, T, embed_dim)
    if use_p_code:
        i = 0
        while f"prosody_embed.{i}.weight" in sd:
            x = x + F.embedding(p_code[:, i, :], sd[f"prosody_embed.{i}.weight"])
            i += 1
    if use_c_code:
        for i in range(n_c):
            x = x + F.embedding(c_code[:, i, :], sd[f"content_embed.{i}.weight"])
    x = wavenet_forward(x.transpose(1, 2), sd, "encoder.", embed_dim, 16, causal=causal, g=timbre.unsqueeze(2))
    return F.conv1d(x, sd["conv_out.weight"], sd["conv_out.bias"])


# ---------------------------------------------------------------------------------------------
# predictor heads (train-step only; fully inside the reference tree, pinned)
# ---------------------------------------------------------------------------------------------


def aa_snakebeta(x, alpha_log, beta_log, filt):
    """Activation1d(SnakeBeta(alpha_logscale=True)): alias_free_torch/act.py:24-29,
    resample.py:28-37 (replicate-pad 5, depthwise conv_transpose stride 2, x2, crop 15/15),
    modules/quantize.py:77-90, alias_free_torch/filter.py:89-96 (replicate-pad 5/6, stride-2 depthwise)."""
    C = x.shape[1]
    u = F.pad(x, (5, 5), mode="replicate")
    u = 2 * F.conv_transpose1d(u, filt.expand(C, -1, -1), stride=2, groups=C)
    u = u[..., 15:-15]
    a = torch.exp(alpha_log).view(1, -1, 1)
    b = torch.exp(beta_log).view(1, -1, 1)
    u = u + (1.0 / (b + 1e-9)) * torch.sin(u * a).pow(2)
    u = F.pad(u, (5, 6), mode="replicate")
    return F.conv1d(u, filt.expand(C, -1, -1), stride=2, groups=C)


def cnnlstm_forward(x, sd, p, n_heads, global_pred=False):
    """CNNLSTM.forward modules/quantize.py:106-125 (3 ResidualUnits :92-104 with dilation 1,2,3 and
    zero 'same' padding, anti-aliased activation, Linear heads; optional time mean)."""
    for j, d in enumerate((1, 2, 3)):
        q = f"{p}model.{j}.block."
        y = aa_snakebeta(x, sd[q + "0.act.alpha"], sd[q + "0.act.beta"], sd[q + "0.upsample.filter"])
        y = F.conv1d(y, conv_weight(sd, q + "1."), sd[q + "1.bias"], dilation=d, padding=3 * d)
        y = aa_snakebeta(y, sd[q + "2.act.alpha"], sd[q + "2.act.beta"], sd[q + "2.upsample.filter"])
        y = F.conv1d(y, conv_weight(sd, q + "3."), sd[q + "3.bias"])
        x = x + y
    q = f"{p}model.3."
    x = aa_snakebeta(x, sd[q + "act.alpha"], sd[q + "act.beta"], sd[q + "upsample.filter"]).transpose(1, 2)
    if global_pred:
        x = x.mean(dim=1)
    return [F.linear(x, sd[f"{p}heads.{i}.weight"], sd[f"{p}heads.{i}.bias"]) for i in range(n_heads)]


def predictors_forward(sd, quantized, timbre):
    """FApredictors.forward_v2 modules/quantize.py:564-606 with build_model's flags
    (modules/commons.py:311-322: use_gr_content_f0=False, use_gr_prosody_phone=False,
    use_gr_residual_f0=True, use_gr_residual_phone=True, use_gr_x_timbre=True); GradientReversal
    (gradient_reversal.py:11-23, alpha = 1) is the identity in the forward pass and negates the gradient: written
    here as 2 * x.detach() - x (same value bit for bit, derivative -1) so autograd through the oracle reproduces it."""
    def gr(x):
        return 2.0 * x.detach() - x

    z_p, z_c, z_r = quantized
    content = cnnlstm_forward(z_c, sd, "phone_predictor.", 1)[0]
    spk = F.linear(timbre, sd["timbre_predictor.weight"], sd["timbre_predictor.bias"])
    f0, uv = cnnlstm_forward(z_p, sd, "f0_predictor.", 2)
    rev_f0, rev_uv = cnnlstm_forward(gr(z_r), sd, "rev_f0_predictor.1.", 2)
    rev_content = cnnlstm_forward(gr(z_r), sd, "rev_content_predictor.1.", 1)[0]
    x_spk = cnnlstm_forward(gr(z_p + z_c + z_r), sd, "rev_timbre_predictor.1.", 1, global_pred=True)[0]
    return (dict(f0=f0, uv=uv, content=content, timbre=spk),
            dict(rev_f0=rev_f0, rev_uv=rev_uv, rev_content=rev_content, x_timbre=x_spk))


# ---------------------------------------------------------------------------------------------
# losses  (third-party STFT/mel semantics: PARITY UNPINNED)
# ---------------------------------------------------------------------------------------------


def mel_spectrogram_loss(x, y, sample_rate=24000,
                         n_mels=(5, 10, 20, 40, 80, 160, 320),
                         window_lengths=(32, 64, 128, 256, 512, 1024, 2048),
                         clamp_eps=1e-5, mag_weight=0.0, log_weight=1.0, pw=1.0):
    """MelSpectrogramLoss.forward dac/nn/loss.py:294-327 with train.py:155-163's arguments.
    x (estimate), y (reference): (B, 1, T)."""
    loss = torch.zeros(())
    for nm, w in zip(n_mels, window_lengths):
        fb = mel_filterbank_slaney(sample_rate, w, nm)
        mx = stft_complex(x.reshape(-1, x.shape[-1]), w, w // 4).abs()
        my = stft_complex(y.reshape(-1, y.shape[-1]), w, w // 4).abs()
        xm = (mx.transpose(1, 2) @ fb.t()).transpose(1, 2)
        ym = (my.transpose(1, 2) @ fb.t()).transpose(1, 2)
        loss = loss + log_weight * (xm.clamp(clamp_eps).pow(pw).log10() - ym.clamp(clamp_eps).pow(pw).log10()).abs().mean()
        loss = loss + mag_weight * (xm - ym).abs().mean()
    return loss


def multiscale_stft_loss(x, y, window_lengths=(2048, 512), clamp_eps=1e-5, mag_weight=1.0, log_weight=1.0, pw=2.0):
    """MultiScaleSTFTLoss.forward dac/nn/loss.py:203-228 (defaults, train.py:154)."""
    loss = torch.zeros(())
    for w in window_lengths:
        mx = stft_complex(x.reshape(-1, x.shape[-1]), w, w // 4).abs()
        my = stft_complex(y.reshape(-1, y.shape[-1]), w, w // 4).abs()
        loss = loss + log_weight * (mx.clamp(clamp_eps).pow(pw).log10() - my.clamp(clamp_eps).pow(pw).log10()).abs().mean()
        loss = loss + mag_weight * (mx - my).abs().mean()
    return loss


def waveform_l1_loss(x, y):
    """L1Loss.forward dac/nn/loss.py:31-48."""
    return (x - y).abs().mean()


def focal_loss(logits, target, gamma=2.0):
    """FocalLoss.forward losses.py:264-276 (train.py:153 gamma=2): the modulation acts on the MEAN cross entropy,
    (1 - exp(-ce))**gamma * ce.  logits (B, C, T) / (N, C), target (B, T) / (N,)."""
    ce = F.cross_entropy(logits, target)
    return (1.0 - torch.exp(-ce)) ** gamma * ce


def meldataset_preprocess(wave):
    """meldataset.py:37-47: torchaudio MelSpectrogram(n_mels=80, n_fft=2048, win 1200, hop 300) with the DEFAULT
    sample rate 16000 (filterbank quirk), all centred frames, (log(1e-5 + mel) + 4) / 4 -> (1, 80, 1 + T // 300)."""
    spec = stft_complex(wave.reshape(1, -1), 2048, 300, 1200).abs().pow(2)
    fb = mel_filterbank_htk(1025, 80, 16000)
    return (torch.log(1e-5 + torch.matmul(spec.transpose(-1, -2), fb).transpose(-1, -2)) + 4) / 4


def reconstruction_loss(x, g_x, eps=1e-7):
    """reconstruction_loss losses.py:65-89: 100*MSE + sum_{s=64..2048} [L1(mel) + sqrt(s/2) * mean_t
    RMS_mel(log diff)], torchaudio MelSpectrogram(sample_rate=16000, n_fft=max(s,512), win=s,
    hop=s//4, n_mels=64).  x, g_x: (B, T)."""
    loss = 100.0 * (x - g_x).pow(2).mean()
    for i in range(6, 12):
        s = 2 ** i
        n_fft = max(s, 512)
        fb = mel_filterbank_htk(n_fft // 2 + 1, 64, 16000)

        def mel(w):
            p = stft_complex(w, n_fft, s // 4, s).abs().pow(2)
            return torch.matmul(p.transpose(-1, -2), fb).transpose(-1, -2)

        sx, sg = mel(x), mel(g_x)
        l1 = (sx - sg).abs().mean()
        l2 = (((torch.log(sx.abs() + eps) - torch.log(sg.abs() + eps)) ** 2).mean(dim=-2) ** 0.5).mean()
        loss = loss + l1 + (s / 2) ** 0.5 * l2
    return loss


# ---------------------------------------------------------------------------------------------
# discriminator (dac/model/discriminator.py), train-step only: 5 multi-period + 3 multi-resolution
# complex-spectrogram discriminators.  The STFT of MRD goes through descript-audiotools (not vendored): its
# semantics are restated from SURVEY.md 8c ([upstream]) -- parity of that front-end is unpinned, like the losses'.
BANDS = ((0.0, 0.1), (0.1, 0.25), (0.25, 0.5), (0.5, 0.75), (0.75, 1.0))


def discriminator_preprocess(y):
    """Discriminator.preprocess :200-205: remove DC, peak-normalise to 0.8."""
    y = y - y.mean(dim=-1, keepdim=True)
    return 0.8 * y / (y.abs().max(dim=-1, keepdim=True)[0] + 1e-9)


def _wn2d(sd, p):
    v, g = sd[p + "weight_v"], sd[p + "weight_g"]
    return g * v / v.flatten(1).norm(dim=1).view(-1, 1, 1, 1)


def mpd_forward(x, sd, p, period):
    """MPD.forward :44-60: reflect-pad to a multiple of the period (a full period when already aligned, :40-42),
    fold to (B, 1, L, period), five (5,1) convs with LeakyReLU(0.1), (3,1) post conv.  Returns the 6 feature maps."""
    t = x.shape[-1]
    x = F.pad(x, (0, period - t % period), mode="reflect")
    x = x.view(x.shape[0], 1, -1, period)
    fmap = []
    for i, stride in enumerate((3, 3, 3, 3, 1)):
        q = f"{p}convs.{i}.0."
        x = F.leaky_relu(F.conv2d(x, _wn2d(sd, q), sd[q + "bias"], stride=(stride, 1), padding=(2, 0)), 0.1)
        fmap.append(x)
    q = f"{p}conv_post."
    x = F.conv2d(x, _wn2d(sd, q), sd[q + "bias"], padding=(1, 0))
    fmap.append(x)
    return fmap


def stft_match_stride(wave2d, window_length):
    """audiotools AudioSignal.stft with STFTParams(window_length, hop = window_length // 4, match_stride=True):
    reflect-pad (pad, pad + right_pad) with pad = (window - hop) // 2, right_pad = ceil(T / hop) * hop - T; periodic
    Hann; torch.stft(centre, reflect); drop two frames at either end.  -> complex (N, F, T_frames)."""
    hop = window_length // 4
    T = wave2d.shape[-1]
    right_pad = math.ceil(T / hop) * hop - T
    pad = (window_length - hop) // 2
    x = F.pad(wave2d.unsqueeze(1), (pad, pad + right_pad), mode="reflect").squeeze(1)
    win = torch.as_tensor(hann_periodic(window_length), dtype=torch.float32)
    s = torch.stft(x, n_fft=window_length, hop_length=hop, window=win, center=True, pad_mode="reflect", return_complex=True)
    return s[..., 2:-2]


def mrd_forward(x, sd, p, window_length, bands=BANDS):
    """MRD.forward :151-170: complex spectrogram (B, 2, T, F) split into 5 frequency bands, per band four (3,9) convs
    (frequency strides 1, 2, 2, 2) and a (3,3) conv, LeakyReLU(0.1); bands concatenated along frequency; (3,3) post conv.
    Returns the 26 feature maps in the reference's order."""
    s = torch.view_as_real(stft_match_stride(x.reshape(x.shape[0], x.shape[-1]), window_length))   # (B, F, T, 2)
    s = s.permute(0, 3, 2, 1)                                                                           # (B, 2, T, F)
    n_fft = window_length // 2 + 1
    fmap, outs = [], []
    for bi, (lo, hi) in enumerate(bands):
        band = s[..., int(lo * n_fft): int(hi * n_fft)]
        for i, (kf, sf) in enumerate(((9, 1), (9, 2), (9, 2), (9, 2), (3, 1))):
            q = f"{p}band_convs.{bi}.{i}.0."
            band = F.leaky_relu(F.conv2d(band, _wn2d(sd, q), sd[q + "bias"], stride=(1, sf), padding=(1, kf // 2)), 0.1)
            fmap.append(band)
        outs.append(band)
    q = f"{p}conv_post."
    y = F.conv2d(torch.cat(outs, dim=-1), _wn2d(sd, q), sd[q + "bias"], padding=(1, 1))
    fmap.append(y)
    return fmap


def discriminator_forward(sd, x, periods=(2, 3, 5, 7, 11), fft_sizes=(2048, 1024, 512)):
    """Discriminator.forward :207-210 (rates = []): list of 8 lists of feature maps; the last map of each is the logit."""
    x = discriminator_preprocess(x)
    out = [mpd_forward(x, sd, f"discriminators.{i}.", p) for i, p in enumerate(periods)]
    out += [mrd_forward(x, sd, f"discriminators.{len(periods) + i}.", w) for i, w in enumerate(fft_sizes)]
    return out


def gan_losses(d_fake, d_real):
    """train.py:282-285 (discriminator), :304-312 (generator: adversarial + feature matching)."""
    loss_d = sum((xf[-1] ** 2).mean() + ((1 - xr[-1]) ** 2).mean() for xf, xr in zip(d_fake, d_real))
    loss_g = sum(((1 - xf[-1]) ** 2).mean() for xf in d_fake)
    loss_feat = sum(F.l1_loss(xf[j], xr[j].detach()) for xf, xr in zip(d_fake, d_real) for j in range(len(xf) - 1))
    return loss_d, loss_g, loss_feat
