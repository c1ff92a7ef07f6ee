"""FA-quantizer (reference: modules/quantize.py:156-454 FAquantizer.forward_v2) and its parts
(dac/nn/quantize.py VectorQuantize / ResidualVectorQuantize, modules/wavenet.py WN,
modules/style_encoder.py StyleEncoder, modules/attentions.py MultiHeadAttention), inference
(eval-mode) forward on the HIP C ABI.  State-dict keys follow the reference module tree.

Launch plan of forward_v2 (B clips):
  log-mel front-end: frames -> windowed-DFT GEMM (MFMA) -> |.|^2 -> mel GEMM with log epilogue,
                     computed ONCE and shared by the 80-bin timbre and 20-bin prosody branches
                     (the reference computes it twice, modules/quantize.py:378,385);
  timbre:  1x1 convs (+Mish epilogue) -> 2x (conv k5 -> GLU+residual) -> QKV convs -> attention ->
           out conv (+residual epilogue) -> fc -> masked mean;
  prosody: 1x1 conv -> 8x (conv k5 -> tanh*sigmoid gate -> 1x1 conv -> res/skip update) -> 1x1 conv;
  VQ:      6 fused fac_vq_fwd steps (1 prosody + 2 content + 3 residual);
  output:  LayerNorm over channels * gamma + beta (one kernel).
"""
import math
import os

import numpy as np
import torch
from torch import nn

from . import dsp, ops
from .wprep import cached_forward
from .layers import ConvWeights, SConv1d, _uniform_


# ------------------------------------------------------------------------------------ VQ
class _Codebook(nn.Module):
    def __init__(self, size, dim):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(size, dim))


class VectorQuantize(nn.Module):
    """dac/nn/quantize.py:13-94.  Keys: in_proj.{weight_g,weight_v,bias}, out_proj.{...}, codebook.weight."""

    def __init__(self, input_dim, codebook_size, codebook_dim):
        super().__init__()
        if codebook_dim != 8:
            raise NotImplementedError("the VQ kernel is specialised for codebook_dim = 8 (modules/commons.py:303)")
        self.codebook_size, self.codebook_dim, self.input_dim = codebook_size, codebook_dim, input_dim
        self.in_proj = ConvWeights(input_dim, codebook_dim, 1, weight_norm=True)
        self.out_proj = ConvWeights(codebook_dim, input_dim, 1, weight_norm=True)
        self.codebook = _Codebook(codebook_size, codebook_dim)

    def _weights(self):
        """(packed in_proj (D,1,32), out_proj weight_v (D,8,1) as stored, its weight-norm scale (D,))."""
        v = self.out_proj.weight_v.detach()
        return self.in_proj.packed(), v, ops.wn_scale(v, self.out_proj.weight_g.detach())

    def forward(self, z):
        B, D, T = z.shape
        codes = torch.empty(B, T, device=z.device, dtype=torch.int64)
        z_e = torch.empty(B, 8, T, device=z.device, dtype=torch.float32)
        out = torch.empty_like(z)
        nt = ops.vq_loss_tiles(T)
        lp = torch.empty(B, nt, device=z.device, dtype=torch.float32)
        w_in, w_out, w_out_scale = self._weights()
        ops.vq_step(z, w_in, self.in_proj.bias.detach(), self.codebook.weight.detach(), w_out, w_out_scale,
                    self.out_proj.bias.detach(), codes, zq_out=out, z_e=z_e, loss_part=lp)
        loss = lp.sum(1) / float(8 * T)
        return out, loss, loss.clone(), codes, z_e


class ResidualVectorQuantize(nn.Module):
    """dac/nn/quantize.py:97-198.  eval: `n_quantizers` codebooks for every sample.  train (:163-168): the first
    int(B * quantizer_dropout) samples keep only `torch.randint(1, n_codebooks + 1)` codebooks (drawn on the CPU's default
    generator exactly like the reference, or handed in as `masks` (n, B) of 0/1), forward and backward on the HIP path
    (`autograd.rvq`: straight-through estimator, commitment / codebook losses with the reference's detach placements);
    `latents` carry no gradient."""

    def __init__(self, input_dim=512, n_codebooks=9, codebook_size=1024, codebook_dim=8, quantizer_dropout=0.0):
        super().__init__()
        self.n_codebooks, self.codebook_size, self.codebook_dim = n_codebooks, codebook_size, codebook_dim
        self.quantizers = nn.ModuleList([VectorQuantize(input_dim, codebook_size, codebook_dim) for _ in range(n_codebooks)])
        self.quantizer_dropout = quantizer_dropout

    def forward(self, z, n_quantizers=None, masks=None):
        if self.training:
            from . import autograd as A
            B, _, T = z.shape
            if masks is None:
                masks = A.draw_quantizer_masks(self.n_codebooks, B, self.quantizer_dropout)
            latents = torch.empty(B, 8 * self.n_codebooks, T, device=z.device, dtype=torch.float32)
            z_q, codes, commit, cbl = A.rvq(self, z, ops.h2d(masks, z.device, torch.float32), latents=latents)
            return z_q, codes, latents, commit, cbl
        n = self.n_codebooks if n_quantizers is None else min(int(n_quantizers), self.n_codebooks)
        B, D, T = z.shape
        dev = z.device
        z_q = torch.zeros_like(z)
        codes = torch.empty(B, n, T, device=dev, dtype=torch.int64)
        latents = torch.empty(B, 8 * n, T, device=dev, dtype=torch.float32)
        nt = ops.vq_loss_tiles(T)
        lp = torch.empty(n, B, nt, device=dev, dtype=torch.float32)
        residual = torch.empty_like(z) if n > 1 else None
        src = z
        for i in range(n):
            q = self.quantizers[i]
            w_in, w_out, w_out_scale = q._weights()
            z_e_i = torch.empty(B, 8, T, device=dev, dtype=torch.float32)
            ops.vq_step(src, w_in, q.in_proj.bias.detach(), q.codebook.weight.detach(), w_out, w_out_scale,
                        q.out_proj.bias.detach(), codes[:, i], residual=residual if i < n - 1 else None,
                        zq_acc=z_q, z_e=z_e_i, loss_part=lp[i])
            latents[:, 8 * i: 8 * (i + 1)] = z_e_i
            src = residual
        per = lp.sum(2) / float(8 * T)       # (n, B)  mean over (8, T) per sample
        loss = per.mean(1).sum()             # mean over batch, summed over quantizers
        return z_q, codes, latents, loss, loss.clone()


# ------------------------------------------------------------------------------------ WaveNet
class WN(nn.Module):
    """modules/wavenet.py:103-166.  Keys: [cond_layer.conv.conv.*,] in_layers.i.conv.conv.*,
    res_skip_layers.i.conv.conv.*.  With gin_channels the conditioning g (B, gin, 1) goes through one
    1x1 conv (2*hidden*n_layers rows) and layer i adds its slice before the tanh/sigmoid gate -- here
    inside the gate kernel (the reference broadcasts it over time first)."""

    def __init__(self, hidden_channels, kernel_size, dilation_rate, n_layers, gin_channels=0, p_dropout=0, causal=False):
        super().__init__()
        self.hidden_channels, self.n_layers, self.gin_channels = hidden_channels, n_layers, gin_channels
        self.in_layers = nn.ModuleList()
        self.res_skip_layers = nn.ModuleList()
        if gin_channels != 0:
            self.cond_layer = SConv1d(gin_channels, 2 * hidden_channels * n_layers, 1, norm="weight_norm")
        for i in range(n_layers):
            self.in_layers.append(SConv1d(hidden_channels, 2 * hidden_channels, kernel_size,
                                          dilation=dilation_rate ** i, norm="weight_norm", causal=causal))
            rs = 2 * hidden_channels if i < n_layers - 1 else hidden_channels
            self.res_skip_layers.append(SConv1d(hidden_channels, rs, 1, norm="weight_norm", causal=causal))

    def forward(self, x, x_mask=None, g=None):
        x = x.clone()
        out = torch.zeros_like(x)
        cond = None
        if g is not None:
            cond = self.cond_layer.run(g.reshape(g.shape[0], -1, 1)).reshape(g.shape[0], -1)   # (B, 2*H*L)
        H2 = 2 * self.hidden_channels
        for i in range(self.n_layers):
            a = self.in_layers[i].run(x)
            acts = ops.gate_tanh_sigmoid(a, None if cond is None else cond[:, i * H2:(i + 1) * H2])
            rs = self.res_skip_layers[i].run(acts)
            ops.wn_res_skip_(rs, x, out, last=(i == self.n_layers - 1))
        return out


# ------------------------------------------------------------------------------------ StyleEncoder
class _PlainConv(nn.Module):
    """nn.Conv1d parameters (weight (C_out,C_in,K), bias) + packed copy."""

    def __init__(self, c_in, c_out, k):
        super().__init__()
        b = 1.0 / math.sqrt(c_in * k)
        self.weight = nn.Parameter(_uniform_(torch.empty(c_out, c_in, k), b))
        self.bias = nn.Parameter(_uniform_(torch.empty(c_out), b))
        self.c_in, self.c_out, self.k = c_in, c_out, k

    def run(self, x, pad=0, **kw):
        if self.k == 1 and pad == 0 and ops.gemm_split_ok(self.c_out, self.c_in, 1, x.shape[0] * x.shape[-1]):
            return ops.conv1d(x, None, self.c_out, 1, bias=self.bias.detach(), pad_left=0, pad_mode=ops.PAD_ZERO, t_out=x.shape[-1],
                              w_split=ops.pack_gemm_weight_split(self.weight.detach()), **kw)      # 1x1, many channels: bf16 pipe
        if (ops.BF16_SPLIT and self.k in (3, 5, 7) and self.c_in % 16 == 0 and self.c_in >= 64 and self.c_out > 32
                and x.shape[0] * x.shape[-1] > 640):      # k = 5 convs of the style encoder: split-bf16 kernel (conv1d_bsplit.hip)
            return ops.conv1d(x, None, self.c_out, self.k, bias=self.bias.detach(), pad_left=pad, pad_mode=ops.PAD_ZERO,
                              t_out=x.shape[-1], w_split=ops.pack_conv_weight_split(self.weight.detach()), **kw)
        return ops.conv1d(x, ops.pack_conv_weight(self.weight.detach()), self.c_out, self.k, bias=self.bias.detach(),
                          pad_left=pad, pad_mode=ops.PAD_ZERO, t_out=x.shape[-1], **kw)


class _Conv1dGLU(nn.Module):
    def __init__(self, c, k):
        super().__init__()
        self.conv1 = _PlainConv(c, 2 * c, k)


class MultiHeadAttention(nn.Module):
    """modules/attentions.py:120-199 without relative window / proximal bias."""

    def __init__(self, channels, out_channels, n_heads):
        super().__init__()
        self.n_heads = n_heads
        self.conv_q = _PlainConv(channels, channels, 1)
        self.conv_k = _PlainConv(channels, channels, 1)
        self.conv_v = _PlainConv(channels, channels, 1)
        self.conv_o = _PlainConv(channels, out_channels, 1)

    def forward(self, x, mask, res=None):
        q, k, v = self.conv_q.run(x), self.conv_k.run(x), self.conv_v.run(x)
        o = ops.attention(q, k, v, mask, self.n_heads)
        return self.conv_o.run(o, res=res)


class StyleEncoder(nn.Module):
    """modules/style_encoder.py:33-91 (eval).  Keys: spectral.{0,3}, temporal.{0,1}.conv1, slf_attn.*, fc."""

    def __init__(self, in_dim=513, hidden_dim=128, out_dim=256):
        super().__init__()
        # indices 0 and 3 of the reference's Sequential(conv, Mish, Dropout, conv, Mish, Dropout)
        self.spectral = nn.ModuleDict({"0": _PlainConv(in_dim, hidden_dim, 1), "3": _PlainConv(hidden_dim, hidden_dim, 1)})
        self.temporal = nn.ModuleList([_Conv1dGLU(hidden_dim, 5), _Conv1dGLU(hidden_dim, 5)])
        self.slf_attn = MultiHeadAttention(hidden_dim, hidden_dim, 2)
        self.fc = _PlainConv(hidden_dim, out_dim, 1)

    def forward(self, x, mask=None):
        """x (B, in_dim, T); mask (B, T) float 0/1 or None (= all frames valid) -> (B, out_dim)."""
        x = self.spectral["0"].run(x, act=ops.ACT_MISH)
        x = self.spectral["3"].run(x, act=ops.ACT_MISH)
        if mask is not None:
            ops.mul_mask_(x, mask)
        for glu in self.temporal:
            a = glu.conv1.run(x, pad=2)
            x = ops.glu_residual(a, x)
        if mask is not None:
            ops.mul_mask_(x, mask)
        x = self.slf_attn(x, mask, res=x)
        x = self.fc.run(x)
        return ops.masked_mean(x, mask)


# ------------------------------------------------------------------------------------ front-end
class LogMelFrontend(nn.Module):
    """FAquantizer.to_mel + preprocess (modules/quantize.py:219-242): torchaudio MelSpectrogram
    (sr 24000, n_fft 2048, win 1200 periodic Hann, hop 300, centre reflect, power 2, HTK 80 mels)
    then (log(1e-5 + mel) + 4) / 4.  Buffer names follow torchaudio so real checkpoints load."""

    def __init__(self, sample_rate=24000, n_fft=2048, win_length=1200, hop_length=300, n_mels=80):
        super().__init__()
        self.n_fft, self.win, self.hop, self.n_mels = n_fft, win_length, hop_length, n_mels
        self.spectrogram = nn.Module()
        self.spectrogram.register_buffer("window", torch.from_numpy(dsp.hann_periodic(win_length)))
        self.mel_scale = nn.Module()
        self.mel_scale.register_buffer("fb", torch.from_numpy(dsp.mel_fbank_htk(n_fft // 2 + 1, n_mels, sample_rate)))
        self._basis = None
        self._basis_split = None
        self._fb_packed = None

    def _consts(self, device):
        if self._basis is None or self._basis.device != device:
            win = self.spectrogram.window.detach().cpu().numpy()
            basis, self._off = dsp.dft_basis(self.n_fft, self.win, win)
            bt = torch.from_numpy(basis).to(device).unsqueeze(-1)
            self._basis = ops.pack_conv_weight(bt)
            self._basis_split = ops.pack_gemm_weight_split(bt) if ops.BF16_SPLIT and ops.GEMM_SPLIT else None
            fb = self.mel_scale.fb.detach().to(device).t().contiguous()  # (n_mels, F)
            self._fb_packed = ops.pack_conv_weight(fb.unsqueeze(-1))
        return self._basis, self._fb_packed, self._off

    def _apply(self, fn, *a, **kw):
        self._basis = None
        return super()._apply(fn, *a, **kw)

    def _load_from_state_dict(self, *a, **kw):
        """The packed DFT basis / filterbank are derived from the `to_mel.*` buffers: a checkpoint loaded after a forward
        must not keep the tables of the old buffers (hooked on the children too: the buffers live one level down)."""
        self._basis = None
        return super()._load_from_state_dict(*a, **kw)

    def forward(self, wave, all_frames=False):
        """wave (B, 1, T) or (B, T) -> normalised log-mel (B, n_mels, T // hop) (the quantizer's crop,
        modules/quantize.py:242) or, with all_frames, all 1 + T // hop centred frames (meldataset.py:45)."""
        w = wave.reshape(wave.shape[0], wave.shape[-1])
        B, T = w.shape
        n_frames = T // self.hop + (1 if all_frames else 0)
        basis, fbp, off = self._consts(w.device)
        F_ = self.n_fft // 2 + 1
        frames = ops.stft_frames(w, self.win, n_frames, self.hop, self.n_fft // 2, off)
        with ops.flop_key("dft"):
            if self._basis_split is not None and ops.gemm_split_ok(2 * F_, self.win, 1, B * n_frames):
                spec = ops.conv1d(frames, None, 2 * F_, 1, pad_left=0, pad_mode=ops.PAD_ZERO, t_out=n_frames, w_split=self._basis_split)
            else:
                spec = ops.conv1d(frames, basis, 2 * F_, 1, pad_left=0, pad_mode=ops.PAD_ZERO, t_out=n_frames)
            power = ops.spec_power(spec, 2)
            return ops.conv1d(power, fbp, self.n_mels, 1, pad_left=0, pad_mode=ops.PAD_ZERO, t_out=n_frames,
                              act=ops.ACT_LOG_MEL)


class _Linear(nn.Module):
    def __init__(self, c_in, c_out):
        super().__init__()
        b = 1.0 / math.sqrt(c_in)
        self.weight = nn.Parameter(_uniform_(torch.empty(c_out, c_in), b))
        self.bias = nn.Parameter(_uniform_(torch.empty(c_out), b))

    def forward(self, x):
        """x (B, C_in) -> (B, C_out) through the conv kernel on the (B, C_in, 1) view."""
        y = ops.conv1d(x.reshape(x.shape[0], x.shape[1], 1), ops.pack_conv_weight(self.weight.detach()),
                       self.weight.shape[0], 1, bias=self.bias.detach(), pad_left=0, pad_mode=ops.PAD_ZERO, t_out=1)
        return y.reshape(x.shape[0], -1)


class TimbreNorm(nn.Module):
    """`nn.LayerNorm(1024, elementwise_affine=False)` of modules/quantize.py:199, called on the (B, T, C) transpose at
    :447 and from outside the module at train.py:450-453 / eval.py:155-158.  Parameter- and buffer-free (no state-dict
    entry, like the reference's).  Runs fac_layernorm_c_affine with gamma = 1, beta = 0 (x_hat * 1 + 0 == x_hat
    bit for bit) on the (B, C, T) storage behind the caller's transpose; differentiable through the same kernel's
    backward.  forward_v2 itself uses the fused norm + gamma/beta launch (one pass instead of three)."""

    def __init__(self, normalized_shape=1024, eps=1e-5):
        super().__init__()
        self.normalized_shape, self.eps, self.elementwise_affine = (int(normalized_shape),), eps, False
        if eps != 1e-5:
            raise NotImplementedError("the LayerNorm kernel is built for eps = 1e-5 (modules/quantize.py:199)")

    def forward(self, x):
        """x (..., C), normalised over the last dimension."""
        from . import autograd as A
        C = self.normalized_shape[0]
        if x.shape[-1] != C:
            raise RuntimeError(f"TimbreNorm: expected last dimension {C}, got {tuple(x.shape)}")
        shape = x.shape
        xt = x.reshape(-1, shape[-2] if x.dim() > 1 else 1, C) if x.dim() != 3 else x
        bct = xt.transpose(1, 2)                      # the caller's x.transpose(1, 2) of a (B, C, T) tensor: already dense
        if not bct.is_contiguous():
            bct = bct.contiguous()
        style = torch.zeros(bct.shape[0], 2 * C, device=x.device, dtype=torch.float32)
        style[:, :C] = 1.0
        if torch.is_grad_enabled() and x.requires_grad:
            y = A.layernorm_affine(bct, style)
        else:
            y = ops.layernorm_c_affine(bct.detach(), style)
        return y.transpose(1, 2).reshape(shape)


# ------------------------------------------------------------------------------------ FAquantizer
def sequence_mask(length, max_length=None):
    """modules/quantize.py:127-131."""
    if max_length is None:
        max_length = int(length.max())
    x = torch.arange(max_length, dtype=length.dtype, device=length.device)
    return x.unsqueeze(0) < length.unsqueeze(1)


QUANT_STREAMS = int(os.environ.get("FAC_QUANT_STREAMS", "3"))     # concurrent chains of FAquantizer's eval forward (1 = serial)


class FAquantizer(nn.Module):
    """modules/quantize.py:156-454 with timbre_norm=True / separate_prosody_encoder=True
    (configs/config.yml:27-46); `forward` is the reference's forward_v2 (:235-237)."""

    def __init__(self, in_dim=1024, n_p_codebooks=1, n_c_codebooks=2, n_t_codebooks=2, n_r_codebooks=3,
                 codebook_size=1024, codebook_dim=8, quantizer_dropout=0.5, causal=False,
                 separate_prosody_encoder=False, timbre_norm=False):
        super().__init__()
        if not (timbre_norm and separate_prosody_encoder):
            raise NotImplementedError("only the shipped configuration (timbre_norm + separate_prosody_encoder) is built")
        rvq = lambda n: ResidualVectorQuantize(in_dim, n, codebook_size, codebook_dim, quantizer_dropout)  # noqa: E731
        self.prosody_quantizer = rvq(n_p_codebooks)
        self.content_quantizer = rvq(n_c_codebooks)
        self.timbre_encoder = StyleEncoder(in_dim=80, hidden_dim=512, out_dim=in_dim)
        self.timbre_linear = _Linear(1024, 1024 * 2)
        with torch.no_grad():
            self.timbre_linear.bias[:1024] = 1
            self.timbre_linear.bias[1024:] = 0
        self.timbre_norm = TimbreNorm(1024)
        self.residual_quantizer = rvq(n_r_codebooks)
        self.melspec_linear = SConv1d(20, 256, 1, causal=causal)
        self.melspec_encoder = WN(hidden_channels=256, kernel_size=5, dilation_rate=1, n_layers=8, gin_channels=0,
                                  p_dropout=0.2, causal=causal)
        self.melspec_linear2 = SConv1d(256, 1024, 1, causal=causal)
        self.to_mel = LogMelFrontend(24000, 2048, 1200, 300, 80)
        self.hop_length = 300
        self.in_dim = in_dim
        self.is_timbre_norm = True
        self.separate_prosody_encoder = True

    def preprocess(self, wave_tensor, n_bins=20):
        return self.to_mel(wave_tensor)[:, :n_bins]

    prob_random_mask_residual = 0.75     # modules/quantize.py:217

    def _forward_train(self, x, wave_segments, full_waves, wave_lens, return_codes, masks=None):
        """forward_v2 in training mode (modules/quantize.py:375-454) with HIP forward + backward.
        Every branch carries gradient: the three RVQs (straight-through, commitment / codebook losses, quantizer
        dropout), the residual path into the encoder latent, LayerNorm + timbre_linear, the timbre encoder
        (Mish / GLU / attention / masked mean, dropout 0.1) and the prosody WaveNet (gates, dropout 0.2).
        masks: optional dict(p=, c=, r= (n, B) quantizer-dropout masks, res= (B,) residual mask, dropout=False to
        switch the Bernoulli dropouts off) for reproducible steps; drawn like the reference otherwise."""
        from . import autograd as A
        from . import autograd_quant as AQ
        import numpy as np
        B = x.shape[0]
        dev = x.device
        use_drop = bool((masks or {}).get("dropout", True))      # WaveNet p = 0.2, StyleEncoder p = 0.1
        mel = self.to_mel(wave_segments)                           # constant input features (no gradient)
        n = min(mel.shape[-1], x.shape[2])
        if x.shape[2] != n:
            x = x[:, :, :n].contiguous()
        masks = masks or {}

        def qmask(key, rvq):
            mk = masks.get(key)
            if mk is None:
                mk = A.draw_quantizer_masks(rvq.n_codebooks, B, rvq.quantizer_dropout)
            return ops.h2d(mk, dev)                     # asynchronous: a blocking copy here costs the host its lead over the device

        # The same three independent chains as the eval forward, side by side (ops.run_chains; autograd replays each chain's backward
        # on the chain's stream).  The chains are ISSUED one after the other by the host, so the random draws keep their order:
        # StyleEncoder dropout, WaveNet dropout, prosody quantizer-dropout draw, content draw (then the residual draw below).
        def timbre_chain():
            if full_waves is None:
                return AQ.style_encoder(self.timbre_encoder, mel, None, use_dropout=use_drop)
            mel_full = self.to_mel(full_waves)
            m = sequence_mask(ops.h2d(wave_lens, dev) // self.hop_length, mel_full.shape[-1]).to(torch.float32).contiguous()
            return AQ.style_encoder(self.timbre_encoder, mel_full, m, use_dropout=use_drop)

        def prosody_chain():
            f0 = A.conv(self.melspec_linear, mel[:, :20].contiguous())
            f0 = A.conv(self.melspec_linear2, AQ.wavenet(self.melspec_encoder, f0, use_dropout=use_drop))
            if f0.shape[2] != n:
                f0 = f0[:, :, :n].contiguous()
            return A.rvq(self.prosody_quantizer, f0, qmask("p", self.prosody_quantizer))

        timbre, (z_p, codes_p, cm_p, cb_p), (z_c, codes_c, cm_c, cb_c) = ops.run_chains(
            [timbre_chain, prosody_chain, lambda: A.rvq(self.content_quantizer, x, qmask("c", self.content_quantizer))], dev, QUANT_STREAMS,
            inputs=[mel, x, full_waves, wave_lens, [v for v in masks.values() if torch.is_tensor(v)]])
        z_r, codes_r, cm_r, cb_r = A.rvq(self.residual_quantizer, A.sub_detached(x, z_p, z_c), qmask("r", self.residual_quantizer))
        res = masks.get("res")
        if res is None:
            res = torch.from_numpy(np.random.choice([0, 1], size=B, p=[self.prob_random_mask_residual,
                                                                       1 - self.prob_random_mask_residual]))
        res = ops.h2d(res, dev, torch.float32).contiguous()
        outs = A.mix_outs(z_p, z_c, z_r, res)
        outs = A.layernorm_affine(outs, A.linear(self.timbre_linear, timbre))
        quantized = [z_p, z_c, z_r]
        commitment, codebook = cm_p + cm_c + cm_r, cb_p + cb_c + cb_r
        if return_codes:
            return outs, quantized, commitment, codebook, timbre, [codes_p, codes_c, codes_r]
        return outs, quantized, commitment, codebook, timbre

    @cached_forward
    def forward(self, x, wave_segments, n_c=1, n_t=2, full_waves=None, wave_lens=None, return_codes=False, masks=None):
        if self.training:
            return self._forward_train(x, wave_segments, full_waves, wave_lens, return_codes, masks)
        mel = self.to_mel(wave_segments)                      # (B, 80, F) computed once
        n = min(mel.shape[-1], x.shape[2])
        if x.shape[2] != n:
            x = x[:, :, :n].contiguous()

        # Three independent chains (round 5): the timbre encoder, the prosody branch (1x1 -> WaveNet -> 1x1 -> prosody RVQ) and the
        # content RVQ depend only on the log-mel features / the latent; at the 160-frame latent rate their ~60 launches have
        # 128 - 256 workgroups each, less than one round of the chip.  Side by side on side streams (ops.run_chains, as the eight
        # discriminators and the predictor heads in training): same kernels, same results, FAC_QUANT_STREAMS=1 runs them in turn.
        def timbre_chain():
            if full_waves is None:
                return self.timbre_encoder(mel, None)
            mel_full = self.to_mel(full_waves)
            m = sequence_mask(ops.h2d(wave_lens, mel_full.device) // self.hop_length, mel_full.shape[-1]).to(torch.float32).contiguous()
            return self.timbre_encoder(mel_full, m)

        def prosody_chain():
            f0 = ops.conv1d(mel[:, :20], self.melspec_linear.w.packed(), 256, 1, bias=self.melspec_linear.w.bias,
                            pad_left=0, pad_mode=ops.PAD_ZERO, t_out=mel.shape[-1])
            f0 = self.melspec_linear2.run(self.melspec_encoder(f0))
            if f0.shape[2] != n:
                f0 = f0[:, :, :n].contiguous()
            return self.prosody_quantizer(f0, 1)

        timbre, (z_p, codes_p, _, cm_p, cb_p), (z_c, codes_c, _, cm_c, cb_c) = ops.run_chains(
            [timbre_chain, prosody_chain, lambda: self.content_quantizer(x, n_c)], x.device, QUANT_STREAMS, inputs=[mel, x, full_waves, wave_lens])
        residual_feature = ops.sub2(x, z_p, z_c)
        z_r, codes_r, _, cm_r, cb_r = self.residual_quantizer(residual_feature, 3)
        outs = ops.add(ops.add(z_p, z_c), z_r)

        style = self.timbre_linear(timbre)                    # (B, 2D) = [gamma | beta]
        outs = ops.layernorm_c_affine(outs, style)

        quantized = [z_p, z_c, z_r]
        commitment = cm_p + cm_c + cm_r
        codebook = cb_p + cb_c + cb_r
        if return_codes:
            return outs, quantized, commitment, codebook, timbre, [codes_p, codes_c, codes_r]
        return outs, quantized, commitment, codebook, timbre

    forward_v2 = forward
