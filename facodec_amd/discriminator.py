"""Discriminator of the training step (reference: dac/model/discriminator.py; built by modules/commons.py:334-340 with
rates=[], periods [2,3,5,7,11], fft_sizes [2048,1024,512]; used by train.py:280-312), HIP forward and backward.

State-dict keys equal the reference's (`discriminators.N.convs.i.0.{weight_g,weight_v,bias}`, `...conv_post.*`,
`discriminators.N.band_convs.b.i.0.*`; Conv2d weights 4-D, old-style weight_norm).

Execution: every Conv2d here has one trivial kernel dimension, so it runs on the 1-D conv kernels --
  MPD  (k,1) convs: 1-D along the folded time axis, the period folded into the batch: tensors (B*period, C, L);
  MRD  (3,k) convs: 1-D two-level-tap convs over the row-concatenated (frame, frequency) signal: tensors (1, C, B*(T+1)*P).
`Discriminator.forward` returns the reference's structure (dac/model/discriminator.py:214-217): a list of 8 lists of
feature maps shaped (B, C, L, period) for MPD and (B, C, T, F) for MRD, so train.py:280-312 (`torch.mean`,
`F.l1_loss` over them) runs unchanged.  Each returned tensor is a zero-copy strided VIEW of the internal map (autograd
flows through it) and carries the internal tensor as `._fac_internal`; `gan_losses()` -- the fused path TrainStep uses --
reads that attribute and never touches the views."""
import os

import torch
from torch import nn

from . import autograd_disc as AD
from . import losses, ops
from .wprep import cached_forward

BANDS = ((0.0, 0.1), (0.1, 0.25), (0.25, 0.5), (0.5, 0.75), (0.75, 1.0))


class _WNConv2d(nn.Module):
    """weight_norm(nn.Conv2d) parameters (old-style names)."""

    def __init__(self, c_in, c_out, kh, kw):
        super().__init__()
        w = torch.empty(c_out, c_in, kh, kw)
        nn.init.kaiming_uniform_(w, a=5 ** 0.5)
        self.weight_g = nn.Parameter(w.flatten(1).norm(dim=1).view(-1, 1, 1, 1))
        self.weight_v = nn.Parameter(w)
        bound = 1.0 / (c_in * kh * kw) ** 0.5
        self.bias = nn.Parameter(torch.empty(c_out).uniform_(-bound, bound))


def _seq(conv):
    """Sequential(conv, LeakyReLU) naming level: key `...0.weight_g`."""
    return nn.Sequential(conv)


class MPD(nn.Module):
    """discriminator.py:27-60."""

    def __init__(self, period):
        super().__init__()
        self.period = period
        ch = [1, 32, 128, 512, 1024, 1024]
        self.convs = nn.ModuleList([_seq(_WNConv2d(ch[i], ch[i + 1], 5, 1)) for i in range(5)])
        self.conv_post = _WNConv2d(1024, 1, 3, 1)
        self.strides = (3, 3, 3, 3, 1)

    def geometry(self, T):
        """Row lengths L_i and pitches P_i of the six tensors x0..x5 (+ logits) for an input of T samples.  All
        B*period rows are laid one after another inside ONE long signal (so the convs run as big single-batch
        launches instead of one short launch tile per row); a row of x_i owns P_i columns, L_i of data followed by zeros
        that act as the convs' zero padding.  P_i = stride_i * P_{i+1} keeps every row aligned to its conv's stride."""
        L = [(T + (self.period - T % self.period)) // self.period]
        for s in self.strides:
            L.append((L[-1] + 4 - 5) // s + 1)
        P = [L[5] + 2] * 6                      # x5 / x4 (stride 1): gap of 2 zeros covers k = 5 and k = 3
        for i in (3, 2, 1, 0):
            P[i] = self.strides[i] * P[i + 1]
        return L, P

    def forward(self, x):
        """x (B, 1, T) preprocessed -> 6 feature maps (1, C, B*period*P_i) with `.rows = (pitch, valid, n_rows)`."""
        B, _, T = x.shape
        L, P = self.geometry(T)
        R = B * self.period
        x = AD.PeriodFold.apply(x, self.period, P[0])
        fmap = []
        for i, (seq, s) in enumerate(zip(self.convs, self.strides)):
            c = seq[0]
            x = AD.PlainConv.apply(x, c.weight_v.squeeze(-1), c.weight_g.squeeze(-1), c.bias, 5, s, 2)
            x = AD.LeakyReLU.apply(x, 0.1, (P[i + 1], L[i + 1]))
            x.rows = (P[i + 1], L[i + 1], R)
            fmap.append(x)
        c = self.conv_post
        x = AD.PlainConv.apply(x, c.weight_v.squeeze(-1), c.weight_g.squeeze(-1), c.bias, 3, 1, 1)
        x = AD.LeakyReLU.apply(x, 1.0, (P[5], L[5]))      # identity on the data columns, zero in the gaps
        x.rows = (P[5], L[5], R)
        fmap.append(x)
        return fmap


class MRD(nn.Module):
    """discriminator.py:101-170.

    Every band runs in a row-concatenated layout (like MPD): one long signal per channel, row r = b*(T+1) + t holds the
    band's F_i valid bins followed by zeros up to the row pitch P_i; the extra row t = T of every clip is all zero.  The
    zeros are the convs' padding along frequency (gap columns) and along time (separator rows), so a (3, k) Conv2d with
    stride (1, s) is ONE 1-D conv along the concatenated axis with two-level taps (k2 = time row at offset k2 * P_in,
    k1 = frequency tap), stride s -- a long GEMM-shaped launch instead of B*T tiny per-frame ones.  P_i = s_i * P_{i+1}
    keeps rows aligned through the strided layers."""

    def __init__(self, window_length, sample_rate=24000, bands=BANDS):
        super().__init__()
        self.window_length = window_length
        n_fft = window_length // 2 + 1
        self.bands = [(int(lo * n_fft), int(hi * n_fft)) for lo, hi in bands]
        ch = 32
        stack = lambda: nn.ModuleList([_seq(_WNConv2d(2, ch, 3, 9)), _seq(_WNConv2d(ch, ch, 3, 9)), _seq(_WNConv2d(ch, ch, 3, 9)),   # noqa: E731
                                       _seq(_WNConv2d(ch, ch, 3, 9)), _seq(_WNConv2d(ch, ch, 3, 3))])
        self.band_convs = nn.ModuleList([stack() for _ in self.bands])
        self.conv_post = _WNConv2d(ch, 1, 3, 3)
        self.fstrides = (1, 2, 2, 2, 1)
        self._scale = None

    @staticmethod
    def geometry(F0):
        """Valid widths F_i and row pitches P_i of the six tensors x0..x5 of one band (x0 = the spectrogram band)."""
        F = [F0, F0]
        for _ in range(3):
            F.append((F[-1] + 8 - 9) // 2 + 1)
        F.append(F[-1])
        p8 = max(F[4] + 1, -(-(F[3] + 4) // 2), -(-(F[2] + 4) // 4), -(-(F[1] + 4) // 8))    # gaps >= 4 before every k = 9 conv
        return F, [8 * p8, 8 * p8, 4 * p8, 2 * p8, p8, p8]

    @staticmethod
    def _conv(c, x, pitch_in, sf, pf):
        """(3, kf) Conv2d with stride (1, sf), padding (1, pf) as a two-level-tap 1-D conv over the concatenated axis."""
        co, ci, _, kf = c.weight_v.shape
        return AD.PlainConv.apply(x, c.weight_v.reshape(co, ci, 3 * kf), c.weight_g.reshape(co, 1, 1), c.bias, 3 * kf, sf,
                                  pitch_in + pf, (kf, pitch_in))

    def forward(self, x):
        """x (B, 1, T) preprocessed -> 26 feature maps (1, C, B*(T'+1)*P_i) with `.rows = (pitch, valid, n_rows, T'+1, T')`."""
        B = x.shape[0]
        dev = x.device
        if self._scale is None or self._scale.basis.device != dev:
            self._scale = losses._SpectralScale(dev, self.window_length, self.window_length, self.window_length // 4)
        spec = AD.Spectrogram.apply(x.reshape(B, x.shape[-1]), self._scale)     # (B, 2F, T')
        T = spec.shape[-1]
        R = B * (T + 1)
        fmap, outs, widths = [], [], []
        for (lo, hi), stack in zip(self.bands, self.band_convs):
            F, P = self.geometry(hi - lo)
            h = AD.SpecCat.apply(spec, lo, hi - lo, P[0])
            for i, (seq, sf) in enumerate(zip(stack, self.fstrides)):
                c = seq[0]
                kf = c.weight_v.shape[-1]
                h = self._conv(c, h, P[i], sf, kf // 2)
                h = AD.LeakyReLU.apply(h, 0.1, (P[i + 1], F[i + 1], T + 1, T))
                h.rows = (P[i + 1], F[i + 1], R, T + 1, T)
                fmap.append(h)
            outs.append(h.reshape(32, R, P[5])[:, :, :F[5]])
            widths.append(F[5])
        Fp = sum(widths)
        Pp = Fp + 2
        cat = torch.nn.functional.pad(torch.cat(outs, dim=2), (0, Pp - Fp)).reshape(1, 32, R * Pp)   # bands side by side (layout copy)
        y = self._conv(self.conv_post, cat, Pp, 1, 1)
        y = AD.LeakyReLU.apply(y, 1.0, (Pp, Fp, T + 1, T))      # identity on the data, zero in gaps / separator rows
        y.rows = (Pp, Fp, R, T + 1, T)
        fmap.append(y)
        self.last_frames = T
        return fmap


class Discriminator(nn.Module):
    """discriminator.py:173-210 (rates = [] only: the MSD branch resamples through audiotools and is unused by the model)."""

    def __init__(self, rates=(), periods=(2, 3, 5, 7, 11), fft_sizes=(2048, 1024, 512), sample_rate=24000, bands=BANDS):
        super().__init__()
        if len(rates):
            raise NotImplementedError("MSD (rates != []) is not used by build_model and is not built")
        self.discriminators = nn.ModuleList([MPD(p) for p in periods] + [MRD(f, sample_rate, bands) for f in fft_sizes])

    def preprocess(self, y):
        return AD.Preprocess.apply(y)

    def _run_all(self, x):
        """The 8 sub-discriminators are independent chains of launches with 200-800 tiles each on 256 CUs (one workgroup per
        CU): run side by side on N_STREAMS streams, one chain's last, partly filled round of workgroups overlaps another
        chain's kernels.  autograd runs every node's backward on the stream of its forward, so the backward passes overlap
        the same way; the streams fork from and join the caller's stream here."""
        if N_STREAMS <= 1 or not x.is_cuda:
            return [d(x) for d in self.discriminators]
        # heaviest chains first (the spectrogram discriminators, then the period discriminators), round robin over the streams
        order = sorted(range(len(self.discriminators)), key=lambda i: not isinstance(self.discriminators[i], MRD))
        res = ops.run_chains([lambda d=self.discriminators[i]: d(x) for i in order], x.device, N_STREAMS, inputs=[x])
        outs = [None] * len(order)
        for i, r in zip(order, res):
            outs[i] = r
        return outs

    @cached_forward
    def forward_internal(self, x):
        """The 8 lists of internal-layout maps (row-concatenated MPD signals, (B*T, C, F) MRD rows)."""
        return self._run_all(self.preprocess(x))

    @cached_forward
    def forward(self, x):
        B = x.shape[0]
        maps = self._run_all(self.preprocess(x))
        return [[_reference_view(d, m, B) for m in dm] for d, dm in zip(self.discriminators, maps)]


def _reference_view(d, m, batch):
    """Zero-copy view of an internal map in the reference's layout: MPD (1, C, B*period*pitch) -> (B, C, L, period)
    (element [b, c, l, p] = row b*period + p, column l), MRD (1, C, B*(T+1)*pitch) -> (B, C, T, F)."""
    if isinstance(d, MPD):
        pitch, valid, rows = m.rows
        c = m.shape[1]
        v = m.as_strided((batch, c, valid, d.period), (d.period * pitch, rows * pitch, 1, pitch))
    else:
        pitch, valid, rows, rpg, vrows = m.rows
        c = m.shape[1]
        v = m.as_strided((batch, c, vrows, valid), (rpg * pitch, rows * pitch, pitch, 1))
    v._fac_internal = m
    return v


def reference_layout(disc, fmaps, batch):
    """Internal feature maps (`forward_internal`) -> contiguous copies in the reference's (B, C, L, period) / (B, C, T, F)."""
    out = []
    for d, maps in zip(disc.discriminators, fmaps):
        conv = []
        for m in maps:
            if isinstance(d, MPD):
                pitch, valid, rows = m.rows
                c = m.shape[1]
                v = m.reshape(c, rows, pitch)[:, :, :valid]                       # (C, B*period, L)
                conv.append(v.reshape(c, batch, d.period, valid).permute(1, 0, 3, 2).contiguous())
            else:
                pitch, valid, rows, rpg, vrows = m.rows
                c = m.shape[1]
                v = m.reshape(c, batch, rpg, pitch)[:, :, :vrows, :valid]          # (C, B, T, F)
                conv.append(v.permute(1, 0, 2, 3).contiguous())
        out.append(conv)
    return out


_MASKS = {}
N_STREAMS = int(os.environ.get("FAC_DISC_STREAMS", "3"))


def _row_mask(t):
    """1 on the data columns of a row-concatenated map, 0 in the gaps (None for plain tensors) + the data count."""
    rows = getattr(t, "rows", None)
    if rows is None:
        return None, t.numel()
    pitch, valid, n_rows = rows[:3]
    rpg, vrows = rows[3:] if len(rows) == 5 else (1, 1)        # MRD: rows_per_group = T + 1 with one zero separator row
    key = (t.shape, pitch, valid, rpg, vrows, t.device)
    if key not in _MASKS:
        pos = torch.arange(t.shape[-1], device=t.device)
        m = ((pos % pitch < valid) & ((pos // pitch) % rpg < vrows)).to(torch.float32)
        _MASKS[key] = m.reshape(1, 1, -1).expand(t.shape).contiguous()
    return _MASKS[key], t.shape[1] * (n_rows // rpg) * vrows * valid


def _internal(t):
    return getattr(t, "_fac_internal", t)


_DTARGETS = {}


def gan_loss_d_batched(d_both):
    """train.py:282-285 for ONE pass of the discriminators over [fake clips | real clips] (batch 2B, the fake half first):
    loss_d = sum_k mean(D_k(fake)^2) + mean((1 - D_k(real))^2) = sum_k sum((logits - target)^2) / n_half with target 0 on the fake
    half and 1 on the real half's data elements.  Half the launches of two separate passes and twice the tiles per launch."""
    loss = None
    for maps in d_both:
        last = _internal(maps[-1])
        mask, cnt_all = _row_mask(last)
        key = (last.shape, getattr(last, "rows", None), last.device)
        tgt = _DTARGETS.get(key)
        if tgt is None:
            tgt = (mask if mask is not None else torch.ones_like(last)).clone()
            half = last.shape[-1] // 2
            assert last.shape[-1] % 2 == 0 and cnt_all % 2 == 0
            tgt[..., :half] = 0.0
            _DTARGETS[key] = tgt
        term = AD.PairMean.apply(last, tgt, 2, cnt_all // 2)
        loss = term if loss is None else loss + term
    return loss


def gan_losses(d_fake, d_real):
    """train.py:282-285 and :304-312 -> (loss_d, loss_g, loss_feature) as autograd scalars, on the internal maps behind
    the views `Discriminator.forward` returns (plain tensors work too).  Means run over the data elements only (gap
    columns of the row-concatenated MPD maps are zero in both operands)."""
    loss_d = loss_g = loss_f = None
    acc = lambda a, b: b if a is None else a + b     # noqa: E731  (scalar adds: bookkeeping)
    for xf, xr in zip(d_fake, d_real):
        xf, xr = [_internal(t) for t in xf], [_internal(t) for t in xr]
        mask, cnt = _row_mask(xf[-1])
        one = mask if mask is not None else torch.ones_like(xf[-1])
        zero = torch.zeros_like(xf[-1])
        loss_d = acc(loss_d, AD.PairMean.apply(xf[-1], zero, 2, cnt) + AD.PairMean.apply(xr[-1], one, 2, cnt))
        loss_g = acc(loss_g, AD.PairMean.apply(xf[-1], one, 2, cnt))
        for j in range(len(xf) - 1):
            _, cj = _row_mask(xf[j])
            loss_f = acc(loss_f, AD.PairMean.apply(xf[j], xr[j].detach(), 0, cj))
    return loss_d, loss_g, loss_f
