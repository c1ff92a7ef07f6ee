"""Parameter-holding leaf modules with the reference's state-dict names, executing on the HIP
C ABI.  The module tree exists so that `state_dict()` / `load_state_dict()` / optimisers see exactly
the reference's parameter names and shapes; the arithmetic lives in
libfacodec_hip.so and is driven by the fused plans in dac_model.py / quantize.py.

Reference counterparts: dac/model/encodec.py (SConv1d :192-228, SConvTranspose1d :231-270,
SLSTM :272-288, NormConv1d :125-139), dac/nn/layers.py (Snake1d :27-33, WNConv1d :9-10).
"""
import math
import os

import torch
from torch import nn

from . import ops


def _uniform_(t, bound):
    with torch.no_grad():
        return t.uniform_(-bound, bound)


class ConvWeights(nn.Module):
    """weight_g / weight_v / bias of an old-style weight-normed conv (or plain weight / bias), in
    torch's layout: Conv1d (C_out, C_in, K); ConvTranspose1d (C_in, C_out, K) with the norm taken
    over dim 0 = C_in.  `packed()` materialises w = g*v/||v|| straight into the MFMA kernel's layout
    (K6); like the reference it is recomputed on every forward unless `freeze_packed` is set."""

    def __init__(self, c_in, c_out, k, weight_norm=True, transposed=False, stride=1, bias=True):
        super().__init__()
        self.c_in, self.c_out, self.k = c_in, c_out, k
        self.transposed, self.stride, self.weight_norm = transposed, stride, weight_norm
        shape = (c_in, c_out, k) if transposed else (c_out, c_in, k)
        fan_in = (c_out if transposed else c_in) * k
        w = _uniform_(torch.empty(shape), 1.0 / math.sqrt(fan_in))
        if weight_norm:
            self.weight_g = nn.Parameter(w.reshape(shape[0], -1).norm(dim=1).reshape(shape[0], 1, 1))
            self.weight_v = nn.Parameter(w)
        else:
            self.weight = nn.Parameter(w)
        self.bias = nn.Parameter(_uniform_(torch.empty(c_out), 1.0 / math.sqrt(fan_in))) if bias else None
        self._packed = None
        self._split = None
        self._rows = None
        self._rows_split = None
        self.freeze_packed = False

    def packed(self):
        if self._packed is not None and self.freeze_packed:
            return self._packed
        v = self.weight_v if self.weight_norm else self.weight
        g = self.weight_g if self.weight_norm else None
        if self.transposed:
            self._packed = ops.pack_convtr_weight(v.detach(), g.detach() if g is not None else None, self.stride,
                                                  out=self._packed)
        else:
            self._packed = ops.pack_conv_weight(v.detach(), g.detach() if g is not None else None, out=self._packed)
        return self._packed

    def packed_rows(self):
        """ConvTranspose1d weights for the all-phases launch (ops.pack_convtr_weight_rows)."""
        if self._rows is not None and self.freeze_packed:
            return self._rows
        v = self.weight_v if self.weight_norm else self.weight
        g = self.weight_g if self.weight_norm else None
        self._rows = ops.pack_convtr_weight_rows(v.detach(), g.detach() if g is not None else None, self.stride, out=self._rows)
        return self._rows

    def packed_rows_split(self):
        """ConvTranspose1d weights for the all-phases launch on the split-bf16 GEMM kernel: (buffer, rows)."""
        if self._rows_split is not None and self.freeze_packed:
            return self._rows_split
        v = self.weight_v if self.weight_norm else self.weight
        g = self.weight_g if self.weight_norm else None
        prev = self._rows_split[0] if self._rows_split is not None else None
        self._rows_split = ops.pack_convtr_weight_rows_split(v.detach(), g.detach() if g is not None else None, self.stride, out=prev)
        return self._rows_split

    def packed_split_strided(self, stride):
        """Split GEMM weights of a strided conv (stride < k <= 2 * stride), ops.pack_gemm_weight_split(in_stride=stride)."""
        if self._split is not None and self.freeze_packed:
            return self._split
        v = self.weight_v if self.weight_norm else self.weight
        g = self.weight_g if self.weight_norm else None
        self._split = ops.pack_gemm_weight_split(v.detach(), g.detach() if g is not None else None, out=self._split, in_stride=stride)
        return self._split

    def packed_split(self):
        """The same weights as three exact bf16 planes (ops.pack_conv_weight_split) for the k = 7 convs."""
        if self._split is not None and self.freeze_packed:
            return self._split
        v = self.weight_v if self.weight_norm else self.weight
        g = self.weight_g if self.weight_norm else None
        self._split = ops.pack_conv_weight_split(v.detach(), g.detach() if g is not None else None, out=self._split)
        return self._split

    def _apply(self, fn, *a, **kw):
        self._packed = None  # device / dtype moves invalidate the packed copies
        self._split = None
        self._rows = None
        self._rows_split = None
        return super()._apply(fn, *a, **kw)


class _Norm(nn.Module):
    """Mirrors the NormConv1d / NormConvTranspose1d naming level (`.conv` / `.convtr`)."""

    def __init__(self, name, weights):
        super().__init__()
        setattr(self, name, weights)


# Short clips through the split GEMM kernel as one flattened signal (SConv1d._run_flat, SConvTranspose1d.run): inference only.
FLAT_SHORT_CLIPS = os.environ.get("FAC_FLAT_SHORT", "1") != "0"
PW_TAILS_TO_384 = os.environ.get("FAC_PW_TAILS_384", "1") != "0"
FLAT_STRIDE1 = os.environ.get("FAC_FLAT_STRIDE1", "1") != "0"      # wide stride-1 k = 7 convs on short clips (SConv1d._run_flat_stride1)


class SConv1d(nn.Module):
    """Causal / asymmetric-padded Conv1d (dac/model/encodec.py:192-228).  State-dict keys:
    conv.conv.{weight_g,weight_v,bias} (norm='weight_norm') or conv.conv.{weight,bias}."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, dilation=1, causal=False,
                 norm="none", pad_mode="reflect", bias=True):
        super().__init__()
        self.conv = _Norm("conv", ConvWeights(in_channels, out_channels, kernel_size, norm == "weight_norm", bias=bias))
        self.kernel_size, self.stride, self.dilation = kernel_size, stride, dilation
        self.causal = causal
        self.pad_mode = ops.PAD_REFLECT if pad_mode == "reflect" else ops.PAD_ZERO

    @property
    def w(self):
        return self.conv.conv

    def run(self, x, alpha_in=None, alpha_out=None, res=None, act=ops.ACT_NONE, alpha_y2=None, want_y=True):
        """alpha_y2: additionally emit snake(y, alpha_y2) for the next Snake->conv (returns (y, y2))."""
        w = self.w
        split = None
        if (ops.BF16_SPLIT and self.kernel_size in (3, 5, 7) and self.stride == 1 and alpha_in is None and w.c_in % 16 == 0
                and w.c_out > 2 and x.shape[0] * x.shape[-1] > 640 and (self.kernel_size == 7 or (w.c_in >= 64 and w.c_out > 32))):
            split = w.packed_split()
        elif (self.kernel_size == 1 and self.stride == 1 and alpha_in is None
              and ops.gemm_split_ok(w.c_out, w.c_in, 1, x.shape[0] * x.shape[-1])
              and not (PW_TAILS_TO_384 and w.c_in == w.c_out and w.c_in in (256, 384) and x.shape[0] * x.shape[-1] >= 65536)):
            # (the C = 256 / 384 ResidualUnit tails stay on the streaming k = 1 kernel: -0.7 ms per B = 32 forward, round 4)
            split = w.packed_split()          # 1x1 with many channels: split-bf16 GEMM (conv1d_gemm_split.hip)
        elif (self.stride == 2 and alpha_in is None and alpha_out is None and res is None and act == ops.ACT_NONE and self.dilation == 1
              and isinstance(x, torch.Tensor)
              and ops.pw_taps_ok(w.c_in, w.c_out, self.kernel_size, 2, False, x.shape[0], -(-x.shape[-1] // 2))):
            pass                              # few channels: the streaming kernel with taps takes the fp32 pack (fac_conv_desc.pw_split)
        elif (self.stride > 1 and alpha_in is None and self.dilation == 1
              and ops.gemm_split_strided_ok(w.c_out, w.c_in, self.kernel_size, self.stride, x.shape[0], -(-x.shape[-1] // self.stride))):
            split = w.packed_split_strided(self.stride)     # downsampling conv: 2 taps over `stride` phase sub-signals
        elif (FLAT_SHORT_CLIPS and self.stride > 1 and alpha_in is None and res is None and self.dilation == 1 and self.causal
              and self.pad_mode == ops.PAD_REFLECT and self.kernel_size == 2 * self.stride and x.shape[-1] % self.stride == 0
              and x.shape[-1] > self.stride and not torch.is_grad_enabled()
              and ops.gemm_split_strided_ok(w.c_out, w.c_in, self.kernel_size, self.stride, 1,
                                            x.shape[0] * (x.shape[-1] // self.stride + 1) - 1)):
            return self._run_flat(x, alpha_out, act, alpha_y2, want_y)
        if (FLAT_SHORT_CLIPS and FLAT_STRIDE1 and split is not None and self.stride == 1 and self.kernel_size == 7 and res is None
                and self.causal and self.pad_mode == ops.PAD_REFLECT and not torch.is_grad_enabled() and isinstance(x, torch.Tensor)
                and x.shape[0] >= 4 and (self.kernel_size - 1) * self.dilation < x.shape[-1] <= 224 and w.c_in * w.c_out >= 1 << 20):
            return self._run_flat_stride1(x, alpha_out, act, alpha_y2, want_y, split)
        if split is not None and self.stride > 1:      # split GEMM over the phase sub-signals: P8 input where it pays (ops.p8_prepass)
            x = ops.p8_prepass(x, 2.0 * w.c_out * self.kernel_size / (4.0 * self.stride))
        return ops.conv1d(x, w.packed() if split is None else None, w.c_out, self.kernel_size, bias=w.bias,
                          stride=self.stride, dilation=self.dilation, pad_mode=self.pad_mode, alpha_in=alpha_in,
                          alpha_out=alpha_out, res=res, act=act, causal=self.causal, alpha_y2=alpha_y2, want_y=want_y,
                          w_split=split)

    def _run_flat(self, x, alpha_out, act, alpha_y2, want_y):
        """Short clips (the 160-frame latent rate): per-clip column tiles would be half empty, so the split GEMM kernel refuses them
        and the launch fell to the fp32 tile at ~72 TFLOP/s.  No new kernel is needed: every clip is reflect-padded on the left
        by k - s = s samples (the causal padding of dac/model/encodec.py:212-222; T % s == 0, so there is no right padding) and
        the padded clips are laid one after another as ONE signal of B (T / s + 1) s samples; the same strided conv without
        padding then computes every real output exactly (output t of clip b is column b (T / s + 1) + t) plus one junk column
        per clip where the window straddles two clips, which is dropped on the way back to (B, C, T / s)."""
        w, s_ = self.w, self.stride
        B, c_in, T = x.shape
        n = T // s_
        xp = torch.nn.functional.pad(x, (s_, 0), mode="reflect")                    # data movement only
        xf = xp.permute(1, 0, 2).reshape(1, c_in, B * (n + 1) * s_)
        t_out = B * (n + 1) - 1
        xf = ops.p8_prepass(xf, 2.0 * w.c_out * self.kernel_size / (4.0 * s_))
        got = ops.conv1d(xf, None, w.c_out, self.kernel_size, bias=w.bias, stride=s_, pad_left=0, pad_mode=ops.PAD_ZERO, t_out=t_out,
                         alpha_out=alpha_out, act=act, alpha_y2=alpha_y2, want_y=want_y, w_split=w.packed_split_strided(s_))

        def back(y):
            if y is None:
                return None
            full = torch.empty(w.c_out, B * (n + 1), device=y.device, dtype=y.dtype)
            full[:, :t_out] = y[0]
            return full.reshape(w.c_out, B, n + 1)[:, :, :n].permute(1, 0, 2).contiguous()

        return (back(got[0]), back(got[1])) if alpha_y2 is not None else back(got)

    def _run_flat_stride1(self, x, alpha_out, act, alpha_y2, want_y, split):
        """The stride-1 counterpart of `_run_flat` (round 5) for the decoder's input conv 1024 -> 1536 k 7 at the 160-frame latent
        rate, which fills 160 of the 256 columns of the split kernel's time tile.  Every clip is reflect-padded on the left by
        P = (k - 1) d (the causal padding of dac/model/encodec.py:212-222, materialised: data movement only) and the padded clips are
        laid one after another as ONE signal of B (T + P) samples; the same conv WITHOUT padding computes output t of clip b at column
        b (T + P) + t -- the same products in the same order as the per-clip launch, so the same bits -- plus P junk columns per
        clip where the window straddles two clips, dropped on the way back.  Measured at B = 32: 0.756 -> 0.517 ms (149 -> 226
        TFLOP/s-eq).  The k = 3 / k = 5 layers at the same rate gain nothing (0.269 -> 0.261, 0.059 -> 0.055 ms: few taps per staged
        column, they are bound by staging, not by tile columns) and stay on the per-clip launch."""
        w = self.w
        B, c_in, T = x.shape
        P = (self.kernel_size - 1) * self.dilation
        xp = torch.nn.functional.pad(x, (P, 0), mode="reflect")
        xf = xp.permute(1, 0, 2).reshape(1, c_in, B * (T + P))
        t_out = B * (T + P) - P
        got = ops.conv1d(xf, None, w.c_out, self.kernel_size, bias=w.bias, stride=1, dilation=self.dilation, pad_left=0,
                         pad_mode=ops.PAD_ZERO, t_out=t_out, alpha_out=alpha_out, act=act, alpha_y2=alpha_y2, want_y=want_y, w_split=split)

        def back(y):
            if y is None:
                return None
            full = torch.empty(w.c_out, B * (T + P), device=y.device, dtype=y.dtype)
            full[:, :t_out] = y[0]
            return full.reshape(w.c_out, B, T + P)[:, :, :T].permute(1, 0, 2).contiguous()

        return (back(got[0]), back(got[1])) if alpha_y2 is not None else back(got)

    def forward(self, x):
        return self.run(x)


class SConvTranspose1d(nn.Module):
    """Causal ConvTranspose1d with right trim (dac/model/encodec.py:231-270).  Keys:
    convtr.convtr.{weight_g (C_in,1,1), weight_v (C_in,C_out,K), bias}."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, causal=False, norm="none"):
        super().__init__()
        self.convtr = _Norm("convtr", ConvWeights(in_channels, out_channels, kernel_size, norm == "weight_norm",
                                                  transposed=True, stride=stride))
        self.stride = stride
        self.causal = causal

    @property
    def w(self):
        return self.convtr.convtr

    def run(self, x, alpha_in=None, alpha_y2=None):
        w = self.w
        if (self.causal and alpha_in is None and isinstance(x, torch.Tensor)
                and ops.pw_taps_ok(w.c_in, w.c_out, 2 * self.stride, self.stride, True, x.shape[0], x.shape[-1])):
            wp = w.packed_rows()              # stride 2, few channels: the streaming kernel with taps (conv1d_pw_split.hip)
        elif ops.convtr_split_ok(w.c_in, w.c_out, self.stride, x.shape[0], x.shape[-1], self.causal, alpha_in):
            wp = w.packed_rows_split()
            x = ops.p8_prepass(x, 2.0 * w.c_out * 2 * self.stride / 4.0)      # all output phases as GEMM rows: 2 s C_out MACs per input sample
        elif (FLAT_SHORT_CLIPS and not torch.is_grad_enabled() and x.shape[-1] < 256
              and ops.convtr_split_ok(w.c_in, w.c_out, self.stride, 1, x.shape[0] * (x.shape[-1] + 1), self.causal, alpha_in)):
            # short clips: ONE signal of B (T + 1) columns with a zero column in front of every clip (the x[t - 1] of its first
            # frame); the s output samples of that column are dropped on the way back (see SConv1d._run_flat)
            B, c_in, T = x.shape
            xf = torch.cat([torch.zeros(B, c_in, 1, device=x.device, dtype=x.dtype), x], -1).permute(1, 0, 2).reshape(1, c_in, B * (T + 1))
            xf = ops.p8_prepass(xf, 2.0 * w.c_out * 2 * self.stride / 4.0)
            got = ops.conv_transpose1d(xf, w.packed_rows_split(), w.c_out, self.stride, bias=w.bias, alpha_y2=alpha_y2, causal=True)

            def back(y):
                return y.reshape(w.c_out, B, (T + 1) * self.stride)[:, :, self.stride:].permute(1, 0, 2).contiguous()

            return (back(got[0]), back(got[1])) if alpha_y2 is not None else back(got)
        else:
            wp = w.packed_rows() if ops.convtr_rows_ok(x.shape[-1], self.stride, self.causal) else w.packed()
        return ops.conv_transpose1d(x, wp, w.c_out, self.stride, bias=w.bias, alpha_in=alpha_in,
                                    alpha_y2=alpha_y2, causal=self.causal)

    def forward(self, x):
        return self.run(x)


class Snake1d(nn.Module):
    """alpha (1, C, 1) of dac/nn/layers.py:27-33.  Normally fused into the neighbouring conv."""

    def __init__(self, channels):
        super().__init__()
        self.alpha = nn.Parameter(torch.ones(1, channels, 1))

    def flat(self):
        return self.alpha.detach().reshape(-1)

    def forward(self, x):
        return ops.snake(x, self.flat())


class _LSTMParams(nn.Module):
    """nn.LSTM's parameter names/shapes (weight_ih_l{k} (4H,H), weight_hh_l{k}, bias_ih_l{k}, bias_hh_l{k})."""

    def __init__(self, hidden, num_layers):
        super().__init__()
        b = 1.0 / math.sqrt(hidden)
        for l in range(num_layers):
            for n, shp in (("weight_ih", (4 * hidden, hidden)), ("weight_hh", (4 * hidden, hidden)),
                           ("bias_ih", (4 * hidden,)), ("bias_hh", (4 * hidden,))):
                setattr(self, f"{n}_l{l}", nn.Parameter(_uniform_(torch.empty(shp), b)))


class SLSTM(nn.Module):
    """dac/model/encodec.py:272-288: multi-layer LSTM over time on (B, C, T) plus skip.
    Input projections run as ONE GEMM per layer on the MFMA conv kernel over the channel-major
    (H, T*BP) work buffer; the recurrence is one resident launch per layer (fac_lstm_layer_fwd_persist) where it
    applies, else fac_lstm_layer_fwd (one launch per step)."""

    def __init__(self, dimension, num_layers=2, skip=True):
        super().__init__()
        self.lstm = _LSTMParams(dimension, num_layers)
        self.dimension, self.num_layers, self.skip = dimension, num_layers, skip

    def forward(self, x, alpha_out=None):
        """alpha_out: Snake alpha applied to the (skip-added) output by the transpose-back kernel, for
        the Snake that follows the LSTM in the Encoder / precedes the first DecoderBlock's ConvTranspose."""
        B, H, T = x.shape
        inp = ops.lstm_to_time_major(x)
        for l in range(self.num_layers):
            p = self.lstm
            w_raw = getattr(p, f"weight_ih_l{l}").detach()
            use_split = ops.gemm_split_ok(4 * H, H, 1, inp.shape[1] * inp.shape[2])
            w_ih = None if use_split else ops.pack_conv_weight(w_raw)
            w_ih_split = ops.pack_gemm_weight_split(w_raw) if use_split else None
            bias = ops.add(getattr(p, f"bias_ih_l{l}").detach(), getattr(p, f"bias_hh_l{l}").detach())
            w_hh = getattr(p, f"weight_hh_l{l}").detach()
            persist = ops.lstm_persist_ok(H, B)
            T_, BP = inp.shape[1], inp.shape[2]
            # one GEMM over every (t, b): the channel-major buffer is a (1, H, T*BP) "signal"
            sig = inp.view(1, H, T_ * BP)
            if use_split:
                sig = ops.p8_prepass(sig, 2.0 * 4 * H / 4.0)
            with ops.flop_scale(B / BP):
                pre = ops.conv1d(sig, w_ih, 4 * H, 1, bias=bias, pad_left=0, t_out=T_ * BP,
                                 pad_mode=ops.PAD_ZERO, w_split=w_ih_split)
                if ops.lstm_persist_split_ok(H, B, T_):      # 17 .. 32 columns: resident, W_hh . h on the bf16 matrix pipe
                    inp = ops.lstm_layer_persist_split(pre.view(4 * H, T_, BP), w_hh, H, B)
                elif persist:     # whole layer in one launch, W_hh resident in registers (lstm_persist.hip)
                    inp = ops.lstm_layer_persist(pre.view(4 * H, T_, BP), w_hh, H, B)
                else:
                    inp = ops.lstm_layer(pre.view(4 * H, T_, BP), ops.pack_lstm_whh(w_hh), H)
        return ops.lstm_from_time_major(inp, x if self.skip else None, B, alpha_out)
