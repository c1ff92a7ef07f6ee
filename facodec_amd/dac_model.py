"""Encoder / Decoder of the codec (reference: dac/model/dac.py:25-165) as fused launch plans.

Module tree = the reference's (same state-dict keys: `block.N...` / `model.N...`), but nothing is
executed layer-by-layer.  Every Snake is folded into a neighbouring conv launch:

  * a Snake whose input has no other consumer (inside a ResidualUnit, dac.py:31-34) is the EPILOGUE
    of the conv producing that input;
  * a Snake in front of a conv whose input IS also needed raw (the ResidualUnit's skip, :38-42) is
    emitted by the producer as a SECOND output: each producer writes y and snake(y, alpha_next).
    The consuming conv then stages a ready-made tensor by pure LDS-DMA -- the profile showed that any
    VALU work (sin!) in the staging waves starves the MFMA waves sharing their SIMD;
  * block-final Snakes (before the strided / transposed conv, before the last conv) have a single
    consumer, so only the pre-activated copy is written.

A ResidualUnit is therefore two launches of the MFMA conv kernel:
    conv k7 (dilated):  input = pre-activated x, epilogue bias + Snake(alpha2)
    conv k1:            epilogue bias + residual(x raw) [+ second output snake(., alpha_next)]
"""
from torch import nn

from . import autograd as A
from . import ops
from .wprep import cached_forward
from .layers import SConv1d, SConvTranspose1d, SLSTM, Snake1d


import os as _os
# channel counts that take the single-launch fp32 ResidualUnit kernel (FAC_FUSED_RU="" / "64" / "64,96": tuning switch)
FUSED_RU_CHANNELS = tuple(int(c) for c in _os.environ.get("FAC_FUSED_RU", "64").split(",") if c)
# (C = 128 is instantiated too but measured slower than k7 + wide-tile k1: 2.56 vs 2.38 ms at T = 24000)


class ResidualUnit(nn.Module):
    """dac/model/dac.py:25-42.  block = [Snake, SConv k7 dil d, Snake, SConv k1]."""

    def __init__(self, dim=16, dilation=1, causal=False):
        super().__init__()
        self.block = nn.Sequential(
            Snake1d(dim),
            SConv1d(dim, dim, kernel_size=7, dilation=dilation, causal=causal, norm="weight_norm"),
            Snake1d(dim),
            SConv1d(dim, dim, kernel_size=1, causal=causal, norm="weight_norm"),
        )

    @property
    def alpha_in(self):
        return self.block[0].flat()

    def run(self, x, x_act, alpha_next=None, want_raw=True):
        """x: raw input (skip path); x_act = snake(x, alpha_in) from the producer.
        Returns (y, snake(y, alpha_next)); y is None when want_raw is False."""
        b = self.block
        k7, k1 = b[1], b[3]
        # big batches take the split-bf16 k = 7 kernel + the streaming k = 1 kernel instead (same time for the k = 7 part,
        # the 1x1 tail at HBM speed: 1.05 vs 1.30 ms per unit at B = 32); the streaming kernel wants >= 8192 column blocks
        streaming = ops.BF16_SPLIT and x.shape[0] * ((x.shape[-1] + 31) // 32) >= 8192
        if k7.w.c_out in FUSED_RU_CHANNELS and x_act.shape[-1] == x.shape[-1] and not streaming:
            # whole unit in one launch: the 1x1 conv runs out of the k7 accumulators (conv1d_fused_ru.hip)
            return_pair = ops.conv1d(x_act, k7.w.packed(), k7.w.c_out, 7, bias=k7.w.bias, dilation=k7.dilation,
                                     alpha_out=b[2].flat(), res=x, w_k1=k1.w.packed(), bias_k1=k1.w.bias,
                                     alpha_y2=alpha_next, want_y=want_raw or alpha_next is None, causal=k7.causal)
            return return_pair if alpha_next is not None else (return_pair, None)
        h = b[1].run(x_act, alpha_out=b[2].flat())
        if h.shape[-1] != x.shape[-1]:  # non-causal trimming of :38-41 never triggers with SConv1d padding
            pad = (x.shape[-1] - h.shape[-1]) // 2
            x = x[..., pad:-pad].contiguous()
        if alpha_next is None:
            return b[3].run(h, res=x), None
        return b[3].run(h, res=x, alpha_y2=alpha_next, want_y=want_raw)

    def forward(self, x):
        if self.training:        # layer-by-layer with autograd (facodec_amd/autograd.py)
            b = self.block
            y = A.conv(b[3], A.snake(b[2], A.conv(b[1], A.snake(b[0], x))))
            return A.add(x, y)
        return self.run(x, ops.snake(x, self.alpha_in))[0]


def _run_units(units, x, x_act, alpha_after):
    """Chains ResidualUnits; the last one only emits the copy pre-activated with `alpha_after`."""
    for j, ru in enumerate(units):
        last = j == len(units) - 1
        nxt = alpha_after if last else units[j + 1].alpha_in
        x, x_act = ru.run(x, x_act, alpha_next=nxt, want_raw=not last)
    return x_act


class EncoderBlock(nn.Module):
    """dac/model/dac.py:45-66.  block = [RU d1, RU d3, RU d9, Snake, strided SConv k=2s]."""

    def __init__(self, dim=16, stride=1, causal=False):
        super().__init__()
        self.block = nn.Sequential(
            ResidualUnit(dim // 2, dilation=1, causal=causal),
            ResidualUnit(dim // 2, dilation=3, causal=causal),
            ResidualUnit(dim // 2, dilation=9, causal=causal),
            Snake1d(dim // 2),
            SConv1d(dim // 2, dim, kernel_size=2 * stride, stride=stride, causal=causal, norm="weight_norm"),
        )

    @property
    def alpha_in(self):
        return self.block[0].alpha_in

    def run(self, x, x_act, alpha_next=None):
        """Returns the strided conv's (y, snake(y, alpha_next)) (just y when alpha_next is None)."""
        b = self.block
        z_act = _run_units([b[0], b[1], b[2]], x, x_act, b[3].flat())
        return b[4].run(z_act, alpha_y2=alpha_next) if alpha_next is not None else (b[4].run(z_act), None)

    def forward(self, x):
        return self.run(x, ops.snake(x, self.alpha_in))[0]


class Encoder(nn.Module):
    """dac/model/dac.py:69-104: wave (B,1,T) -> latent (B, d_latent, ceil(T / prod(strides)))."""

    def __init__(self, d_model=64, strides=(2, 4, 8, 8), d_latent=64, causal=False, lstm=2):
        super().__init__()
        layers = [SConv1d(1, d_model, kernel_size=7, causal=causal, norm="weight_norm")]
        for s in strides:
            d_model *= 2
            layers.append(EncoderBlock(d_model, stride=s, causal=causal))
        self.use_lstm = lstm
        if lstm:
            layers.append(SLSTM(d_model, lstm))
        layers += [Snake1d(d_model), SConv1d(d_model, d_latent, kernel_size=3, causal=causal, norm="weight_norm")]
        self.block = nn.Sequential(*layers)
        self.enc_dim = d_model
        self.boundary_hook = None          # training only: callable(child module, activation entering it), see train.GeneratorStep

    def _forward_train(self, x):
        """Training mode: one autograd node per ResidualUnit (A.res_unit), every Snake but the one after the LSTM fused into a
        conv epilogue as a second output, no tensor with two autograd consumers."""
        mods = list(self.block)
        x = A.conv(mods[0], x)
        for m in mods[1:-2]:
            if self.boundary_hook is not None:
                self.boundary_hook(m, x)            # x enters top-level child m (train.py: progressive gradient exchange)
            if isinstance(m, EncoderBlock):
                b = m.block
                x, xa = A.snake_dual(x, b[0].block[0].alpha)
                for j in range(3):
                    nxt = b[j + 1].block[0].alpha if j < 2 else b[3].alpha      # next unit's first Snake / the block's Snake
                    x, xa = A.res_unit(b[j], x, xa, nxt)
                x = A.conv(b[4], xa)                                            # strided conv on the pre-activated tensor
            else:
                x = A.slstm(m, x)
        return A.conv(mods[-1], A.snake(mods[-2], x))

    @cached_forward
    def forward(self, x):
        if self.training:
            return self._forward_train(x)
        mods = list(self.block)
        blocks = [m for m in mods if isinstance(m, EncoderBlock)]
        final_alpha = mods[-2].flat()
        x, x_act = mods[0].run(x, alpha_y2=blocks[0].alpha_in)
        for i, blk in enumerate(blocks):
            if i + 1 < len(blocks):
                nxt = blocks[i + 1].alpha_in
            else:
                nxt = None if self.use_lstm else final_alpha   # the LSTM wants the raw tensor
            x, x_act = blk.run(x, x_act, alpha_next=nxt)
        if self.use_lstm:
            x_act = mods[-3](x, alpha_out=final_alpha)           # SLSTM + skip, Snake on the way out
        return mods[-1].run(x_act)


class DecoderBlock(nn.Module):
    """dac/model/dac.py:107-128.  block = [Snake, SConvTranspose k=2s, RU d1, RU d3, RU d9]."""

    def __init__(self, input_dim=16, output_dim=8, stride=1, causal=False):
        super().__init__()
        self.block = nn.Sequential(
            Snake1d(input_dim),
            SConvTranspose1d(input_dim, output_dim, kernel_size=2 * stride, stride=stride, causal=causal,
                             norm="weight_norm"),
            ResidualUnit(output_dim, dilation=1, causal=causal),
            ResidualUnit(output_dim, dilation=3, causal=causal),
            ResidualUnit(output_dim, dilation=9, causal=causal),
        )

    @property
    def alpha_in(self):
        return self.block[0].flat()

    def run(self, x_act, alpha_after):
        """x_act = snake(x, alpha_in); returns the block output pre-activated with `alpha_after`."""
        b = self.block
        y, y_act = b[1].run(x_act, alpha_y2=b[2].alpha_in)
        return _run_units([b[2], b[3], b[4]], y, y_act, alpha_after)

    def forward(self, x):
        raise NotImplementedError("DecoderBlock is executed through Decoder.forward's fused plan")


class Decoder(nn.Module):
    """dac/model/dac.py:131-165: latent (B, input_channel, F) -> wave (B, d_out, F * prod(rates))."""

    def __init__(self, input_channel, channels, rates, d_out=1, causal=False, lstm=2):
        super().__init__()
        layers = [SConv1d(input_channel, channels, kernel_size=7, causal=causal, norm="weight_norm")]
        self.use_lstm = lstm
        if lstm:
            layers.append(SLSTM(channels, num_layers=lstm))
        out_dim = channels
        for i, s in enumerate(rates):
            in_dim, out_dim = channels // 2 ** i, channels // 2 ** (i + 1)
            layers.append(DecoderBlock(in_dim, out_dim, s, causal=causal))
        layers += [Snake1d(out_dim), SConv1d(out_dim, d_out, kernel_size=7, causal=causal, norm="weight_norm"),
                   nn.Tanh()]
        self.model = nn.Sequential(*layers)
        self.boundary_hook = None

    def _forward_train(self, x):
        """Training mode: see Encoder._forward_train.  The last ResidualUnit of a block emits the copy pre-activated with the
        NEXT block's (or the output conv's) Snake."""
        mods = list(self.model)
        blocks = [m for m in mods if isinstance(m, DecoderBlock)]
        x = A.conv(mods[0], x)
        xa = None
        for m in mods[1:-3]:
            if self.boundary_hook is not None:
                self.boundary_hook(m, x)            # x enters top-level child m (for a block after the first: together with xa)
            if isinstance(m, DecoderBlock):
                b = m.block
                if xa is None:
                    xa = A.snake(b[0], x)                                       # first block: after the LSTM (or the input conv)
                x = A.conv_tr(b[1], xa)
                x, xa = A.snake_dual(x, b[2].block[0].alpha)
                i = blocks.index(m)
                after = blocks[i + 1].block[0].alpha if i + 1 < len(blocks) else mods[-3].alpha
                for j in range(3):
                    nxt = b[j + 3].block[0].alpha if j < 2 else after
                    x, xa = A.res_unit(b[j + 2], x, xa, nxt)
            else:
                x = A.slstm(m, x)
        return A.conv(mods[-2], xa, act=ops.ACT_TANH)

    @cached_forward
    def forward(self, x):
        if self.training:
            return self._forward_train(x)
        mods = list(self.model)
        blocks = [m for m in mods if isinstance(m, DecoderBlock)]
        final_alpha = mods[-3].flat()
        if self.use_lstm:
            x = mods[0].run(x)
            x_act = mods[1](x, alpha_out=blocks[0].alpha_in)
        else:
            _, x_act = mods[0].run(x, alpha_y2=blocks[0].alpha_in, want_y=False)
        for i, blk in enumerate(blocks):
            nxt = blocks[i + 1].alpha_in if i + 1 < len(blocks) else final_alpha
            x_act = blk.run(x_act, nxt)
        return mods[-2].run(x_act, act=ops.ACT_TANH)
