"""Encoder / Decoder of the codec (reference: dac/model/dac.py:25-165) as fused launch plans.

Module tree = the reference's (same state-dict keys: `block.N...` / `model.N...`), but nothing is
executed layer-by-layer: each ResidualUnit is two launches of the MFMA conv kernel --
    conv k7 (dilated):  Snake(alpha1) prologue, bias + Snake(alpha2) epilogue
    conv k1:            bias + residual-add epilogue
-- every other Snake is the prologue of the conv that consumes it, the final tanh is an epilogue,
so no activation ever makes a separate trip through HBM.
"""
from torch import nn

from . import ops
from .layers import SConv1d, SConvTranspose1d, SLSTM, Snake1d


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

    def forward(self, x):
        b = self.block
        h = b[1].run(x, alpha_in=b[0].flat(), alpha_out=b[2].flat())
        if h.shape[-1] != x.shape[-1]:  # non-causal trimming of :38-41 never triggers with SConv1d padding
            pad = (x.shape[-1] - h.shape[-1]) // 2
            x = x[..., pad:-pad].contiguous()
        return b[3].run(h, res=x)


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

    def forward(self, x):
        b = self.block
        x = b[2](b[1](b[0](x)))
        return b[4].run(x, alpha_in=b[3].flat())


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

    def forward(self, x):
        mods = list(self.block)
        x = mods[0].run(x)
        for m in mods[1:-2]:
            x = m(x)
        return mods[-1].run(x, alpha_in=mods[-2].flat())


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

    def forward(self, x):
        b = self.block
        x = b[1].run(x, alpha_in=b[0].flat())
        return b[4](b[3](b[2](x)))


class Decoder(nn.Module):
    """dac/model/dac.py:131-165: latent (B, input_channel, F) -> wave (B, d_out, F * prod(rates))."""

    def __init__(self, input_channel, channels, rates, d_out=1, causal=False, lstm=2):
        super().__init__()
        layers = [SConv1d(input_channel, channels, kernel_size=7, causal=causal, norm="weight_norm")]
        if lstm:
            layers.append(SLSTM(channels, num_layers=lstm))
        out_dim = channels
        for i, s in enumerate(rates):
            in_dim, out_dim = channels // 2 ** i, channels // 2 ** (i + 1)
            layers.append(DecoderBlock(in_dim, out_dim, s, causal=causal))
        layers += [Snake1d(out_dim), SConv1d(out_dim, d_out, kernel_size=7, causal=causal, norm="weight_norm"),
                   nn.Tanh()]
        self.model = nn.Sequential(*layers)

    def forward(self, x):
        mods = list(self.model)
        x = mods[0].run(x)
        for m in mods[1:-3]:
            x = m(x)
        return mods[-2].run(x, alpha_in=mods[-3].flat(), act=ops.ACT_TANH)
