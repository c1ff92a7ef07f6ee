"""Predictor heads of the training step, forward pass (reference: modules/quantize.py:29-125 SnakeBeta /
ResidualUnit / CNNLSTM, :456-606 FApredictors.forward_v2; alias_free_torch/* Activation1d;
gradient_reversal.py).  State-dict keys follow the reference module tree (checked against a dump of the
real reference).

Launch plan of a CNNLSTM: 3 x [anti-aliased SnakeBeta (one fused kernel: 2x up, activation, 2x down) ->
conv k7 dilated, zero 'same' padding (MFMA conv kernel) -> anti-aliased SnakeBeta -> conv k1 with the
residual add in its epilogue] -> anti-aliased SnakeBeta -> Linear heads as 1x1 convs on the (B, C, T)
layout (transposed back to (B, T, out) for the caller).
"""
import torch
from torch import nn

from . import dsp, ops
from .layers import ConvWeights
from .quantize import _Linear


class SnakeBeta(nn.Module):
    """modules/quantize.py:29-90 with alpha_logscale=True: alpha, beta (C,) stored in log scale."""

    def __init__(self, in_features, alpha=1.0, alpha_trainable=True, alpha_logscale=True):
        super().__init__()
        if not alpha_logscale:
            raise NotImplementedError("only the log-scale SnakeBeta used by the predictors is built")
        self.alpha = nn.Parameter(torch.zeros(in_features) * alpha)
        self.beta = nn.Parameter(torch.zeros(in_features) * alpha)


class _Filter(nn.Module):
    def __init__(self):
        super().__init__()
        self.register_buffer("filter", dsp.kaiser_sinc_filter1d(0.25, 0.3, 12))


class _Down(nn.Module):
    def __init__(self):
        super().__init__()
        self.lowpass = _Filter()


class Activation1d(nn.Module):
    """alias_free_torch/act.py:7-29 (ratio 2, 12-tap Kaiser-sinc).  Keys: act.{alpha,beta},
    upsample.filter, downsample.lowpass.filter."""

    def __init__(self, activation):
        super().__init__()
        self.act = activation
        self.upsample = _Filter()
        self.downsample = _Down()

    def forward(self, x):
        return ops.aa_snakebeta(x, self.act.alpha.detach(), self.act.beta.detach(),
                                self.upsample.filter.reshape(-1))


class _WNConv(ConvWeights):
    """weight_norm(nn.Conv1d) with keys weight_g / weight_v / bias directly on the block entry."""


class ResidualUnit(nn.Module):
    """modules/quantize.py:92-104: x + [act -> WNConv k7 dil d (zero pad 3d) -> act -> WNConv k1]."""

    def __init__(self, dim=16, dilation=1):
        super().__init__()
        self.dilation = dilation
        self.block = nn.Sequential(
            Activation1d(SnakeBeta(dim)), _WNConv(dim, dim, 7, weight_norm=True),
            Activation1d(SnakeBeta(dim)), _WNConv(dim, dim, 1, weight_norm=True))

    def forward(self, x):
        b, d, T = self.block, self.dilation, x.shape[-1]
        y = b[0](x)
        y = ops.conv1d(y, b[1].packed(), b[1].c_out, 7, bias=b[1].bias, dilation=d, pad_left=3 * d,
                       pad_mode=ops.PAD_ZERO, t_out=T)
        y = b[2](y)
        return ops.conv1d(y, b[3].packed(), b[3].c_out, 1, bias=b[3].bias, pad_left=0, pad_mode=ops.PAD_ZERO,
                          t_out=T, res=x)


class CNNLSTM(nn.Module):
    """modules/quantize.py:106-125 (despite the name there is no LSTM in it)."""

    def __init__(self, indim, outdim, head, global_pred=False):
        super().__init__()
        self.global_pred = global_pred
        # index 4 of the reference Sequential is an einops Rearrange (no parameters)
        self.model = nn.Sequential(ResidualUnit(indim, dilation=1), ResidualUnit(indim, dilation=2),
                                   ResidualUnit(indim, dilation=3), Activation1d(SnakeBeta(indim)))
        self.heads = nn.ModuleList([_Linear(indim, outdim) for _ in range(head)])

    def forward(self, x):
        x = self.model(x)                                   # (B, C, T)
        if self.global_pred:
            pooled = ops.masked_mean(x, None)               # mean over time -> (B, C)
            return [h(pooled) for h in self.heads]
        outs = []
        for h in self.heads:
            y = ops.conv1d(x, ops.pack_conv_weight(h.weight.detach()), h.weight.shape[0], 1, bias=h.bias.detach(),
                           pad_left=0, pad_mode=ops.PAD_ZERO, t_out=x.shape[-1])
            outs.append(y.transpose(1, 2).contiguous())     # (B, T, out) like "b c t -> b t c" + Linear
        return outs


class GradientReversal(nn.Module):
    """gradient_reversal.py:29-35: identity in the forward pass (the -alpha gradient comes with backward)."""

    def __init__(self, alpha):
        super().__init__()
        self.alpha = alpha

    def forward(self, x):
        return x


class FApredictors(nn.Module):
    """modules/quantize.py:456-606 for timbre_norm=True: `forward` is forward_v2(quantized, timbre)."""

    def __init__(self, in_dim=1024, use_gr_content_f0=False, use_gr_prosody_phone=False, use_gr_residual_f0=False,
                 use_gr_residual_phone=False, use_gr_timbre_content=True, use_gr_timbre_prosody=True,
                 use_gr_x_timbre=False, norm_f0=True, timbre_norm=False, use_gr_content_global_f0=False):
        super().__init__()
        if not timbre_norm:
            raise NotImplementedError("only the shipped configuration (timbre_norm=True) is built")
        self.f0_predictor = CNNLSTM(in_dim, 1, 2)
        self.phone_predictor = CNNLSTM(in_dim, 1024, 1)
        self.timbre_predictor = _Linear(in_dim, 20000)
        self.use_gr_content_f0, self.use_gr_prosody_phone = use_gr_content_f0, use_gr_prosody_phone
        self.use_gr_residual_f0, self.use_gr_residual_phone = use_gr_residual_f0, use_gr_residual_phone
        self.use_gr_x_timbre = use_gr_x_timbre
        self.rev_f0_predictor = nn.Sequential(GradientReversal(1.0), CNNLSTM(in_dim, 1, 2))
        self.rev_content_predictor = nn.Sequential(GradientReversal(1.0), CNNLSTM(in_dim, 1024, 1))
        self.rev_timbre_predictor = nn.Sequential(GradientReversal(1.0), CNNLSTM(in_dim, 20000, 1, global_pred=True))
        self.global_f0_predictor = _Linear(in_dim, 1)                       # never used in forward (:500)
        if use_gr_content_global_f0:
            self.rev_global_f0_predictor = nn.Sequential(GradientReversal(1.0), CNNLSTM(in_dim, 1, 1, global_pred=True))

    @staticmethod
    def _sum(parts, like):
        acc = None
        for p in parts:
            acc = p if acc is None else ops.add(acc, p)
        return acc if acc is not None else torch.zeros_like(like)

    def forward(self, quantized, timbre):
        if self.training:            # HIP forward + backward (facodec_amd/autograd_pred.py)
            from . import autograd_pred
            return autograd_pred.predictors(self, quantized, timbre)
        prosody, content, residual = quantized[0], quantized[1], quantized[2]
        content_pred = self.phone_predictor(content)[0]
        spk_pred = self.timbre_predictor(timbre)
        f0_pred, uv_pred = self.f0_predictor(prosody)
        rev_in = self._sum(([content] if self.use_gr_content_f0 else []) + ([residual] if self.use_gr_residual_f0 else []), prosody)
        rev_f0_pred, rev_uv_pred = self.rev_f0_predictor(rev_in)
        rev_in = self._sum(([prosody] if self.use_gr_prosody_phone else []) + ([residual] if self.use_gr_residual_phone else []), content)
        rev_content_pred = self.rev_content_predictor(rev_in)[0]
        x_spk_pred = None
        if self.use_gr_x_timbre:
            x_spk_pred = self.rev_timbre_predictor(ops.add(ops.add(prosody, content), residual))[0]
        preds = {"f0": f0_pred, "uv": uv_pred, "content": content_pred, "timbre": spk_pred}
        rev_preds = {"rev_f0": rev_f0_pred, "rev_uv": rev_uv_pred, "rev_content": rev_content_pred, "x_timbre": x_spk_pred}
        return preds, rev_preds

    forward_v2 = forward
