"""Training-mode predictor heads (modules/quantize.py:92-125 ResidualUnit / CNNLSTM, :456-606 FApredictors.forward_v2,
gradient_reversal.py) with HIP forward and backward.  The heads' LOSSES need external targets (phonemes from a CTC
model, F0 / UV, speaker ids: train.py:314-356) and are not part of this build; what is here are the differentiable
heads themselves, so a caller that has targets can train them."""
import os

import torch
from torch.autograd import Function

from . import _lib, ops
from . import autograd as A
from . import autograd_quant as AQ


class _AASnakeBeta(Function):
    """Activation1d(SnakeBeta) (alias_free_torch/act.py:24-29): 2x up, x + sin^2(x e^a)/(e^b + 1e-9), 2x down."""

    @staticmethod
    def forward(ctx, x, alpha, beta, filt):
        ctx.save_for_backward(x, alpha, beta, filt)
        return ops.aa_snakebeta(x.detach(), alpha.detach(), beta.detach(), filt)

    @staticmethod
    def backward(ctx, dy):
        x, alpha, beta, filt = ctx.saved_tensors
        B, c, T = x.shape
        xd = x.detach().contiguous()
        dx = torch.empty_like(xd)
        da, db = torch.empty(c, device=x.device), torch.empty(c, device=x.device)
        scratch = torch.empty(2 * B * c * ((T + 255) // 256), device=x.device)
        _lib.check(_lib.load().fac_aa_snakebeta_bwd(ops._ptr(xd), ops._ptr(alpha.detach()), ops._ptr(beta.detach()), ops._ptr(filt),
                                                    ops._ptr(dy.contiguous()), ops._ptr(dx), ops._ptr(da), ops._ptr(db),
                                                    ops._ptr(scratch), B, c, T, ops._stream()), "fac_aa_snakebeta_bwd")
        return dx, da, db, None


class _GradReverse(Function):
    """gradient_reversal.py:11-27: identity forward, -alpha * grad backward."""

    @staticmethod
    def forward(ctx, x, alpha):
        ctx.alpha = alpha
        return x.view_as(x)

    @staticmethod
    def backward(ctx, d):
        return ops.rows_fma(d.contiguous(), torch.full((d.shape[0],), -float(ctx.alpha), device=d.device)), None


def activation(m, x):
    """Activation1d module `m` (act.alpha / act.beta log-scale, upsample.filter)."""
    return _AASnakeBeta.apply(x, m.act.alpha, m.act.beta, m.upsample.filter.reshape(-1).contiguous())


def _wnconv(w, x, k, dilation):
    """weight-normed Conv1d with zero 'same' padding (modules/quantize.py:96-101)."""
    return A._Conv.apply(x, w.weight_v, w.weight_g, w.bias, (k, 1, dilation, ops.PAD_ZERO, False, ops.ACT_NONE))


def residual_unit(m, x):
    b = m.block
    y = _wnconv(b[1], activation(b[0], x), 7, m.dilation)
    y = _wnconv(b[3], activation(b[2], y), 1, 1)
    return A.add(x, y)


def cnnlstm(m, x):
    """CNNLSTM.forward (modules/quantize.py:119-125) -> list of per-head outputs ((B, T, out) or (B, out))."""
    for i in range(3):
        x = residual_unit(m.model[i], x)
    x = activation(m.model[3], x)
    if m.global_pred:
        pooled = AQ._MaskedMean.apply(x, None)
        return [A.linear(h, pooled) for h in m.heads]
    outs = []
    for h in m.heads:
        y = A._Conv.apply(x, h.weight.unsqueeze(-1), None, h.bias, (1, 1, 1, ops.PAD_ZERO, True, ops.ACT_NONE))
        outs.append(y.transpose(1, 2))
    return outs


PRED_STREAMS = int(os.environ.get("FAC_PRED_STREAMS", "3"))


def predictors(m, quantized, timbre):
    """FApredictors.forward_v2 (modules/quantize.py:564-606), timbre_norm configuration."""
    prosody, content, residual = quantized
    rev = lambda t: _GradReverse.apply(t, 1.0)   # noqa: E731

    def total(parts, like):
        acc = None
        for p in parts:
            acc = p if acc is None else A.add(acc, p)
        return acc if acc is not None else torch.zeros_like(like)

    rin_f0 = total(([content] if m.use_gr_content_f0 else []) + ([residual] if m.use_gr_residual_f0 else []), prosody)
    rin_ph = total(([prosody] if m.use_gr_prosody_phone else []) + ([residual] if m.use_gr_residual_phone else []), content)
    # four (five) independent heads, 1 024 channels x (B x 160) columns each: 160 workgroups per launch on 256 CUs -- run them
    # side by side (ops.run_chains; FAC_PRED_STREAMS=1: one after the other)
    chains = [lambda: cnnlstm(m.phone_predictor, content), lambda: cnnlstm(m.f0_predictor, prosody),
              lambda: cnnlstm(m.rev_f0_predictor[1], rev(rin_f0)), lambda: cnnlstm(m.rev_content_predictor[1], rev(rin_ph))]
    if m.use_gr_x_timbre:
        chains.append(lambda: cnnlstm(m.rev_timbre_predictor[1], rev(A.add(A.add(prosody, content), residual))))
    res = ops.run_chains(chains, content.device, PRED_STREAMS, inputs=[prosody, content, residual, rin_f0, rin_ph, timbre])
    content_pred = res[0][0]
    f0_pred, uv_pred = res[1]
    rev_f0_pred, rev_uv_pred = res[2]
    rev_content_pred = res[3][0]
    x_spk_pred = res[4][0] if m.use_gr_x_timbre else None
    spk_pred = A.linear(m.timbre_predictor, timbre)
    preds = {"f0": f0_pred, "uv": uv_pred, "content": content_pred, "timbre": spk_pred}
    rev_preds = {"rev_f0": rev_f0_pred, "rev_uv": rev_uv_pred, "rev_content": rev_content_pred, "x_timbre": x_spk_pred}
    return preds, rev_preds
