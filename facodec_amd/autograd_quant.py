"""Training-mode WaveNet (modules/wavenet.py:138-166) and StyleEncoder (modules/style_encoder.py:63-91,
modules/attentions.py:158-199) with HIP forward and backward: the side branches of FAquantizer.forward_v2.
Dropout masks are drawn with torch's generator (plumbing) and applied / back-propagated by `fac_mul_scaled`."""
import ctypes as C

import torch
from torch.autograd import Function

from . import _lib, ops
from . import autograd as A


def _call(name, *args):
    _lib.check(getattr(_lib.load(), name)(*args, ops._stream()), name)


def _p(t):
    return ops._ptr(t)


class _Gate(Function):
    @staticmethod
    def forward(ctx, a):
        ctx.save_for_backward(a)
        return ops.gate_tanh_sigmoid(a.detach())

    @staticmethod
    def backward(ctx, d):
        (a,) = ctx.saved_tensors
        B, c2, T = a.shape
        da = torch.empty_like(a)
        _call("fac_gate_bwd", _p(a.detach()), _p(d.contiguous()), _p(da), B, c2 // 2, T)
        return da


class _Mish(Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        y = torch.empty_like(x)
        _call("fac_mish_fwd", _p(x.detach().contiguous()), _p(y), x.numel())
        return y

    @staticmethod
    def backward(ctx, d):
        (x,) = ctx.saved_tensors
        dx = torch.empty_like(x)
        _call("fac_mish_bwd", _p(x.detach().contiguous()), _p(d.contiguous()), _p(dx), x.numel())
        return dx


class _GLU(Function):
    """res + a[:, :C] * sigmoid(a[:, C:])."""

    @staticmethod
    def forward(ctx, a, res):
        ctx.save_for_backward(a)
        return ops.glu_residual(a.detach(), res.detach())

    @staticmethod
    def backward(ctx, d):
        (a,) = ctx.saved_tensors
        B, c2, T = a.shape
        d = d.contiguous()
        da = torch.empty_like(a)
        _call("fac_glu_bwd", _p(a.detach()), _p(d), _p(da), B, c2 // 2, T)
        return da, d


class _Mul(Function):
    """x * m * scale with a constant m of x's shape (dropout keep-mask)."""

    @staticmethod
    def forward(ctx, x, m, scale):
        ctx.save_for_backward(m)
        ctx.scale = scale
        y = torch.empty_like(x)
        _call("fac_mul_scaled", _p(x.detach().contiguous()), _p(m), _p(y), C.c_float(scale), x.numel())
        return y

    @staticmethod
    def backward(ctx, d):
        (m,) = ctx.saved_tensors
        dx = torch.empty_like(d)
        _call("fac_mul_scaled", _p(d.contiguous()), _p(m), _p(dx), C.c_float(ctx.scale), d.numel())
        return dx, None, None


class _MulMask(Function):
    """x (B, C, T) * mask (B, T)."""

    @staticmethod
    def forward(ctx, x, mask):
        ctx.save_for_backward(mask)
        return ops.mul_mask_(x.detach().clone(), mask)

    @staticmethod
    def backward(ctx, d):
        (mask,) = ctx.saved_tensors
        return ops.mul_mask_(d.contiguous().clone(), mask), None


class _MaskedMean(Function):
    @staticmethod
    def forward(ctx, x, mask):
        ctx.mask, ctx.shape = mask, x.shape
        return ops.masked_mean(x.detach(), mask)

    @staticmethod
    def backward(ctx, d):
        B, c, T = ctx.shape
        dx = torch.empty(B, c, T, device=d.device)
        _call("fac_masked_mean_bwd", _p(d.contiguous()), _p(ctx.mask), _p(dx), B, c, T)
        return dx, None


class _Attention(Function):
    """softmax(q k^T / sqrt(dk), mask) [dropout] v with the probability matrix materialised."""

    @staticmethod
    def forward(ctx, q, k, v, mask, n_heads, keep, scale):
        B, c, T = q.shape
        dk = c // n_heads
        qd, kd, vd = q.detach().contiguous(), k.detach().contiguous(), v.detach().contiguous()
        P = torch.empty(B, n_heads, T, T, device=q.device)
        _call("fac_attention_probs", _p(qd), _p(kd), _p(mask), _p(P), B, n_heads, dk, T)
        P_used = P
        if keep is not None:
            P_used = torch.empty_like(P)
            _call("fac_mul_scaled", _p(P), _p(keep), _p(P_used), C.c_float(scale), P.numel())
        o = torch.empty_like(qd)
        _call("fac_attention_pv", _p(P_used), _p(vd), _p(o), B, n_heads, dk, T)
        ctx.save_for_backward(qd, kd, vd, P, P_used, keep)
        ctx.cfg = (mask, n_heads, scale)
        return o

    @staticmethod
    def backward(ctx, dO):
        q, k, v, P, P_used, keep = ctx.saved_tensors
        mask, H, scale = ctx.cfg
        B, c, T = q.shape
        dk = c // H
        dO = dO.contiguous()
        dv, dP = torch.empty_like(v), torch.empty_like(P)
        _call("fac_attention_bwd_pv", _p(P_used), _p(v), _p(dO), _p(dv), _p(dP), B, H, dk, T)
        if keep is not None:
            _call("fac_mul_scaled", _p(dP), _p(keep), _p(dP), C.c_float(scale), dP.numel())
        dq, dkk = torch.empty_like(q), torch.empty_like(k)
        _call("fac_attention_bwd_qk", _p(P), _p(dP), _p(q), _p(k), _p(mask), _p(dq), _p(dkk), B, H, dk, T)
        return dq, dkk, dv, None, None, None, None


class _ResSkip(Function):
    """WaveNet layer tail (modules/wavenet.py:159-165): x += rs[:, :H]; out += rs[:, H:] (last layer: out += rs)."""

    @staticmethod
    def forward(ctx, x, out, rs, last):
        ctx.last, ctx.H = last, x.shape[1]
        xn, on = x.detach().clone(), out.detach().clone()
        ops.wn_res_skip_(rs.detach().contiguous(), xn, on, last)
        return xn, on

    @staticmethod
    def backward(ctx, dx, dout):
        if ctx.last:
            return None, dout, dout, None
        return dx, dout, torch.cat([dx, dout], dim=1), None


def dropout(x, p, enabled=True):
    if not enabled or p <= 0.0:
        return x
    keep = torch.empty_like(x).bernoulli_(1.0 - p)
    return _Mul.apply(x, keep, 1.0 / (1.0 - p))


def plain_conv(m, x, act=ops.ACT_NONE):
    """_PlainConv (nn.Conv1d weight / bias, zero 'same' padding) with autograd."""
    return A._Conv.apply(x, m.weight, None, m.bias, (m.k, 1, 1, ops.PAD_ZERO, False, act))


def wavenet(m, x, p_dropout=0.2, use_dropout=True):
    """WN.forward in training mode, g = None, mask of ones."""
    out = torch.zeros_like(x)
    for i in range(m.n_layers):
        acts = dropout(_Gate.apply(A.conv(m.in_layers[i], x)), p_dropout, use_dropout)
        rs = A.conv(m.res_skip_layers[i], acts)
        x, out = _ResSkip.apply(x, out, rs, i == m.n_layers - 1)
    return out


def style_encoder(m, mel, mask=None, p_dropout=0.1, use_dropout=True):
    """StyleEncoder.forward in training mode.  mel (B, 80, T) (no gradient), mask (B, T) float or None."""
    x = dropout(_Mish.apply(plain_conv(m.spectral["0"], mel)), p_dropout, use_dropout)
    x = dropout(_Mish.apply(plain_conv(m.spectral["3"], x)), p_dropout, use_dropout)
    if mask is not None:
        x = _MulMask.apply(x, mask)
    for glu in m.temporal:
        a = plain_conv(glu.conv1, x)
        if use_dropout and p_dropout > 0:
            gated = dropout(_GLU.apply(a, torch.zeros_like(x)), p_dropout, True)
            x = A.add(x, gated)
        else:
            x = _GLU.apply(a, x)
    if mask is not None:
        x = _MulMask.apply(x, mask)
    att = m.slf_attn
    q, k, v = plain_conv(att.conv_q, x), plain_conv(att.conv_k, x), plain_conv(att.conv_v, x)
    keep, scale = None, 1.0
    if use_dropout and p_dropout > 0:
        B, c, T = q.shape
        keep = torch.empty(B, att.n_heads, T, T, device=q.device).bernoulli_(1.0 - p_dropout)
        scale = 1.0 / (1.0 - p_dropout)
    o = plain_conv(att.conv_o, _Attention.apply(q, k, v, mask, att.n_heads, keep, scale))
    x = A.add(x, dropout(o, p_dropout, use_dropout))
    x = plain_conv(m.fc, x)
    return _MaskedMean.apply(x, mask)
