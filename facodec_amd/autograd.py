"""Training-mode execution of the encoder / decoder stacks: torch.autograd.Function wrappers whose forward AND
backward run on the HIP C ABI (first part of BASELINE.json configs[2]; the reference relies on ATen autograd for
dac/model/dac.py + dac/model/encodec.py).

Layer by layer (no cross-layer fusion yet: every Function keeps what its backward needs):
  conv        y = SConv1d(x)                  bwd: fac_conv1d_fwd on flipped weights + fac_pad_fold_bwd,
                                                   fac_conv1d_bwd_weight, fac_weight_norm_bwd, fac_bias_grad
  conv_tr     y = SConvTranspose1d(x)         bwd: strided forward conv, weight-gradient kernel with swapped roles
  snake       y = x + sin^2(a x)/(a + 1e-9)   bwd: fac_snake_bwd
  tanh                                        bwd: fac_tanh_bwd
  lstm        SLSTM (dac/model/encodec.py:282-288): training forward stores gates / cell states; BPTT = per step one
              W_hh^T GEMV (the split-reduction conv kernel) + fac_lstm_gate_bwd, then three big GEMMs for dW_hh,
              dW_ih and the input gradient.
torch only allocates, views and (for the reversed recurrence bookkeeping) slices; sums of two gradients use fac_add.
"""
import torch
from torch.autograd import Function

from . import ops


def _wn(w):
    """(v, g) of a ConvWeights holder (g None for a plain weight)."""
    return (w.weight_v, w.weight_g) if w.weight_norm else (w.weight, None)


class _Conv(Function):
    @staticmethod
    def forward(ctx, x, v, g, bias, cfg):
        k, stride, dilation, pad_mode, causal, act = cfg
        vd, gd = v.detach(), (g.detach() if g is not None else None)
        y = ops.conv1d(x.detach(), ops.pack_conv_weight(vd, gd), v.shape[0], k, bias=bias.detach() if bias is not None else None,
                       stride=stride, dilation=dilation, pad_mode=pad_mode, causal=causal, act=act)
        ctx.cfg = cfg
        ctx.save_for_backward(x, v, g, bias, y if act == ops.ACT_TANH else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, v, g, bias, y = ctx.saved_tensors
        k, stride, dilation, pad_mode, causal, act = ctx.cfg
        dy = dy.contiguous()
        if act == ops.ACT_TANH:
            dy = ops.tanh_bwd(y, dy)
        vd, gd = v.detach(), (g.detach() if g is not None else None)
        dx = ops.conv1d_bwd_data(dy, vd, gd, x.shape[-1], stride=stride, dilation=dilation, pad_mode=pad_mode, causal=causal) \
            if ctx.needs_input_grad[0] else None
        dw = ops.conv1d_bwd_weight(x.detach(), dy, k, stride=stride, dilation=dilation, pad_mode=pad_mode, causal=causal)
        if g is not None:
            dv, dg = ops.weight_norm_bwd(vd, gd, dw)
        else:
            dv, dg = dw, None
        db = ops.bias_grad(dy) if bias is not None else None
        return dx, dv, dg, db, None


class _ConvTr(Function):
    @staticmethod
    def forward(ctx, x, v, g, bias, stride):
        vd, gd = v.detach(), (g.detach() if g is not None else None)
        y = ops.conv_transpose1d(x.detach(), ops.pack_convtr_weight(vd, gd, stride), v.shape[1], stride,
                                 bias=bias.detach() if bias is not None else None)
        ctx.stride = stride
        ctx.save_for_backward(x, v, g, bias)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, v, g, bias = ctx.saved_tensors
        dy = dy.contiguous()
        vd, gd = v.detach(), (g.detach() if g is not None else None)
        dx, dw = ops.conv_transpose1d_bwd(x.detach(), dy, vd, gd, ctx.stride)
        if g is not None:
            dv, dg = ops.weight_norm_bwd(vd, gd, dw)
        else:
            dv, dg = dw, None
        return dx, dv, dg, (ops.bias_grad(dy) if bias is not None else None), None


class _Snake(Function):
    @staticmethod
    def forward(ctx, x, alpha):
        ctx.save_for_backward(x, alpha)
        return ops.snake(x.detach(), alpha.detach().reshape(-1))

    @staticmethod
    def backward(ctx, dy):
        x, alpha = ctx.saved_tensors
        dx, da = ops.snake_bwd(x.detach(), alpha.detach().reshape(-1), dy.contiguous())
        return dx, da.reshape(alpha.shape)


class _Add(Function):
    @staticmethod
    def forward(ctx, a, b):
        return ops.add(a.detach(), b.detach())

    @staticmethod
    def backward(ctx, dy):
        return dy, dy


class _LSTM(Function):
    """SLSTM with skip; parameters flat per layer: (w_ih, w_hh, b_ih, b_hh) * L."""

    @staticmethod
    def forward(ctx, x, skip, *params):
        L = len(params) // 4
        B, H, T = x.shape
        inp = ops.lstm_to_time_major(x.detach())                      # (H, T, BP)
        BP = inp.shape[2]
        saved = []
        for l in range(L):
            w_ih, w_hh, b_ih, b_hh = (p.detach() for p in params[4 * l: 4 * l + 4])
            pre = ops.conv1d(inp.view(1, H, T * BP), ops.pack_conv_weight(w_ih), 4 * H, 1, bias=ops.add(b_ih, b_hh),
                             pad_left=0, t_out=T * BP, pad_mode=ops.PAD_ZERO)
            gates = torch.empty(4 * H, T, BP, device=x.device)
            cs = torch.empty(H, T, BP, device=x.device)
            yT = ops.lstm_layer(pre.view(4 * H, T, BP), ops.pack_lstm_whh(w_hh), H, save=(gates, cs))
            saved.append((inp, yT, gates, cs))
            inp = yT
        ctx.saved = saved
        ctx.skip, ctx.dims = skip, (B, H, T, BP)
        ctx.save_for_backward(x, *params)
        return ops.lstm_from_time_major(inp, x.detach() if skip else None, B)

    @staticmethod
    def backward(ctx, dy):
        x, *params = ctx.saved_tensors
        B, H, T, BP = ctx.dims
        L = len(params) // 4
        dy = dy.contiguous()
        d_out = ops.lstm_to_time_major(dy)                            # gradient w.r.t. the top layer's h sequence
        grads = [None] * (4 * L)
        for l in reversed(range(L)):
            w_ih, w_hh, _, _ = (p.detach() for p in params[4 * l: 4 * l + 4])
            inp, yT, gates, cs = ctx.saved[l]
            dgates = torch.empty(4 * H, T, BP, device=dy.device)
            dc = torch.zeros(H, BP, device=dy.device)
            whh_t = ops.pack_conv_weight(w_hh.t().contiguous().unsqueeze(-1))        # (H out, 4H in, 1)
            rec = None
            for t in reversed(range(T)):
                ops.lstm_gate_bwd(d_out[:, t], rec, gates[:, t], cs[:, t], cs[:, t - 1] if t > 0 else None, dc,
                                  dgates[:, t], H, BP, T * BP, first=(t == T - 1))
                if t > 0:   # W_hh^T dgates_t feeds dh_{t-1}
                    rec = ops.conv1d(dgates[:, t].unsqueeze(0), whh_t, H, 1, pad_left=0, pad_mode=ops.PAD_ZERO, t_out=BP)
            dg_flat = dgates.view(1, 4 * H, T * BP)
            # dW_hh = sum_t dgates_t h_{t-1}^T  (h_{-1} = 0): pair dgates[:, 1:] with yT[:, :-1]
            if T > 1:
                grads[4 * l + 1] = ops.conv1d_bwd_weight(yT[:, :-1].contiguous().view(1, H, (T - 1) * BP),
                                                         dgates[:, 1:].contiguous().view(1, 4 * H, (T - 1) * BP), 1,
                                                         pad_mode=ops.PAD_ZERO).reshape(4 * H, H)
            else:
                grads[4 * l + 1] = torch.zeros_like(w_hh)
            grads[4 * l] = ops.conv1d_bwd_weight(inp.view(1, H, T * BP), dg_flat, 1, pad_mode=ops.PAD_ZERO).reshape(4 * H, H)
            db = ops.bias_grad(dg_flat)
            grads[4 * l + 2], grads[4 * l + 3] = db, db.clone()
            # gradient w.r.t. this layer's input sequence: W_ih^T dgates, one GEMM over every (t, b)
            d_out = ops.conv1d(dg_flat, ops.pack_conv_weight(w_ih.t().contiguous().unsqueeze(-1)), H, 1, pad_left=0,
                               pad_mode=ops.PAD_ZERO, t_out=T * BP).view(H, T, BP)
        dx = ops.lstm_from_time_major(d_out, dy if ctx.skip else None, B)
        return (dx, None, *grads)


# ------------------------------------------------------------------------------------------ module-level helpers
def conv(m, x, act=ops.ACT_NONE):
    """SConv1d module `m` applied with autograd."""
    v, g = _wn(m.w)
    return _Conv.apply(x, v, g, m.w.bias, (m.kernel_size, m.stride, m.dilation, m.pad_mode, m.causal, act))


def conv_tr(m, x):
    if not m.causal:
        raise NotImplementedError("training path: non-causal SConvTranspose1d backward is not built yet")
    v, g = _wn(m.w)
    return _ConvTr.apply(x, v, g, m.w.bias, m.stride)


def snake(m, x):
    return _Snake.apply(x, m.alpha)


def add(a, b):
    return _Add.apply(a, b)


def slstm(m, x):
    p = m.lstm
    flat = []
    for l in range(m.num_layers):
        flat += [getattr(p, f"weight_ih_l{l}"), getattr(p, f"weight_hh_l{l}"), getattr(p, f"bias_ih_l{l}"),
                 getattr(p, f"bias_hh_l{l}")]
    return _LSTM.apply(x, m.skip, *flat)
