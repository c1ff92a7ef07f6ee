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


import os as _os
_LSTM_BWD_FUSED = _os.environ.get("FAC_LSTM_BWD_FUSED", "1") != "0"


def _split_ok(c_out, c_in, k, stride, x):
    if k == 1 and stride == 1:          # 1x1 with many channels: split-bf16 GEMM (conv1d_gemm_split.hip)
        return ops.gemm_split_ok(c_out, c_in, 1, x.shape[0] * x.shape[-1]) and not ops.pw_split_tail_ok(c_in, c_out, x.shape[0] * x.shape[-1])
    if k in (3, 5):                     # WaveNet / style-encoder k = 5, encoder output conv k = 3: enough channels only
        return (ops.BF16_SPLIT and stride == 1 and c_in % 16 == 0 and c_in >= 64 and c_out % 16 == 0 and c_out > 32
                and x.shape[0] * x.shape[-1] > 640)
    return (ops.BF16_SPLIT and k == 7 and stride == 1 and c_in % 16 == 0 and c_out % 16 == 0 and c_out > 2
            and x.shape[0] * x.shape[-1] > 640)


def _wn(w):
    """(v, g) of a ConvWeights holder (g None for a plain weight)."""
    return (w.weight_v, w.weight_g) if w.weight_norm else (w.weight, None)


class _Conv(Function):
    @staticmethod
    def forward(ctx, x, v, g, bias, cfg):
        k, stride, dilation, pad_mode, causal, act = cfg
        vd, gd = v.detach(), (g.detach() if g is not None else None)
        sc = ops.wn_scale(vd, gd) if gd is not None else None          # g / ||v||: once per forward, re-used by the backward
        if (stride == 2 and dilation == 1 and act == ops.ACT_NONE
                and ops.pw_taps_ok(v.shape[1], v.shape[0], k, 2, False, x.shape[0], -(-x.shape[-1] // 2))):
            wp, ws = ops.pack_conv_weight(vd, gd, scale=sc), None        # few channels: the streaming kernel with taps takes the fp32 pack
        elif stride > 1 and dilation == 1 and ops.gemm_split_strided_ok(v.shape[0], v.shape[1], k, stride, x.shape[0], -(-x.shape[-1] // stride)):
            wp, ws = None, ops.pack_gemm_weight_split(vd, gd, in_stride=stride, scale=sc)     # downsampling conv on the split GEMM kernel
        elif _split_ok(v.shape[0], v.shape[1], k, stride, x):       # k = 7 / k = 1 convs: fp32-grade split on the bf16 pipe
            wp, ws = None, ops.pack_conv_weight_split(vd, gd, scale=sc)
        else:
            wp, ws = ops.pack_conv_weight(vd, gd, scale=sc), None
        ctx.scale = sc
        xin = x.detach()
        n_out = -(-x.shape[-1] // stride)
        if (stride > 1 and dilation == 1 and causal and pad_mode == ops.PAD_REFLECT and act == ops.ACT_NONE and x.shape[-1] % stride == 0
                and x.shape[-1] > stride and ops.flat_strided_ok(v.shape[0], v.shape[1], k, stride, x.shape[0], n_out)):
            # short clips (the 160-frame latent rate): every clip reflect-padded on the left by k - s = s samples (the causal padding
            # of dac/model/encodec.py:212-222; T % s == 0: no right padding), all of them as one signal on the split GEMM kernel
            y = ops.conv1d_flat_strided(torch.nn.functional.pad(xin, (stride, 0), mode="reflect"),
                                        ops.pack_gemm_weight_split(vd, gd, in_stride=stride, scale=sc), v.shape[0], k, stride,
                                        bias=bias.detach() if bias is not None else None)
            ctx.cfg = cfg
            ctx.save_for_backward(x, v, g, bias, None)
            return y
        if ws is not None and stride > 1 and dilation == 1:
            # split GEMM over the phase sub-signals: plane inputs where the pass pays for itself (ops.p8_prepass; round 6: the training
            # launches take the same pre-pass as the inference ones -- same bf16 operands in the same order, same bits)
            xin = ops.p8_prepass(xin, 2.0 * v.shape[0] * k / (4.0 * stride))
        y = ops.conv1d(xin, wp, v.shape[0], k, bias=bias.detach() if bias is not None else None,
                       stride=stride, dilation=dilation, pad_mode=pad_mode, causal=causal, act=act, w_split=ws)
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
        dx = ops.conv1d_bwd_data(dy, vd, gd, x.shape[-1], stride=stride, dilation=dilation, pad_mode=pad_mode, causal=causal,
                                 scale=ctx.scale) if ctx.needs_input_grad[0] else None
        db = None
        if bias is not None:
            dw, db = ops.conv1d_bwd_weight(x.detach(), dy, k, stride=stride, dilation=dilation, pad_mode=pad_mode, causal=causal, want_db=True)
        else:
            dw = ops.conv1d_bwd_weight(x.detach(), dy, k, stride=stride, dilation=dilation, pad_mode=pad_mode, causal=causal)
        if g is not None:
            dv, dg = ops.weight_norm_bwd(vd, gd, dw)
        else:
            dv, dg = dw, None
        return dx, dv, dg, db, None


class _ConvTr(Function):
    @staticmethod
    def forward(ctx, x, v, g, bias, stride):
        vd, gd = v.detach(), (g.detach() if g is not None else None)
        ctx.stride = stride
        ctx.save_for_backward(x, v, g, bias)
        if ops.flat_convtr_ok(v.shape[0], v.shape[1], stride, x.shape[0], x.shape[-1] + 1):
            # short clips: a zero column in front of every clip (the x[t - 1] of its first frame), one flattened signal; the s output
            # samples of that column are dropped
            xz = torch.nn.functional.pad(x.detach(), (1, 0))
            return ops.conv_transpose1d_flat(xz, vd, gd, stride, bias=bias.detach() if bias is not None else None)[:, :, stride:].contiguous()
        wt = ops.convtr_weight_for(vd, gd, stride, x.shape[-1], batch=x.shape[0])
        xin = x.detach()
        if isinstance(wt, tuple):                     # all-phases launch on the split GEMM kernel: 2 s C_out MACs per input sample
            xin = ops.p8_prepass(xin, 2.0 * v.shape[1] * 2 * stride / 4.0)
        y = ops.conv_transpose1d(xin, wt, v.shape[1], stride, bias=bias.detach() if bias is not None else None)
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


class _SnakeDual(Function):
    """(y, snake(y, alpha)) for a tensor that feeds a ResidualUnit: the raw tensor is the unit's skip input, the pre-activated copy
    the input of its k = 7 conv (dac/model/dac.py:31-42).  One backward launch returns dy_raw + d snake: the engine never sees a
    tensor with two consumers, so there is no separate fan-in add."""

    @staticmethod
    def forward(ctx, y, alpha):
        ctx.save_for_backward(y, alpha)
        return y, ops.snake(y.detach(), alpha.detach().reshape(-1))

    @staticmethod
    def backward(ctx, dy, dya):
        y, alpha = ctx.saved_tensors
        if dya is None:
            return dy, None
        dx, da, _ = ops.snake_bwd_fused(y.detach(), alpha.detach().reshape(-1), dya,          # dya: possibly a window of padded rows
                                        add=dy.contiguous() if dy is not None else None)
        return dx, da.reshape(alpha.shape)


def snake_dual(y, alpha):
    return _SnakeDual.apply(y, alpha)


class _ResUnit(Function):
    """A whole ResidualUnit (dac/model/dac.py:25-42) as ONE autograd node with two launches forward:
        h, ha = conv k7(xa) + b7, snake(h, a2)                    (second output from the conv's epilogue)
        y, ya = conv k1(ha) + b1 + x, snake(y, a_next)            (residual and the NEXT Snake fused into the epilogue)
    x: the unit's raw input (skip), xa = snake(x, a1) from the producer (previous unit's `ya`, or snake_dual); a_next: alpha of
    the Snake that follows the unit (next unit's first Snake, or the block's own).  Backward: Snake backward kernels that also
    add the skip gradient and leave the bias gradients (fac_snake_bwd_fused), conv data / weight gradients, weight-norm.
    Saves xa, h, ha (y is the output)."""

    @staticmethod
    def forward(ctx, x, xa, v7, g7, b7, a2, v1, g1, b1, a_next, cfg):
        dilation, pad_mode, causal = cfg
        c = v7.shape[0]
        xd, xad = x.detach(), xa.detach()
        v7d, g7d, v1d, g1d = v7.detach(), g7.detach(), v1.detach(), g1.detach()
        s7, s1 = ops.wn_scale(v7d, g7d), ops.wn_scale(v1d, g1d)      # once per forward, re-used by the backward
        if _split_ok(c, v7.shape[1], 7, 1, xad):
            wp, ws = None, ops.pack_conv_weight_split(v7d, g7d, scale=s7)
        else:
            wp, ws = ops.pack_conv_weight(v7d, g7d, scale=s7), None
        h, ha = ops.conv1d(xad, wp, c, 7, bias=b7.detach(), dilation=dilation, pad_mode=pad_mode, causal=causal,
                           alpha_y2=a2.detach().reshape(-1), w_split=ws)
        if _split_ok(c, c, 1, 1, ha):
            wp1, ws1 = None, ops.pack_conv_weight_split(v1d, g1d, scale=s1)
        else:
            wp1, ws1 = ops.pack_conv_weight(v1d, g1d, scale=s1), None
        y, ya = ops.conv1d(ha, wp1, c, 1, bias=b1.detach(), pad_mode=pad_mode, causal=causal, res=xd,
                           alpha_y2=a_next.detach().reshape(-1), w_split=ws1)
        ctx.cfg = cfg
        ctx.scales = (s7, s1)
        ctx.save_for_backward(xa, h, ha, y, v7, g7, a2, v1, g1, a_next)
        return y, ya

    @staticmethod
    def backward(ctx, dy, dya):
        xa, h, ha, y, v7, g7, a2, v1, g1, a_next = ctx.saved_tensors
        dilation, pad_mode, causal = ctx.cfg
        T = y.shape[-1]
        v7d, g7d, v1d, g1d = v7.detach(), g7.detach(), v1.detach(), g1.detach()
        # gradient of y: raw consumer (next unit's skip) + the following Snake; also the k1 bias gradient
        if dya is not None:
            dyt, da_next, db1 = ops.snake_bwd_fused(y.detach(), a_next.detach().reshape(-1), dya,     # (possibly a window of padded rows)
                                                    add=dy.contiguous() if dy is not None else None, want_bias=True)
            da_next = da_next.reshape(a_next.shape)
        else:
            dyt, da_next, db1 = dy.contiguous(), None, ops.bias_grad(dy.contiguous())
        s7, s1 = ctx.scales
        dha = ops.conv1d_bwd_data(dyt, v1d, g1d, T, pad_mode=pad_mode, causal=causal, scale=s1)
        dv1, dg1 = ops.weight_norm_bwd(v1d, g1d, ops.conv1d_bwd_weight(ha.detach(), dyt, 1, pad_mode=pad_mode, causal=causal))
        dh, da2, db7 = ops.snake_bwd_fused(h.detach(), a2.detach().reshape(-1), dha, want_bias=True)
        # the gradient of xa goes to the Snake backward of the producing node (the previous unit, or snake_dual), which reads rows
        # with a stride: no un-padding copy
        dxa = ops.conv1d_bwd_data(dh, v7d, g7d, T, dilation=dilation, pad_mode=pad_mode, causal=causal, scale=s7, allow_view=True)
        dv7, dg7 = ops.weight_norm_bwd(v7d, g7d, ops.conv1d_bwd_weight(xa.detach(), dh, 7, dilation=dilation, pad_mode=pad_mode,
                                                                       causal=causal))
        return dyt, dxa, dv7, dg7, db7, da2.reshape(a2.shape), dv1, dg1, db1, da_next, None


def res_unit(ru, x, xa, alpha_next):
    """ResidualUnit module `ru` in training mode: (y, snake(y, alpha_next))."""
    b = ru.block
    k7, k1 = b[1], b[3]
    if not (k7.w.weight_norm and k1.w.weight_norm):
        raise NotImplementedError("fused ResidualUnit: weight-normed convs only (dac/model/dac.py:25-42)")
    return _ResUnit.apply(x, xa, k7.w.weight_v, k7.w.weight_g, k7.w.bias, b[2].alpha, k1.w.weight_v, k1.w.weight_g, k1.w.bias,
                          alpha_next, (k7.dilation, k7.pad_mode, k7.causal))


class _Add(Function):
    @staticmethod
    def forward(ctx, a, b):
        return ops.add(a.detach(), b.detach())

    @staticmethod
    def backward(ctx, dy):
        # NOT `return dy, dy`.  The two inputs' producers may run their backward on DIFFERENT streams (concurrent chains,
        # ops.run_chains), and the autograd engine accumulates into a buffered gradient IN PLACE once that tensor is uniquely owned.
        # One tensor object handed to two consumers then gets written on one stream (the later accumulation for consumer A) while
        # kernels of the other stream (consumer B's own accumulation, already released on the host) still read it.  Round 5 hit
        # exactly that: with the content RVQ on a side stream the residual path's gradient -- hence the encoder's -- came out 2.6 %
        # wrong at B = 16, deterministically, and right under AMD_SERIALIZE_KERNEL=3 (tools/tune/chain_probe.py is the minimal
        # cross-stream fan-in, which is fine; the shared object was the missing ingredient).  A copy per add of latent-rate tensors
        # is noise.
        return dy, dy.clone()


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
            with ops.flop_scale(B / BP):
                use_split = ops.gemm_split_ok(4 * H, H, 1, T * BP)
                sig = inp.view(1, H, T * BP)
                if use_split:
                    sig = ops.p8_prepass(sig, 2.0 * 4 * H / 4.0)
                pre = ops.conv1d(sig, None if use_split else ops.pack_conv_weight(w_ih), 4 * H, 1,
                                 bias=ops.add(b_ih, b_hh), pad_left=0, t_out=T * BP, pad_mode=ops.PAD_ZERO,
                                 w_split=ops.pack_gemm_weight_split(w_ih) if use_split else None)
                gates = torch.empty(4 * H, T, BP, device=x.device)
                cs = torch.empty(H, T, BP, device=x.device)
                if ops.lstm_persist_ok(H, B):      # whole layer in one launch, W_hh resident in registers
                    yT = ops.lstm_layer_persist(pre.view(4 * H, T, BP), w_hh, H, B, save=(gates, cs))
                else:
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
        with ops.flop_scale(B / BP):                                  # batch padding is not algorithmic work
            return _LSTM._backward(ctx, dy, x, params, B, H, T, BP)

    @staticmethod
    def _backward(ctx, dy, x, params, B, H, T, BP):
        L = len(params) // 4
        dy = dy.contiguous()
        d_out = ops.lstm_to_time_major(dy)                            # gradient w.r.t. the top layer's h sequence
        grads = [None] * (4 * L)
        for l in reversed(range(L)):
            w_ih, w_hh, _, _ = (p.detach() for p in params[4 * l: 4 * l + 4])
            inp, yT, gates, cs = ctx.saved[l]
            if _LSTM_BWD_FUSED:       # whole recurrence of the layer in one C call, two launches per step (lstm.hip)
                dgates = ops.lstm_layer_bwd(d_out.contiguous(), w_hh, gates, cs, H, batch=B)
            else:
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
            if ops.gemm_split_ok(H, 4 * H, 1, T * BP):
                d_out = ops.conv1d(dg_flat, None, H, 1, pad_left=0, pad_mode=ops.PAD_ZERO, t_out=T * BP,
                                   w_split=ops.pack_gemm_weight_split_t(w_ih)).view(H, T, BP)
            else:
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


class _LayerNormAffine(Function):
    """outs = LayerNorm_C(x) * gamma + beta with style = [gamma | beta] (modules/quantize.py:444-449)."""

    @staticmethod
    def forward(ctx, x, style):
        ctx.save_for_backward(x, style)
        return ops.layernorm_c_affine(x.detach(), style.detach().contiguous())

    @staticmethod
    def backward(ctx, dout):
        x, style = ctx.saved_tensors
        return ops.layernorm_c_affine_bwd(x.detach(), style.detach().contiguous(), dout.contiguous())


def layernorm_affine(x, style):
    return _LayerNormAffine.apply(x, style)


class _RVQ(Function):
    """ResidualVectorQuantize.forward in training mode (dac/nn/quantize.py:127-198) with the per-sample quantizer
    masks given explicitly (mask (n, B) of 0/1: the reference draws them from torch.randint, :163-168).
    params per quantizer: in_proj (v, g, bias), codebook, out_proj (v, g, bias).
    Returns z_q, commitment (scalar), codebook loss (scalar); codes via the `codes` buffer (B, n, T)."""

    @staticmethod
    def forward(ctx, z, mask, codes, latents, *params):
        n = len(params) // 7
        B, D, T = z.shape
        zd = z.detach()
        z_q = torch.zeros_like(zd)
        residuals, z_es = [], []
        nt = ops.vq_loss_tiles(T)
        lp = torch.empty(n, B, nt, device=z.device)
        src = zd
        for i in range(n):
            v_in, g_in, b_in, cb, v_out, g_out, b_out = (p.detach() for p in params[7 * i: 7 * i + 7])
            residuals.append(src)
            z_e = torch.empty(B, 8, T, device=z.device)
            nxt = torch.empty_like(zd)
            ops.vq_step(src, ops.pack_conv_weight(v_in, g_in), b_in, cb, v_out, ops.wn_scale(v_out, g_out), b_out,
                        codes[:, i], residual=nxt, zq_acc=z_q, mask=mask[i].contiguous(), z_e=z_e, loss_part=lp[i])
            # the fused kernel writes nxt = src - z_q_i (reading src = z_in): keep src untouched for the backward
            z_es.append(z_e)
            if latents is not None:
                latents[:, 8 * i: 8 * (i + 1)] = z_e
            src = nxt
        per = lp.sum(2) / float(8 * T)                 # (n, B): mse per sample; commitment == codebook loss in value
        loss = (per * mask).mean(1).sum()
        ctx.n, ctx.dims = n, (B, D, T)
        ctx.saved = (residuals, z_es, codes, mask)
        ctx.save_for_backward(*params)
        return z_q, loss, loss.clone()

    @staticmethod
    def backward(ctx, d_zq, g_commit, g_cb):
        params = ctx.saved_tensors
        residuals, z_es, codes, mask = ctx.saved
        n = ctx.n
        B, D, T = ctx.dims
        d_zq = d_zq.contiguous()
        grads = [None] * (7 * n)
        d_res = None                                   # gradient w.r.t. the residual entering quantizer i+1
        for i in reversed(range(n)):
            v_in, g_in, b_in, cb, v_out, g_out, b_out = (p.detach() for p in params[7 * i: 7 * i + 7])
            mi = mask[i].contiguous()
            # upstream of z_q_i: the masked sum and (negatively) the residual chain
            G = ops.rows_fma(d_zq, mi, d_res, -1.0)
            ci = codes[:, i]
            _, z_st = ops.vq_latent_bwd(z_es[i], cb, ci, want_dze=False, want_zst=True)
            # out_proj (8 -> D, 1x1, weight-normed)
            d_zst = ops.conv1d_bwd_data(G, v_out, g_out, T, pad_mode=ops.PAD_ZERO)
            dw_out = ops.conv1d_bwd_weight(z_st, G, 1, pad_mode=ops.PAD_ZERO)
            grads[7 * i + 4], grads[7 * i + 5] = ops.weight_norm_bwd(v_out, g_out, dw_out)
            grads[7 * i + 6] = ops.bias_grad(G)
            wc = (mi * (g_commit / B)).contiguous()
            wb = (mi * (g_cb / B)).contiguous()
            d_ze, _ = ops.vq_latent_bwd(z_es[i], cb, ci, d_zst=d_zst, wc=wc)
            grads[7 * i + 3] = ops.vq_codebook_grad(z_es[i], cb, ci, wb)
            # in_proj (D -> 8)
            d_in = ops.conv1d_bwd_data(d_ze, v_in, g_in, T, pad_mode=ops.PAD_ZERO)
            dw_in = ops.conv1d_bwd_weight(residuals[i], d_ze, 1, pad_mode=ops.PAD_ZERO)
            grads[7 * i], grads[7 * i + 1] = ops.weight_norm_bwd(v_in, g_in, dw_in)
            grads[7 * i + 2] = ops.bias_grad(d_ze)
            d_res = d_in if d_res is None else ops.add(d_res, d_in)
        return (d_res, None, None, None, *grads)


def rvq(m, z, mask=None, latents=None):
    """ResidualVectorQuantize module `m` in training mode -> (z_q, codes, commitment, codebook_loss).
    mask (n, B) float 0/1 (quantizer dropout); None = every quantizer active for every sample.
    latents: optional (B, 8 n, T) buffer that receives the projected latents z_e_i (dac/nn/quantize.py:195; no gradient)."""
    n = m.n_codebooks
    B, _, T = z.shape
    if mask is None:
        mask = torch.ones(n, B, device=z.device)
    codes = torch.empty(B, n, T, device=z.device, dtype=torch.int64)
    flat = []
    for q in m.quantizers:
        flat += [q.in_proj.weight_v, q.in_proj.weight_g, q.in_proj.bias, q.codebook.weight,
                 q.out_proj.weight_v, q.out_proj.weight_g, q.out_proj.bias]
    z_q, commit, cbl = _RVQ.apply(z, mask, codes, latents, *flat)
    return z_q, codes, commit, cbl


class _SubDetached(Function):
    """x - a - b where a and b are treated as constants (modules/quantize.py:411: x - z_p.detach() - z_c.detach())."""

    @staticmethod
    def forward(ctx, x, a, b):
        return ops.sub2(x.detach(), a.detach(), b.detach())

    @staticmethod
    def backward(ctx, d):
        return d, None, None


class _MixOuts(Function):
    """outs = z_p.detach() + z_c.detach() + z_r * res_mask[b]  (modules/quantize.py:402-435): only z_r carries gradient."""

    @staticmethod
    def forward(ctx, z_p, z_c, z_r, res_mask):
        ctx.save_for_backward(res_mask)
        return ops.add(ops.add(z_p.detach(), z_c.detach()), ops.rows_fma(z_r.detach(), res_mask))

    @staticmethod
    def backward(ctx, d):
        (res_mask,) = ctx.saved_tensors
        return None, None, ops.rows_fma(d.contiguous(), res_mask), None


def sub_detached(x, a, b):
    return _SubDetached.apply(x, a, b)


def mix_outs(z_p, z_c, z_r, res_mask):
    return _MixOuts.apply(z_p, z_c, z_r, res_mask)


def linear(m, x):
    """_Linear module (weight (out, in), bias) with autograd, as a 1x1 conv on the (B, C, 1) view."""
    y = _Conv.apply(x.reshape(x.shape[0], x.shape[1], 1), m.weight.unsqueeze(-1), None, m.bias,
                    (1, 1, 1, ops.PAD_ZERO, True, ops.ACT_NONE))
    return y.reshape(x.shape[0], -1)


def draw_quantizer_masks(n_codebooks, batch, dropout=0.5, generator=None):
    """The per-sample quantizer-dropout masks of ResidualVectorQuantize.forward in training mode
    (dac/nn/quantize.py:163-168, 181-183) as a (n, B) float tensor."""
    nq = torch.ones(batch) * n_codebooks + 1
    drop = torch.randint(1, n_codebooks + 1, (batch,), generator=generator)
    n_drop = int(batch * dropout)
    nq[:n_drop] = drop[:n_drop].to(nq.dtype)
    return torch.stack([(torch.full((batch,), float(i)) < nq).to(torch.float32) for i in range(n_codebooks)])
