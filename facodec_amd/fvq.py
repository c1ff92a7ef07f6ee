"""The variant the north star names: quantize/fvq.py `FactorizedVectorQuantize` and quantize/rvq.py
`ResidualVQ` (dead code in the reference -- nothing imports them -- but the same nearest-code search as
the live dac/nn/quantize.py path, fvq.py:101-116 == quantize.py:78-94).  Same fused `fac_vq_fwd` kernel;
only the parameter names differ (weight-normed nn.Linear projections: in_proj/out_proj.weight_g (out,1),
weight_v (out,in); codebook under `_codebook.weight`)."""
import math

import torch
from torch import nn
from torch.autograd import Function

from . import ops
from .layers import _uniform_


class _WNLinear(nn.Module):
    """weight_norm(nn.Linear(c_in, c_out)): weight_g (c_out, 1), weight_v (c_out, c_in), bias (c_out)."""

    def __init__(self, c_in, c_out):
        super().__init__()
        b = 1.0 / c_in ** 0.5
        v = _uniform_(torch.empty(c_out, c_in), b)
        self.weight_g = nn.Parameter(v.norm(dim=1, keepdim=True))
        self.weight_v = nn.Parameter(v)
        self.bias = nn.Parameter(_uniform_(torch.empty(c_out), b))


class _Embedding(nn.Module):
    def __init__(self, n, d):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(n, d))


class FactorizedVectorQuantize(nn.Module):
    """quantize/fvq.py:16-116.  forward(z (B, D, T)) -> (z_q (B, D, T), indices (B, T) int64, commit_loss (B,));
    eval mode returns a zero loss exactly like :66-74."""

    def __init__(self, dim, codebook_size, codebook_dim, commitment, **kwargs):
        super().__init__()
        if codebook_dim != 8 or dim == codebook_dim:
            raise NotImplementedError("the VQ kernel is specialised for dim != codebook_dim = 8")
        self.codebook_size, self.codebook_dim, self.commitment, self.dim = codebook_size, codebook_dim, commitment, dim
        self.in_proj = _WNLinear(dim, codebook_dim)
        self.out_proj = _WNLinear(codebook_dim, dim)
        self._codebook = _Embedding(codebook_size, codebook_dim)

    @property
    def codebook(self):
        return self._codebook

    def forward(self, z):
        if self.training and torch.is_grad_enabled():
            B, _, T = z.shape
            codes = torch.empty(B, T, device=z.device, dtype=torch.int64)
            ip, op = self.in_proj, self.out_proj
            out, loss = _FVQ.apply(z, float(self.commitment), codes, ip.weight_v.unsqueeze(-1), ip.weight_g.unsqueeze(-1), ip.bias,
                                   self._codebook.weight, op.weight_v.unsqueeze(-1), op.weight_g.unsqueeze(-1), op.bias)
            return out, codes, loss
        B, D, T = z.shape
        w_in = ops.pack_conv_weight(self.in_proj.weight_v.detach().unsqueeze(-1), self.in_proj.weight_g.detach())
        v_out = self.out_proj.weight_v.detach()
        s_out = ops.wn_scale(v_out, self.out_proj.weight_g.detach())
        codes = torch.empty(B, T, device=z.device, dtype=torch.int64)
        out = torch.empty_like(z)
        lp = torch.empty(B, ops.vq_loss_tiles(T), device=z.device, dtype=torch.float32)
        ops.vq_step(z, w_in, self.in_proj.bias.detach(), self._codebook.weight.detach(), v_out, s_out,
                    self.out_proj.bias.detach(), codes, zq_out=out, loss_part=lp)
        if self.training:
            mse = lp.sum(1) / float(8 * T)
            loss = mse * self.commitment + mse
        else:
            loss = torch.zeros(B, device=z.device)
        return out, codes, loss

    def embed_code(self, embed_id):
        return self._codebook.weight[embed_id]

    def decode_code(self, embed_id):
        return self.embed_code(embed_id).transpose(1, 2)


class _FVQ(Function):
    """FactorizedVectorQuantize.forward in training mode (quantize/fvq.py:59-84) as ONE autograd node: the fused search kernel
    forward; backward = out_proj (1x1) data / weight gradients -> straight-through estimator (:76-78) + commitment term against
    z_q.detach() (:67-70) -> in_proj gradients; codebook loss against z_e.detach() (:71) by the deterministic per-code gather.
    Projection weights arrive as (out, in, 1) views of the Linear parameters (autograd undoes the view).
    Returns (z_q (B, D, T), commit_loss (B,))."""

    @staticmethod
    def forward(ctx, z, commitment, codes, v_in, g_in, b_in, cb, v_out, g_out, b_out):
        B, D, T = z.shape
        zd = z.detach().contiguous()
        out = torch.empty_like(zd)
        z_e = torch.empty(B, 8, T, device=z.device, dtype=torch.float32)
        lp = torch.empty(B, ops.vq_loss_tiles(T), device=z.device, dtype=torch.float32)
        ops.vq_step(zd, ops.pack_conv_weight(v_in.detach(), g_in.detach()), b_in.detach(), cb.detach(), v_out.detach(),
                    ops.wn_scale(v_out.detach(), g_out.detach()), b_out.detach(), codes, zq_out=out, z_e=z_e, loss_part=lp)
        mse = lp.sum(1) / float(8 * T)
        ctx.commitment, ctx.T = commitment, T
        ctx.codes = codes
        ctx.save_for_backward(zd, z_e, v_in, g_in, cb, v_out, g_out)
        ctx.mark_non_differentiable(codes)
        return out, mse * commitment + mse

    @staticmethod
    def backward(ctx, d_out, d_loss):
        zd, z_e, v_in, g_in, cb, v_out, g_out = (t.detach() for t in ctx.saved_tensors)
        T, codes = ctx.T, ctx.codes
        G = d_out.contiguous()
        _, z_st = ops.vq_latent_bwd(z_e, cb, codes, want_dze=False, want_zst=True)
        d_zst = ops.conv1d_bwd_data(G, v_out, g_out, T, pad_mode=ops.PAD_ZERO)
        dv_out, dg_out = ops.weight_norm_bwd(v_out, g_out, ops.conv1d_bwd_weight(z_st, G, 1, pad_mode=ops.PAD_ZERO))
        db_out = ops.bias_grad(G)
        d_loss = d_loss.contiguous()
        d_ze, _ = ops.vq_latent_bwd(z_e, cb, codes, d_zst=d_zst, wc=(d_loss * ctx.commitment).contiguous())
        d_cb = ops.vq_codebook_grad(z_e, cb, codes, d_loss)
        d_z = ops.conv1d_bwd_data(d_ze, v_in, g_in, T, pad_mode=ops.PAD_ZERO)
        dv_in, dg_in = ops.weight_norm_bwd(v_in, g_in, ops.conv1d_bwd_weight(zd, d_ze, 1, pad_mode=ops.PAD_ZERO))
        return d_z, None, None, dv_in, dg_in, ops.bias_grad(d_ze), d_cb, dv_out, dg_out, db_out


class _Sub(Function):
    """residual - quantized (quantize/rvq.py:59), gradient to both."""

    @staticmethod
    def forward(ctx, a, b):
        return _sub(a.detach(), b.detach())

    @staticmethod
    def backward(ctx, d):
        d = d.contiguous()
        return d, ops.rows_fma(d, torch.full((d.shape[0],), -1.0, device=d.device))


class _MaskedAcc(Function):
    """acc + q * mask[b] (quantize/rvq.py:61); acc may be None for the first layer."""

    @staticmethod
    def forward(ctx, acc, q, mask):
        ctx.save_for_backward(mask)
        ctx.has_acc = acc is not None
        return ops.rows_fma(q.detach().contiguous(), mask, None if acc is None else acc.detach())

    @staticmethod
    def backward(ctx, d):
        (mask,) = ctx.saved_tensors
        d = d.contiguous()
        return (d if ctx.has_acc else None), ops.rows_fma(d, mask), None


class ResidualVQ(nn.Module):
    """quantize/rvq.py:12-81.  `codebook_size` is log2 of the number of codes (:21).  Returns (quantized_out (B, D, T),
    all_indices (N, B, T), all_losses (N,), all_quantized (N, B, D, T)) like :70-73.

    Training (:36-46): sample b keeps `n_quantizers[b]` codebooks; the first int(B * quantizer_dropout) samples take theirs
    from `torch.randint(1, N + 1)` ('linear') or `2 ** torch.randint(1, int(log2 N))` ('exp') -- drawn here with the same call
    on the CPU's default generator (same draws for the same seed), or handed in as `dropout` (the (B,) result of that call)
    so a test can replay the reference's draw.  With `dropout_type=None` the reference's training forward dies on an unbound
    local (:45 reads `dropout`, which only the two typed branches assign); the same error is raised here."""

    def __init__(self, *, num_quantizers, codebook_size, **kwargs):
        super().__init__()
        sizes = [codebook_size] * num_quantizers if isinstance(codebook_size, int) else list(codebook_size)
        self.layers = nn.ModuleList([FactorizedVectorQuantize(codebook_size=2 ** s, **kwargs) for s in sizes])
        self.num_quantizers = num_quantizers
        self.quantizer_dropout = kwargs.get("quantizer_dropout", 0.0)
        self.dropout_type = kwargs.get("dropout_type", None)

    def _draw_n_quantizers(self, B, dropout):
        nq = torch.ones((B,)) * self.num_quantizers + 1
        if dropout is None:
            if self.dropout_type == "linear":
                dropout = torch.randint(1, self.num_quantizers + 1, (B,))
            elif self.dropout_type == "exp":
                dropout = torch.pow(2, torch.randint(1, int(math.log2(self.num_quantizers)), (B,)))
            else:
                raise UnboundLocalError("local variable 'dropout' referenced before assignment (quantize/rvq.py:45: training "
                                        "needs dropout_type 'linear' or 'exp')")
        n_dropout = int(B * self.quantizer_dropout)
        nq[:n_dropout] = torch.as_tensor(dropout)[:n_dropout].to(nq.dtype)
        return nq

    def _forward_train(self, x, dropout):
        B = x.shape[0]
        nq = self._draw_n_quantizers(B, dropout)
        residual, quantized_out = x, None
        idxs, losses, quants = [], [], []
        for i, layer in enumerate(self.layers):
            mask = (torch.full((B,), float(i)) < nq).to(x.device, torch.float32)
            q, idx, loss = layer(residual)
            residual = _Sub.apply(residual, q)
            quantized_out = _MaskedAcc.apply(quantized_out, q, mask)
            losses.append((loss * mask).mean())     # B-element host-side bookkeeping of the returned per-layer scalar
            idxs.append(idx)
            quants.append(q)
        return quantized_out, torch.stack(idxs), torch.stack(losses), torch.stack(quants)

    def forward(self, x, n_quantizers=None, dropout=None):
        if self.training:
            return self._forward_train(x, dropout)
        n = self.num_quantizers if n_quantizers is None else int(n_quantizers)
        residual, quantized_out = x, None
        idxs, losses, quants = [], [], []
        for layer in list(self.layers)[:n]:
            q, idx, loss = layer(residual)
            residual = _sub(residual, q)
            quantized_out = q if quantized_out is None else ops.add(quantized_out, q)
            idxs.append(idx)
            losses.append(loss.mean())
            quants.append(q)
        return quantized_out, torch.stack(idxs), torch.stack(losses), torch.stack(quants)


def _sub(a, b):
    """a - b through the fused sub kernel (a - b - 0)."""
    z = getattr(_sub, "_zero", None)
    if z is None or z.shape != a.shape or z.device != a.device:
        z = torch.zeros_like(a)
        _sub._zero = z
    return ops.sub2(a, b, z)
