"""The variant the north star names: quantize/fvq.py `FactorizedVectorQuantize` and quantize/rvq.py
`ResidualVQ` (dead code in the reference -- nothing imports them -- but the same nearest-code search as
the live dac/nn/quantize.py path, fvq.py:101-116 == quantize.py:78-94).  Same fused `fac_vq_fwd` kernel;
only the parameter names differ (weight-normed nn.Linear projections: in_proj/out_proj.weight_g (out,1),
weight_v (out,in); codebook under `_codebook.weight`)."""
import torch
from torch import nn

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
        B, D, T = z.shape
        w_in = ops.pack_conv_weight(self.in_proj.weight_v.detach().unsqueeze(-1), self.in_proj.weight_g.detach())
        v_out = self.out_proj.weight_v.detach()
        s_out = ops.wn_scale(v_out, self.out_proj.weight_g.detach())
        codes = torch.empty(B, T, device=z.device, dtype=torch.int64)
        out = torch.empty_like(z)
        lp = torch.empty(B, (T + 63) // 64, device=z.device, dtype=torch.float32)
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


class ResidualVQ(nn.Module):
    """quantize/rvq.py:12-81, eval path (no quantizer dropout).  `codebook_size` is log2 of the number of
    codes (:21).  Returns (quantized_out (B, D, T), all_indices (N, B, T), all_losses (N,),
    all_quantized (N, B, D, T)) like :70-73."""

    def __init__(self, *, num_quantizers, codebook_size, **kwargs):
        super().__init__()
        sizes = [codebook_size] * num_quantizers if isinstance(codebook_size, int) else list(codebook_size)
        self.layers = nn.ModuleList([FactorizedVectorQuantize(codebook_size=2 ** s, **kwargs) for s in sizes])
        self.num_quantizers = num_quantizers

    def forward(self, x, n_quantizers=None):
        if self.training:
            raise NotImplementedError("train-mode ResidualVQ (quantizer dropout) is not built yet")
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
