"""Train-mode golden vectors of row a11 from the REAL reference (quantize/fvq.py:36-86 FactorizedVectorQuantize with its
detach placements + straight-through estimator, quantize/rvq.py:27-73 ResidualVQ with the train-time `n_quantizers`
dropout): forward values AND the gradients torch autograd gives the reference.  Build container only.

Run:  python tests/golden/make_golden_fvq_train.py       (seconds)      -> tests/golden/fvq_train.npz

The reference draws the dropout with torch.randint (rvq.py:38-44); the draw is recorded here (torch.randint patched to
return DRAW) and handed to the product / oracle as an argument.  The scalar that is back-propagated is
sum(z_q * W) + sum(loss_b * U) (+ sum(all_quantized * V) for the residual stack) with fixed formula weights, so every output
carries a non-trivial upstream gradient."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)

import make_golden as MG  # noqa: E402
from facodec_amd import synth  # noqa: E402

DRAW = {"linear": [2, 1, 3, 1], "exp": [1, 1, 1, 1]}    # 'exp': randint(1, int(log2(4)) = 2) -> always 1, then 2 ** 1 (rvq.py:41-44)


def upstream(shape, seed):
    n = int(np.prod(shape))
    k = torch.arange(n, dtype=torch.float64)
    return torch.sin(0.37 * k + seed).reshape(shape).float() / float(np.sqrt(n))


def main():
    MG.install_shims()
    sys.path.insert(0, MG.REF)
    from quantize.fvq import FactorizedVectorQuantize
    from quantize.rvq import ResidualVQ
    out = {}
    g = torch.Generator().manual_seed(11)

    # ---------------------------------------------------------------- one FactorizedVectorQuantize, train mode, with gradients
    vq = FactorizedVectorQuantize(dim=64, codebook_size=1024, codebook_dim=8, commitment=0.15).train()
    synth.load_synthetic(vq, seed=4, prefix="fvq.")
    z = torch.randn(3, 64, 50, generator=g).requires_grad_()
    zq, idx, loss = vq(z)
    (zq * upstream(zq.shape, 1)).sum().add((loss * upstream(loss.shape, 2)).sum()).backward()
    out.update(fvq_z=z.detach().numpy(), fvq_zq=zq.detach().numpy(), fvq_idx=idx.numpy().astype(np.int16), fvq_loss=loss.detach().numpy(),
               fvq_dz=z.grad.numpy())
    for n, p in vq.named_parameters():
        out["fvq_grad." + n] = p.grad.numpy()

    # ---------------------------------------------------------------- ResidualVQ, train mode, both dropout types
    for kind, nq in (("linear", 3), ("exp", 4)):
        rv = ResidualVQ(num_quantizers=nq, codebook_size=10, dim=64, codebook_dim=8, commitment=0.15, quantizer_dropout=0.75,
                        dropout_type=kind).train()
        synth.load_synthetic(rv, seed=6, prefix="rvq.")
        x = torch.randn(4, 64, 50, generator=g).requires_grad_()
        real = torch.randint
        calls = []

        def fake(lo, hi, size, **kw):
            calls.append((lo, hi, tuple(size)))
            assert all(lo <= v < hi for v in DRAW[kind]) and tuple(size) == (4,)
            return torch.tensor(DRAW[kind])

        torch.randint = fake
        try:
            q_out, all_idx, all_loss, all_q = rv(x)
        finally:
            torch.randint = real
        assert len(calls) == 1
        s = (q_out * upstream(q_out.shape, 3)).sum() + (all_loss * upstream(all_loss.shape, 4)).sum() + (all_q * upstream(all_q.shape, 5)).sum()
        s.backward()
        p = f"rvq_{kind}_"
        out.update({p + "x": x.detach().numpy(), p + "draw": np.array(DRAW[kind]), p + "randint_args": np.array(calls[0][:2]),
                    p + "out": q_out.detach().numpy(), p + "idx": all_idx.numpy().astype(np.int16), p + "losses": all_loss.detach().numpy(),
                    p + "quantized_probe": all_q.detach()[:, :, ::4, ::5].numpy(), p + "dx": x.grad.numpy()})
        for n, prm in rv.named_parameters():
            out[p + "grad." + n] = prm.grad.numpy()
    np.savez_compressed(os.path.join(HERE, "fvq_train.npz"), **out)
    print({k: (v.shape, float(np.abs(v).max())) for k, v in out.items() if k.endswith(("dz", "dx", "losses", "loss"))})


if __name__ == "__main__":
    torch.manual_seed(0)
    main()
