"""Code-mismatch triage (facodec_amd/diagnostics.py): near-tie flips and their cascades vs genuine mismatches.  CPU only."""
import torch

from facodec_amd.diagnostics import classify_code_mismatches
from facodec_amd.quantize import ResidualVectorQuantize


def _rvq():
    torch.manual_seed(0)
    rvq = ResidualVectorQuantize(input_dim=16, n_codebooks=3, codebook_size=8, codebook_dim=8)
    for q in rvq.quantizers:
        q.codebook.weight.data.normal_()
    return rvq


def _nearest(rvq, lat):
    """(B, 8n, T) latents -> (B, n, T) nearest codes in fp64 (dac/nn/quantize.py:78-94)."""
    B, c, T = lat.shape
    out = torch.empty(B, c // 8, T, dtype=torch.long)
    for i in range(c // 8):
        e = torch.nn.functional.normalize(lat[:, 8 * i: 8 * i + 8].double().permute(0, 2, 1).reshape(-1, 8), dim=1)
        cb = torch.nn.functional.normalize(rvq.quantizers[i].codebook.weight.detach().double(), dim=1)
        out[:, i] = (e @ cb.t()).argmax(1).reshape(B, T)
    return out


def test_no_mismatch_and_genuine_mismatch():
    rvq = _rvq()
    lat = torch.randn(2, 24, 5)
    codes = _nearest(rvq, lat)
    assert classify_code_mismatches(rvq, lat, codes, codes.numpy()) == dict(mismatches=0, near_tie=0, cascade=0, genuine=0, worst_gap=0.0)
    wrong = codes.clone()
    wrong[1, 0, 3] = (wrong[1, 0, 3] + 1) % 8            # a clearly farther code in stage 0, and stage 2 of the same frame
    wrong[1, 2, 3] = (wrong[1, 2, 3] + 3) % 8
    r = classify_code_mismatches(rvq, lat, codes, wrong)
    assert r["mismatches"] == 2 and r["genuine"] == 2 and r["near_tie"] == 0 and r["worst_gap"] > 1e-3


def test_near_tie_flip_with_cascade():
    rvq = _rvq()
    cb = torch.nn.functional.normalize(rvq.quantizers[1].codebook.weight.detach(), dim=1)
    lat = torch.randn(1, 24, 4)
    # stage 1 of frame 2: a latent on the bisector of codes 3 and 5 (equidistant up to rounding), scaled arbitrarily
    lat[0, 8:16, 2] = 2.5 * (cb[3] + cb[5])
    got = _nearest(rvq, lat)
    assert int(got[0, 1, 2]) in (3, 5)
    expected = got.clone()
    expected[0, 1, 2] = 8 - int(got[0, 1, 2])            # the other one of the tied pair
    expected[0, 2, 2] = (expected[0, 2, 2] + 1) % 8      # the later stage saw a different residual: cascade
    r = classify_code_mismatches(rvq, lat, got, expected)
    assert r["mismatches"] == 2 and r["near_tie"] == 1 and r["cascade"] == 1 and r["genuine"] == 0 and r["worst_gap"] <= 1e-5
