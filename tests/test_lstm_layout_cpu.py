"""Host-side check of the operand layouts documented at the top of facodec_amd/csrc/lstm_persist.hip (no GPU): the index
formulas of `pack_whh16_kernel`, `frag_index` and `k_of_unit` are restated here and run through an emulation of
v_mfma_f32_16x16x4_f32 (A: lane l = row l%16, k l/16; B: lane l = k l/16, col l%16) -- the 16 waves' partial products must add up
to W_hh h (forward) and to the gate-quarter partial products of W_hh^T dgates (BPTT), and the 8 units x 16 columns a workgroup
publishes per step must be four whole 128-byte lines of their own (what the fresh-region exchange relies on).  The GPU tests in
tests/test_lstm_persist.py hold the kernels themselves to the oracle."""
import numpy as np
import pytest


def k_of_unit(u):
    u16, p, e = u & 15, (u & 15) >> 3, u & 7
    return (u & ~15) + ((e >> 1) << 2) + 2 * p + (e & 1)


def unit_of_k(k):
    r = k & 15
    jj, kq = r >> 2, r & 3
    return (k & ~15) + 8 * (kq >> 1) + 2 * jj + (kq & 1)


def frag_index(unit, col, H):
    KS, kw = H >> 6, H >> 4
    k = k_of_unit(unit)
    w, rem = divmod(k, kw)
    s, kq = rem >> 2, rem & 3
    j4, jj = s >> 2, s & 3
    cb, c16 = col >> 4, col & 15
    return ((((cb * 16 + w) * (KS >> 2) + j4) * 64 + kq * 16 + c16) << 2) + jj


def pack_whh16(w, H, transposed):
    n, KS, kw = 4 * H * H, H >> 6, H >> 4
    o = np.arange(n)
    comp, lane, rest = o & 3, (o >> 2) & 63, o >> 8
    j = rest % (KS >> 1)
    rest //= KS >> 1
    wv, blk = rest & 15, rest >> 4
    s, rb = 2 * j + (comp >> 1), comp & 1
    r32 = rb * 16 + (lane & 15)
    k = wv * kw + 4 * s + (lane >> 4)
    uk = np.array([unit_of_k(int(x)) for x in k])
    if not transposed:
        return w[(r32 >> 3) * H + blk * 8 + (r32 & 7), uk]
    return w[(blk & 3) * H + uk, (blk >> 2) * 32 + r32]


def workgroup_product(packed, blk, frag, H, ncb):
    """What wave_product + the LDS reduction of one workgroup compute: (32, ncb*16)."""
    KS = H >> 6
    p4, f4 = packed.reshape(-1, 64, 4), frag.reshape(-1, 64, 4)
    out = np.zeros((32, ncb * 16))
    lanes = np.arange(64)
    for wave in range(16):
        for cb in range(ncb):
            for j in range(KS // 4):
                b4 = f4[(cb * 16 + wave) * (KS // 4) + j]
                for jj in range(4):
                    s = 4 * j + jj
                    a = p4[(blk * 16 + wave) * (KS // 2) + (s >> 1)]
                    for rb in range(2):
                        av = a[:, 2 * (s & 1) + rb]
                        A, B = np.zeros((16, 4)), np.zeros((4, 16))
                        A[lanes % 16, lanes // 16] = av
                        B[lanes // 16, lanes % 16] = b4[:, jj]
                        out[rb * 16:(rb + 1) * 16, cb * 16:(cb + 1) * 16] += A @ B
    return out


def test_unit_permutation_is_a_bijection():
    assert all(unit_of_k(k_of_unit(u)) == u for u in range(2048))


@pytest.mark.parametrize("H,ncb", [(512, 1), (512, 2)])
def test_fragment_layouts_reproduce_the_matrix_products(H, ncb):
    rng = np.random.default_rng(0)
    nc = ncb * 16
    w, h, dg = rng.standard_normal((4 * H, H)), rng.standard_normal((H, nc)), rng.standard_normal((4 * H, nc))
    frag = np.zeros(H * nc)
    for u in range(H):
        for c in range(nc):
            frag[frag_index(u, c, H)] = h[u, c]
    pk, ref = pack_whh16(w, H, False), w @ h
    for ub in (0, 5, H // 8 - 1):                      # forward: workgroup ub owns units ub*8.., rows q*8 + u
        out = workgroup_product(pk, ub, frag, H, ncb)
        for q in range(4):
            assert np.allclose(out[q * 8:(q + 1) * 8], ref[q * H + ub * 8:q * H + ub * 8 + 8])
    pkt = pack_whh16(w, H, True)
    for ub, q in ((0, 0), (3, 2), (H // 32 - 1, 3)):   # BPTT: workgroup (ub, q) -> quarter-q part of W_hh^T dgates for 32 units
        fr = np.zeros(H * nc)
        for u in range(H):
            for c in range(nc):
                fr[frag_index(u, c, H)] = dg[q * H + u, c]
        out = workgroup_product(pkt, ub * 4 + q, fr, H, ncb)
        assert np.allclose(out, w[q * H:(q + 1) * H, ub * 32:ub * 32 + 32].T @ dg[q * H:(q + 1) * H])


@pytest.mark.parametrize("H", [512, 1024, 1536])
def test_a_workgroup_publishes_whole_cache_lines(H):
    for ub in range(H // 8):
        idx = sorted(frag_index(ub * 8 + u, c, H) for u in range(8) for c in range(16))
        assert idx == list(range(idx[0], idx[0] + 128)) and idx[0] % 32 == 0, ub      # 128 floats = four 128-byte lines
