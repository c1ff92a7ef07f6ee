"""Host-side checks of kernel index arithmetic (no GPU).  First part: the operand layouts documented at the top of facodec_amd/csrc/lstm_persist.hip: the index
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


# ----------------------------------------------------------------- bf16 x 3 resident forward (lstm_fwd_persist_split_kernel)
def _split3(x):
    """fp32 -> three round-to-nearest-even bf16 terms (as float32 arrays), the ls_split3 of the kernel."""
    import torch
    t = torch.as_tensor(np.asarray(x, dtype=np.float32))
    hi = t.to(torch.bfloat16).float()
    r1 = t - hi
    mid = r1.to(torch.bfloat16).float()
    lo = (r1 - mid).to(torch.bfloat16).float()
    return hi.numpy(), mid.numpy(), lo.numpy()


def pack_whh_split(w, H):
    """pack_whh_split_kernel: packed[((ub*8 + wave)*NK + s)*3 + plane][lane][i]; lane: row l%32 = gate*8 + unit, k = (wave*NK + s)*16 + 8*(l/32) + i."""
    NK = H // 128
    planes = _split3(w)
    out = np.zeros(((H // 8) * 8 * NK * 3, 64, 8), dtype=np.float32)
    lanes = np.arange(64)
    l31, kq = lanes & 31, lanes >> 5
    for ub in range(H // 8):
        rows = (l31 >> 3) * H + ub * 8 + (l31 & 7)
        for wv in range(8):
            for s in range(NK):
                k0 = (wv * NK + s) * 16 + 8 * kq
                for p in range(3):
                    out[((ub * 8 + wv) * NK + s) * 3 + p] = planes[p][rows[:, None], k0[:, None] + np.arange(8)[None, :]]
    return out


def exchange_pieces(h, H):
    """What the workgroups publish for one step: hs[plane][k/16][lane][i], lane = 32*((k%16)/8) + column, i = k%8; returns the array
    and, per workgroup, the byte ranges it writes in each plane."""
    planes = _split3(h)                                   # (H, 32) each
    hs = np.zeros((3, H // 16, 64, 8), dtype=np.float32)
    spans = []
    for ub in range(H // 8):
        S, half = ub >> 1, ub & 1
        for p in range(3):
            for col in range(32):
                hs[p, S, 32 * half + col] = planes[p][ub * 8:ub * 8 + 8, col]
        start = ((S * 64) + 32 * half) * 16               # byte offset inside a plane
        spans.append((start, start + 32 * 16))
    return hs, spans


def split_workgroup_product(packed, hs, ub, H):
    """Eight waves x NK steps x six bf16 products (TA / TB of the kernel) through an emulated v_mfma_f32_32x32x16_bf16
    (A: lane l = row l%32, k 8*(l/32)+i; B: lane l = col l%32, same k)."""
    NK = H // 128
    TA, TB = (1, 2, 0, 1, 0, 0), (1, 0, 2, 0, 1, 0)
    lanes = np.arange(64)
    out = np.zeros((32, 32))
    for wv in range(8):
        for s in range(NK):
            for ta, tb in zip(TA, TB):
                a = packed[((ub * 8 + wv) * NK + s) * 3 + ta]          # (64, 8)
                b = hs[tb, wv * NK + s]                               # (64, 8)
                A, B = np.zeros((32, 16)), np.zeros((16, 32))
                for i in range(8):
                    A[lanes & 31, 8 * (lanes >> 5) + i] = a[:, i]
                    B[8 * (lanes >> 5) + i, lanes & 31] = b[:, i]
                out += A.astype(np.float64) @ B.astype(np.float64)
    return out


@pytest.mark.parametrize("H", [512, 1024])
def test_split_resident_layouts_reproduce_w_hh_times_h(H):
    rng = np.random.default_rng(1)
    w = (rng.standard_normal((4 * H, H)) / np.sqrt(H)).astype(np.float32)
    h = np.tanh(rng.standard_normal((H, 32))).astype(np.float32)
    packed = pack_whh_split(w, H)
    hs, spans = exchange_pieces(h, H)
    ref = w.astype(np.float64) @ h.astype(np.float64)
    for ub in (0, 1, H // 16 + 1, H // 8 - 1):
        got = split_workgroup_product(packed, hs, ub, H)
        rows = np.array([(r >> 3) * H + ub * 8 + (r & 7) for r in range(32)])
        # six of the nine cross products of exactly split operands: fp32-grade agreement with the exact product
        assert np.abs(got - ref[rows]).max() < 2e-6 * np.abs(ref).max()
    # every workgroup writes, per plane, 512 contiguous bytes = four whole 128-byte lines, and no two workgroups share a line
    assert all(a % 128 == 0 and b - a == 512 for a, b in spans)
    assert len({a for a, _ in spans}) == len(spans) and sorted(a for a, _ in spans) == list(range(0, H * 64, 512))
    # the three terms of a split are an exact decomposition
    hi, mid, lo = _split3(h)
    assert np.array_equal((hi.astype(np.float64) + mid + lo).astype(np.float32), h)


# ----------------------------------------------------------------- tile walk of conv1d_bsplit_kernel (host-side restatement)
def _decode(v, n_tiles, n_t_tiles, B):
    """The XCD-aware decode of conv1d_bsplit.hip: virtual block id v -> (co tile, clip, time tile)."""
    q8, r8 = n_tiles >> 3, n_tiles & 7
    xcd, within = v & 7, v >> 3
    tid = (xcd * (q8 + 1) if xcd < r8 else r8 * (q8 + 1) + (xcd - r8) * q8) + within
    tt, rest = tid % n_t_tiles, tid // n_t_tiles
    return rest // B, rest % B, tt


@pytest.mark.parametrize("n_t_tiles,B,co_tiles,grid", [(188, 32, 2, 256), (4, 32, 12, 256), (7, 3, 5, 64), (1, 1, 9, 8), (94, 32, 3, 248)])
def test_tile_walk_visits_every_tile_once_and_stays_on_its_xcd(n_t_tiles, B, co_tiles, grid):
    """Workgroup b of a persistent launch handles v = b, b + grid, ... (grid a multiple of 8): together they must cover every
    (co tile, clip, time tile) exactly once, every workgroup must stay on one XCD's contiguous range (its blockIdx & 7), and the
    tiles of an XCD must form one contiguous id range -- what keeps a weight slab in that XCD's L2."""
    n_tiles = n_t_tiles * B * co_tiles
    seen = {}
    for b in range(min(grid, n_tiles)):
        for v in range(b, n_tiles, grid):
            assert (v & 7) == (b & 7)
            t = _decode(v, n_tiles, n_t_tiles, B)
            assert t not in seen
            seen[t] = b & 7
    assert len(seen) == n_tiles and all(0 <= c < co_tiles and 0 <= bb < B and 0 <= tt < n_t_tiles for c, bb, tt in seen)
    flat = sorted(((c * B + bb) * n_t_tiles + tt, x) for (c, bb, tt), x in seen.items())
    xs = [x for _, x in flat]
    assert xs == sorted(xs)                      # ids grouped by XCD in increasing order: contiguous ranges


def test_resident_split_lstm_is_refused_when_its_exchange_scratch_would_be_large():
    """the bf16 x 3 resident LSTM needs one fresh exchange region per step (T * H * 32 * 6 bytes per layer call):
    45 MB at the benchmark's 160 frames, 2.5 GB at 9 000 frames.  Above the budget the policy answers False before it touches the
    library, and the caller (layers.SLSTM.forward) takes the per-step kernels, which need no scratch."""
    from facodec_amd import ops
    assert 160 * 1536 * 32 * 6 < ops.LSTM_PERSIST_SPLIT_MAX_SCRATCH < 9000 * 1536 * 32 * 6
    assert ops.lstm_persist_split_ok(1536, 32, T=9000) is False
    assert ops.lstm_persist_split_ok(1024, 32, T=6000) is False          # 1.1 GB


def test_add_backward_never_hands_one_tensor_to_two_consumers():
    """Round 5's cross-stream autograd hazard: the engine accumulates IN PLACE into a buffered gradient once it is
    uniquely owned, so a backward that returns one tensor object for two inputs lets a consumer on one stream write what a consumer
    on another stream is still reading.  `A.add`'s backward must return two distinct tensors with equal values."""
    import torch
    from facodec_amd import autograd as A
    dy = torch.arange(12, dtype=torch.float32).reshape(3, 4)
    ga, gb = A._Add.backward(None, dy)
    assert torch.equal(ga, dy) and torch.equal(gb, dy)
    assert ga.data_ptr() != gb.data_ptr()
    # and no other Function of the training path passes an incoming gradient through to two inputs
    import inspect
    import re
    from facodec_amd import autograd_disc, autograd_pred, autograd_quant
    for mod in (A, autograd_disc, autograd_pred, autograd_quant):
        src = inspect.getsource(mod)
        for m in re.finditer(r"def backward\(ctx, (\w+)[^)]*\):(.*?)(?=\n    @staticmethod|\nclass |\ndef |\Z)", src, re.S):
            name, body = m.group(1), m.group(2)
            for ret in re.findall(r"return ([^\n]+)", body):
                parts = [p.strip() for p in ret.split(",")]
                assert parts.count(name) <= 1, (mod.__name__, ret)
