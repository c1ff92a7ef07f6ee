"""The one-launch-per-layer SLSTM recurrence and BPTT (facodec_amd/csrc/lstm_persist.hip) against the CPU oracle
(dac/model/encodec.py:272-288 restated in oracle/) and against the per-step kernels of lstm.hip on the same inputs."""
import pytest
import torch

from facodec_amd import synth

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = torch.as_tensor(a).detach().double().cpu(), torch.as_tensor(b).detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def _g(seed=0):
    return torch.Generator().manual_seed(seed)


@pytest.fixture(scope="module")
def cuda():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def ops():
    from facodec_amd import ops
    old, old_split = ops.LSTM_PERSIST_MAX_BATCH, ops.LSTM_PERSIST_SPLIT
    ops.LSTM_PERSIST_MAX_BATCH = 32        # the policy stops at 16 columns; the kernels are held to parity up to 32
    ops.LSTM_PERSIST_SPLIT = False         # ... and 17 .. 32 columns go to the bf16 x 3 kernel, which has its own tests below
    yield ops
    ops.LSTM_PERSIST_MAX_BATCH, ops.LSTM_PERSIST_SPLIT = old, old_split


@pytest.fixture
def split_ops(ops):
    ops.LSTM_PERSIST_SPLIT = True
    yield ops
    ops.LSTM_PERSIST_SPLIT = False


@pytest.fixture(scope="module")
def O():
    from oracle import facodec_oracle
    return facodec_oracle


@pytest.mark.parametrize("B,H,T", [(16, 1536, 12), (17, 512, 9), (32, 1024, 7), (1, 512, 1), (5, 1024, 2)])
def test_resident_slstm_against_oracle(B, H, T, O, ops, cuda):
    from facodec_amd.layers import SLSTM
    assert ops.lstm_persist_ok(H, B), "the resident kernel must be the one that runs here"
    m = SLSTM(H, 2)
    sd = synth.load_synthetic(m, seed=5)
    x = torch.randn(B, H, T, generator=_g(H + T))
    y = O.slstm(x, sd, "lstm.", 2)
    with torch.no_grad():
        yg = m.to(cuda)(x.to(cuda))
    assert rel(yg, y) < 1e-5


@pytest.mark.parametrize("B,H,T", [(16, 1536, 160), (16, 1024, 160), (24, 512, 33), (32, 1536, 160), (32, 1024, 40)])
def test_resident_layer_matches_per_step_kernels(B, H, T, ops, cuda):
    """Forward (h sequence, saved gates / cell states) and BPTT (dgates) of one layer, resident vs per-step launch path."""
    g = _g(B + H)
    BP = 32 * ((B + 31) // 32)
    pre = torch.zeros(4 * H, T, BP)
    pre[:, :, :B] = torch.randn(4 * H, T, B, generator=g)
    w_hh = (torch.rand(4 * H, H, generator=g) * 2 - 1) / H ** 0.5
    pre, w_hh = pre.to(cuda), w_hh.to(cuda)
    gates_a, cs_a = torch.empty(4 * H, T, BP, device=cuda), torch.empty(H, T, BP, device=cuda)
    gates_b, cs_b = torch.empty_like(gates_a), torch.empty_like(cs_a)
    ya = ops.lstm_layer_persist(pre, w_hh, H, B, save=(gates_a, cs_a))
    yb = ops.lstm_layer(pre, ops.pack_lstm_whh(w_hh), H, save=(gates_b, cs_b))
    assert rel(ya[:, :, :B], yb[:, :, :B]) < 2e-6
    assert rel(gates_a[:, :, :B], gates_b[:, :, :B]) < 2e-6
    assert rel(cs_a[:, :, :B], cs_b[:, :, :B]) < 2e-6
    assert float(ya[:, :, 16 * ((B + 15) // 16):].abs().max() if BP > 16 * ((B + 15) // 16) else 0.0) == 0.0
    d_out = torch.zeros(H, T, BP)
    d_out[:, :, :B] = torch.randn(H, T, B, generator=g)
    d_out = d_out.to(cuda)
    dga = ops.lstm_layer_bwd(d_out, w_hh, gates_b, cs_b, H, batch=B)
    dgb = ops.lstm_layer_bwd(d_out, w_hh, gates_b, cs_b, H, batch=None)
    assert torch.isfinite(dga).all()
    assert rel(dga[:, :, :B], dgb[:, :, :B]) < 5e-6
    assert float(dga[:, :, B:].abs().max() if BP > B else 0.0) == 0.0      # zero dy columns stay exactly zero


def test_resident_layers_replay_in_a_graph(ops, cuda):
    """The exchange flags carry their own epoch: the same captured launch replays without a host-side reset."""
    B, H, T = 16, 512, 20
    g = _g(3)
    pre = torch.zeros(4 * H, T, 32)
    pre[:, :, :B] = torch.randn(4 * H, T, B, generator=g)
    w_hh = ((torch.rand(4 * H, H, generator=g) * 2 - 1) / H ** 0.5).to(cuda)
    pre = pre.to(cuda)
    ref = ops.lstm_layer_persist(pre, w_hh, H, B).clone()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        ops.lstm_layer_persist(pre, w_hh, H, B)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s):
            out = ops.lstm_layer_persist(pre, w_hh, H, B)
            out2 = ops.lstm_layer_persist(pre, w_hh, H, B)          # a second resident launch behind the first
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, ref) and torch.equal(out2, ref)


def test_slstm_training_step_resident_vs_per_step(ops, cuda):
    """The whole SLSTM autograd node (two layers, skip): input and parameter gradients with the resident kernels against
    the per-step path."""
    from facodec_amd import autograd as A
    from facodec_amd.layers import SLSTM
    B, H, T = 5, 512, 11
    m = SLSTM(H, 2)
    synth.load_synthetic(m, seed=9)
    m = m.to(cuda)
    x = torch.randn(B, H, T, generator=_g(1)).to(cuda)
    r = torch.randn(B, H, T, generator=_g(2)).to(cuda)

    def run(persist):
        old = ops.LSTM_PERSIST
        ops.LSTM_PERSIST = persist
        try:
            for p in m.parameters():
                p.grad = None
            xx = x.clone().requires_grad_()
            y = A.slstm(m, xx)
            (y * r).sum().backward()
            return y.detach(), xx.grad, {n: p.grad.clone() for n, p in m.named_parameters()}
        finally:
            ops.LSTM_PERSIST = old

    ya, dxa, ga = run(True)
    yb, dxb, gb = run(False)
    assert rel(ya, yb) < 2e-6 and rel(dxa, dxb) < 1e-5
    for n in ga:
        assert rel(ga[n], gb[n]) < 1e-5, n


# ------------------------------------------------------------------ resident forward with bf16 x 3 operands (17 .. 32 columns)
@pytest.mark.parametrize("B,H,T", [(32, 1536, 160), (32, 1024, 160), (20, 512, 33), (17, 1024, 5), (32, 1536, 1), (32, 512, 2)])
def test_split_resident_layer_matches_per_step_kernels(B, H, T, split_ops, cuda):
    """fac_lstm_layer_fwd_persist_split (W_hh . h as six bf16 products of exactly split operands, fp32 accumulation) against the
    per-step fp32-MFMA kernel on the same layer: fp32-grade agreement over the whole sequence, twice the same bits."""
    ops = split_ops
    assert ops.lstm_persist_split_ok(H, B), "the split resident kernel must be the one that runs here"
    g = _g(B + H + T)
    pre = torch.zeros(4 * H, T, 32)
    pre[:, :, :B] = torch.randn(4 * H, T, B, generator=g)
    w_hh = (torch.rand(4 * H, H, generator=g) * 2 - 1) / H ** 0.5
    pre, w_hh = pre.to(cuda), w_hh.to(cuda)
    ya = ops.lstm_layer_persist_split(pre, w_hh, H, B)
    yb = ops.lstm_layer(pre, ops.pack_lstm_whh(w_hh), H)
    assert torch.isfinite(ya).all()
    assert rel(ya[:, :, :B], yb[:, :, :B]) < 1e-5
    assert torch.equal(ops.lstm_layer_persist_split(pre, w_hh, H, B), ya)
    # against float64 on the host, beside the fp32 kernel's own error (the bar of the split conv kernels)
    if T <= 33:
        w64, p64 = w_hh.double().cpu(), pre.double().cpu()
        h, c, out = torch.zeros(H, 32, dtype=torch.float64), torch.zeros(H, 32, dtype=torch.float64), []
        for t in range(T):
            gts = p64[:, t] + w64 @ h
            i, f, gg, o = gts[:H].sigmoid(), gts[H:2 * H].sigmoid(), gts[2 * H:3 * H].tanh(), gts[3 * H:].sigmoid()
            c = f * c + i * gg
            h = o * c.tanh()
            out.append(h)
        ref = torch.stack(out, 1)
        e_split, e_fp32 = rel(ya[:, :, :B], ref[:, :, :B]), rel(yb[:, :, :B], ref[:, :, :B])
        assert e_split < 1.5 * e_fp32 + 1e-7, (e_split, e_fp32)


@pytest.mark.parametrize("B,H,T", [(32, 1024, 7), (17, 512, 9), (24, 1536, 12)])
def test_split_resident_slstm_against_oracle(B, H, T, O, split_ops, cuda):
    from facodec_amd.layers import SLSTM
    assert split_ops.lstm_persist_split_ok(H, B)
    m = SLSTM(H, 2)
    sd = synth.load_synthetic(m, seed=5)
    x = torch.randn(B, H, T, generator=_g(H + T))
    y = O.slstm(x, sd, "lstm.", 2)
    with torch.no_grad():
        yg = m.to(cuda)(x.to(cuda))
    assert rel(yg, y) < 1e-5


def test_split_resident_layer_replays_in_a_graph(split_ops, cuda):
    ops = split_ops
    B, H, T = 32, 512, 20
    g = _g(4)
    pre = torch.randn(4 * H, T, 32, generator=g).to(cuda)
    w_hh = ((torch.rand(4 * H, H, generator=g) * 2 - 1) / H ** 0.5).to(cuda)
    ref = ops.lstm_layer_persist_split(pre, w_hh, H, B).clone()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        ops.lstm_layer_persist_split(pre, w_hh, H, B)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s):
            out = ops.lstm_layer_persist_split(pre, w_hh, H, B)
            out2 = ops.lstm_layer_persist(pre[:, :, :32].contiguous(), w_hh, H, 16)     # an fp32 resident launch behind it
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, ref) and torch.isfinite(out2).all()


def test_poisoned_step_moves_nothing_and_clean_step_is_unchanged(cuda):
    """a resident-LSTM wait that gives up leaves meaningless results behind; the optimiser must not apply them.  The
    abort word rides behind the per-parameter flags (`FlatAdamW._poison`, written by fac_lstm_abort_flag, exchanged with the
    flags) and a positive value clears every flag in front of the masked AdamW step (fac_mask_flags_if): no parameter, moment or
    step count moves.  With the word at 0 (this process never timed out) the step is the one torch.optim.AdamW takes."""
    from facodec_amd import ops as _ops
    from facodec_amd.optim import FlatAdamW
    assert _ops.lstm_persist_ok(512, 4) and _ops.lstm_timeouts() == 0         # arms the device's counter word on the way
    g = _g(11)
    ref = [torch.randn(33, 7, generator=g).requires_grad_(), torch.randn(19, generator=g).requires_grad_()]
    ours = [torch.nn.Parameter(r.detach().clone().to(cuda)) for r in ref]
    opt = FlatAdamW(ours, lr=1e-2)
    grads = [torch.randn(33, 7, generator=g), torch.randn(19, generator=g)]

    def backward():
        for p, gr in zip(ours, grads):
            p.grad.copy_(gr.to(cuda))
        opt.mark_grads()

    before = [p.detach().clone() for p in ours]
    backward()
    opt.exchange_for_step()
    assert float(opt._poison) == 0.0                    # what the device reports about itself
    opt._poison.fill_(1.0)                              # ... and what a rank whose recurrence bailed out would have contributed
    opt._step_kernels()
    opt.end_step()
    assert all(torch.equal(a, p.detach()) for a, p in zip(before, ours)) and opt.param_steps == [0, 0]
    assert float(opt.m.abs().max()) == 0.0 and float(opt.v.abs().max()) == 0.0
    backward()
    opt.step()
    t = torch.optim.AdamW(ref, lr=1e-2, betas=(0.9, 0.98), eps=1e-9, weight_decay=0.1)
    for r, gr in zip(ref, grads):
        r.grad = gr.clone()
    t.step()
    assert opt.param_steps == [1, 1]
    for r, p in zip(ref, ours):
        assert rel(p, r) < 1e-6
