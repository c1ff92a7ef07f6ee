"""N>1 path on CPU: world_size-2 gloo processes exercise the bench scaffold (sharding, barrier,
MAX-over-ranks timing, SUM of units).  The data path itself has no collective."""
import os
import socket
import sys

import torch
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _portable(x):
    """Tensors cross the result queue BY VALUE (numpy bytes): torch's default reduction hands over a shared-memory file
    descriptor through a socket to the producing process, and a worker that has already exited resets that connection
    (the suite's one flaky failure: ConnectionResetError in q.get)."""
    if torch.is_tensor(x):
        return ("__tensor__", x.detach().cpu().numpy().copy())
    if isinstance(x, dict):
        return {k: _portable(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return type(x)(_portable(v) for v in x)
    return x


def _restore(x):
    if isinstance(x, tuple) and len(x) == 2 and isinstance(x[0], str) and x[0] == "__tensor__":
        return torch.from_numpy(x[1])
    if isinstance(x, dict):
        return {k: _restore(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return type(x)(_restore(v) for v in x)
    return x


def _run_ranks(worker, world=2, *args):
    """Spawns `world` gloo ranks of `worker(rank, world, port, q, *args)`; returns their results ordered by rank (each worker
    puts ONE tuple whose first element is its rank)."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, world, port, q) + args) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((_restore(q.get(timeout=180)) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def _worker(rank, world, port, q):
    sys.path.insert(0, REPO)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import time
    from facodec_amd import benchutil, synth
    r, lr, w = benchutil.init_distributed("gloo")
    lo, hi = benchutil.shard_clips(7, w, r)
    clips = synth.synth_clips(hi - lo, 2400, seed=0, rank=r)
    calls = []

    def step():
        calls.append(1)
        time.sleep(0.02 * (r + 1))   # rank 1 is slower: MAX must pick it up

    dt = benchutil.timed_steps(step, steps=3, warmup=1, sync_fn=lambda: None)
    units = benchutil.aggregate_units((hi - lo) * 3)
    q.put(_portable((r, lo, hi, len(calls), dt, units, float(clips.abs().max()))))
    torch.distributed.destroy_process_group()


def test_two_rank_gloo_scaffold():
    res = _run_ranks(_worker)
    (r0, lo0, hi0, c0, dt0, u0, m0), (r1, lo1, hi1, c1, dt1, u1, m1) = res
    assert (lo0, hi0, lo1, hi1) == (0, 4, 4, 7)          # disjoint cover of the 7 clips
    assert c0 == c1 == 4                                  # 1 warm-up + exactly 3 timed steps
    assert abs(dt0 - dt1) < 1e-9 and dt0 >= 3 * 0.04 * 0.9   # both report the slow rank's time
    assert u0 == u1 == 21.0                               # (4 + 3) clips x 3 steps
    assert m0 == m1 == 1.0                                # peak-normalised clips, different per rank


def test_shard_clips_edge_cases():
    sys.path.insert(0, REPO)
    from facodec_amd.benchutil import shard_clips
    assert [shard_clips(5, 8, r) for r in range(8)] == [(0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (5, 5), (5, 5), (5, 5)]
    assert shard_clips(0, 2, 1) == (0, 0)
    assert [shard_clips(64, 2, r) for r in range(2)] == [(0, 32), (32, 64)]


def _grad_worker(rank, world, port, q):
    sys.path.insert(0, REPO)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from facodec_amd import benchutil
    from facodec_amd.optim import FlatAdamW
    benchutil.init_distributed("gloo")
    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.randn(5, 3)), torch.nn.Parameter(torch.randn(7)), torch.nn.Parameter(torch.randn(2, 2))]
    opt = FlatAdamW(params)
    assert params[0].data.data_ptr() == opt.p.data_ptr()          # parameters live in the arena
    params[0].grad = torch.full((5, 3), float(rank + 1))
    params[1].grad = torch.arange(7, dtype=torch.float32) * (rank + 1)
    # params[2] has no gradient on any rank: counts as zeros
    opt.gather_grads()
    opt.all_reduce_mean()                                           # ONE collective over the whole arena
    q.put(_portable((rank, opt.g.clone())))
    torch.distributed.destroy_process_group()


def test_two_rank_gradient_arena_all_reduce():
    """Data-parallel exchange of the training step (facodec_amd/optim.py): every rank ends with the mean gradient,
    one all-reduce per model key, unused parameters as zeros."""
    res = _run_ranks(_grad_worker)
    g0, g1 = res[0][1], res[1][1]
    assert torch.equal(g0, g1)
    assert torch.allclose(g0[:15], torch.full((15,), 1.5))
    assert torch.allclose(g0[15:22], torch.arange(7, dtype=torch.float32) * 1.5)
    assert torch.equal(g0[22:], torch.zeros(4))


def _overlap_worker(rank, world, port, q):
    sys.path.insert(0, REPO)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from facodec_amd import benchutil
    from facodec_amd.optim import FlatAdamW
    benchutil.init_distributed("gloo")
    torch.manual_seed(0)
    keys = {}
    for name, shapes in (("decoder", [(4, 3), (5,)]), ("unused", [(6,)]), ("encoder", [(2, 2)])):
        keys[name] = FlatAdamW([torch.nn.Parameter(torch.randn(*s)) for s in shapes])
    # backward: autograd accumulates straight into the arenas through the .grad views
    for i, p in enumerate(keys["decoder"].params):
        (p * float(rank + 1) * (i + 1)).sum().backward()
    assert keys["decoder"].params[0].grad.data_ptr() == keys["decoder"].g.data_ptr()      # views, no gather copy
    assert keys["decoder"].params_without_grad() == [] and keys["unused"].params_without_grad() == [0]
    # the decoder's exchange leaves from "inside backward" (a hook): no knowledge of earlier steps is needed or used
    keys["decoder"].launch_all_reduce(from_hook=True)
    early_first_step = keys["decoder"]._work is not None
    (keys["encoder"].params[0] * float(rank + 1)).sum().backward()
    # end of backward: every key launches in the same fixed order on every rank, including the key without any gradient
    for k in ("decoder", "unused", "encoder"):
        keys[k].launch_all_reduce()
    for k in ("encoder", "decoder", "unused"):          # waited for in a different order (the optimiser-step order)
        keys[k].wait_all_reduce()
    res = {k: o.g.clone() for k, o in keys.items()}
    logs = {k: list(o._launch_log) for k, o in keys.items()}
    # second iteration: the ranks reach DIFFERENT parameter sets (rank 0 only parameter 0, rank 1 only parameter 1) and both
    # launch from the hook: same collectives in the same order, unreached parameters count as zeros
    for o in keys.values():
        o.zero_grad()
    (keys["decoder"].params[rank] * 2.0).sum().backward()
    keys["decoder"].launch_all_reduce(from_hook=True)
    keys["decoder"].wait_all_reduce()
    q.put(_portable((rank, res, early_first_step, logs, keys["decoder"].g.clone(), keys["decoder"]._flags.clone())))
    torch.distributed.destroy_process_group()


def test_two_rank_async_exchange_order_and_empty_key():
    """Asynchronous exchange per model key: gradients live in the arena (views), keys launch in a fixed order and are waited for
    in another, a key whose parameters get no gradient still takes part (zeros) so the collectives stay matched, a launch from
    inside backward needs no history, and two ranks that reach DIFFERENT parameters neither deadlock nor drift: every rank ends
    with the mean (zeros for the rank that did not reach a parameter) and with the union of the flags."""
    res = _run_ranks(_overlap_worker)
    (_, g0, e0, log0, d0, f0), (_, g1, e1, log1, d1, f1) = res
    for k in g0:
        assert torch.equal(g0[k], g1[k])
    assert torch.allclose(g0["decoder"][:12], torch.full((12,), 1.5)) and torch.allclose(g0["decoder"][12:], torch.full((5,), 3.0))
    assert torch.equal(g0["unused"], torch.zeros(6)) and torch.allclose(g0["encoder"], torch.full((4,), 1.5))
    assert e0 and e1 and log0 == log1
    assert log0["decoder"] == [(0, "hook")] and log0["unused"] == [(0, "end")]
    assert torch.equal(d0, d1) and torch.allclose(d0, torch.full((17,), 1.0))          # (2 + 0) / 2 for both parameters
    assert torch.equal(f0 > 0, torch.tensor([True, True])) and torch.equal(f0, f1)


class _Chain(torch.nn.Module):
    """Four 'blocks' in a chain, like the encoder / decoder: x -> b0 -> b1 -> b2 -> b3."""

    def __init__(self):
        super().__init__()
        self.blocks = torch.nn.ModuleList([torch.nn.Linear(8, 8) for _ in range(4)])

    def forward(self, x, hook=None):
        for k, b in enumerate(self.blocks):
            if hook is not None:
                hook(k, x)
            x = torch.tanh(b(x))
        return x


def _bucket_worker(rank, world, port, q):
    sys.path.insert(0, REPO)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from facodec_amd import benchutil
    from facodec_amd.optim import FlatAdamW
    benchutil.init_distributed("gloo")
    out = {}
    for mode in ("one_bucket", "progressive"):
        torch.manual_seed(0)
        net = _Chain()
        # 72 floats per block (64 weights + 8 biases): 300-byte buckets hold one block each, counted from the end
        opt = FlatAdamW(net.parameters(), bucket_bytes=0 if mode == "one_bucket" else 280)
        first = [2 * k for k in range(4)]                   # index of block k's first parameter (weight, bias per block)
        events = []

        def hook(k, x, opt=opt, events=events):             # gradient of the activation entering block k: blocks k + 1 .. are final
            if x.requires_grad and k + 1 < 4:
                def fire(g, k=k):
                    opt.launch_all_reduce(from_param=first[k + 1], from_hook=True)
                    events.append((k, opt._next_bucket))
                x.register_hook(fire)

        opt.zero_grad(unbind=True)
        x = torch.randn(5, 8, generator=torch.Generator().manual_seed(10 + rank)).requires_grad_()   # a different batch per rank
        net(x, hook if mode == "progressive" else None).pow(2).sum().backward()
        in_flight_at_end = len(opt._works)
        opt.launch_all_reduce()
        opt.wait_all_reduce()
        out[mode] = (opt.g.clone(), list(opt._launch_log), events, in_flight_at_end, len(opt.buckets))
    # a gradient that shows up after its bucket has left is an error, not a silent skip
    opt.zero_grad(unbind=True)
    net(torch.randn(2, 8)).sum().backward()
    opt.launch_all_reduce(from_param=first[3], from_hook=True)
    net.blocks[3].weight.grad = torch.ones(8, 8)            # "late": a new tensor on a parameter of the launched bucket
    try:
        opt.gather_grads()
        late_raises = False
    except RuntimeError:
        late_raises = True
    opt._works, opt._flags_final = [], False                # (the other rank raised at the same point: nothing left to match)
    q.put(_portable((rank, out, late_raises)))
    torch.distributed.destroy_process_group()


def test_two_rank_bucketed_progressive_exchange():
    """The arena leaves in buckets counted from its end; buckets whose parameters are final are launched from activation-gradient
    hooks while backward is still running (what train.py installs at the encoder / decoder block boundaries).  Same averaged
    gradients as one collective over the whole arena, same launch sequence on both ranks although their batches differ, only the
    first bucket (the remainder) is left for the end, and a late gradient raises."""
    res = _run_ranks(_bucket_worker)
    (_, o0, late0), (_, o1, late1) = res
    for o in (o0, o1):
        assert torch.allclose(o["one_bucket"][0], o["progressive"][0], atol=1e-7)
        assert o["one_bucket"][4] == 1 and o["progressive"][4] == 4
        assert o["progressive"][1] == [(0, "hook"), (1, "hook"), (2, "hook"), (3, "end")]
        assert o["progressive"][2] == [(2, 1), (1, 2), (0, 3)]         # boundary of block k fires -> buckets of blocks k + 1 .. are out
        assert o["progressive"][3] == 3                                # three collectives already in flight when backward ends
    assert torch.equal(o0["progressive"][0], o1["progressive"][0]) and o0["progressive"][1] == o1["progressive"][1]
    assert late0 and late1


def _accumulation_worker(rank, world, port, q):
    sys.path.insert(0, REPO)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from facodec_amd import benchutil
    from facodec_amd.optim import FlatAdamW
    benchutil.init_distributed("gloo")
    torch.manual_seed(0)
    net = _Chain()
    opt = FlatAdamW(net.parameters(), bucket_bytes=280)
    x = torch.randn(5, 8, generator=torch.Generator().manual_seed(20 + rank))
    # (a) two backward passes before one step, the first one's buckets already in flight: the second accumulates IN PLACE into
    # the bound views (no foreign tensor for _rebind to notice) -- the post-accumulate hook sees it, step raises
    opt.zero_grad(unbind=True)
    net(x).pow(2).sum().backward()
    opt.launch_all_reduce(from_param=2 * 2, from_hook=True)         # buckets of blocks 2 and 3 have left
    net(x).pow(2).sum().backward()
    late = sorted(set(opt._late_accum))
    try:
        opt.exchange_for_step()
        raised = False
    except RuntimeError:
        raised = True
    opt.launch_all_reduce()                                         # (both ranks raised at the same point: finish the key's collectives)
    opt.wait_all_reduce()
    opt.zero_grad(unbind=True)
    # (b) accumulation done right: both backward passes first, the exchange after the last one
    net(x).pow(2).sum().backward()
    net(x).pow(2).sum().backward()
    opt.exchange_for_step()
    acc = opt.g.clone()
    opt.end_step(zero_grad=True, advance_lr=False)
    # (c) step(zero_grad=False) -- what TrainStep uses -- followed by another backward + step WITHOUT zero_grad: every bucket is
    # exchanged again (it used to re-exchange the flags only and apply un-averaged gradients)
    opt.zero_grad(unbind=True)
    net(x).pow(2).sum().backward()
    opt.exchange_for_step()
    first = opt.g.clone()
    opt.end_step(zero_grad=False, advance_lr=False)
    net(x).pow(2).sum().backward()                                  # adds this rank's gradient onto the averaged one
    opt.exchange_for_step()
    second, log = opt.g.clone(), list(opt._launch_log)
    opt.end_step(zero_grad=False, advance_lr=False)
    q.put(_portable((rank, late, raised, acc, first, second, log, [list(e) for e in opt.exchange_log])))
    torch.distributed.destroy_process_group()


def test_two_rank_accumulation_and_repeated_step_without_zero_grad():
    """(a) a second backward while the first one's buckets are in flight is caught by the post-accumulate hook (the
    in-place add into the bound view is invisible to `_rebind`) and raised by the step; (b) accumulating over two backward passes
    with the exchange after the last one gives the mean of the summed gradients on both ranks; (c) after `step(zero_grad=False)` a
    further backward + step exchanges every bucket again: both ranks hold the same arena."""
    (_, late0, r0, acc0, f0, s0, log0, elog0), (_, late1, r1, acc1, f1, s1, log1, elog1) = _run_ranks(_accumulation_worker)
    assert r0 and r1 and late0 == late1 == [4, 5, 6, 7]            # weights and biases of blocks 2, 3 (launched), not 0, 1
    assert torch.equal(acc0, acc1) and torch.equal(f0, f1)
    assert torch.allclose(acc0, 2 * f0, atol=1e-6)                  # two identical passes accumulated, then averaged
    assert torch.equal(s0, s1)                                      # the second step's exchange really averaged
    assert torch.allclose(s0, 2 * f0, atol=1e-6)                    # averaged first gradient + mean of the second pass
    assert log0 == log1 == [(0, "end"), (1, "end"), (2, "end"), (3, "end")]
    assert elog0[-1] == log0 and elog0[-2] == log0


def _flag_worker(rank, world, port, q):
    sys.path.insert(0, REPO)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from facodec_amd import benchutil
    from facodec_amd.optim import FlatAdamW
    benchutil.init_distributed("gloo")
    out = {}
    for mode, dp in (("arena", True), ("ddp_wrapped", False)):
        torch.manual_seed(0)
        params = [torch.nn.Parameter(torch.randn(3, 2)), torch.nn.Parameter(torch.randn(4)), torch.nn.Parameter(torch.randn(5))]
        opt = FlatAdamW(params, data_parallel=dp)
        (params[0] * float(rank + 1)).sum().backward()                      # every rank reaches parameter 0
        if rank == 0:
            (params[1] * 3.0).sum().backward()                              # only rank 0 reaches parameter 1
        # parameter 2: no rank
        local = list(opt._touched)
        opt.all_reduce_mean()
        out[mode] = (local, opt._flags.clone(), opt.g.clone())
    # in-place writes into the views are invisible to the hooks until declared
    opt.zero_grad()
    params[2].grad.add_(1.0)
    hidden = list(opt._touched)
    opt.mark_grads([params[2]])
    q.put(_portable((rank, out, hidden, list(opt._touched))))
    torch.distributed.destroy_process_group()


def test_two_rank_touched_flags_travel_with_the_gradients():
    """which parameters are stepped must be the UNION over ranks -- a parameter reached on rank 0 only
    receives the averaged gradient on both ranks and must be stepped on both.  The flags ride at the tail of the arena
    through the same all-reduce (or, for DDP-wrapped modules whose gradients DDP averages, through a MAX of the flags)."""
    res = _run_ranks(_flag_worker)
    (_, o0, hid0, mk0), (_, o1, _, _) = res
    assert o0["arena"][0] == [True, True, False] and o1["arena"][0] == [True, False, False]      # local views differ
    for mode in ("arena", "ddp_wrapped"):
        f0, f1 = o0[mode][1], o1[mode][1]
        assert torch.equal(f0 > 0, torch.tensor([True, True, False])) and torch.equal(f0 > 0, f1 > 0), (mode, f0, f1)
    assert torch.equal(o0["arena"][2], o1["arena"][2])                                          # averaged gradients agree
    assert torch.allclose(o0["arena"][2][6:10], torch.full((4,), 1.5))                          # (3 + 0) / 2 on BOTH ranks
    assert torch.allclose(o1["ddp_wrapped"][2][6:10], torch.zeros(4))                           # no arena exchange in DDP mode
    assert hid0 == [False, False, False] and mk0 == [False, False, True]


def test_unbound_gradients_are_folded_in_one_copy():
    """FlatAdamW.zero_grad(unbind=True): autograd keeps its gradient tensors (`.grad` is None during backward, no
    add-into-the-view launch per parameter); `gather_grads` / `launch_all_reduce` / `step` fold them into the arena, mark the
    parameters and bind the views again.  Same arena contents as the bound mode."""
    sys.path.insert(0, REPO)
    from facodec_amd.optim import FlatAdamW
    torch.manual_seed(3)
    shapes = [(4, 3), (5,), (2, 2, 2), (6,)]

    def run(unbind):
        torch.manual_seed(4)
        params = [torch.nn.Parameter(torch.randn(*s)) for s in shapes]
        opt = FlatAdamW(params)
        opt.zero_grad(unbind=unbind)
        assert all((p.grad is None) == unbind for p in params)
        loss = sum((p * float(i + 1)).pow(2).sum() for i, p in enumerate(params[:3]))      # params[3] is not reached
        loss = loss + params[0].sum()                                                        # two contributions to params[0]
        loss.backward()
        if unbind:
            assert params[0].grad.data_ptr() != opt.g.data_ptr()                             # autograd's own tensor
            assert params[3].grad is None
        opt.gather_grads()
        assert all(p.grad.data_ptr() == opt.g.data_ptr() + 4 * off for p, (off, _) in zip(params, opt.slices))
        assert opt.params_without_grad() == [3]
        return opt.g.clone()

    assert torch.equal(run(True), run(False))


def _unbound_worker(rank, world, port, q):
    sys.path.insert(0, REPO)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from facodec_amd import benchutil
    from facodec_amd.optim import FlatAdamW
    benchutil.init_distributed("gloo")
    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.randn(4, 3)), torch.nn.Parameter(torch.randn(5)), torch.nn.Parameter(torch.randn(6))]
    opt = FlatAdamW(params)
    log = []
    for it in range(2):
        opt.zero_grad(unbind=True)                       # what TrainStep does: autograd hands its gradient tensors over
        assert all(p.grad is None for p in params)
        (params[0] * float(rank + 1)).sum().backward()
        if rank == 0:                                    # params[1] is reached on rank 0 only; params[2] on no rank
            (params[1] * 3.0).sum().backward()
        opt.launch_all_reduce()                          # folds the stolen tensors into the arena, then ONE collective
        opt.wait_all_reduce()
        assert all(p.grad.data_ptr() == opt.g.data_ptr() + 4 * off for p, (off, _) in zip(params, opt.slices))   # bound again
        log.append((opt.g.clone(), opt._flags.clone()))
        opt._expected = tuple(opt._touched)
        opt._flags_final = False
    q.put(_portable((rank, log)))
    torch.distributed.destroy_process_group()


def test_two_rank_unbound_gradients_fold_and_exchange():
    """zero_grad(unbind=True) under two ranks: the gradients autograd kept are folded into the arena by launch_all_reduce, the
    exchange averages them, the per-parameter flags say which parameters ANY rank reached, and the views are bound again."""
    res = _run_ranks(_unbound_worker)
    for it in range(2):
        (g0, f0), (g1, f1) = res[0][1][it], res[1][1][it]
        assert torch.equal(g0, g1) and torch.equal(f0, f1)
        assert torch.allclose(g0[:12], torch.full((12,), 1.5))                  # mean of 1 and 2
        assert torch.allclose(g0[12:17], torch.full((5,), 1.5))                 # 3 on rank 0, nothing on rank 1 -> mean 1.5
        assert torch.equal(g0[17:], torch.zeros(6))
        assert (f0[:2] > 0).all() and f0[2] == 0


class _FusedRecurrence(torch.autograd.Function):
    """h_t = tanh(W h_{t-1} + x_t) over T steps as ONE autograd node (what the resident LSTM launch is to the engine)."""

    @staticmethod
    def forward(ctx, x, W):
        hs, h = [], torch.zeros(x.shape[1], W.shape[0])
        for t in range(x.shape[0]):
            h = torch.tanh(h @ W.t() + x[t])
            hs.append(h)
        out = torch.stack(hs)
        ctx.save_for_backward(x, W, out)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, W, out = ctx.saved_tensors
        dx, dW, dh = torch.zeros_like(x), torch.zeros_like(W), torch.zeros(x.shape[1], W.shape[0])
        for t in reversed(range(x.shape[0])):
            g = (dout[t] + dh) * (1 - out[t] ** 2)
            h_prev = out[t - 1] if t > 0 else torch.zeros_like(out[0])
            dW += g.t() @ h_prev
            dx[t] = g
            dh = g @ W
        return dx, dW


class _ChainWithRecurrence(torch.nn.Module):
    """x -> lin0 -> recurrence (fused node or one node per step) -> lin1 -> lin2, like conv stack -> SLSTM -> conv stack."""

    def __init__(self, fused):
        super().__init__()
        self.lin0, self.lin1, self.lin2 = (torch.nn.Linear(8, 8) for _ in range(3))
        self.W = torch.nn.Parameter(torch.randn(8, 8) * 0.3)
        self.fused = fused

    def children_in_order(self):
        return [self.lin0, "rec", self.lin1, self.lin2]

    def forward(self, x, hook=None):          # x (T, B, 8)
        for k, m in enumerate(self.children_in_order()):
            if hook is not None:
                hook(k, x)
            if m == "rec":
                if self.fused:
                    x = _FusedRecurrence.apply(x, self.W)
                else:                          # the per-step fall-back: T small nodes instead of one
                    hs, h = [], torch.zeros(x.shape[1], 8)
                    for t in range(x.shape[0]):
                        h = torch.tanh(h @ self.W.t() + x[t])
                        hs.append(h)
                    x = torch.stack(hs)
            else:
                x = torch.tanh(m(x))
        return x


def _fallback_worker(rank, world, port, q):
    sys.path.insert(0, REPO)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from facodec_amd import benchutil
    from facodec_amd.optim import FlatAdamW
    benchutil.init_distributed("gloo")
    out = {}
    for mode in ("both_fused", "rank1_falls_back"):
        torch.manual_seed(0)
        net = _ChainWithRecurrence(fused=(mode == "both_fused" or rank == 0))
        params = list(net.lin0.parameters()) + [net.W] + list(net.lin1.parameters()) + list(net.lin2.parameters())
        opt = FlatAdamW(params, bucket_bytes=280)
        first = [0, 2, 3, 5]                                    # first parameter index of child k (lin0 | W | lin1 | lin2)
        events = []

        def hook(k, x, opt=opt, events=events):                 # gradient of the activation entering child k: children k + 1 .. final
            if x.requires_grad and k + 1 < 4:
                def fire(g, k=k):
                    opt.launch_all_reduce(from_param=first[k + 1], from_hook=True)
                    events.append((k, opt._next_bucket))
                x.register_hook(fire)

        opt.zero_grad(unbind=True)
        x = torch.randn(6, 3, 8, generator=torch.Generator().manual_seed(20 + rank)).requires_grad_()
        net(x, hook).pow(2).sum().backward()
        opt.exchange_for_step()
        out[mode] = (opt.g.clone(), list(opt._launch_log), list(events), len(opt.buckets))
        opt.end_step()
    q.put(_portable((rank, out)))
    torch.distributed.destroy_process_group()


def test_two_rank_bucket_order_when_one_rank_falls_back_to_per_step_lstm():
    """a rank whose resident LSTM is unusable (fac_lstm_persist_ok = 0: timeout seen, device shared) runs the
    recurrence as one node PER STEP instead of one node per layer.  The exchange's launch points are gradient hooks on the
    activations entering the top-level children, so the graph INSIDE a child must not matter: both ranks issue the same buckets
    from the same points in the same order, and the averaged gradient equals the run where both ranks use the fused node."""
    (_, o0), (_, o1) = _run_ranks(_fallback_worker)
    for mode in ("both_fused", "rank1_falls_back"):
        assert o0[mode][1] == o1[mode][1] and o0[mode][2] == o1[mode][2], (mode, o0[mode][1], o1[mode][1])
        assert torch.equal(o0[mode][0], o1[mode][0])
    assert o0["both_fused"][1] == o0["rank1_falls_back"][1] and o0["both_fused"][2] == o0["rank1_falls_back"][2]
    assert o0["both_fused"][3] >= 3 and any(w == "hook" for _, w in o0["both_fused"][1])
    assert torch.allclose(o0["both_fused"][0], o0["rank1_falls_back"][0], atol=1e-6)
