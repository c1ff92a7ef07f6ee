"""Flat-arena AdamW + ExponentialLR + gradient clipping + data-parallel gradient averaging (reference:
optimizers.py:72-108 `build_optimizer` -> one torch AdamW(betas (0.9, 0.98), eps 1e-9, weight_decay 0.1) and
ExponentialLR(gamma 0.999996) per model key, wrapped in `MultiOptimizer` :11-70; train.py:362-374 clips each key at
1000 and steps; accelerate's DDP averages gradients across ranks, overlapped with backward).

Per model key there are four contiguous arenas (parameters, gradients, first / second moments):
  * every `p.data` is a view of the parameter arena and every `p.grad` a view of the gradient arena, so autograd
    accumulates straight into the arena -- there is no bucket-assembly copy;
  * the data-parallel exchange is an all-reduce(mean) of the gradient arena itself, cut into BUCKETS of <= 64 MB on parameter
    boundaries, counted from the END of the arena (the parameters backward reaches first): few, large collectives for the
    point-to-point xGMI links, and what is still outstanding when backward ends is at most the first bucket -- the remainder --
    not a 145 - 777 MB key.  Buckets are launched asynchronously in that fixed order (`launch_all_reduce(from_param=...)`), from
    gradient hooks at points of the graph where the bucket is final BY CONSTRUCTION (train.py), never from a per-rank guess;
    `step` launches whatever is left and waits.  A gradient that shows up after its bucket was launched is a hard error;
  * WHICH parameters are stepped is decided on the device.  One flag per parameter ("autograd produced a gradient for
    it on this rank") is exchanged by one more small collective behind the key's last bucket, so after the exchange a
    flag is > 0 iff ANY rank reached the parameter; `fac_adamw_step_masked` steps exactly those on every rank (their
    slice holds the averaged gradient everywhere) with a per-parameter step count kept on the device, and leaves the
    others alone -- no weight decay, no moment update -- like torch's AdamW skips `grad is None`.  Ranks therefore
    cannot drift apart when a parameter is reached on one rank only, and no host read sits between backward and step;
  * the step is `fac_grad_norm_clip` (two-stage reduction) + `fac_adamw_step_masked` (one launch per key).

Gradients written by hand.  Autograd's accumulation marks a parameter through a post-accumulate-grad hook; assigning a
new tensor to `p.grad` is folded into the arena and marked by `gather_grads()` / `step()`; an IN-PLACE write into the
pre-bound view (`p.grad.copy_(...)`, `p.grad.add_(...)`) is invisible to both -- call `mark_grads()` (all parameters, or
the ones given) after writing gradients that way, otherwise those parameters are skipped like `grad is None`.
"""
import ctypes
import os

import torch
import torch.distributed as dist

from . import _lib, ops


def _dist_on():
    return dist.is_available() and dist.is_initialized()


class _NativeWork:
    """Handle of a collective issued on the exchange stream: wait() makes the CURRENT stream wait for it (what a torch work
    handle's wait() does for the process group's stream)."""

    def __init__(self, event):
        self.event = event

    def wait(self):
        torch.cuda.current_stream().wait_event(self.event)


class NativeRccl:
    """FAC_NATIVE_RCCL=1: the arenas are exchanged by `fac_allreduce_arena` (include/facodec_hip.h) -- ncclAllReduce(ncclAvg) called
    through the C ABI on an exchange stream this class owns -- instead of torch.distributed's all_reduce.  One communicator per
    process, shared by all model keys (every rank issues the keys' buckets in the same order).  The 128-byte unique id is made
    by rank 0 and shipped through the torch.distributed process group (any backend), the only thing that group is used for here."""

    _instance = None

    @classmethod
    def get(cls):
        if cls._instance is None and os.environ.get("FAC_NATIVE_RCCL") == "1" and _dist_on() and torch.cuda.is_available():
            cls._instance = cls()
        return cls._instance

    def __init__(self):
        import ctypes as C
        lib = _lib.load()
        if not lib.fac_rccl_available():
            raise _lib.FacodecHipError("FAC_NATIVE_RCCL=1 but librccl.so.1 could not be loaded")
        world, rank = dist.get_world_size(), dist.get_rank()
        buf = C.create_string_buffer(128)
        if rank == 0:
            _lib.check(lib.fac_rccl_unique_id(buf), "fac_rccl_unique_id")
        box = [buf.raw if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(box, src=0)
        self._id = C.create_string_buffer(box[0], 128)
        self.comm = C.c_void_p()
        _lib.check(lib.fac_rccl_comm_init(C.byref(self.comm), self._id, world, rank), "fac_rccl_comm_init")
        self.stream = torch.cuda.Stream()
        self.calls = 0

    def all_reduce_mean(self, t):
        """Asynchronous in-place mean over the ranks of a contiguous fp32 tensor: ordered after everything queued on the current
        stream, runs on the exchange stream; -> handle with wait()."""
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
        ready = torch.cuda.Event()
        ready.record()
        self.stream.wait_event(ready)
        _lib.check(_lib.load().fac_allreduce_arena(self.comm, t.data_ptr(), t.numel(), 1, self.stream.cuda_stream), "fac_allreduce_arena")
        done = torch.cuda.Event()
        done.record(self.stream)
        t.record_stream(self.stream)
        self.calls += 1
        return _NativeWork(done)


GATHER_COPY = os.environ.get("FAC_GATHER_COPY", "1") != "0"


def _fold(dst, src):
    """dst[j].copy_(src[j]) for all j.  On the GPU: fac_gather_copy (a few launches, ~0.1 ms of host time for 300 tensors) for the
    contiguous fp32 pairs; torch._foreach_copy_ (measured 17 us of host time per tensor, a hipMemcpyAsync for every second one)
    for whatever is left, on the CPU, and with FAC_GATHER_COPY=0."""
    rest_d, rest_s = dst, src
    if GATHER_COPY and dst[0].is_cuda:
        ok = [s.is_cuda and s.dtype == torch.float32 and d.dtype == torch.float32 and s.is_contiguous() and d.is_contiguous()
              and s.numel() == d.numel() and s.numel() > 0 for d, s in zip(dst, src)]
        fd = [d for d, k in zip(dst, ok) if k]
        fs = [s for s, k in zip(src, ok) if k]
        if fd:
            n = len(fd)
            ps = (ctypes.c_void_p * n)(*[s.data_ptr() for s in fs])
            pd = (ctypes.c_void_p * n)(*[d.data_ptr() for d in fd])
            ne = (ctypes.c_int64 * n)(*[s.numel() for s in fs])
            _lib.check(_lib.load().fac_gather_copy(ps, pd, ne, n, ops._stream()), "fac_gather_copy")
        rest_d = [d for d, k in zip(dst, ok) if not k]
        rest_s = [s for s, k in zip(src, ok) if not k]
    if rest_d:
        if hasattr(torch, "_foreach_copy_"):                # (takes its slow path by itself for tensors that differ in dtype / layout)
            torch._foreach_copy_(rest_d, rest_s)
        else:
            for a, b in zip(rest_s, rest_d):
                b.copy_(a)


class FlatAdamW:
    def __init__(self, params, lr=1e-4, betas=(0.9, 0.98), eps=1e-9, weight_decay=0.1, gamma=0.999996, max_norm=1000.0,
                 data_parallel=True, bucket_bytes=None):
        """data_parallel=False: the gradients are averaged by somebody else (the modules are wrapped in torch's
        DistributedDataParallel as train.py:110-111 does) -- no arena all-reduce, only the tiny flag exchange.
        bucket_bytes: size of the exchange buckets (default FAC_BUCKET_MB or 64 MB; 0 = the whole arena as one)."""
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev = self.params[0].device
        n = sum(p.numel() for p in self.params)
        P = len(self.params)
        self.n, self.data_parallel = n, data_parallel
        self.p = torch.empty(n, device=dev, dtype=torch.float32)
        # gradients | per-parameter flags | one "poison" word: a resident-LSTM wait of some rank gave up during this step
        # (ops.lstm_timeouts).  The word travels with the flags, and a positive value after the exchange clears every
        # flag on every rank before the masked AdamW step -- a step whose gradients came out of a bailed-out recurrence is skipped
        # everywhere, with no host read between backward and step.
        self._gx = torch.zeros(n + P + 1, device=dev, dtype=torch.float32)
        self.g = self._gx[:n]
        self._flags = self._gx[n:n + P]
        self._flagsx = self._gx[n:]                                        # flags + poison: what the flag collective carries
        self._poison = self._gx[n + P:]
        self.m = torch.zeros(n, device=dev, dtype=torch.float32)
        self.v = torch.zeros(n, device=dev, dtype=torch.float32)
        self._scratch = torch.empty(1024, device=dev, dtype=torch.float32)
        self._bc = torch.empty(2 * P, device=dev, dtype=torch.float32)
        self._steps_dev = torch.zeros(P, device=dev, dtype=torch.int32)    # torch AdamW keeps one step count per parameter
        self.norm = torch.zeros(2, device=dev, dtype=torch.float32)
        off = 0
        self.slices = []
        self._views = []                                      # the arena views `p.grad` is bound to
        self._touched = [False] * P
        for i, p in enumerate(self.params):
            k = p.numel()
            self.p[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.p[off:off + k].view_as(p)          # the module now reads its weights from the arena
            self._views.append(self.g[off:off + k].view_as(p))
            p.grad = self._views[i]                          # ... and autograd accumulates into the arena
            p.register_post_accumulate_grad_hook(self._mark(i))
            self.slices.append((off, k))
            off += k
        self._offsets = torch.tensor([s[0] for s in self.slices] + [n], dtype=torch.int64).to(dev)
        self.lr, self.base_lr = lr, lr
        self.betas, self.eps, self.wd, self.gamma, self.max_norm = betas, eps, weight_decay, gamma, max_norm
        self.lr_epochs = 0                                    # ExponentialLR.last_epoch
        if bucket_bytes is None:
            bucket_bytes = int(float(os.environ.get("FAC_BUCKET_MB", "64")) * 2 ** 20)
        # buckets = parameter index ranges [lo, hi), built from the END of the arena backwards; buckets[0] is launched first
        self.buckets, hi, size = [], P, 0
        for i in reversed(range(P)):
            size += 4 * self.slices[i][1]
            if i == 0 or (bucket_bytes > 0 and size >= bucket_bytes):
                self.buckets.append((i, hi))
                hi, size = i, 0
        self._works = []                                      # pending asynchronous collectives of this step, in launch order
        self._next_bucket = 0                                 # buckets [0, _next_bucket) have been launched this step
        self._at_launch = [False] * P                         # `_touched` of a bucket's parameters when it was launched
        self._late_accum = []                                 # parameters accumulated into while their bucket was in flight
        self._need_scale = False
        self._flags_final = False                             # flags uploaded (and exchanged) for the pending step
        self._flags_pat, self._flags_cache = None, None
        self._expected = None                                 # parameters the previous step's backward reached (reporting only)
        self.exchange_log = []                                # per step: [(bucket, "hook" | "end"), ...]  ([] = nothing exchanged)
        self._launch_log = []
        self.time_exchange, self._wait_events = False, None
        self._ptr_cache = None                                # (parameter arena base, gradient arena base, per-parameter addresses)

    def _mark(self, i):
        """Post-accumulate hook: fires on EVERY accumulation into parameter i (a tied parameter, a module used twice in one graph, a
        second backward before `step`).  An accumulation into a bucket whose exchange has already been launched is an in-place add
        into memory the asynchronous collective may be reading: recorded here, raised by `step` / `wait_all_reduce` (an exception
        inside the autograd engine's thread would leave the collectives of the other ranks unmatched at an arbitrary point)."""
        def hook(_p):
            if self._next_bucket and i >= self.buckets[self._next_bucket - 1][0] and (self._works or self._flags_final):
                self._late_accum.append(i)
            self._touched[i] = True
        return hook

    def _raise_late(self):
        late = sorted(set(self._late_accum) | set(self._late_gradients() if self._exchanging() else []))
        if late:
            self._late_accum = []
            raise RuntimeError(f"FlatAdamW: parameters {late[:8]} received a gradient after the exchange of their bucket was launched "
                               "(a second backward / a shared parameter / a launch point that is too early for this graph; "
                               "FAC_EARLY_EXCHANGE=0 launches after backward, and gradient accumulation over several backward "
                               "passes needs the exchange launched after the last one)")

    # ------------------------------------------------------------------------------------------ torch-optimizer shims
    @property
    def param_groups(self):
        """What the reference's loop reads from an optimiser / scheduler (train.py:384 `get_last_lr()[0]`)."""
        return [dict(params=self.params, lr=self.lr, betas=tuple(self.betas), eps=self.eps, weight_decay=self.wd,
                     initial_lr=self.base_lr)]

    def get_last_lr(self):
        return [self.lr]

    @property
    def param_steps(self):
        """Per-parameter step counts (device-side state; reading them synchronises -- checkpoints and tests only)."""
        return [int(s) for s in self._steps_dev.tolist()]

    @property
    def steps(self):
        return max(self.param_steps)

    # ------------------------------------------------------------------------------------------ gradients
    @property
    def _work(self):
        """The pending collectives of this step (None when nothing is in flight)."""
        return self._works[-1] if self._works else None

    def _drain(self):
        """A pending asynchronous exchange must finish before anybody writes into the arena."""
        if self._works:
            self.wait_all_reduce()

    def zero_grad(self, set_to_none=False, unbind=False):
        """One memset of the gradient arena (+ flags); the `.grad` views stay bound (set_to_none is accepted and ignored:
        unbinding would only force a re-bind).
        unbind=True (what TrainStep uses): `.grad` is left None for the coming backward, so autograd's AccumulateGrad
        keeps ("steals") each produced gradient tensor instead of launching one add-into-the-view kernel per parameter
        (~800 small launches per step); `launch_all_reduce` / `step` then fold all of them into the arena with one
        multi-tensor copy and bind the views again.
        (Round 6: no `_rebind()` in here any more -- folding foreign gradient tensors into an arena that is zeroed on the next line
        was ~1 ms of Python per key at the one point of the step where the device has nothing queued, tools/gpu_idle.py.)"""
        self._drain()
        self._gx.zero_()
        self._touched = [False] * len(self.params)
        self._reset_exchange_state()
        if unbind:
            for p in self.params:
                p.grad = None
        else:
            views = self._views
            for i, p in enumerate(self.params):
                if p.grad is not views[i]:
                    p.grad = views[i]

    def _reset_exchange_state(self):
        """Nothing of this key is in flight or marked launched: the next `launch_all_reduce` starts from bucket 0 again.  (`step`
        used to clear only `_flags_final`; after `step(zero_grad=False)` a further backward + exchange then skipped every bucket
        -- `_next_bucket == len(buckets)` -- and re-exchanged the flags only: un-averaged gradients, ranks drifting apart.)"""
        self._flags_final = False
        self._next_bucket = 0
        self._launch_log = []
        self._at_launch = [False] * len(self.params)
        self._late_accum = []

    def _rebind(self, lo=0, hi=None):
        """`p.grad` must be the arena view (someone may have set it to None or to a foreign tensor), `p.data` must still
        live in the parameter arena (a later module.to()/.float() would silently detach the optimiser from the model).
        Foreign gradient tensors are folded into the arena -- all of them in one multi-tensor copy -- and marked.
        [lo, hi): only these parameters (a bucket about to be launched).  A foreign gradient on a parameter whose bucket is
        ALREADY in flight means the bucket was launched before its gradients were final: that is an error, not a fold."""
        base_p, base_g = self.p.data_ptr(), self.g.data_ptr()
        hi = len(self.params) if hi is None else hi
        launched_from = self.buckets[self._next_bucket - 1][0] if self._next_bucket else len(self.params)
        in_flight = bool(self._works or self._flags_final)
        params, views, touched = self.params, self._views, self._touched
        # This loop runs once per key and step at the one point where the device has nothing queued (tools/gpu_idle.py: one 3.5 - 4.4 ms
        # gap per step; tools/tune/host_profile.py: 4.2 ms of it in here for ~1 500 parameters in round 6's first form).  Every torch
        # call costs 0.15 - 0.5 us, so: parameter / view addresses are cached (the arenas never move), and what autograd guarantees
        # about a gradient (device, dtype and shape of its parameter) is not re-tested per tensor -- only its layout is.
        cache = self._ptr_cache
        if cache is None or cache[0] != base_p or cache[1] != base_g:
            cache = self._ptr_cache = (base_p, base_g, [base_p + 4 * o for o, _ in self.slices], [base_g + 4 * o for o, _ in self.slices])
        pptr, vptr = cache[2], cache[3]
        idx, src, sptr = [], [], []
        for i in range(lo, hi):
            p = params[i]
            if p.data_ptr() != pptr[i]:
                raise RuntimeError("FlatAdamW: a parameter no longer lives in the optimiser's arena (was the module moved or cast after "
                                   "the optimiser was built?); build FlatAdamW after the model is on its final device")
            g = p.grad
            v = views[i]
            if g is None:
                p.grad = v
            elif g is not v:
                gp = g.data_ptr()
                if gp != vptr[i]:                      # foreign gradient tensor: fold it in once, then rebind
                    if i >= launched_from and in_flight:
                        raise RuntimeError(f"FlatAdamW: the gradient of parameter {i} arrived after the exchange of its bucket was launched "
                                           "(the launch point is too early for this graph; FAC_EARLY_EXCHANGE=0 launches after backward)")
                    idx.append(i)
                    src.append(g)
                    sptr.append(gp)
                    p.grad = v
                    touched[i] = True
        if src:
            self._fold_into_views(idx, src, sptr, vptr)

    def _fold_into_views(self, idx, src, sptr, vptr):
        """views[i].copy_(g) for the stolen gradient tensors: contiguous ones on the GPU through fac_gather_copy with the cached view
        addresses (one table entry per tensor, no per-tensor torch calls beyond the layout test); the rest as `_fold` does."""
        views, slices = self._views, self.slices
        if not (GATHER_COPY and self.g.is_cuda):
            _fold([views[i] for i in idx], src)
            return
        ok = [g.is_contiguous() and slices[i][1] > 0 for i, g in zip(idx, src)]
        if not all(ok):
            _fold([views[i] for i, k in zip(idx, ok) if not k], [g for g, k in zip(src, ok) if not k])
            idx = [i for i, k in zip(idx, ok) if k]
            sptr = [q for q, k in zip(sptr, ok) if k]
        n = len(idx)
        if n:
            ps = (ctypes.c_void_p * n)(*sptr)
            pd = (ctypes.c_void_p * n)(*[vptr[i] for i in idx])
            ne = (ctypes.c_int64 * n)(*[slices[i][1] for i in idx])
            _lib.check(_lib.load().fac_gather_copy(ps, pd, ne, n, ops._stream()), "fac_gather_copy")

    def gather_grads(self, mark_all=False):
        """Compatibility with callers that assign `.grad` tensors by hand: folds them into the arena (no-op for views).
        mark_all: also count every parameter as having a gradient (for in-place writes into the views)."""
        self._rebind()
        if mark_all:
            self.mark_grads()

    def mark_grads(self, params=None):
        """Declare hand-written (in-place) gradients: all parameters, or the given ones."""
        if params is None:
            self._touched = [True] * len(self.params)
        else:
            ids = {id(p) for p in params}
            for i, p in enumerate(self.params):
                if id(p) in ids:
                    self._touched[i] = True
        self._flags_final = False

    def backward_complete(self):
        """True once every parameter the previous step's backward reached has been reached again (reporting / tests; launch
        decisions do not depend on it)."""
        return self._expected is not None and all(t or not e for t, e in zip(self._touched, self._expected))

    def _upload_flags(self):
        """Local flags -> the flag tail of the arena.  The usage pattern of the model is static after the first step, so the
        pattern is kept as a device tensor and re-used (a device-to-device copy, no host transfer in the steady state)."""
        pat = tuple(self._touched)
        if pat != self._flags_pat:
            self._flags_cache = torch.tensor(pat, dtype=torch.float32).to(self._gx.device)
            self._flags_pat = pat
        self._flags.copy_(self._flags_cache)
        if self._gx.is_cuda:
            _lib.check(_lib.load().fac_lstm_abort_flag(ops._ptr(self._poison), ops._stream()), "fac_lstm_abort_flag")

    def _exchanging(self):
        # a single rank has nothing to exchange; FAC_FORCE_ALLREDUCE=1 still issues the collectives (smoke-tests the RCCL path --
        # AVG op, async work handles, stream hand-over -- on a one-GPU box: the mean over one rank is the identity)
        return _dist_on() and (dist.get_world_size() > 1 or os.environ.get("FAC_FORCE_ALLREDUCE") == "1")

    def launch_all_reduce(self, from_param=0, from_hook=False, only_if_complete=False):
        """Data-parallel exchange (RCCL over xGMI under backend 'nccl'; the collectives run on the process group's stream, after
        everything already queued on the current stream).  Launches, in the fixed end-of-arena-first order, every bucket not yet
        launched whose parameters all have index >= from_param, each as one asynchronous all-reduce(mean) of its slice of the
        gradient arena; behind the key's last bucket one small collective carries the per-parameter flags.

        from_param > 0 is for calls from inside backward (gradient hooks): the caller asserts that the parameters from there on are
        FINAL at this point of the graph on every rank (train.py derives that from the order autograd runs the nodes in, not from
        which parameters happened to receive a gradient on this rank), so every rank issues the same collectives in the same order.
        `only_if_complete` is accepted for older callers and ignored."""
        if self._flags_final or not self._exchanging():
            return
        where = "hook" if from_hook else "end"
        if not self.data_parallel:          # gradients already averaged by a DDP wrapper: only the flags travel (MAX: any rank)
            if from_param > 0:
                return
            self._rebind()
            self._upload_flags()
            self._flags_final = True
            self._launch_log.append(("flags", where))
            self._works.append(dist.all_reduce(self._flagsx, op=dist.ReduceOp.MAX, async_op=True))
            return
        if self.g.is_cuda and self._next_bucket < len(self.buckets) and self.buckets[self._next_bucket][0] >= from_param:
            ops.join_side_streams(self.g.device)    # parameter gradients written by chains on side streams (discriminators, heads)
        native = NativeRccl.get() if self.g.is_cuda else None      # FAC_NATIVE_RCCL=1: fac_allreduce_arena instead of torch's all_reduce
        nccl = native is not None or dist.get_backend() == "nccl"   # RCCL: the mean is part of the collective; gloo (CPU scaffold) has no AVG
        op = dist.ReduceOp.AVG if nccl else dist.ReduceOp.SUM
        reduce = (lambda t: native.all_reduce_mean(t)) if native is not None else (lambda t: dist.all_reduce(t, op=op, async_op=True))
        while self._next_bucket < len(self.buckets) and self.buckets[self._next_bucket][0] >= from_param:
            lo, hi = self.buckets[self._next_bucket]
            self._rebind(lo, hi)                    # gradients autograd kept outside the arena (zero_grad(unbind=True))
            self._at_launch[lo:hi] = self._touched[lo:hi]
            e_lo, e_hi = self.slices[lo][0], self.slices[hi - 1][0] + self.slices[hi - 1][1]
            self._works.append(reduce(self.g[e_lo:e_hi]))
            self._launch_log.append((self._next_bucket, where))
            self._next_bucket += 1
            self._need_scale = self._need_scale or not nccl
        if self._next_bucket == len(self.buckets):
            self._upload_flags()
            self._flags_final = True
            self._works.append(reduce(self._flagsx))

    def _late_gradients(self):
        """Parameters of already launched buckets that were marked after the launch (in-place accumulation into the bound views
        while the collective may be reading them)."""
        if not self._next_bucket:
            return []
        lo = self.buckets[self._next_bucket - 1][0]
        return [i for i in range(lo, len(self.params)) if self._touched[i] and not self._at_launch[i]]

    def wait_all_reduce(self):
        if self._works:
            works, self._works = self._works, []              # cleared first: an exception in wait() must not wedge the next step
            timed = self.time_exchange and self._gx.is_cuda
            if timed:                                         # how long the compute stream stalls for the collectives: what the
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)   # overlap failed to hide
                e0.record()
            for w in works:
                w.wait()                                      # makes the current stream wait for the collective
            if timed:
                e1.record()
                self._wait_events = (e0, e1)
            if self._need_scale:
                self._gx.mul_(1.0 / dist.get_world_size())
                self._need_scale = False

    def all_reduce_mean(self):
        self.launch_all_reduce()
        self.wait_all_reduce()

    def exchange_wait_ms(self):
        """Stall of the compute stream at the last `wait_all_reduce` (needs `time_exchange = True`; synchronises)."""
        if self._wait_events is None:
            return None
        e0, e1 = self._wait_events
        e1.synchronize()
        return e0.elapsed_time(e1)

    def broadcast_parameters(self, src=0):
        """Construction-time parameter broadcast of DistributedDataParallel (train.py:110-111 via accelerator.prepare) as ONE
        collective over the parameter arena: every rank starts from rank `src`'s weights."""
        if _dist_on() and dist.get_world_size() > 1:
            dist.broadcast(self.p, src=src)

    # ------------------------------------------------------------------------------------------ step
    def exchange_for_step(self):
        """Everything of `step` that precedes the two kernels: late-gradient check, fold foreign gradient tensors into the arena,
        launch whatever has not been launched, wait, final flags.  (Separate so that the CPU / gloo tests can drive the exchange
        protocol of a whole step without the HIP library.)"""
        self._raise_late()
        self._rebind()                                         # raises as well if a stolen gradient belongs to a launched bucket
        self.all_reduce_mean()                                 # whatever has not been launched yet, then wait for everything
        self._raise_late()                                     # an accumulation that raced the collectives just waited for
        if not self._flags_final:                              # single rank (or no process group): local flags
            self._upload_flags()

    def end_step(self, zero_grad=True, advance_lr=True):
        self._expected = tuple(self._touched)
        self.exchange_log.append(list(self._launch_log))
        if len(self.exchange_log) > 64:
            del self.exchange_log[:-64]
        self._reset_exchange_state()                           # also without zero_grad(): the next backward + exchange starts over
        if advance_lr:
            self.scheduler_step()
        if zero_grad:
            self.zero_grad()

    def step(self, zero_grad=True, advance_lr=True):
        self.exchange_for_step()
        self._step_kernels()
        self.end_step(zero_grad, advance_lr)

    def _step_kernels(self):
        """The device side of a step, behind the exchange: poison check (a positive word = some rank's resident LSTM bailed out during
        this step: every flag is cleared, nothing is stepped anywhere), gradient-norm clip, masked AdamW."""
        lib = _lib.load()
        st = ops._stream()
        _lib.check(lib.fac_mask_flags_if(ops._ptr(self._flags), len(self.params), ops._ptr(self._poison), st), "fac_mask_flags_if")
        clip = None
        if self.max_norm is not None:
            _lib.check(lib.fac_grad_norm_clip(ops._ptr(self.g), self.n, self.max_norm, ops._ptr(self._scratch), ops._ptr(self.norm), st),
                       "fac_grad_norm_clip")
            clip = self.norm
        _lib.check(lib.fac_adamw_step_masked(ops._ptr(self.p), ops._ptr(self.g), ops._ptr(self.m), ops._ptr(self.v), self.n,
                                             ops._ptr(self._offsets), len(self.params), ops._ptr(self._flags), ops._ptr(self._steps_dev),
                                             ops._ptr(self._bc), self.lr, self.betas[0], self.betas[1], self.eps, self.wd,
                                             ops._ptr(clip), st), "fac_adamw_step_masked")

    def scheduler_step(self):
        """ExponentialLR.step(), once per iteration (train.py:372-374)."""
        self.lr *= self.gamma
        self.lr_epochs += 1

    def grad_norm(self):
        """Pre-clip gradient norm of the last step (device scalar; a copy: the next step overwrites the buffer)."""
        return self.norm[0].clone()

    def params_without_grad(self):
        """Indices of the parameters the last backward did not reach on THIS rank."""
        return [i for i, t in enumerate(self._touched) if not t]

    # ------------------------------------------------------------------------------------------ checkpoint state
    def state_dict(self):
        """torch.optim.AdamW.state_dict() layout (what the reference's checkpoints carry per key, optimizers.py:17-20):
        state[i] = {step, exp_avg, exp_avg_sq} for the parameters that have been stepped, one param group."""
        state = {}
        steps = self.param_steps
        for i, (p, (off, k)) in enumerate(zip(self.params, self.slices)):
            if steps[i] > 0:
                state[i] = dict(step=torch.tensor(float(steps[i])),
                                exp_avg=self.m[off:off + k].view_as(p).clone(), exp_avg_sq=self.v[off:off + k].view_as(p).clone())
        group = dict(lr=self.lr, betas=tuple(self.betas), eps=self.eps, weight_decay=self.wd, amsgrad=False, maximize=False,
                     foreach=None, capturable=False, differentiable=False, fused=None, initial_lr=self.base_lr,
                     params=list(range(len(self.params))))
        return dict(state=state, param_groups=[group])

    def load_state_dict(self, sd):
        """Validates everything BEFORE touching any state (torch's load_state_dict fails without side effects, and the
        reference's MultiOptimizer prints 'Unloaded' and carries on with the optimiser as it was, optimizers.py:27-32)."""
        group = sd["param_groups"][0]
        if len(group["params"]) != len(self.params):
            raise ValueError(f"optimizer state has {len(group['params'])} parameters, this key has {len(self.params)}")
        lr, base_lr = float(group["lr"]), float(group.get("initial_lr", self.base_lr))
        betas, eps, wd = tuple(float(b) for b in group["betas"]), float(group["eps"]), float(group["weight_decay"])
        todo = []
        for idx, pid in enumerate(group["params"]):
            st = sd["state"].get(pid)
            if st is None:
                continue
            off, k = self.slices[idx]
            for name in ("exp_avg", "exp_avg_sq"):
                if st[name].numel() != k:
                    raise ValueError(f"optimizer state {name}[{pid}] has {st[name].numel()} elements, parameter {idx} has {k}")
            todo.append((idx, off, k, st["exp_avg"], st["exp_avg_sq"], int(float(st["step"]))))
        self._drain()
        self.lr, self.base_lr, self.betas, self.eps, self.wd = lr, base_lr, betas, eps, wd
        self.m.zero_()
        self.v.zero_()
        steps = [0] * len(self.params)
        for idx, off, k, m, v, s in todo:
            self.m[off:off + k].copy_(m.reshape(-1))
            self.v[off:off + k].copy_(v.reshape(-1))
            steps[idx] = s
        self._steps_dev.copy_(torch.tensor(steps, dtype=torch.int32))

    def scheduler_state_dict(self):
        """torch ExponentialLR.state_dict() keys the reference's checkpoints hold (optimizers.py:22-25)."""
        return dict(gamma=self.gamma, base_lrs=[self.base_lr], last_epoch=self.lr_epochs, _step_count=self.lr_epochs + 1,
                    _last_lr=[self.lr])

    def load_scheduler_state_dict(self, sd):
        gamma, base_lr, epochs = float(sd["gamma"]), float(sd["base_lrs"][0]), int(sd["last_epoch"])
        self.gamma, self.base_lr, self.lr_epochs = gamma, base_lr, epochs
        self.lr = float(sd["_last_lr"][0]) if "_last_lr" in sd else self.base_lr * self.gamma ** self.lr_epochs


class _SchedulerView:
    """`optimizer.schedulers[key]` of optimizers.py:11-16: the object train.py:384 asks for `get_last_lr()`."""

    def __init__(self, opt):
        self._opt = opt

    def get_last_lr(self):
        return self._opt.get_last_lr()

    def step(self, *args):
        self._opt.scheduler_step()

    def state_dict(self):
        return self._opt.scheduler_state_dict()

    def load_state_dict(self, sd):
        self._opt.load_scheduler_state_dict(sd)


class MultiOptimizer:
    """optimizers.py:11-70 over FlatAdamW: `step(key)`, `scheduler(key=)`, `zero_grad(key)`, and the (key, state) list
    formats of `state_dict` / `scheduler_state_dict` that modules/commons.py:446-471 `load_checkpoint` and train.py's
    checkpoint writer exchange.  `step` includes FlatAdamW's own clip at the key's max_norm: after the caller's
    `clip_grad_norm_` with the same bound (train.py:290,362-365) the second clip coefficient is 1."""

    def __init__(self, optimizers):
        self.optimizers = dict(optimizers)
        self.schedulers = {k: _SchedulerView(o) for k, o in self.optimizers.items()}
        self.keys = list(self.optimizers)

    def state_dict(self):
        return [(k, self.optimizers[k].state_dict()) for k in self.keys]

    def scheduler_state_dict(self):
        return [(k, self.optimizers[k].scheduler_state_dict()) for k in self.keys]

    def load_state_dict(self, state_dict):
        for k, val in state_dict:
            try:
                self.optimizers[k].load_state_dict(val)
            except (KeyError, ValueError, RuntimeError):      # the reference prints and goes on (optimizers.py:27-32)
                print("Unloaded %s" % k)

    def load_scheduler_state_dict(self, state_dict):
        for k, val in state_dict:
            try:
                self.optimizers[k].load_scheduler_state_dict(val)
            except (KeyError, ValueError):
                print("Unloaded %s" % k)

    def step(self, key=None, scaler=None):
        if scaler is not None:
            raise NotImplementedError("GradScaler is not supported: the step is fp32 (train.py runs without mixed precision)")
        for k in ([key] if key is not None else self.keys):
            self.optimizers[k].step(zero_grad=False, advance_lr=False)

    def zero_grad(self, key=None):
        for k in ([key] if key is not None else self.keys):
            self.optimizers[k].zero_grad()

    def scheduler(self, *args, key=None):
        for k in ([key] if key is not None else self.keys):
            self.optimizers[k].scheduler_step()


def _unwrap(m):
    return m.module if isinstance(m, torch.nn.parallel.DistributedDataParallel) else m


def sync_module_states(model_dict, src=0):
    """What DistributedDataParallel's constructor does for train.py:110-111: every rank starts from rank `src`'s parameters
    and buffers.  For modules NOT wrapped in DDP (the arena path); per key one broadcast of the parameters flattened into
    one tensor, one of the float buffers."""
    if not (_dist_on() and dist.get_world_size() > 1):
        return
    for m in model_dict.values():
        tensors = [p.data for p in _unwrap(m).parameters()] + [b for b in _unwrap(m).buffers() if b.is_floating_point()]
        if not tensors:
            continue
        flat = torch.cat([t.reshape(-1).to(torch.float32) for t in tensors])
        dist.broadcast(flat, src=src)
        off = 0
        with torch.no_grad():
            for t in tensors:
                t.copy_(flat[off:off + t.numel()].view_as(t))
                off += t.numel()


def build_optimizer(model_dict, scheduler_params_dict=None, lr=1e-4, type="AdamW", broadcast=True):
    """optimizers.py:72-108 (AdamW only; ScaledAdam belongs to the unused transformer_modules tree).  A key wrapped in
    DistributedDataParallel (train.py:110-111) keeps DDP's own gradient averaging; an unwrapped key gets the arena
    all-reduce, and -- broadcast=True, process group initialised -- starts from rank 0's parameters like DDP would."""
    if type != "AdamW":
        raise ValueError("Unknown optimizer type: %s" % type)
    max_norm = {"discriminator": 10.0}                       # train.py:290 vs :362-365
    ddp = torch.nn.parallel.DistributedDataParallel
    if broadcast:
        sync_module_states({k: m for k, m in model_dict.items() if not isinstance(m, ddp)})
    return MultiOptimizer({k: FlatAdamW(m.parameters(), lr=lr, max_norm=max_norm.get(k, 1000.0),
                                        data_parallel=not isinstance(m, ddp)) for k, m in model_dict.items()})
