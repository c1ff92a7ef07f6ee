"""Batched weight preparation: the weight-norm scales and packed weight layouts of a module tree re-materialised by a handful of
launches at the start of a forward / train step instead of one or two small launches per tensor inside it.

The reference recomputes w = g * v / ||v|| in a pre-forward hook of every weight-normed conv, on every forward
(dac/model/encodec.py:42-51, dac/nn/layers.py:9-14); this library did the same with `ops.wn_scale` + `ops.pack_*` per layer: ~290
launches of a configs[1] forward, ~1 700 of a train step, each 5 - 13 us of a serial stream.  A `WeightCache` keeps that contract
-- every forward sees weights derived from the parameters' CURRENT contents -- with different mechanics:

* inside an open region (`begin()` .. `end()`, or `with cache:`) the decorated `ops` functions (`@prepared`) look their arguments up
  in the cache.  A call the cache has not seen runs as before (into a buffer the cache then owns) and is remembered;
* the next `begin()` re-runs everything remembered as ONE recorded batch (fac_prep_* in include/facodec_hip.h: one launch per
  phase and kernel family, the same device functions as the single launches, bit-identical results) on the current stream, and
  only what that batch wrote counts as fresh for the region.  Calls that find a fresh entry return its buffer and launch nothing;
* nothing is ever served outside a region, a region always starts with a re-materialisation, and a lookup is keyed by the
  argument tensors' addresses: a parameter that was re-pointed simply misses.  Entries unused for a few regions are dropped.

Who opens regions: `train.TrainStep` / `GeneratorStep` (one cache for the generator side, one for the discriminator, which is
re-materialised again after its optimiser step), and the no-grad forwards of Encoder / Decoder / FAquantizer / Discriminator
(`cached_forward`).  Forwards that build a graph outside TrainStep keep the per-layer launches: their packed weights are saved for
a backward whose time the module cannot know.
"""
import os
import threading

import torch

from . import _lib

ENABLED = os.environ.get("FAC_WEIGHT_BATCH", "1") != "0"
KEEP_EPOCHS = 4             # an entry nobody asked for in this many regions is dropped from the batch

_ACTIVE = []                # caches with an open region (module-global on purpose: autograd's worker thread must see them)
_TLS = threading.local()    # .stack: entries being computed on this thread; a nested call (wn_scale inside a pack) makes the outer
                            # one depend on it


def _stack():
    st = getattr(_TLS, "stack", None)
    if st is None:
        st = _TLS.stack = []
    return st


def _visible(me):
    """Caches whose region this thread may use: its own, and those opened for every thread (a train step: autograd's worker thread
    runs the backward nodes).  Another thread's no-grad forward is nobody else's business -- its replay is ordered on ITS stream."""
    return [c for c in _ACTIVE if c.owner == me or c.any_thread]


class _Entry:
    __slots__ = ("fn", "args", "kw", "ret", "buf", "phase", "fresh", "used", "children")

    def __init__(self, fn, args, kw):
        self.fn, self.args, self.kw = fn, args, kw
        self.ret = self.buf = None
        self.phase, self.fresh, self.used = 0, -1, -1
        self.children = []


def _buf_of(ret):
    return ret[0] if isinstance(ret, tuple) else ret


def _key_part(a):
    if isinstance(a, torch.Tensor):
        return (a.data_ptr(), tuple(a.shape))
    return a


class WeightCache:
    """See the module docstring.  `params`: the tensors whose derived layouts this cache may own (anything else passes through)."""

    def __init__(self, params, name="", any_thread=False):
        self.name = name
        self.any_thread = any_thread      # served to every thread while the region is open (TrainStep), or to the opening thread only
        self.owner = None
        self._params = [p for p in params]
        self.claims = None
        self.entries = {}
        self.by_buf = {}                  # address of an owned buffer -> its entry
        self.epoch = 0
        self.plan = -1
        self.planned = []
        self.dirty = False
        self.depth = 0
        self.stats = dict(rebuilds=0, replays=0, hits=0, misses=0, dropped=0)

    # ------------------------------------------------------------------------------------------ regions
    def begin(self):
        if self.depth:
            self.depth += 1
            return self
        if self.claims is None:
            self.claims = {p.data_ptr() for p in self._params if p.is_cuda}
        self.epoch += 1
        capturing = torch.cuda.is_current_stream_capturing() if torch.cuda.is_available() else False
        if not capturing:
            self._prune()
            if self.dirty:
                self._rebuild()
        if self.plan >= 0:
            from . import ops
            _lib.check(_lib.load().fac_prep_replay(self.plan, ops._stream()), "fac_prep_replay")
            self.stats["replays"] += 1
            ep = self.epoch
            for e in self.planned:
                e.fresh = ep
        self.depth = 1
        self.owner = threading.get_ident()
        _ACTIVE.append(self)
        return self

    def end(self):
        if self.depth > 1:
            self.depth -= 1
            return
        if self.depth == 1:
            self.depth = 0
            _ACTIVE.remove(self)

    __enter__ = begin

    def __exit__(self, *exc):
        self.end()
        return False

    def close(self):
        """Frees the device tables (waits for the device: a replay may still be running)."""
        if self.plan >= 0:
            torch.cuda.synchronize()
            _lib.load().fac_prep_free(self.plan)
            self.plan, self.planned = -1, []

    def __deepcopy__(self, memo):
        return WeightCache([], self.name, self.any_thread)     # a copied module starts its own cache (cached_forward rebuilds it)

    def __reduce__(self):
        return (WeightCache, ([], self.name, self.any_thread))

    def __del__(self):
        try:
            if self.plan >= 0:
                _lib.load().fac_prep_free(self.plan)       # hipFree waits for the device
        except Exception:
            pass

    # ------------------------------------------------------------------------------------------ lookups
    def lookup(self, name, fn, args, kw):
        out_kw = kw.pop("out", None) if "out" in kw else None
        del out_kw                                            # the cache's buffer takes the place of the caller's
        key = (name,) + tuple(_key_part(a) for a in args) + tuple((k, _key_part(kw[k])) for k in sorted(kw))
        e = self.entries.get(key)
        if e is not None and e.fresh == self.epoch:
            e.used = self.epoch
            self.stats["hits"] += 1
            st = _stack()
            if st:
                st[-1].children.append(e)
            return e.ret
        self.stats["misses"] += 1
        if e is None:
            e = _Entry(fn, args, dict(kw))
            for a in list(args) + list(kw.values()):        # an input another entry produces: that entry comes first
                if isinstance(a, torch.Tensor):
                    src = self.by_buf.get(a.data_ptr())
                    if src is not None:
                        e.children.append(src)
        e.used = self.epoch
        st = _stack()
        st.append(e)
        try:
            ret = fn(*args, out=e.buf, **kw)                  # the single launch(es), into the entry's buffer once it has one
        finally:
            st.pop()
        if e.buf is None:
            e.ret, e.buf = ret, _buf_of(ret)
            self.entries[key] = e
            self.by_buf.setdefault(e.buf.data_ptr(), e)       # (a forwarding function returns its inner call's buffer: the inner entry keeps it)
            self.claims.add(e.buf.data_ptr())
            self.dirty = True
        if st:
            st[-1].children.append(e)
        return e.ret

    # ------------------------------------------------------------------------------------------ the batch
    def _phases(self):
        memo = {}

        def phase(e):
            k = id(e)
            if k not in memo:
                memo[k] = 0                                   # (cycles cannot occur: a child is created before its parent returns)
                memo[k] = 1 + max((phase(c) for c in e.children), default=-1)
            return memo[k]

        for e in self.entries.values():
            e.phase = phase(e)

    def _prune(self):
        if not self.entries:
            return
        horizon = self.epoch - KEEP_EPOCHS
        keep = set()

        def mark(e):
            if id(e) in keep:
                return
            keep.add(id(e))
            for c in e.children:
                mark(c)

        for e in self.entries.values():
            if e.used >= horizon:
                mark(e)
        if len(keep) == len(self.entries):
            return
        for k in [k for k, e in self.entries.items() if id(e) not in keep]:
            e = self.entries.pop(k)
            self.stats["dropped"] += 1
        alive = {e.buf.data_ptr() for e in self.entries.values()}
        for ptr in [ptr for ptr in self.by_buf if ptr not in alive]:
            del self.by_buf[ptr]
            self.claims.discard(ptr)
        for ptr, e in list(self.by_buf.items()):
            if id(e) not in keep:                             # the buffer lives on under another entry (a forwarding function's)
                self.by_buf[ptr] = next(x for x in self.entries.values() if x.buf.data_ptr() == ptr)
        self.dirty = True

    def _rebuild(self):
        """Records every entry's launches (children first) as one plan.  The functions run exactly as in a lookup, except that the C
        side records instead of launching and every nested lookup hits."""
        lib = _lib.load()
        if self.plan >= 0:
            torch.cuda.current_stream().synchronize()         # a replay of the old tables may still be running on this stream
            lib.fac_prep_free(self.plan)
            self.plan, self.planned = -1, []
        self.dirty = False
        if not self.entries:
            return
        alive = {id(e) for e in self.entries.values()}
        for e in self.entries.values():
            e.children = list({id(c): c for c in e.children if id(c) in alive and c is not e}.values())
        self._phases()
        order = sorted(self.entries.values(), key=lambda e: e.phase)
        saved = [(e, e.fresh) for e in order]
        for e in order:
            e.fresh = self.epoch                              # nested lookups must hit while recording
        was_active = self in _ACTIVE
        self.owner = threading.get_ident()
        st = _stack()
        if not was_active:
            _ACTIVE.append(self)
        _lib.check(lib.fac_prep_begin(), "fac_prep_begin")
        try:
            for e in order:
                _lib.check(lib.fac_prep_set_phase(e.phase), "fac_prep_set_phase")
                kids = e.children
                st.append(e)
                try:
                    e.fn(*e.args, out=e.buf, **e.kw)
                finally:
                    st.pop()
                e.children = kids                             # (the recording pass re-appended them)
            plan = lib.fac_prep_end()
        except Exception:
            lib.fac_prep_abort()
            for e, f in saved:
                e.fresh = f
            raise
        finally:
            if not was_active:
                _ACTIVE.remove(self)
        for e, f in saved:
            e.fresh = f
        if plan < 0:
            raise _lib.FacodecHipError("fac_prep_end: " + _lib.last_error())
        self.plan, self.planned = plan, order
        self.stats["rebuilds"] += 1

    def info(self):
        import ctypes
        nj, nl = ctypes.c_int(0), ctypes.c_int(0)
        if self.plan >= 0:
            _lib.load().fac_prep_info(self.plan, ctypes.byref(nj), ctypes.byref(nl))
        return dict(entries=len(self.entries), jobs=nj.value, launches=nl.value, **self.stats)


def prepared(fn):
    """Decorator of the `ops` weight-preparation functions (first argument: the weight tensor; keyword `out`)."""
    name = fn.__name__

    def wrapper(*args, **kw):
        if not _ACTIVE:
            return fn(*args, **kw)
        v = args[0]
        if not isinstance(v, torch.Tensor) or not v.is_cuda or not v.is_contiguous() or v.dtype != torch.float32:
            return fn(*args, **kw)
        ptr = v.data_ptr()
        me = threading.get_ident()
        for c in _ACTIVE:
            if ptr in c.claims and (c.owner == me or c.any_thread):
                for a in args[1:]:
                    if isinstance(a, torch.Tensor) and (a.data_ptr() not in c.claims or not a.is_contiguous()):
                        return fn(*args, **kw)               # an operand the cache does not own (a temporary): its address means nothing
                for k, a in kw.items():
                    if k != "out" and isinstance(a, torch.Tensor) and (a.data_ptr() not in c.claims or not a.is_contiguous()):
                        return fn(*args, **kw)
                return c.lookup(name, fn, args, dict(kw))
        return fn(*args, **kw)

    wrapper.__name__ = name
    wrapper.__doc__ = fn.__doc__
    wrapper.__wrapped__ = fn
    return wrapper


def cached_forward(method):
    """No-grad forward of a top-level module inside a region of the module's own cache (built on first use over its parameters).
    Forwards that build a graph (their packed weights are saved for a backward the module cannot see the end of), stream captures
    and calls made inside somebody else's region pass through."""

    def forward(self, *args, **kw):
        if (not ENABLED or torch.is_grad_enabled() or not torch.cuda.is_available() or (_ACTIVE and _visible(threading.get_ident()))
                or torch.cuda.is_current_stream_capturing()):
            return method(self, *args, **kw)
        cache = self.__dict__.get("_wcache")
        first = next(self.parameters(), None)
        if first is None or not first.is_cuda:
            return method(self, *args, **kw)
        if cache is not None and cache.claims is not None and first.data_ptr() not in cache.claims:
            drop_cache(self)                                  # the module was moved / its parameters replaced: start over
            cache = None
        if cache is None:
            cache = self.__dict__["_wcache"] = WeightCache(list(self.parameters()), type(self).__name__)
        with cache:
            return method(self, *args, **kw)

    forward.__name__ = method.__name__
    forward.__doc__ = method.__doc__
    forward.__wrapped__ = method
    return forward


def drop_cache(module):
    """Forget a module's cache (after its parameters were moved / replaced wholesale)."""
    c = module.__dict__.pop("_wcache", None)
    if c is not None:
        c.close()
