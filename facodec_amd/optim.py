"""Flat-arena AdamW + ExponentialLR + gradient clipping + data-parallel gradient averaging (reference:
optimizers.py:72-108 `build_optimizer` -> one torch AdamW(betas (0.9, 0.98), eps 1e-9, weight_decay 0.1) and
ExponentialLR(gamma 0.999996) per model key, wrapped in `MultiOptimizer` :11-70; train.py:362-374 clips each key at
1000 and steps; accelerate's DDP averages gradients across ranks, overlapped with backward).

Per model key there are four contiguous arenas (parameters, gradients, first / second moments):
  * every `p.data` is a view of the parameter arena and every `p.grad` a view of the gradient arena, so autograd
    accumulates straight into the arena -- there is no bucket-assembly copy;
  * the data-parallel exchange is ONE all-reduce(mean) of the gradient arena per key (145 / 64 / 342 / 170 MB:
    few, large collectives for the point-to-point xGMI links), launched asynchronously (`launch_all_reduce`) as soon
    as that key's backward is complete and waited for only in `step` -- RCCL runs it on its own stream under the
    rest of the backward pass;
  * the step is `fac_grad_norm_clip` (two-stage reduction) + `fac_adamw_step` (one launch per run of parameters
    that received a gradient; parameters whose gradient autograd never produced are skipped exactly like torch's
    AdamW skips `grad is None` -- no weight decay, no moment update).
"""
import os

import torch
import torch.distributed as dist

from . import _lib, ops


class FlatAdamW:
    def __init__(self, params, lr=1e-4, betas=(0.9, 0.98), eps=1e-9, weight_decay=0.1, gamma=0.999996, max_norm=1000.0):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev = self.params[0].device
        n = sum(p.numel() for p in self.params)
        self.p = torch.empty(n, device=dev, dtype=torch.float32)
        self.g = torch.zeros(n, device=dev, dtype=torch.float32)
        self.m = torch.zeros(n, device=dev, dtype=torch.float32)
        self.v = torch.zeros(n, device=dev, dtype=torch.float32)
        self._scratch = torch.empty(1024, device=dev, dtype=torch.float32)
        self.norm = torch.zeros(2, device=dev, dtype=torch.float32)
        off = 0
        self.slices = []
        self._touched = [False] * len(self.params)
        for i, p in enumerate(self.params):
            k = p.numel()
            self.p[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.p[off:off + k].view_as(p)          # the module now reads its weights from the arena
            p.grad = self.g[off:off + k].view_as(p)          # ... and autograd accumulates into the arena
            p.register_post_accumulate_grad_hook(self._mark(i))
            self.slices.append((off, k))
            off += k
        self.lr, self.base_lr = lr, lr
        self.betas, self.eps, self.wd, self.gamma, self.max_norm = betas, eps, weight_decay, gamma, max_norm
        self.param_steps = [0] * len(self.params)            # torch AdamW keeps one step count per parameter
        self.lr_epochs = 0                                    # ExponentialLR.last_epoch
        self._work = None
        self._need_scale = False
        self._expected = None                                 # parameters the previous step's backward reached

    def _mark(self, i):
        def hook(_p):
            self._touched[i] = True
        return hook

    @property
    def steps(self):
        return max(self.param_steps)

    # ------------------------------------------------------------------------------------------ gradients
    def zero_grad(self):
        """One memset of the gradient arena; the `.grad` views stay bound (torch's zero_grad(set_to_none) would unbind)."""
        self._rebind()
        self.g.zero_()
        self._touched = [False] * len(self.params)

    def _rebind(self):
        """`p.grad` must be the arena view (someone may have set it to None or to a foreign tensor), `p.data` must still
        live in the parameter arena (a later module.to()/.float() would silently detach the optimiser from the model)."""
        base_p, base_g = self.p.data_ptr(), self.g.data_ptr()
        for i, (p, (off, k)) in enumerate(zip(self.params, self.slices)):
            if p.data_ptr() != base_p + 4 * off:
                raise RuntimeError("FlatAdamW: a parameter no longer lives in the optimiser's arena (was the module moved or cast after "
                                   "the optimiser was built?); build FlatAdamW after the model is on its final device")
            if p.grad is None:
                p.grad = self.g[off:off + k].view_as(p)
            elif p.grad.data_ptr() != base_g + 4 * off:       # foreign gradient tensor: fold it in once, then rebind
                self.g[off:off + k].copy_(p.grad.reshape(-1))
                p.grad = self.g[off:off + k].view_as(p)
                self._touched[i] = True

    def gather_grads(self):
        """Compatibility with callers that assign `.grad` tensors by hand: folds them into the arena (no-op for views)."""
        self._rebind()

    def backward_complete(self):
        """True once every parameter the previous step's backward reached has been reached again (the usage pattern of
        the model is static, so all ranks agree): the arena is final and may be handed to the collective early."""
        return self._expected is not None and all(t or not e for t, e in zip(self._touched, self._expected))

    def launch_all_reduce(self, only_if_complete=False):
        """Data-parallel exchange: one asynchronous all-reduce(mean) of the whole arena (RCCL over xGMI under backend
        'nccl'; it runs on the process group's stream, after everything already queued on the current stream).
        only_if_complete: called from inside backward (gradient hooks) -- launch only when `backward_complete()`."""
        if self._work is not None or not (dist.is_available() and dist.is_initialized()):
            return
        # a single rank has nothing to exchange; FAC_FORCE_ALLREDUCE=1 still issues the collective (smoke-tests the RCCL path --
        # AVG op, async work handle, stream hand-over -- on a one-GPU box: the mean over one rank is the identity)
        if dist.get_world_size() == 1 and os.environ.get("FAC_FORCE_ALLREDUCE") != "1":
            return
        if only_if_complete and not self.backward_complete():
            return
        if dist.get_backend() == "nccl":          # RCCL: the mean is part of the collective
            self._work = dist.all_reduce(self.g, op=dist.ReduceOp.AVG, async_op=True)
        else:   # gloo (CPU test scaffold, or two ranks sharing one GPU in a smoke run) has no AVG
            self._work = dist.all_reduce(self.g, op=dist.ReduceOp.SUM, async_op=True)
            self._need_scale = True

    def wait_all_reduce(self):
        if self._work is not None:
            self._work.wait()                                 # makes the current stream wait for the collective
            self._work = None
            if self._need_scale:
                self.g.mul_(1.0 / dist.get_world_size())
                self._need_scale = False

    def all_reduce_mean(self):
        self.launch_all_reduce()
        self.wait_all_reduce()

    # ------------------------------------------------------------------------------------------ step
    def _active_runs(self):
        """Maximal runs of consecutive parameters that received a gradient and share a step count -> (offset, n, step)."""
        runs = []
        for i, (off, k) in enumerate(self.slices):
            if not self._touched[i]:
                continue
            st = self.param_steps[i] + 1
            if runs and runs[-1][0] + runs[-1][1] == off and runs[-1][2] == st:
                runs[-1][1] += k
            else:
                runs.append([off, k, st])
        return runs

    def step(self, zero_grad=True, advance_lr=True):
        lib = _lib.load()
        n = self.p.numel()
        st = ops._stream()
        self._rebind()
        self.all_reduce_mean()
        clip = None
        if self.max_norm is not None:
            _lib.check(lib.fac_grad_norm_clip(ops._ptr(self.g), n, self.max_norm, ops._ptr(self._scratch), ops._ptr(self.norm), st),
                       "fac_grad_norm_clip")
            clip = self.norm
        for off, k, step in self._active_runs():
            sl = slice(off, off + k)
            _lib.check(lib.fac_adamw_step(ops._ptr(self.p[sl]), ops._ptr(self.g[sl]), ops._ptr(self.m[sl]), ops._ptr(self.v[sl]), k,
                                          self.lr, self.betas[0], self.betas[1], self.eps, self.wd, step, ops._ptr(clip), st),
                       "fac_adamw_step")
        for i, t in enumerate(self._touched):
            if t:
                self.param_steps[i] += 1
        self._expected = tuple(self._touched)
        if advance_lr:
            self.scheduler_step()
        if zero_grad:
            self.zero_grad()

    def scheduler_step(self):
        """ExponentialLR.step(), once per iteration (train.py:372-374)."""
        self.lr *= self.gamma
        self.lr_epochs += 1

    def grad_norm(self):
        """Pre-clip gradient norm of the last step (device scalar; a copy: the next step overwrites the buffer)."""
        return self.norm[0].clone()

    def params_without_grad(self):
        """Indices of the parameters the last backward did not reach."""
        return [i for i, t in enumerate(self._touched) if not t]

    # ------------------------------------------------------------------------------------------ checkpoint state
    def state_dict(self):
        """torch.optim.AdamW.state_dict() layout (what the reference's checkpoints carry per key, optimizers.py:17-20):
        state[i] = {step, exp_avg, exp_avg_sq} for the parameters that have been stepped, one param group."""
        state = {}
        for i, (p, (off, k)) in enumerate(zip(self.params, self.slices)):
            if self.param_steps[i] > 0:
                state[i] = dict(step=torch.tensor(float(self.param_steps[i])),
                                exp_avg=self.m[off:off + k].view_as(p).clone(), exp_avg_sq=self.v[off:off + k].view_as(p).clone())
        group = dict(lr=self.lr, betas=tuple(self.betas), eps=self.eps, weight_decay=self.wd, amsgrad=False, maximize=False,
                     foreach=None, capturable=False, differentiable=False, fused=None, initial_lr=self.base_lr,
                     params=list(range(len(self.params))))
        return dict(state=state, param_groups=[group])

    def load_state_dict(self, sd):
        group = sd["param_groups"][0]
        if len(group["params"]) != len(self.params):
            raise ValueError(f"optimizer state has {len(group['params'])} parameters, this key has {len(self.params)}")
        self.lr = float(group["lr"])
        self.base_lr = float(group.get("initial_lr", self.base_lr))
        self.betas, self.eps, self.wd = tuple(group["betas"]), float(group["eps"]), float(group["weight_decay"])
        self.m.zero_()
        self.v.zero_()
        self.param_steps = [0] * len(self.params)
        for idx, pid in enumerate(group["params"]):
            st = sd["state"].get(pid)
            if st is None:
                continue
            off, k = self.slices[idx]
            self.m[off:off + k].copy_(st["exp_avg"].reshape(-1))
            self.v[off:off + k].copy_(st["exp_avg_sq"].reshape(-1))
            self.param_steps[idx] = int(float(st["step"]))

    def scheduler_state_dict(self):
        """torch ExponentialLR.state_dict() keys the reference's checkpoints hold (optimizers.py:22-25)."""
        return dict(gamma=self.gamma, base_lrs=[self.base_lr], last_epoch=self.lr_epochs, _step_count=self.lr_epochs + 1,
                    _last_lr=[self.lr])

    def load_scheduler_state_dict(self, sd):
        self.gamma = float(sd["gamma"])
        self.base_lr = float(sd["base_lrs"][0])
        self.lr_epochs = int(sd["last_epoch"])
        self.lr = float(sd["_last_lr"][0]) if "_last_lr" in sd else self.base_lr * self.gamma ** self.lr_epochs


class MultiOptimizer:
    """optimizers.py:11-70 over FlatAdamW: `step(key)`, `scheduler(key=)`, `zero_grad(key)`, and the (key, state) list
    formats of `state_dict` / `scheduler_state_dict` that modules/commons.py:446-471 `load_checkpoint` and train.py's
    checkpoint writer exchange."""

    def __init__(self, optimizers):
        self.optimizers = dict(optimizers)
        self.schedulers = self.optimizers                     # the LR schedule lives in the same object
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
        for k in ([key] if key is not None else self.keys):
            self.optimizers[k].step(zero_grad=False, advance_lr=False)

    def zero_grad(self, key=None):
        for k in ([key] if key is not None else self.keys):
            self.optimizers[k].zero_grad()

    def scheduler(self, *args, key=None):
        for k in ([key] if key is not None else self.keys):
            self.optimizers[k].scheduler_step()


def build_optimizer(model_dict, scheduler_params_dict=None, lr=1e-4, type="AdamW"):
    """optimizers.py:72-108 (AdamW only; ScaledAdam belongs to the unused transformer_modules tree)."""
    if type != "AdamW":
        raise ValueError("Unknown optimizer type: %s" % type)
    max_norm = {"discriminator": 10.0}                       # train.py:290 vs :362-365
    return MultiOptimizer({k: FlatAdamW(m.parameters(), lr=lr, max_norm=max_norm.get(k, 1000.0)) for k, m in model_dict.items()})
