"""Flat-arena AdamW + ExponentialLR + gradient clipping + data-parallel gradient averaging (reference:
optimizers.py:72-108 `build_optimizer` -> one torch AdamW(betas (0.9, 0.98), eps 1e-9, weight_decay 0.1) and
ExponentialLR(gamma 0.999996) per model key; train.py:362-374 clips each key at 1000 and steps; accelerate's DDP
averages gradients across ranks).

Per model key: parameters are re-pointed into ONE contiguous arena, so the step is one fused kernel
(`fac_adamw_step`), the gradient norm one two-stage reduction, and the data-parallel exchange ONE all-reduce of the
gradient arena over RCCL (few, large collectives: 145 / 64 / 342 MB for encoder / quantizer / decoder)."""
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
        for p in self.params:
            k = p.numel()
            self.p[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.p[off:off + k].view_as(p)          # the module now reads its weights from the arena
            self.slices.append((off, k))
            off += k
        self.lr, self.betas, self.eps, self.wd, self.gamma, self.max_norm = lr, betas, eps, weight_decay, gamma, max_norm
        self.steps = 0

    def gather_grads(self):
        """Copies every .grad into the gradient arena (bucket assembly; missing gradients count as zero, like DDP's
        find_unused_parameters)."""
        for p, (off, k) in zip(self.params, self.slices):
            if p.grad is None:
                self.g[off:off + k].zero_()
            else:
                self.g[off:off + k].copy_(p.grad.reshape(-1))

    def all_reduce_mean(self):
        """Data-parallel exchange: one all-reduce of the whole arena (RCCL over xGMI under backend 'nccl')."""
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(self.g, op=dist.ReduceOp.SUM)
            self.g.div_(dist.get_world_size())

    def step(self, zero_grad=True):
        lib = _lib.load()
        n = self.p.numel()
        st = ops._stream()
        self.gather_grads()
        self.all_reduce_mean()
        clip = None
        if self.max_norm is not None:
            _lib.check(lib.fac_grad_norm_clip(ops._ptr(self.g), n, self.max_norm, ops._ptr(self._scratch), ops._ptr(self.norm), st),
                       "fac_grad_norm_clip")
            clip = self.norm
        self.steps += 1
        _lib.check(lib.fac_adamw_step(ops._ptr(self.p), ops._ptr(self.g), ops._ptr(self.m), ops._ptr(self.v), n, self.lr,
                                      self.betas[0], self.betas[1], self.eps, self.wd, self.steps, ops._ptr(clip), st),
                   "fac_adamw_step")
        self.lr *= self.gamma                                 # ExponentialLR, stepped once per iteration (train.py:372-374)
        if zero_grad:
            for p in self.params:
                p.grad = None

    def grad_norm(self):
        """Pre-clip gradient norm of the last step (device scalar)."""
        return self.norm[0]
