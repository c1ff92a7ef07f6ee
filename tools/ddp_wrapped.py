#!/usr/bin/env python
"""The reference's own data-parallel wrapping on this build's modules: every model[k] goes through
DistributedDataParallel(find_unused_parameters=True, broadcast_buffers=True) as train.py:49-50,110-111 does
(accelerator.prepare), the optimisers are built AFTER wrapping (train.py:113-116), and one iteration of train.py:265-312,
357-370 is executed literally on the wrappers (module calls, torch reductions on the discriminator's feature maps,
optimizer.zero_grad / backward / clip_grad_norm_ / optimizer.step(key) / optimizer.scheduler(key=)).  Predictor terms are
left out (their targets come from external networks).

Checks, on every rank:
  * construction-time broadcast: ranks start from DIFFERENT weights (seed = rank); after wrapping all equal rank 0's;
  * `model.quantizer.module.timbre_linear / .timbre_norm` (train.py:450-453) reachable through the wrapper;
  * the gradients DDP leaves in `.grad` (views of the optimiser's arena) equal the ones of this build's own data-parallel
    path (TrainStep: one asynchronous all-reduce of the arena per key) on the same clips, same masks;
  * parameters identical across ranks after the optimiser steps.

Two ranks sharing one GPU need gloo (RCCL refuses two ranks on one device):

    FAC_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
        --master-port 29519 tools/ddp_wrapped.py
"""
import json
import os
import sys

import torch
import torch.distributed as dist
import torch.nn.functional as F
from torch.nn.parallel import DistributedDataParallel as DDP

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from facodec_amd import benchutil, losses, synth  # noqa: E402
from facodec_amd.commons import Munch, build_model, default_model_params  # noqa: E402
from facodec_amd.optim import build_optimizer  # noqa: E402
from facodec_amd.train import TrainStep  # noqa: E402

KEYS = ("encoder", "quantizer", "decoder", "discriminator")
GEN = ("encoder", "quantizer", "decoder")


def _build(dev, seed):
    model = build_model(default_model_params())
    for k in KEYS:
        synth.load_synthetic(model[k], seed=seed, prefix=k + ".")
        model[k].to(dev)
        model[k].train()
    return model


def _flat(module):
    return torch.cat([p.detach().reshape(-1) for p in module.parameters()])


def main():
    rank, local_rank, world = benchutil.init_distributed()
    dev = torch.device(f"cuda:{local_rank % torch.cuda.device_count()}")
    torch.cuda.set_device(dev)
    report = {"world": world, "backend": dist.get_backend()}

    # ---- arena path first (its own modules, rank 0's weights on every rank), for the expected gradients
    B, T = 2, 12000
    masks = dict(p=torch.ones(1, B), c=torch.ones(2, B), r=torch.ones(3, B), res=torch.ones(B), dropout=False)
    masks = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in masks.items()}
    wav_seg = synth.synth_clips(B, T, seed=7, rank=rank).to(dev)
    waves = synth.synth_clips(B, 2 * T, seed=8, rank=rank).to(dev).reshape(B, -1)
    wave_lens = torch.tensor([2 * T, 2 * T - 2400], dtype=torch.int64, device=dev)

    ref_model = _build(dev, seed=0)
    w0 = {k: _flat(ref_model[k]).clone() for k in KEYS}
    ref_step = TrainStep(ref_model, lr=1e-4)
    ref_out = ref_step(wav_seg, masks=masks, full_waves=waves, wave_lens=wave_lens)
    want = {k: ref_step.opt[k].g.clone() for k in KEYS}
    want_p = {k: ref_step.opt[k].p.clone() for k in KEYS}

    # ---- the reference's wrapping: different weights per rank, DDP broadcasts rank 0's
    model = _build(dev, seed=rank)
    ddp = Munch()
    for k in KEYS:
        ddp[k] = DDP(model[k], device_ids=[dev.index], find_unused_parameters=True, broadcast_buffers=True)
    report["broadcast_ok"] = all(torch.equal(_flat(ddp[k].module), w0[k]) for k in KEYS)
    optimizer = build_optimizer({k: ddp[k] for k in KEYS}, lr=1e-4)
    report["ddp_keys_skip_arena_exchange"] = all(not optimizer.optimizers[k].data_parallel for k in KEYS)
    mel_criterion = losses.MelSpectrogramLoss(n_mels=[5, 10, 20, 40, 80, 160, 320], window_lengths=[32, 64, 128, 256, 512, 1024, 2048],
                                              mel_fmin=[0] * 7, mel_fmax=[None] * 7, pow=1.0, mag_weight=0.0, clamp_eps=1e-5,
                                              sample_rate=24000)

    # train.py:265-277
    z = ddp.encoder(wav_seg)
    z, quantized, commitment_loss, codebook_loss, timbre = ddp.quantizer(z, wav_seg, n_c=2, full_waves=waves, wave_lens=wave_lens,
                                                                         masks=masks)
    pred_wave = ddp.decoder(z)
    wav_seg_target = wav_seg
    # :279-292
    d_fake = ddp.discriminator(pred_wave.detach())
    d_real = ddp.discriminator(wav_seg_target)
    loss_d = 0
    for x_fake, x_real in zip(d_fake, d_real):
        loss_d += torch.mean(x_fake[-1] ** 2)
        loss_d += torch.mean((1 - x_real[-1]) ** 2)
    optimizer.zero_grad()
    loss_d.backward()
    grad_norm_d = torch.nn.utils.clip_grad_norm_(ddp.discriminator.parameters(), 10.0)
    got = {"discriminator": optimizer.optimizers["discriminator"].g.clone()}      # (clipped in place by the line above)
    clip_d = min(1.0, 10.0 / (float(grad_norm_d) + 1e-6))
    optimizer.step("discriminator")
    optimizer.scheduler(key="discriminator")
    # :294-312
    mel_loss = mel_criterion(pred_wave, wav_seg_target)
    d_fake = ddp.discriminator(pred_wave)
    d_real = ddp.discriminator(wav_seg_target)
    loss_g = 0
    for x_fake in d_fake:
        loss_g += torch.mean((1 - x_fake[-1]) ** 2)
    loss_feature = 0
    for i in range(len(d_fake)):
        for j in range(len(d_fake[i]) - 1):
            loss_feature += F.l1_loss(d_fake[i][j], d_real[i][j].detach())
    # :357-374 without the predictor terms
    loss_gen_all = mel_loss * 15.0 + loss_feature * 1.0 + loss_g * 1.0 + commitment_loss * 0.25 + codebook_loss * 1.0
    optimizer.zero_grad()
    loss_gen_all.backward()
    for k in GEN:
        got[k] = optimizer.optimizers[k].g.clone()
    # train.py:450-453 through the wrapper (before the optimiser moves timbre_linear)
    with torch.no_grad():
        style2 = ddp.quantizer.module.timbre_linear(timbre.detach()).unsqueeze(2)
        gamma, beta = style2.chunk(2, 1)
        x = (quantized[0] + quantized[1] + quantized[2]).detach()
        x = x.transpose(1, 2)
        x = ddp.quantizer.module.timbre_norm(x)
        x = x.transpose(1, 2)
        x = x * gamma + beta
    report["timbre_norm_vs_forward"] = float((x - z.detach()).abs().max() / z.detach().abs().max())

    norms = {k: float(torch.nn.utils.clip_grad_norm_(ddp[k].parameters(), 1000.0)) for k in GEN}
    for k in GEN:
        optimizer.step(k)
    for k in GEN:
        optimizer.scheduler(key=k)
    report["last_lr"] = optimizer.schedulers["encoder"].get_last_lr()[0]            # train.py:384

    torch.cuda.synchronize()
    rel = {}
    for k in KEYS:
        w = want[k] * (clip_d if k == "discriminator" else 1.0)
        rel[k] = dict(max=float((got[k] - w).abs().max() / w.abs().max()),
                      norm=abs(float(got[k].norm()) - float(w.norm())) / float(w.norm()))
    report["grad_vs_arena_path"] = rel
    report["loss_rel"] = {"loss_d": abs(float(loss_d) - float(ref_out["loss_d"])) / abs(float(ref_out["loss_d"])),
                          "loss_gen_all": abs(float(loss_gen_all) - float(ref_out["loss"])) / abs(float(ref_out["loss"]))}
    # first AdamW step: update = lr * g / (|g| + eps), so an entry whose gradient is rounding noise may move by up to lr in
    # either path -- the mean over the arena is the meaningful figure (lr = 1e-4)
    report["param_after_vs_arena_path"] = {k: dict(max=float((optimizer.optimizers[k].p - want_p[k]).abs().max()),
                                                   mean=float((optimizer.optimizers[k].p - want_p[k]).abs().mean())) for k in KEYS}
    sums = torch.stack([optimizer.optimizers[k].p.double().sum() for k in KEYS]).to(dev)
    gathered = [torch.zeros_like(sums) for _ in range(world)]
    dist.all_gather(gathered, sums)
    report["params_identical_across_ranks"] = all(torch.equal(g, gathered[0]) for g in gathered)
    report["gen_grad_norms"] = norms
    ok = (report["broadcast_ok"] and report["ddp_keys_skip_arena_exchange"] and report["params_identical_across_ranks"]
          and report["timbre_norm_vs_forward"] < 1e-5 and all(v < 1e-5 for v in report["loss_rel"].values())
          and all(r["max"] < 3e-3 and r["norm"] < 2e-4 for r in rel.values())
          and all(v["mean"] < 2e-7 for v in report["param_after_vs_arena_path"].values()))
    report["ok"] = bool(ok)
    flag = torch.tensor([1.0 if ok else 0.0], device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        report["ok_all_ranks"] = bool(flag.item() == 1.0)
        print(json.dumps(report))
    dist.destroy_process_group()
    sys.exit(0 if flag.item() == 1.0 else 1)


if __name__ == "__main__":
    main()
