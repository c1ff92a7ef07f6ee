"""Times fac_vq_fwd at the benchmark's shape (B = 32 clips x 160 frames, D = 1024, 1024 x 8 codebook; dac/nn/quantize.py:34-94)
with HIP events: algorithmic bytes = latent read + residual write + accumulator read / write = 4 B D T 4 bytes per launch.
  python tools/vq_bench.py [B T D]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from facodec_amd import ops, synth  # noqa: E402
from facodec_amd.quantize import VectorQuantize  # noqa: E402


def main():
    B, T, D = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (32, 160, 1024)
    dev = torch.device("cuda:0")
    q = VectorQuantize(D, 1024, 8).eval()
    synth.load_synthetic(q, seed=5)
    q = q.to(dev)
    z = torch.randn(B, D, T, device=dev)
    w_in, w_out, sc = q._weights()
    codes = torch.empty(B, T, device=dev, dtype=torch.int64)
    z_e = torch.empty(B, 8, T, device=dev)
    res, acc = torch.empty_like(z), torch.zeros_like(z)
    lp = torch.empty(B, ops.vq_loss_tiles(T), device=dev)

    def run():
        ops.vq_step(z, w_in, q.in_proj.bias.detach(), q.codebook.weight.detach(), w_out, sc, q.out_proj.bias.detach(), codes,
                    residual=res, zq_acc=acc, z_e=z_e, loss_part=lp)
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    n = 50
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / n
    alg = 4 * B * D * T * 4
    print(f"fac_vq_fwd B={B} T={T} D={D}: {us:.1f} us per launch, algorithmic {alg / 1e6:.1f} MB -> {alg / us / 1e6:.2f} TB/s "
          f"= {alg / us / 1e6 / 8.0:.3f} of 8 TB/s; tiles {ops.vq_loss_tiles(T)} x {B} workgroups")


if __name__ == "__main__":
    main()
