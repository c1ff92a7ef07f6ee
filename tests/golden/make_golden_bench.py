"""Reference-made fixtures AT THE BENCHMARK'S OWN SIZES, from the REAL reference
imported in the build container (it never travels):

  codec_b32.npz   BASELINE.json configs[1]: the 32 clips x 2 s of bench.py's own timed batch (synth.synth_clips(32, 48000,
                  seed=0)) through model.encoder -> model.quantizer(return_codes=True) -> model.decoder in eval mode:
                  all 32 x 6 x 160 code indices (int16, 61 KB), their sha256, and probes of four clips (latent, quantizer
                  output, waveform, timbre).
  train_b16.npz   configs[2]: ONE train.py:188-374 iteration on 16 segments x 2 s (160 frames) cropped from 16 padded
                  utterances of 2.4 s: the 17 loss scalars and the five pre-clip gradient norms (tests/golden/
                  make_golden_train.py's `iteration`, the same code that made train_step.npz at B = 4 x 0.25 s), plus the
                  recorded random draws.  Inputs are regenerated from facodec_amd/synth.py by the test; only results are stored.

The autograd graph of the reference at 16 x 2 s does not fit this container's 62 GB (its TorchScript Snake alone keeps three
tensors per activation): `DiskOffload` is a torch.autograd.graph.saved_tensors_hooks pair that parks every saved tensor of
>= 16 MB in a file under /tmp (deduplicated by content hash) and reads it back when backward asks for it -- the reference's
code and arithmetic are untouched, only where its saved activations wait changes.

Run:  python tests/golden/make_golden_bench.py [b32] [train16]      (about 2 + 30 minutes on 8 cores, < 60 GB RAM, ~100 GB of /tmp)
"""
import hashlib
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)

import make_golden as MG  # noqa: E402
import make_golden_train as MGT  # noqa: E402
from facodec_amd import synth  # noqa: E402

PROBE_CLIPS = [0, 9, 17, 31]

# configs[2] inputs: 16 utterances padded to 2.4 s, lengths in frames of 300 samples; crops of 160 frames
B16 = 16
T_FULL16 = 72000
WAVE_LENS16 = [72000, 60000, 66000, 72000, 54000, 69000, 72000, 63000, 57000, 72000, 70500, 61500, 72000, 52500, 67500, 72000]
CROP16 = [40, 11, 0, 80, 20, 55, 3, 47, 30, 79, 64, 5, 17, 14, 33, 71]
DRAWS16 = {"p": [1] * 16, "c": [1, 2, 2, 1, 1, 2, 1, 2] + [1] * 8, "r": [2, 1, 3, 3, 1, 2, 3, 1] + [1] * 8}
RES_MASK16 = [1, 0, 1, 1, 1, 1, 0, 1, 1, 1, 0, 1, 1, 1, 1, 0]


class DiskOffload:
    def __init__(self, root="/tmp/facodec_golden_offload", min_bytes=16 << 20):
        import shutil
        shutil.rmtree(root, ignore_errors=True)
        os.makedirs(root)
        self.root, self.min_bytes, self.bytes_written, self.bytes_saved = root, min_bytes, 0, 0

    def pack(self, t):
        nbytes = t.numel() * t.element_size()
        if nbytes < self.min_bytes or t.device.type != "cpu" or t.dtype not in (torch.float32, torch.int64, torch.bool):
            return t
        import xxhash
        a = t.detach().contiguous().numpy()
        h = xxhash.xxh3_128(a.view(np.uint8).reshape(-1)).hexdigest()
        path = os.path.join(self.root, h + ".bin")
        if not os.path.exists(path):
            a.tofile(path)
            self.bytes_written += nbytes
        self.bytes_saved += nbytes
        return (path, tuple(t.shape), a.dtype)

    def unpack(self, obj):
        if torch.is_tensor(obj):
            return obj
        path, shape, dtype = obj
        return torch.from_numpy(np.fromfile(path, dtype=dtype)).reshape(shape)

    def close(self):
        import shutil
        shutil.rmtree(self.root, ignore_errors=True)


def b32():
    build_model, recursive_munch = MG.ref_imports()
    with torch.no_grad():
        model = build_model(recursive_munch(MG.model_params()))
        for k in ("encoder", "quantizer", "decoder"):
            synth.load_synthetic(model[k], seed=0, prefix=k + ".")
            model[k].eval()
        wave = synth.synth_clips(32, 48000, seed=0)                    # bench.py's timed batch on rank 0
        t0 = time.time()
        z = model.encoder(wave)
        outs, quantized, commit, cbl, timbre, codes = model.quantizer(z, wave, n_c=2, return_codes=True)
        y = model.decoder(outs)
        dt = time.time() - t0
    allc = torch.cat(codes, 1).numpy().astype(np.int16)               # (32, 6, 160): prosody | content x 2 | residual x 3
    probe_t = np.arange(0, 48000, 47)
    pc = PROBE_CLIPS
    np.savez_compressed(
        os.path.join(HERE, "codec_b32.npz"), codes=allc, codes_sha256=np.array(hashlib.sha256(allc.tobytes()).hexdigest()),
        probe_clips=np.array(pc), z_probe=z[pc][:, ::8, :].numpy(), outs_probe=outs[pc][:, ::8, :].numpy(),
        wave_probe=y[pc][:, 0, probe_t].numpy(), probe_t=probe_t, timbre=timbre[pc].numpy(), commitment=np.float32(commit),
        codebook=np.float32(cbl), wave_absmax=np.float32(y.abs().max()), z_absmax=np.float32(z.abs().max()),
        outs_absmax=np.float32(outs.abs().max()), reference_cpu_seconds=np.float32(dt), reference_cpu_threads=np.int64(torch.get_num_threads()))
    print(f"[b32] reference forward of 32 x 2 s: {dt:.1f} s on {torch.get_num_threads()} threads = {64.0 / dt:.2f} audio-s/s; "
          f"codes sha256 {hashlib.sha256(allc.tobytes()).hexdigest()[:16]}")


def train16():
    model, _ = MGT.build_reference_in_train_mode()
    cfg = dict(B=B16, SEG_FRAMES=160, T_FULL=T_FULL16, WAVE_LENS=WAVE_LENS16, CROP_START=CROP16, DROPOUT_DRAWS=DRAWS16,
               RES_MASK=RES_MASK16, wave_seed=41, target_seed=77, probes=False)
    for b, (n, s) in enumerate(zip(WAVE_LENS16, CROP16)):
        assert n % 300 == 0 and (s + 160) * 300 <= n, (b, n, s)
    t0 = time.time()
    off = DiskOffload()
    try:
        with torch.autograd.graph.saved_tensors_hooks(off.pack, off.unpack):
            out, _ = MGT.iteration(model, cfg)
    finally:
        print(f"[train16] saved-tensor offload: {off.bytes_saved / 2 ** 30:.1f} GiB requested, {off.bytes_written / 2 ** 30:.1f} GiB written")
        off.close()
    dt = time.time() - t0
    keep = {k: v for k, v in out.items() if isinstance(v, np.floating) or k in (
        "mask_p", "mask_c", "mask_r", "mask_res", "wave_lens", "crop_start", "seg_frames", "f0_targets", "real_norm", "phones", "speaker",
        "params_without_grad")}
    keep.update(wave_seed=np.int64(41), t_full=np.int64(T_FULL16), pred_wave_probe=out["pred_wave_probe"][:, ::29],
                z_probe=out["z_probe"][:, ::8, ::4], reference_cpu_seconds=np.float32(dt),
                reference_cpu_threads=np.int64(torch.get_num_threads()))
    np.savez_compressed(os.path.join(HERE, "train_b16.npz"), **keep)
    print(f"[train16] reference iteration of 16 x 2 s: {dt:.1f} s on {torch.get_num_threads()} threads")
    print(json.dumps({k: float(v) for k, v in keep.items() if isinstance(v, np.floating)}, indent=1))


if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count())
    what = sys.argv[1:] or ["b32", "train16"]
    if "b32" in what:
        b32()
    if "train16" in what:
        train16()
