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

  codec_b32_decidable.npz   which of those 30 720 indices the reference itself decides: the same batch through the reference in
                  fp32 / all threads, fp32 / 1 thread and fp64; per position the three answers, the fp64 top-2 gap, `decidable` =
                  all three agree (b32_decidable() below; + codec_b32_decidable_report.json)

  codec_b32x4_decidable.npz  the same for four batches (seeds 0..3) with the reference's own margin noise in the definition of
                  `decidable` (b32x4_decidable() below; + codec_b32x4_decidable_report.json; ~12 minutes)

Run:  python tests/golden/make_golden_bench.py [b32] [b32_decidable] [b32x4_decidable] [train16]   (about 2 + 25 + 30 minutes on 8 cores, < 60 GB RAM, ~100 GB of /tmp)
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


# ------------------------------------------------------------------------------------------------------------------------------
# Which of the 30 720 code indices of configs[1] does the reference itself decide?  
# The reference is not bit-reproducible against itself (8 threads vs 1 thread differ by 8.8e-7), so ONE fp32 run is
# not "the" answer at a frame whose two best codes are nearly tied.  b32_decidable() runs the real reference on the timed batch
# three ways -- fp32 on all threads (the run codec_b32.npz holds), fp32 on ONE thread (another summation order inside oneDNN /
# MKL) and fp64 (same fp32-valued weights and inputs, every operation in double) -- and stores, per (clip, codebook, frame):
# the three answers, the fp64 run's gap between its best and second-best code (dac/nn/quantize.py:86-91 distance) and the
# margin each fp32 run saw between those same two codes.  `decidable` = the three runs agree.  Nothing here is tuned to this
# build: the script never imports facodec_amd beyond the synthetic weights / clips.
def _run_reference_with_vq_capture(dtype, threads, tag, seed=0):
    build_model, recursive_munch = MG.ref_imports()
    from dac.nn.quantize import VectorQuantize
    torch.set_num_threads(threads)
    old = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        with torch.no_grad():
            model = build_model(recursive_munch(MG.model_params()))
            for k in ("encoder", "quantizer", "decoder"):
                synth.load_synthetic(model[k], seed=0, prefix=k + ".")           # fp32-valued weights, exact in fp64
                model[k].to(dtype).eval()
            wave = synth.synth_clips(32, 48000, seed=seed).to(dtype)
            cap = []                                                               # call order = prosody, content x 2, residual x 3
            hooks = [m.register_forward_hook(lambda mod, args, out: cap.append((mod, out[4].detach().clone(), out[3].detach().clone())))
                     for m in model.quantizer.modules() if isinstance(m, VectorQuantize)]
            t0 = time.time()
            z = model.encoder(wave)
            outs, quantized, commit, cbl, timbre, codes = model.quantizer(z, wave, n_c=2, return_codes=True)
            dt = time.time() - t0
            for h in hooks:
                h.remove()
        assert len(cap) == 6, len(cap)
        allc = torch.cat(codes, 1).numpy().astype(np.int16)
        assert all(torch.equal(cap[i][2], torch.from_numpy(allc[:, i].astype(np.int64))) for i in range(6))
        print(f"[b32_decidable] {tag}: {dt:.1f} s, sha256 {hashlib.sha256(allc.tobytes()).hexdigest()[:16]}", flush=True)
        return dict(codes=allc, z_e=[c[1].double() for c in cap], codebooks=[c[0].codebook.weight.detach().double() for c in cap],
                    z=z.double(), seconds=dt)
    finally:
        torch.set_default_dtype(old)


def _pair_margin(z_e, codebook, first, second):
    """d(second) - d(first) of dac/nn/quantize.py:86-91's distance, evaluated in fp64 on the given projected latents (B, 8, T)."""
    e = torch.nn.functional.normalize(z_e.permute(0, 2, 1).reshape(-1, z_e.shape[1]), dim=1)
    c = torch.nn.functional.normalize(codebook, dim=1)
    d = e.pow(2).sum(1, keepdim=True) - 2 * e @ c.t() + c.pow(2).sum(1, keepdim=True).t()
    idx = torch.arange(d.shape[0])
    return (d[idx, second.reshape(-1)] - d[idx, first.reshape(-1)]).reshape(first.shape), d


def b32_decidable():
    ncpu = os.cpu_count()
    runs = {"f32_mt": _run_reference_with_vq_capture(torch.float32, ncpu, f"fp32, {ncpu} threads"),
            "f32_1t": _run_reference_with_vq_capture(torch.float32, 1, "fp32, 1 thread"),
            "f64": _run_reference_with_vq_capture(torch.float64, ncpu, f"fp64, {ncpu} threads")}
    held = np.load(os.path.join(HERE, "codec_b32.npz"))
    assert np.array_equal(runs["f32_mt"]["codes"], held["codes"]), "the fp32 all-threads run no longer reproduces codec_b32.npz"
    B, n, T = runs["f64"]["codes"].shape
    gap64 = np.zeros((B, n, T), np.float64)
    second64 = np.zeros((B, n, T), np.int16)
    margins = {k: np.zeros((B, n, T), np.float64) for k in ("f32_mt", "f32_1t")}
    for i in range(n):
        r = runs["f64"]
        _, d = _pair_margin(r["z_e"][i], r["codebooks"][i], torch.zeros(B, T, dtype=torch.int64), torch.zeros(B, T, dtype=torch.int64))
        top2 = torch.topk(-d, 2, dim=1)
        best = top2.indices[:, 0].reshape(B, T)
        assert torch.equal(best, torch.from_numpy(r["codes"][:, i].astype(np.int64))), i
        sec = top2.indices[:, 1].reshape(B, T)
        gap64[:, i] = (top2.values[:, 0] - top2.values[:, 1]).reshape(B, T).numpy()
        second64[:, i] = sec.numpy()
        for k in margins:          # what the fp32 run saw between the SAME two codes (meaningful where its upstream stages agree with fp64's)
            m, _ = _pair_margin(runs[k]["z_e"][i], runs[k]["codebooks"][i], best, sec)
            margins[k][:, i] = m.numpy()
    agree = (runs["f32_mt"]["codes"] == runs["f32_1t"]["codes"]) & (runs["f32_mt"]["codes"] == runs["f64"]["codes"])
    # a stage's input is final only if every stage feeding it agrees too: prosody -> nothing; content i -> content < i;
    # residual i -> prosody, content, residual < i (modules/quantize.py:398-417: residual input = x - z_p - z_c)
    feeds = {0: [], 1: [], 2: [1], 3: [0, 1, 2], 4: [0, 1, 2, 3], 5: [0, 1, 2, 3, 4]}
    upstream_agree = np.stack([np.all(agree[:, feeds[i]], axis=1) if feeds[i] else np.ones((B, T), bool) for i in range(n)], 1)
    clean = agree & upstream_agree
    noise = {k: np.abs(margins[k] - gap64)[upstream_agree] for k in margins}
    zrel = {k: float((runs[k]["z"] - runs["f64"]["z"]).abs().max() / runs["f64"]["z"].abs().max()) for k in margins}
    report = {
        "positions": int(agree.size), "decidable": int(agree.sum()), "undecidable": int((~agree).sum()),
        "undecidable_with_agreeing_upstream": int((~agree & upstream_agree).sum()),
        "undecidable_positions_clip_codebook_frame": np.argwhere(~agree).tolist(),
        "fp64_gap_at_undecidable": [float(gap64[tuple(p)]) for p in np.argwhere(~agree)],
        "smallest_fp64_gap_among_decidable": float(gap64[clean].min()),
        "decidable_with_fp64_gap_below_1e-6": int((gap64[clean] < 1e-6).sum()),
        "reference_margin_noise_fp32_vs_fp64": {k: {"max": float(v.max()), "p99.9": float(np.quantile(v, 0.999)), "median": float(np.median(v))} for k, v in noise.items()},
        "reference_latent_rel_err_fp32_vs_fp64": zrel,
        "seconds": {k: float(r["seconds"]) for k, r in runs.items()}, "threads": ncpu,
    }
    np.savez_compressed(
        os.path.join(HERE, "codec_b32_decidable.npz"), codes_f32_mt=runs["f32_mt"]["codes"], codes_f32_1t=runs["f32_1t"]["codes"],
        codes_f64=runs["f64"]["codes"], second_f64=second64, gap_f64=gap64.astype(np.float32), decidable=agree,
        margin_f32_mt=margins["f32_mt"].astype(np.float32), margin_f32_1t=margins["f32_1t"].astype(np.float32),
        z_e_f64_probe=np.stack([r.numpy() for r in runs["f64"]["z_e"]], 1)[PROBE_CLIPS].astype(np.float32),
        report=np.array(json.dumps(report)))
    json.dump(report, open(os.path.join(HERE, "codec_b32_decidable_report.json"), "w"), indent=1)
    print(json.dumps(report, indent=1))


# ------------------------------------------------------------------------------------------------------------------------------
# The same self-examination of the reference on FOUR batches (seeds 0..3 of synth.synth_clips(32, 48000): 122 880 code indices),
# and a definition of "decidable" that has the reference's own arithmetic noise in it: per batch, the largest
# difference the reference's fp32 runs show against its fp64 run in the MARGIN between the fp64 run's two best codes (positions
# whose upstream stages agree) is that batch's `noise`; a position is decidable iff the three runs agree AND the fp64 top-2 gap is
# at least that noise.  A position the runs agree on with a smaller gap agrees "by luck rather than by margin": the reference's
# own fp32 rounding could have flipped it, so either of the fp64 run's two best codes is the reference's answer there.
def _decidability_of_one_batch(seed, ncpu):
    runs = {"f32_mt": _run_reference_with_vq_capture(torch.float32, ncpu, f"seed {seed}: fp32, {ncpu} threads", seed),
            "f32_1t": _run_reference_with_vq_capture(torch.float32, 1, f"seed {seed}: fp32, 1 thread", seed),
            "f64": _run_reference_with_vq_capture(torch.float64, ncpu, f"seed {seed}: fp64, {ncpu} threads", seed)}
    B, n, T = runs["f64"]["codes"].shape
    gap64 = np.zeros((B, n, T), np.float64)
    second64 = np.zeros((B, n, T), np.int16)
    margins = {k: np.zeros((B, n, T), np.float64) for k in ("f32_mt", "f32_1t")}
    for i in range(n):
        r = runs["f64"]
        _, d = _pair_margin(r["z_e"][i], r["codebooks"][i], torch.zeros(B, T, dtype=torch.int64), torch.zeros(B, T, dtype=torch.int64))
        top2 = torch.topk(-d, 2, dim=1)
        best = top2.indices[:, 0].reshape(B, T)
        assert torch.equal(best, torch.from_numpy(r["codes"][:, i].astype(np.int64))), i
        sec = top2.indices[:, 1].reshape(B, T)
        gap64[:, i] = (top2.values[:, 0] - top2.values[:, 1]).reshape(B, T).numpy()
        second64[:, i] = sec.numpy()
        for k in margins:
            m, _ = _pair_margin(runs[k]["z_e"][i], runs[k]["codebooks"][i], best, sec)
            margins[k][:, i] = m.numpy()
    agree = (runs["f32_mt"]["codes"] == runs["f32_1t"]["codes"]) & (runs["f32_mt"]["codes"] == runs["f64"]["codes"])
    feeds = {0: [], 1: [], 2: [1], 3: [0, 1, 2], 4: [0, 1, 2, 3], 5: [0, 1, 2, 3, 4]}
    upstream_agree = np.stack([np.all(agree[:, feeds[i]], axis=1) if feeds[i] else np.ones((B, T), bool) for i in range(n)], 1)
    noise = max(float(np.abs(margins[k] - gap64)[upstream_agree].max()) for k in margins)
    return runs, gap64, second64, agree, upstream_agree, noise


def b32x4_decidable(seeds=(0, 1, 2, 3)):
    ncpu = os.cpu_count()
    keep = {k: [] for k in ("codes_f32_mt", "codes_f32_1t", "codes_f64", "second_f64", "gap_f64", "agree", "decidable")}
    noises, report = [], {"seeds": list(seeds), "batches": []}
    for seed in seeds:
        runs, gap64, second64, agree, upstream_agree, noise = _decidability_of_one_batch(seed, ncpu)
        if seed == 0:
            held = np.load(os.path.join(HERE, "codec_b32_decidable.npz"))
            assert all(np.array_equal(runs[k]["codes"], held["codes_" + k]) for k in runs), "seed 0 no longer reproduces codec_b32_decidable.npz"
        decidable = agree & (gap64 >= noise)
        for k in ("f32_mt", "f32_1t", "f64"):
            keep["codes_" + k].append(runs[k]["codes"])
        keep["second_f64"].append(second64)
        keep["gap_f64"].append(gap64.astype(np.float32))
        keep["agree"].append(agree)
        keep["decidable"].append(decidable)
        noises.append(noise)
        below = agree & ~decidable
        report["batches"].append({
            "seed": int(seed), "positions": int(agree.size), "runs_agree": int(agree.sum()), "runs_disagree": int((~agree).sum()),
            "runs_disagree_positions_clip_codebook_frame": np.argwhere(~agree).tolist(),
            "fp64_gap_where_runs_disagree": [float(gap64[tuple(p)]) for p in np.argwhere(~agree)],
            "margin_noise_fp32_vs_fp64_max": noise,
            "agree_but_gap_below_noise": int(below.sum()),
            "agree_but_gap_below_noise_positions": np.argwhere(below).tolist(),
            "agree_but_gap_below_noise_fp64_gaps": [float(gap64[tuple(p)]) for p in np.argwhere(below)],
            "decidable": int(decidable.sum()),
            "smallest_fp64_gap_among_decidable": float(gap64[decidable].min()),
            "fp32_runs_bit_identical": bool(np.array_equal(runs["f32_mt"]["codes"], runs["f32_1t"]["codes"])),
            "fp32_vs_fp64_code_differences": int((runs["f32_mt"]["codes"] != runs["f64"]["codes"]).sum()),
            "seconds": {k: float(r["seconds"]) for k, r in runs.items()}})
        print(json.dumps(report["batches"][-1]), flush=True)
    report["rule"] = ("decidable = the reference's three runs agree AND its fp64 top-2 gap >= the batch's measured fp32-vs-fp64 margin noise; "
                      "see facodec_amd.diagnostics.check_codes_decidable_noise")
    np.savez_compressed(os.path.join(HERE, "codec_b32x4_decidable.npz"), seeds=np.array(seeds), noise=np.array(noises, np.float64),
                        report=np.array(json.dumps(report)), **{k: np.stack(v) for k, v in keep.items()})
    json.dump(report, open(os.path.join(HERE, "codec_b32x4_decidable_report.json"), "w"), indent=1)


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
    if "b32_decidable" in what:
        b32_decidable()
    if "b32x4_decidable" in what:
        b32x4_decidable()
    if "train16" in what:
        train16()
