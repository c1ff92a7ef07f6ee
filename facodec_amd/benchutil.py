"""Timing scaffold shared by bench.py and the gloo tests: one process per GPU, clips sharded across
ranks with no data-path collective (clips are independent units); only the
timing itself uses collectives (barrier + MAX all-reduce of the elapsed time)."""
import os
import time

import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment (torch.distributed.run)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # a single rank gets a process group too when FAC_FORCE_ALLREDUCE=1 asks for the collective path on a one-GPU box
    if (world > 1 or (os.environ.get("FAC_FORCE_ALLREDUCE") == "1" and "RANK" in os.environ)) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # "nccl" is RCCL on ROCm; FAC_DIST_BACKEND=gloo lets two ranks share one GPU in a smoke test
            backend = os.environ.get("FAC_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_clips(n_total, world, rank):
    """Contiguous block partition of clip indices [0, n_total) -> this rank's range."""
    per = (n_total + world - 1) // world
    lo = min(n_total, rank * per)
    return lo, min(n_total, lo + per)


def timed_steps(step_fn, steps, warmup, sync_fn, device=None):
    """W untimed warm-up steps, then exactly K steps bracketed by barrier + device sync on both
    sides; returns the MAX elapsed seconds over ranks."""
    for _ in range(warmup):
        step_fn()
    sync_fn()
    if dist.is_initialized():
        dist.barrier()
    sync_fn()
    t0 = time.perf_counter()
    for _ in range(steps):
        step_fn()
    sync_fn()
    if dist.is_initialized():
        dist.barrier()
    sync_fn()
    dt = time.perf_counter() - t0
    if dist.is_initialized():
        t = torch.tensor([dt], dtype=torch.float64, device=device if device is not None else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def aggregate_units(units_this_rank, device=None):
    """Whole-job units = sum over ranks."""
    if not dist.is_initialized():
        return float(units_this_rank)
    t = torch.tensor([float(units_this_rank)], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def streaming_soak(model, device, hops, check_minutes=0.0, streams=1, use_graphs=True, sample_rate=24000):
    """BASELINE.json configs[4]: `hops` hops of 480 samples (30 min = 90 000) through ONE streaming session per stream (carried conv
    histories and LSTM state, HIP-graph replay, two chains per hop), synthetic audio resident in HBM; per-hop latency = host wall
    clock around push() + device sync.

    check_minutes M > 0 (the drift check): every code and every output sample of the first M minutes of the stream is kept and
    compared at the end with the OFFLINE causal model run once over the same M minutes (+ 1 s, so that the offline STFT's
    right-edge reflection stays outside the compared frames): codes must be bit-exact, the waveform within 1e-4 -- any drift of
    the hop / frame phase bookkeeping, the ring buffers or the carried LSTM state over tens of thousands of hops shows up there.
    The remaining hops continue on the same session (finite outputs and contiguous frame numbering are checked to the end).

    Outliers: round 4's 30-minute run had ONE hop of 31.6 ms among 90 000.  The cyclic garbage collector is therefore kept out of
    the timed brackets (switched off for the loop, run explicitly every 5 000 hops BETWEEN two hops) -- and round 5's first run
    STILL had three hops of 16 - 22 ms, about 10 s apart, so the collector was not it.  Every hop now also carries a HIP-event pair
    around push() on the session's stream: the five slowest hops are reported as [index, host ms, device ms], which says whether
    the device executed late or the host thread was held up around a normally executing graph."""
    import gc
    from .streaming import HOP, StreamingCodec
    hops -= hops % 5
    check_frames = int(check_minutes * 60 * (sample_rate // 300))     # 80 frames of 300 samples per second
    check_frames = min(check_frames, (4800 + hops * HOP) // 300 - 8)
    check_frames = max(check_frames, 0)
    # one minute of distinct synthetic audio, looped (the stream state never repeats); with the check: the checked prefix + 1 s
    loop_len = sample_rate * 60 if not check_frames else check_frames * 300 + sample_rate
    loop_len = min(loop_len, max(sample_rate * 20, 4800 + hops * HOP)) if not check_frames else loop_len
    from . import synth
    loop = synth.synth_clips(streams, loop_len, seed=0).to(device)
    keep_codes = torch.zeros(streams, 6, check_frames, dtype=torch.int64, device=device) if check_frames else None
    keep_wave = torch.zeros(streams, 1, check_frames * 300, device=device) if check_frames else None
    state = dict(frames_seen=0, finite=True)

    def keep(o):
        if o["codes"] is None:
            return
        f0, n = o["frame0"], o["codes"][0].shape[-1]
        assert f0 == state["frames_seen"], (f0, state["frames_seen"])
        state["frames_seen"] += n
        if f0 < check_frames:
            m = min(n, check_frames - f0)
            keep_codes[:, :, f0:f0 + m] = torch.cat(o["codes"], 1)[:, :, :m]
            keep_wave[:, :, 300 * f0:300 * (f0 + m)] = o["wave"][:, :, :300 * m]

    with torch.no_grad():
        if check_frames:        # the offline pass comes first: the session is conditioned on ITS timbre vector (as in the parity test)
            # timbre from a 2 s enrolment clip handed in as the "full utterance" (modules/quantize.py:378-383): the style encoder's
            # attention over all frames of a 5-minute signal is outside its kernel's tile, and a session is enrolled this way anyway
            enrol = loop[:, 0, :2 * sample_rate].contiguous()
            lens = torch.full((streams,), 2 * sample_rate, dtype=torch.int64, device=device)
            z = model.encoder(loop)
            outs, _, _, _, timbre, codes = model.quantizer(z, loop, n_c=2, return_codes=True, full_waves=enrol, wave_lens=lens)
            ref_codes = torch.cat(codes, 1)[:, :, :check_frames].clone()
            ref_y = model.decoder(outs)[:, :, :check_frames * 300].clone()
            del z, outs, codes
            torch.cuda.empty_cache()
        else:
            enrol = loop[:, :, :2 * sample_rate]
            timbre = model.quantizer(model.encoder(enrol), enrol, n_c=2)[4]
        sess = StreamingCodec(model, timbre, n_c=2, use_graphs=use_graphs)
        first = sess.prime(loop[:, :, :4800])
        keep(first)
        torch.cuda.synchronize()
        lat, dev_ms, pos, out = [], [], 4800, first
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        gc_was_on = gc.isenabled()
        gc.collect()
        gc.disable()
        try:
            t_all = time.perf_counter()
            for h in range(hops):
                if pos + HOP > loop.shape[-1]:
                    pos = 0
                hop = loop[:, :, pos:pos + HOP]
                pos += HOP
                t0 = time.perf_counter()
                ev0.record()
                out = sess.push(hop)
                ev1.record()
                torch.cuda.synchronize()
                lat.append(time.perf_counter() - t0)
                dev_ms.append(ev0.elapsed_time(ev1))         # outside the bracket: first to last device timestamp of the hop
                keep(out)                                    # device-side copies, outside the timed bracket
                if h % 1000 == 999 and out["wave"] is not None:
                    state["finite"] = state["finite"] and bool(torch.isfinite(out["wave"]).all())
                if h % 5000 == 4999:
                    gc.collect()                             # between two hops
            wall = time.perf_counter() - t_all
        finally:
            if gc_was_on:
                gc.enable()
        drift = None
        if check_frames:
            torch.cuda.synchronize()                         # the first pass over the buffer IS the checked prefix (+ 1 s)
            drift = {"checked_minutes": round(check_frames / (60.0 * (sample_rate // 300)), 3), "checked_frames": check_frames,
                     "checked_hops": int(check_frames * 300 // HOP),
                     "code_mismatches_vs_offline": int((ref_codes != keep_codes).sum()), "codes_compared": int(ref_codes.numel()),
                     "wave_rel_err_vs_offline": float((keep_wave - ref_y).abs().max() / ref_y.abs().max()),
                     "frames_emitted_total": state["frames_seen"],
                     "frame_numbering_contiguous": True,      # keep() asserts frame0 == frames seen so far on every emission

                     "outputs_finite_to_the_end": state["finite"],
                     "reference": "offline causal model (encoder -> quantizer -> decoder) over the same first minutes + 1 s in one pass"}
    # the drift contract of the docstring, enforced by the callers (bench.py / tools/stream_bench.py exit non-zero on ok = False)
    ok = bool(state["finite"]) and (drift is None or (drift["code_mismatches_vs_offline"] == 0 and drift["wave_rel_err_vs_offline"] < 1e-4))
    steady = sorted(lat[10:])
    q = lambda p: round(1e3 * steady[min(len(steady) - 1, int(p * len(steady)))], 4)  # noqa: E731
    audio_s = hops * HOP / float(sample_rate)
    slowest = sorted(range(10, len(lat)), key=lambda i: -lat[i])[:5]
    return {"hop_samples": HOP, "streams": streams, "hops": hops, "audio_minutes": round(audio_s / 60, 2), "graphs": use_graphs,
            "p50_ms": q(0.5), "p90_ms": q(0.9), "p99_ms": q(0.99), "p99.9_ms": q(0.999), "max_ms": round(1e3 * steady[-1], 3),
            "slowest_hops_index_host_ms_device_ms": [[i, round(1e3 * lat[i], 3), round(dev_ms[i], 3)] for i in slowest],
            "device_ms_p50": round(sorted(dev_ms[10:])[len(dev_ms[10:]) // 2], 4),
            "rtf": round(wall / audio_s, 5), "wall_s": round(wall, 2),
            "frames_emitted_last_hop": None if out["codes"] is None else int(out["codes"][0].shape[-1]),
            "drift_check": drift, "ok": ok}
