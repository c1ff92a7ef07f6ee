"""Code-index mismatch triage (SURVEY.md 7 "hard parts"): codes are arg-max decisions, so two fp32-grade evaluations of the
same network may legitimately differ where the best and second-best code of a latent are (nearly) tied, and every later
stage of that residual chain then sees a different residual.  `classify_code_mismatches` separates those cases from real
errors: per (stream, clip, frame) it looks at the FIRST residual stage that differs and measures, on this run's own
projected latent, how much farther the expected code is than the chosen one (dac/nn/quantize.py:78-94 distance: squared
distance between the L2-normalised latent and the L2-normalised code).  A gap below `tie_tol` is a near-tie flip, stages
after it are its cascade; anything else is a genuine mismatch."""
import hashlib

import numpy as np
import torch

from .quantize import ResidualVectorQuantize

REFERENCE_RUNS = ("f32_mt", "f32_1t", "f64")


def check_codes_decidable(codes, fixture):
    """Exact code check of configs[1] with the only exception the REFERENCE ITSELF makes.

    fixture: tests/golden/codec_b32_decidable.npz (tests/golden/make_golden_bench.py b32_decidable): the real reference on the
    timed batch in fp32 / all threads, fp32 / one thread and fp64; `decidable[b, i, t]` = the three runs chose the same code.
    codes: (B, 6, T) -- prosody | content x 2 | residual x 3, the order FAquantizer.forward_v2 runs them
    (modules/quantize.py:398-417), or the list of the three streams it returns.

    Rule (no tolerance anywhere):
      * a decidable position must equal the reference's code;
      * a frame (b, t) that holds an undecidable position must, as a whole 6-code column, equal the column of ONE of the
        reference's own runs -- so an undecidable position takes one of the reference's answers, and the later residual stages
        of that frame (their input depends on that choice, dac/nn/quantize.py:173-193) follow THAT run, nothing else.
    -> dict(ok, decidable_mismatches, undecidable_frames_not_a_reference_column, differs_from_fp32_reference, equals_run, ...)."""
    if isinstance(codes, (list, tuple)):
        codes = torch.cat([torch.as_tensor(c) for c in codes], 1)
    got = torch.as_tensor(codes).cpu().numpy().astype(np.int64)
    runs = {k: fixture["codes_" + k].astype(np.int64) for k in REFERENCE_RUNS}
    dec = fixture["decidable"].astype(bool)
    assert got.shape == dec.shape, (got.shape, dec.shape)
    ref = runs["f32_mt"]
    diff = got != ref
    undec_frames = (~dec).any(1)                                           # (B, T)
    column_of_a_run = np.zeros_like(undec_frames)
    for r in runs.values():
        column_of_a_run |= (got == r).all(1)
    bad_cols = undec_frames & ~column_of_a_run
    bad_dec = diff & dec
    sha = lambda a: hashlib.sha256(a.astype(np.int16).tobytes()).hexdigest()   # noqa: E731
    got_sha = sha(got)
    return dict(
        ok=bool(bad_dec.sum() == 0 and bad_cols.sum() == 0),
        positions=int(dec.size), decidable=int(dec.sum()),
        decidable_mismatches=int(bad_dec.sum()),
        decidable_mismatch_positions=np.argwhere(bad_dec)[:16].tolist(),
        # how close the reference's own call was at those positions: its fp64 top-2 gap there (the first differing stage of a frame is
        # the flip, later stages of the same frame follow it)
        decidable_mismatch_fp64_gaps=[float(fixture["gap_f64"][tuple(p)]) for p in np.argwhere(bad_dec)[:16]] if "gap_f64" in fixture else None,
        undecidable_positions=np.argwhere(~dec).tolist(),
        undecidable_frames=int(undec_frames.sum()),
        undecidable_frames_not_a_reference_column=int(bad_cols.sum()),
        differs_from_fp32_reference=int(diff.sum()),
        equals_run=[k for k, r in runs.items() if np.array_equal(got, r)],
        sha256=got_sha, sha256_equal=got_sha == sha(ref), sha256_equal_fp64_reference=got_sha == sha(runs["f64"]))


# which earlier stages decide the INPUT of stage i (modules/quantize.py:398-417: prosody | content x 2 on the latent, residual x 3 on
# x - z_p - z_c; inside an RVQ every stage sees the residual the earlier ones left, dac/nn/quantize.py:173-193)
FEEDS = {0: (), 1: (), 2: (1,), 3: (0, 1, 2), 4: (0, 1, 2, 3), 5: (0, 1, 2, 3, 4)}


def check_codes_decidable_noise(codes, fixture, batch=0):
    """The exact code check with the reference's OWN arithmetic noise in the definition of "decidable" (VERDICT r5 item 5).

    fixture: tests/golden/codec_b32x4_decidable.npz (make_golden_bench.py b32x4_decidable): for each of four 32-clip batches the
    real reference in fp32 / all threads, fp32 / one thread and fp64, the fp64 run's second-best code and top-2 gap per position, and
    `noise[batch]` = the largest difference that batch's fp32 runs show against its fp64 run in the margin between those two codes.
    `decidable` = the three runs agree AND the fp64 gap >= noise[batch]: a position the runs agree on with a smaller gap agrees by
    luck (the reference's own fp32 rounding could have flipped it), so there either of the fp64 run's two best codes is the
    reference's answer.  codes: (32, 6, 160) or the three streams FAquantizer.forward_v2 returns, for batch `batch` of the fixture.

    Rule, stage by stage in the order the quantizers run (no tolerance anywhere).  R = the reference runs whose codes equal the
    product's on every stage feeding this one (FEEDS):
      * R not empty: the code must be the answer of a run in R -- or, where all three runs agree with an fp64 gap below the batch's
        noise, the fp64 run's second-best code (a NOISE FLIP; listed with its gap);
      * R empty because an earlier stage of the frame took a noise flip: nothing the reference computed says what follows (no run
        went that way) -- counted as that flip's cascade, not compared;
      * R empty for any other reason: an earlier stage was already wrong; counted as a mismatch as well.
    -> dict(ok, mismatches, noise_flips [[clip, stage, frame, fp64 gap] ...], cascade_positions, differs_from_fp32 / _fp64, ...)."""
    if isinstance(codes, (list, tuple)):
        codes = torch.cat([torch.as_tensor(c) for c in codes], 1)
    got = torch.as_tensor(codes).cpu().numpy().astype(np.int64)
    runs = [fixture["codes_" + k][batch].astype(np.int64) for k in REFERENCE_RUNS]
    agree, dec = fixture["agree"][batch].astype(bool), fixture["decidable"][batch].astype(bool)
    second, gap = fixture["second_f64"][batch].astype(np.int64), fixture["gap_f64"][batch]
    assert got.shape == dec.shape, (got.shape, dec.shape)
    B, n, T = got.shape
    eq = [got == r for r in runs]                                            # per run (B, n, T)
    bad = np.zeros((B, n, T), bool)
    flip = np.zeros((B, n, T), bool)
    cascade = np.zeros((B, n, T), bool)
    flipped_upstream = np.zeros((B, n, T), bool)                             # a stage feeding this one took a noise flip
    for i in range(n):
        cons = [np.ones((B, T), bool) for _ in runs]
        for j in FEEDS[i]:
            cons = [c & e[:, j] for c, e in zip(cons, eq)]
            flipped_upstream[:, i] |= flip[:, j] | flipped_upstream[:, j]
        any_cons = cons[0] | cons[1] | cons[2]
        answered = (cons[0] & eq[0][:, i]) | (cons[1] & eq[1][:, i]) | (cons[2] & eq[2][:, i])
        all_cons = cons[0] & cons[1] & cons[2]
        noise_ok = all_cons & agree[:, i] & ~dec[:, i] & (got[:, i] == second[:, i])
        flip[:, i] = noise_ok & ~answered
        cascade[:, i] = ~any_cons & flipped_upstream[:, i]
        bad[:, i] = ~(answered | noise_ok | cascade[:, i])
    sha = lambda a: hashlib.sha256(a.astype(np.int16).tobytes()).hexdigest()   # noqa: E731
    flips = [[int(b), int(i), int(t), float(gap[b, i, t])] for b, i, t in np.argwhere(flip)]
    bad_pos = np.argwhere(bad)
    return dict(
        ok=bool(bad.sum() == 0), batch=int(batch), positions=int(dec.size), decidable=int(dec.sum()),
        noise=float(fixture["noise"][batch]), mismatches=int(bad.sum()), decidable_mismatches=int((bad & dec).sum()),
        mismatch_positions=bad_pos[:16].tolist(), mismatch_fp64_gaps=[float(gap[tuple(p)]) for p in bad_pos[:16]],
        noise_flips=flips, cascade_positions=int(cascade.sum()),
        runs_agree_but_gap_below_noise=int((agree & ~dec).sum()), runs_disagree=int((~agree).sum()),
        differs_from_fp32=int((~eq[0]).sum()), differs_from_fp64=int((~eq[2]).sum()),
        differs_from_fp64_positions_and_gaps=[[int(b), int(i), int(t), float(gap[b, i, t])] for b, i, t in np.argwhere(~eq[2])[:8]],
        equals_run=[k for k, e in zip(REFERENCE_RUNS, eq) if e.all()], sha256=sha(got))


class LatentCapture:
    """Context manager: records the projected latents (B, 8 n, T) every ResidualVectorQuantize under `module` returns."""

    def __init__(self, module):
        self.rvqs = [(n, m) for n, m in module.named_modules() if isinstance(m, ResidualVectorQuantize)]
        self.latents = {}
        self._handles = []

    def __enter__(self):
        for name, m in self.rvqs:
            self._handles.append(m.register_forward_hook(lambda mod, args, out, name=name: self.latents.__setitem__(name, out[2].detach())))
        return self

    def __exit__(self, *exc):
        for h in self._handles:
            h.remove()
        return False


def _distances(lat, codebook):
    """lat (N, 8), codebook (Kc, 8) -> (N, Kc) squared distances between the normalised vectors (fp64 on the host)."""
    e = torch.nn.functional.normalize(lat.double(), dim=1)
    c = torch.nn.functional.normalize(codebook.double(), dim=1)
    return e.pow(2).sum(1, keepdim=True) - 2 * e @ c.t() + c.pow(2).sum(1, keepdim=True).t()


def flipped_frames(got, expected):
    """(B, n, T) codes -> (B, T) bool: frames at which any stage differs."""
    return (got.cpu().long() != torch.as_tensor(expected).long()).any(1)


def classify_faquantizer_codes(capture, codes, expected_list, tie_tol=1e-5):
    """The three code streams FAquantizer.forward_v2 returns (prosody, content, residual: modules/quantize.py:398-437) against
    the reference's; a frame whose prosody or content code flipped feeds the residual quantizer a different input, so its
    residual mismatches are counted as that flip's cascade.  capture: the LatentCapture the forward ran under.
    -> {quantizer name: classify_code_mismatches result}."""
    rvqs, report, upstream = dict(capture.rvqs), {}, None
    for (name, _), c, e in zip(capture.rvqs, codes, expected_list):
        is_residual = name.startswith("residual")
        report[name] = classify_code_mismatches(rvqs[name], capture.latents[name], c, e, tie_tol, upstream if is_residual else None)
        f = flipped_frames(c, e)
        upstream = f if upstream is None else (upstream | f)
    return report


def classify_code_mismatches(rvq, latents, got, expected, tie_tol=1e-5, upstream_flips=None):
    """rvq: the ResidualVectorQuantize that produced `got` (B, n, T) with projected latents (B, 8 n, T); expected: the
    reference's codes, same shape.  upstream_flips: optional (B, T) bool -- frames at which a quantizer FEEDING this one already
    differs (modules/quantize.py:411: the residual quantizer's input is x - z_p - z_c, frame by frame), whose mismatches here are
    that flip's cascade (`flipped_frames`).  -> dict(mismatches, near_tie, cascade, genuine, worst_gap)."""
    got, expected = got.cpu().long(), torch.as_tensor(expected).long()
    assert got.shape == expected.shape, (got.shape, expected.shape)
    diff = got != expected
    out = dict(mismatches=int(diff.sum()), near_tie=0, cascade=0, genuine=0, worst_gap=0.0)
    if not out["mismatches"]:
        return out
    lat = latents.cpu()
    B, n, T = got.shape
    for b, t in diff.any(1).nonzero().tolist():
        stages = diff[b, :, t].nonzero().flatten().tolist()
        if upstream_flips is not None and bool(upstream_flips[b, t]):
            out["cascade"] += len(stages)
            continue
        i = stages[0]
        d = _distances(lat[b, 8 * i: 8 * i + 8, t][None], rvq.quantizers[i].codebook.weight.detach().cpu())[0]
        gap = float(d[expected[b, i, t]] - d[got[b, i, t]])
        out["worst_gap"] = max(out["worst_gap"], abs(gap))
        if abs(gap) <= tie_tol:
            out["near_tie"] += 1
            out["cascade"] += len(stages) - 1
        else:
            out["genuine"] += len(stages)
    return out
