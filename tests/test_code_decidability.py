"""configs[1]'s exact code check: the fixture that says which of the 30 720 code indices the REFERENCE
decides (tests/golden/codec_b32_decidable.npz: the real reference on bench.py's timed batch in fp32 / all threads, fp32 / one
thread, fp64 -- tests/golden/make_golden_bench.py b32_decidable), the rule built on it
(facodec_amd.diagnostics.check_codes_decidable), and the CPU oracle under that rule.  The GPU product is held to the same rule
by tests/test_train_golden.py::test_batch32_codes_against_reference_golden and by bench.py before anything is timed."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

from facodec_amd import synth
from facodec_amd.diagnostics import check_codes_decidable


@pytest.fixture(scope="module")
def fx(golden_dir):
    return np.load(os.path.join(golden_dir, "codec_b32_decidable.npz"))


def test_fixture_is_consistent_with_the_reference_codes_held_so_far(fx, golden_dir):
    held = np.load(os.path.join(golden_dir, "codec_b32.npz"))
    assert np.array_equal(fx["codes_f32_mt"], held["codes"])
    assert hashlib.sha256(fx["codes_f32_mt"].astype(np.int16).tobytes()).hexdigest() == str(held["codes_sha256"])
    agree = (fx["codes_f32_mt"] == fx["codes_f32_1t"]) & (fx["codes_f32_mt"] == fx["codes_f64"])
    assert np.array_equal(agree, fx["decidable"])
    rep = json.loads(str(fx["report"]))
    assert rep["positions"] == 32 * 6 * 160 and rep["decidable"] == int(agree.sum())
    # what the reference says about itself on this batch: ONE frame where fp32 and fp64 choose differently (first residual stage of
    # clip 15, frame 90: fp64 top-2 gap 3.9e-7) and the third residual stage of the same frame following it; every other position
    # has an fp64 gap > 1e-6
    assert np.argwhere(~agree).tolist() == [[15, 3, 90], [15, 5, 90]]
    assert float(fx["gap_f64"][15, 3, 90]) < 1e-6 < rep["smallest_fp64_gap_among_decidable"]
    assert (fx["gap_f64"] >= 0).all()


def test_rule_has_no_allowance(fx):
    f32, f64 = fx["codes_f32_mt"].astype(np.int64), fx["codes_f64"].astype(np.int64)
    for run, name in ((f32, "f32_mt"), (f64, "f64")):
        v = check_codes_decidable(run, fx)
        assert v["ok"] and name in v["equals_run"] and v["decidable_mismatches"] == 0
    assert check_codes_decidable(f32, fx)["sha256_equal"] and check_codes_decidable(f64, fx)["sha256_equal_fp64_reference"]
    # the list of three streams FAquantizer.forward_v2 returns is accepted as well
    assert check_codes_decidable([torch.from_numpy(f64[:, :1]), torch.from_numpy(f64[:, 1:3]), torch.from_numpy(f64[:, 3:])], fx)["ok"]
    c = f32.copy()
    c[3, 2, 17] = (c[3, 2, 17] + 1) % 1024                      # one decidable position off by one code: fails
    v = check_codes_decidable(c, fx)
    assert not v["ok"] and v["decidable_mismatches"] == 1 and v["decidable_mismatch_positions"] == [[3, 2, 17]]
    c = f32.copy()
    c[15, 3, 90] = f64[15, 3, 90]                               # fp64's answer at the undecidable stage but fp32's cascade: no run's column
    v = check_codes_decidable(c, fx)
    assert not v["ok"] and v["decidable_mismatches"] == 0 and v["undecidable_frames_not_a_reference_column"] == 1
    c = f64.copy()
    c[15, 5, 90] = 7                                            # an undecidable position with an answer nobody gave
    assert not check_codes_decidable(c, fx)["ok"]
    c = f32.copy()
    c[15, 4, 90] = (c[15, 4, 90] + 1) % 1024                    # a decidable position of the undecidable frame
    v = check_codes_decidable(c, fx)
    assert not v["ok"] and v["decidable_mismatches"] == 1


def test_cpu_oracle_obeys_the_rule_on_the_clips_around_the_undecidable_frame(fx):
    """oracle/facodec_oracle.py (torch-CPU fp32) on clips 12..15 of the timed batch: equal on every decidable position; at the
    frame the reference does not decide it takes one run's column (measured: the fp32 runs')."""
    from oracle import facodec_oracle as O
    from facodec_amd.commons import build_model
    import bench
    model = build_model(bench.default_model_params())
    sds = {}
    for k in ("encoder", "quantizer"):
        synth.load_synthetic(model[k], seed=0, prefix=k + ".")
        sds[k] = {n: v.detach().clone() for n, v in model[k].state_dict().items()}
    del model
    sel = [12, 13, 14, 15]
    wave = synth.synth_clips(32, 48000, seed=0)[sel]
    with torch.no_grad():
        z = O.encoder_forward(sds["encoder"], wave)
        codes = O.quantizer_forward(sds["quantizer"], z, wave, n_c=2)[5]
    sub = {k: fx[k][sel] for k in ("codes_f32_mt", "codes_f32_1t", "codes_f64", "decidable")}
    v = check_codes_decidable(codes, sub)
    assert v["ok"] and v["decidable_mismatches"] == 0, v
    assert v["undecidable_positions"] == [[3, 3, 90], [3, 5, 90]]


# ------------------------------------------------------------------------------------------------------------------------------
# Four batches, and the reference's own margin noise inside the definition of "decidable" 
@pytest.fixture(scope="module")
def fx4(golden_dir):
    return np.load(os.path.join(golden_dir, "codec_b32x4_decidable.npz"))


def test_four_batch_fixture_is_consistent(fx, fx4):
    from facodec_amd.diagnostics import REFERENCE_RUNS
    assert fx4["seeds"].tolist() == [0, 1, 2, 3] and fx4["codes_f64"].shape == (4, 32, 6, 160)
    for k in REFERENCE_RUNS:                      # batch 0 is bench.py's timed batch: the same runs codec_b32_decidable.npz holds
        assert np.array_equal(fx4["codes_" + k][0], fx["codes_" + k])
    agree = (fx4["codes_f32_mt"] == fx4["codes_f32_1t"]) & (fx4["codes_f32_mt"] == fx4["codes_f64"])
    assert np.array_equal(agree, fx4["agree"])
    for b in range(4):
        noise = float(fx4["noise"][b])
        assert 1e-7 < noise < 1e-4                # fp32-vs-fp64 margin noise of the reference itself: a few 1e-6
        assert np.array_equal(fx4["decidable"][b], agree[b] & (fx4["gap_f64"][b].astype(np.float64) >= noise))
    # batch 0: the three positions round 5 called "decidable by luck" (fp64 gaps 1.26e-6 / 1.44e-6 / 1.67e-6 < noise 4.1e-6) are
    # undecidable by construction now
    lucky = np.argwhere(fx4["agree"][0] & ~fx4["decidable"][0]).tolist()
    assert lucky == [[5, 2, 157], [7, 5, 141], [28, 5, 8]]
    rep = json.loads(str(fx4["report"]))
    assert [b["seed"] for b in rep["batches"]] == [0, 1, 2, 3] and all(b["fp32_runs_bit_identical"] for b in rep["batches"])


def test_noise_rule_accepts_every_reference_run_and_both_sides_of_a_noise_tie(fx4):
    from facodec_amd.diagnostics import check_codes_decidable_noise as rule
    for b in range(4):
        for k in ("f32_mt", "f32_1t", "f64"):
            v = rule(fx4["codes_" + k][b], fx4, b)
            assert v["ok"] and v["mismatches"] == 0 and not v["noise_flips"] and k in v["equals_run"], (b, k, v)
    f64 = fx4["codes_f64"][0].astype(np.int64)
    # a noise-undecidable position may take the fp64 run's SECOND-best code; what follows in that frame is not compared
    c = f64.copy()
    c[5, 2, 157] = fx4["second_f64"][0][5, 2, 157]
    c[5, 4, 157] = (c[5, 4, 157] + 3) % 1024             # downstream of the flip (content -> residual): cascade, not a mismatch
    v = rule(c, fx4, 0)
    assert v["ok"] and v["noise_flips"] == [[5, 2, 157, pytest.approx(1.6653e-06, rel=1e-3)]] and v["cascade_positions"] == 3
    # ... but not a third code
    c = f64.copy()
    c[5, 2, 157] = (fx4["second_f64"][0][5, 2, 157] + 1) % 1024
    assert c[5, 2, 157] != f64[5, 2, 157]
    v = rule(c, fx4, 0)
    assert not v["ok"] and v["mismatches"] >= 1 and v["mismatch_positions"][0] == [5, 2, 157]
    # the prosody stream does not depend on the content flip: an error there is still an error in that frame
    c = f64.copy()
    c[5, 2, 157] = fx4["second_f64"][0][5, 2, 157]
    c[5, 0, 157] = (c[5, 0, 157] + 1) % 1024
    assert not rule(c, fx4, 0)["ok"]


def test_noise_rule_has_no_allowance_on_decidable_positions(fx4):
    from facodec_amd.diagnostics import check_codes_decidable_noise as rule
    for b in range(4):
        ref = fx4["codes_f32_mt"][b].astype(np.int64)
        dec = fx4["decidable"][b]
        pos = np.argwhere(dec)[1234 + 97 * b]
        c = ref.copy()
        c[tuple(pos)] = fx4["second_f64"][b][tuple(pos)]    # even the runner-up is wrong where the reference decides
        v = rule(c, fx4, b)
        assert not v["ok"] and v["decidable_mismatches"] >= 1 and v["mismatch_positions"][0] == pos.tolist()
    # the frame the reference's runs split on (batch 0, clip 15, frame 90): one run's answers along the chain, nothing mixed
    f32, f64 = fx4["codes_f32_mt"][0].astype(np.int64), fx4["codes_f64"][0].astype(np.int64)
    c = f32.copy()
    c[15, 3, 90] = f64[15, 3, 90]                         # fp64's first residual code, fp32's later ones
    v = rule(c, fx4, 0)
    assert not v["ok"] and [15, 5, 90] in v["mismatch_positions"]
    c = f32.copy()
    c[15, 3:, 90] = f64[15, 3:, 90]                       # the whole fp64 residual chain on the fp32 prosody / content codes (equal there)
    assert rule(c, fx4, 0)["ok"]
