"""Batched weight preparation (facodec_amd/wprep.py, fac_prep_* of include/facodec_hip.h): the per-forward re-materialisation of
w = g * v / ||v|| (dac/model/encodec.py:42-51, dac/nn/layers.py:9-14) as a few recorded launches must give the bits of the
per-tensor launches, must follow the parameters' current contents, and must never serve anything outside a region."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu


def _weights(cuda, seed=0):
    g = torch.Generator().manual_seed(seed)
    mk = lambda *s: torch.randn(*s, generator=g).to(cuda)       # noqa: E731
    return dict(v7=mk(192, 192, 7), g7=mk(192, 1, 1).abs() + 0.5, v1=mk(96, 96, 1), g1=mk(96, 1, 1).abs() + 0.5,
                vs=mk(256, 128, 8), gs=mk(256, 1, 1).abs() + 0.5, vt=mk(384, 192, 8), gt=mk(384, 1, 1).abs() + 0.5,
                vd=mk(32, 32, 27), gd=mk(32, 1, 1).abs() + 0.5, vn=mk(8, 1024, 1), gn=mk(8, 1, 1).abs() + 0.5, raw=mk(512, 256, 1))


def _all_forms(ops, w):
    """Every recordable function on shapes of the model: (name, result tensor)."""
    out = []
    sc = ops.wn_scale(w["v7"], w["g7"])
    out.append(("wn_scale", sc))
    out.append(("pack_conv_weight", ops.pack_conv_weight(w["v7"], w["g7"])))
    out.append(("pack_conv_weight(scale)", ops.pack_conv_weight(w["vn"], w["gn"], scale=ops.wn_scale(w["vn"], w["gn"]))))
    out.append(("pack_conv_weight(raw)", ops.pack_conv_weight(w["raw"])))
    out.append(("pack_conv_weight_split k7", ops.pack_conv_weight_split(w["v7"], w["g7"], scale=sc)))
    out.append(("pack_conv_weight_split k1", ops.pack_conv_weight_split(w["v1"], w["g1"])))
    out.append(("pack_gemm_weight_split", ops.pack_gemm_weight_split(w["raw"])))
    out.append(("pack_gemm_weight_split strided", ops.pack_gemm_weight_split(w["vs"], w["gs"], in_stride=4)))
    out.append(("pack_convtr_weight", ops.pack_convtr_weight(w["vt"], w["gt"], 4)))
    out.append(("pack_convtr_weight_rows", ops.pack_convtr_weight_rows(w["vt"], w["gt"], 4)))
    out.append(("pack_convtr_weight_rows_split", ops.pack_convtr_weight_rows_split(w["vt"], w["gt"], 4)[0]))
    out.append(("pack_conv_weight_split2", ops.pack_conv_weight_split2(w["vd"], w["gd"], 9)))
    fl = ops.flipped_weight(w["v7"], w["g7"], sc)
    out.append(("flipped_weight", fl))
    out.append(("pack_conv_weight_split(flipped)", ops.pack_conv_weight_split(fl)))
    out.append(("pack_conv_weight_bwd", ops.pack_conv_weight_bwd(w["v7"], w["g7"], sc)))
    return out


def test_recorded_batch_gives_the_bits_of_the_single_launches_and_follows_the_parameters(cuda):
    from facodec_amd import ops, wprep
    w = _weights(cuda)
    cache = wprep.WeightCache(list(w.values()), "test")
    ref = [(n, t.clone()) for n, t in _all_forms(ops, w)]                    # no region: the single launches
    with cache:
        first = [(n, t.clone()) for n, t in _all_forms(ops, w)]             # region 1: every call is new (single launches, remembered)
    assert cache.stats["hits"] == 0 and cache.stats["misses"] > 0
    for (n, a), (_, b) in zip(ref, first):
        assert torch.equal(a, b), n
    with cache:                                                             # region 2: one recorded batch, then hits only
        misses = cache.stats["misses"]
        second = [(n, t.clone()) for n, t in _all_forms(ops, w)]
        assert cache.stats["misses"] == misses and cache.stats["hits"] > 0
    info = cache.info()
    assert info["rebuilds"] == 1 and info["replays"] == 1 and 0 < info["launches"] <= 12 and info["jobs"] >= len(ref), info
    for (n, a), (_, b) in zip(ref, second):
        assert torch.equal(a, b), n
    with torch.no_grad():                                                   # the parameters move (an optimiser step) ...
        for k in w:
            w[k].mul_(1.0 + 0.01 * (len(k) % 3)).add_(0.001)
    new_ref = [(n, t.clone()) for n, t in _all_forms(ops, w)]
    assert not torch.equal(new_ref[1][1], ref[1][1])
    with cache:                                                             # ... and the next region re-materialises from them
        third = [(n, t.clone()) for n, t in _all_forms(ops, w)]
    assert cache.info()["rebuilds"] == 1
    for (n, a), (_, b) in zip(new_ref, third):
        assert torch.equal(a, b), n
    torch.cuda.synchronize()
    cache.close()


def test_nothing_is_served_outside_a_region_and_unused_entries_are_dropped(cuda):
    from facodec_amd import ops, wprep
    w = _weights(cuda, seed=1)
    cache = wprep.WeightCache(list(w.values()), "test")
    for _ in range(2):
        with cache:
            inside = ops.pack_conv_weight_split(w["v7"], w["g7"])
    with torch.no_grad():
        w["v7"].mul_(2.0).add_(0.3)
    outside = ops.pack_conv_weight_split(w["v7"], w["g7"])                  # no region: a single launch from the new contents
    assert outside.data_ptr() != inside.data_ptr() and not torch.equal(outside, inside)
    with cache:
        again = ops.pack_conv_weight_split(w["v7"], w["g7"])
        assert torch.equal(again, outside)
        temp = w["v7"].clone()                                             # a tensor the cache does not own passes through
        assert ops.pack_conv_weight_split(temp, w["g7"]).data_ptr() != again.data_ptr()
    n = len(cache.entries)
    for _ in range(2 * wprep.KEEP_EPOCHS + 3):                              # regions that ask for something else (a nested scale outlives its pack by one horizon)
        with cache:
            ops.wn_scale(w["v1"], w["g1"])
    assert cache.stats["dropped"] >= n and len(cache.entries) == 1, (cache.stats, len(cache.entries))
    torch.cuda.synchronize()
    cache.close()


def test_forward_with_the_batch_equals_the_per_layer_launches(cuda):
    """configs[0]-shaped eval forward (B = 2): latent, codes and waveform with FAC_WEIGHT_BATCH on are the bits of the per-layer
    launches, also after the weights were changed in place between two forwards."""
    from facodec_amd import synth, wprep
    from facodec_amd.commons import build_model, default_model_params
    model = build_model(default_model_params())
    for k in ("encoder", "quantizer", "decoder"):
        synth.load_synthetic(model[k], seed=0, prefix=k + ".")
        model[k].eval().to(cuda)
    wave = synth.synth_clips(2, 48000, seed=0).to(cuda)

    def fwd():
        with torch.no_grad():
            z = model.encoder(wave)
            outs, _, _, _, timbre, codes = model.quantizer(z, wave, n_c=2, return_codes=True)
            return z.clone(), [c.clone() for c in codes], model.decoder(outs).clone()

    saved = wprep.ENABLED
    try:
        wprep.ENABLED = False
        ref = fwd()
        wprep.ENABLED = True
        runs = [fwd() for _ in range(3)]
        enc = model.encoder.__dict__["_wcache"].info()
        assert enc["replays"] == 2 and enc["hits"] > 0 and enc["launches"] <= 8, enc
        for z, codes, y in runs:
            assert torch.equal(z, ref[0]) and torch.equal(y, ref[2])
            assert all(torch.equal(a, b) for a, b in zip(codes, ref[1]))
        with torch.no_grad():
            for p in list(model.encoder.parameters())[:40] + list(model.decoder.parameters())[:40]:
                p.mul_(1.01)
        got = fwd()
        wprep.ENABLED = False
        ref2 = fwd()
        assert not torch.equal(ref2[2], ref[2])
        assert torch.equal(got[0], ref2[0]) and torch.equal(got[2], ref2[2])
    finally:
        wprep.ENABLED = saved


def test_train_steps_identical_with_and_without_the_batch(cuda):
    """configs[2] at its real size, three seeded iterations with the optimisers on: losses, gradient norms and every parameter arena
    after the last step agree to the last bit between FAC_WEIGHT_BATCH=0 and the default, and the default's steady-state step
    re-materialises its ~1 000 weight layouts with at most 12 launches per region.  Separate processes: the switch is read once."""
    script = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "tune", "wb_check.py")
    outs, infos = [], []
    for val in ("0", "1"):
        r = subprocess.run([sys.executable, script, "3"], env=dict(os.environ, FAC_WEIGHT_BATCH=val), capture_output=True, text=True,
                           timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("WB ")][-1][3:]))
        infos.append(json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("WBINFO ")][-1][7:]))
    assert outs[0] == outs[1], outs
    assert infos[0] == {} and set(infos[1]) == {"gen", "disc"}, infos
    for k, i in infos[1].items():
        assert i["hits"] > 0 and 0 < i["launches"] <= 12 and i["rebuilds"] <= 3, (k, i)


def test_regions_belong_to_their_thread_and_copies_start_over(cuda):
    """A no-grad forward's region is served to the thread that opened it only (its replay is ordered on that thread's stream); a
    region opened for every thread (a train step: autograd's worker runs the backward nodes) is served to all; a deep-copied module
    does not inherit the plan of the original."""
    import copy
    import threading
    from facodec_amd import ops, wprep
    w = _weights(cuda, seed=2)
    seen = {}

    def other(tag):
        seen[tag] = ops.pack_conv_weight_split(w["v7"], w["g7"]).data_ptr()

    for any_thread in (False, True):
        cache = wprep.WeightCache(list(w.values()), "test", any_thread=any_thread)
        for _ in range(2):
            with cache:
                mine = ops.pack_conv_weight_split(w["v7"], w["g7"]).data_ptr()
        with cache:
            assert ops.pack_conv_weight_split(w["v7"], w["g7"]).data_ptr() == mine
            t = threading.Thread(target=other, args=(any_thread,))
            t.start()
            t.join()
        assert (seen[any_thread] == mine) == any_thread
        clone = copy.deepcopy(cache)
        assert clone.plan == -1 and not clone.entries and cache.plan >= 0
        torch.cuda.synchronize()
        cache.close()
