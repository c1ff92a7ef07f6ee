"""Drop-in boundary: build_model() surface and state-dict names/shapes equal the reference's
(tests/golden/state_shapes.json was dumped from the imported reference, modules/commons.py:283-348)."""
import json
import os

import torch

from facodec_amd.commons import (Munch, build_model, default_model_params, default_redecoder_params,
                                 recursive_munch)


def test_state_dict_keys_and_shapes_match_reference(golden_dir):
    ref = json.load(open(os.path.join(golden_dir, "state_shapes.json")))
    model = build_model(default_model_params())
    assert list(model.keys()) == ["encoder", "quantizer", "decoder", "discriminator", "fa_predictors"]   # commons.py:342-348
    # buffers real torchaudio checkpoints carry (SURVEY 3.3 [upstream]); absent from the shimmed dump
    extra_ok = {"to_mel.spectrogram.window", "to_mel.mel_scale.fb"}
    for k in ("encoder", "quantizer", "decoder", "discriminator", "fa_predictors"):
        own = {n: list(v.shape) for n, v in model[k].state_dict().items()}
        assert set(ref[k]) - set(own) == set(), k
        assert set(own) - set(ref[k]) <= extra_ok, k
        for n, shp in ref[k].items():
            assert own[n] == shp, (k, n)


def test_param_counts():
    model = build_model(default_model_params())
    n = {k: sum(p.numel() for p in model[k].parameters()) for k in model}
    assert abs(n["encoder"] - 36.28e6) < 0.05e6
    assert abs(n["decoder"] - 85.54e6) < 0.05e6
    assert abs(n["quantizer"] - 15.89e6) < 0.05e6
    assert abs(n["fa_predictors"] - 194.26e6) < 0.05e6


def test_munch_and_unknown_stage():
    import pytest
    m = recursive_munch({"a": {"b": [1, {"c": 2}]}})
    assert isinstance(m, Munch) and m.a.b[1].c == 2
    with pytest.raises(ValueError):
        build_model(default_model_params(), stage="nope")


def test_timbre_linear_bias_init():
    q = build_model(default_model_params()).quantizer
    assert torch.all(q.timbre_linear.bias[:1024] == 1) and torch.all(q.timbre_linear.bias[1024:] == 0)


def test_redecoder_and_encoder_stages_match_reference(golden_dir):
    """modules/commons.py:385-439: stage='redecoder' -> (encoder=Redecoder, decoder=non-causal, no LSTM);
    stage='encoder' -> (encoder, quantizer).  Names/shapes against the dump from the real reference."""
    ref = json.load(open(os.path.join(golden_dir, "state_shapes.json")))
    model = build_model(default_redecoder_params(), stage="redecoder")
    assert set(model.keys()) == {"encoder", "decoder"}
    for k in ("encoder", "decoder"):
        own = {n: list(v.shape) for n, v in model[k].state_dict().items()}
        assert own == ref["redecoder." + k], k
    assert not any("lstm" in n for n in model.decoder.state_dict())
    enc = build_model(default_redecoder_params(), stage="encoder")
    assert set(enc.keys()) == {"encoder", "quantizer"}
    assert {n: list(v.shape) for n, v in enc.encoder.state_dict().items()} == ref["encoder"]
