"""Pin oracle/sbv2_oracle.py against the fixtures produced by tests/golden/make_golden.py from
`transformers` (the third-party code the reference's export scripts run).  CPU only."""
import ast
import os

import numpy as np
import pytest

import sbv2_oracle as O
from sbv2_api_amd import synth


def _load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name))
    cfg = ast.literal_eval(str(z["cfg"])) if "cfg" in z else None
    return z, cfg


def test_log_bucket_table(golden_dir):
    z = np.load(os.path.join(golden_dir, "deberta_buckets.npz"))
    for key in z.files:
        S, b, m = (int(t[1:]) for t in key.split("_"))
        got = O.build_relative_position(S, b, m)
        assert np.array_equal(got, z[key].astype(np.int64)), key


def test_spline_inverse(golden_dir):
    z = np.load(os.path.join(golden_dir, "spline_inverse.npz"))
    y = O.rq_spline_inverse(z["x"], z["uw"], z["uh"], z["ud"], 5.0)
    np.testing.assert_allclose(y, z["y"], atol=2e-5, rtol=0)
    # tails are the identity
    assert y[2] == z["x"][2] and y[3] == z["x"][3]


@pytest.mark.parametrize("name", ["deberta_tiny_S24.npz", "deberta_tiny_S5.npz"])
def test_deberta_tiny(golden_dir, name):
    z, cfg = _load(golden_dir, name)
    W = synth.make_deberta_weights(cfg, int(z["seed"]))
    out = O.deberta_forward(W, cfg, z["input_ids"])
    np.testing.assert_allclose(out, z["output"], atol=2e-5, rtol=0)


@pytest.mark.parametrize("name", ["vits_tiny_blocks.npz", "vits_full_blocks.npz"])
def test_vits_blocks(golden_dir, name):
    z, cfg = _load(golden_dir, name)
    W = synth.make_vits_weights(cfg, int(z["seed"]))
    x, g = z["enc_x"], z["dp_g"]
    # one encoder layer, no speaker conditioning (cond_layer_idx out of range)
    y = O.encoder(W, "enc_p.encoder.", x, None, dict(cfg, cond_layer_idx=99), 1)
    np.testing.assert_allclose(y, z["enc_y"], atol=5e-5, rtol=0)
    np.testing.assert_allclose(O.duration_predictor(W, cfg, x, g), z["dp_logw"], atol=5e-5, rtol=0)
    logw = O.stochastic_duration_predictor(W, cfg, x, g, z["sdp_noise"])
    np.testing.assert_allclose(logw, z["sdp_logw"], atol=2e-4, rtol=0)
    pcm = O.hifigan(W, cfg, z["dec_z"], g)
    np.testing.assert_allclose(pcm, z["dec_pcm"], atol=2e-5, rtol=0)


def test_deberta_full(golden_dir):
    """Full ku-nlp-large shape, S=64: `hidden_states[-3][0]` of transformers' DebertaV2Model."""
    z, cfg = _load(golden_dir, "deberta_full_S64.npz")
    W = synth.make_deberta_weights(cfg, int(z["seed"]))
    out = O.deberta_forward(W, cfg, z["input_ids"])
    np.testing.assert_allclose(out, z["output"], atol=1e-4, rtol=0)


def test_torch_conv_backend_matches_numpy():
    """The cpu_baseline leg of bench.py runs the oracle with torch's CPU convolutions; same results as the numpy path."""
    cfg = O.VITS_TINY
    W = synth.make_vits_weights(cfg, 9)
    z = synth.hash_normal(3, cfg["inter"] * 19).reshape(cfg["inter"], 19)
    g = synth.hash_normal(4, cfg["gin"])
    a = O.hifigan(W, cfg, z, g)
    O.set_conv_backend("torch")
    try:
        b = O.hifigan(W, cfg, z, g)
    finally:
        O.set_conv_backend("numpy")
    np.testing.assert_allclose(a, b, atol=2e-6, rtol=0)
