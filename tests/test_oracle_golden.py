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


@pytest.mark.parametrize("name", ["deberta_tiny_S24.npz", "deberta_tiny_S5.npz", "deberta_tiny_conv_S24.npz", "deberta_tiny_conv_tanh_S9.npz",
                                  "deberta_tiny_conv_masked_S12.npz"])
def test_deberta_tiny(golden_dir, name):
    """hidden_states[-3][0] of transformers' DebertaV2Model, without and with the ConvLayer after layer 0 (gelu k3, tanh k5, masked tail)."""
    z, cfg = _load(golden_dir, name)
    W = synth.make_deberta_weights(cfg, int(z["seed"]))
    out = O.deberta_forward(W, cfg, z["input_ids"], z["attention_mask"])
    keep = z["attention_mask"] > 0      # rows of masked tokens are never consumed (word2ph only repeats real tokens)
    np.testing.assert_allclose(out[keep], z["output"][keep], atol=2e-5, rtol=0)
    if "conv" in name:
        assert cfg["conv_kernel_size"] > 0 and "deberta.encoder.conv.conv.weight" in W
        if "masked" in name:
            np.testing.assert_allclose(out[~keep], z["output"][~keep], atol=2e-5, rtol=0)   # ConvLayer output * mask == 0 after layer 0 ...


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


@pytest.mark.parametrize("name", ["deberta_full_S64.npz", "deberta_full_S64_noconv.npz"])
def test_deberta_full(golden_dir, name):
    """Full ku-nlp-large shape, S=64: `hidden_states[-3][0]` of transformers' DebertaV2Model (with / without the ConvLayer)."""
    z, cfg = _load(golden_dir, name)
    assert (cfg.get("conv_kernel_size", 0) > 0) == ("noconv" not in name)
    W = synth.make_deberta_weights(cfg, int(z["seed"]))
    out = O.deberta_forward(W, cfg, z["input_ids"])
    np.testing.assert_allclose(out, z["output"], atol=1e-4, rtol=0)


def test_duration_path_expansion(golden_dir):
    """a6: ceil(exp(logw) * length_scale) -> clamp_min(sum, 1) -> monotonic path -> expansion of m_p / logs_p, as executed by
    transformers' VitsModel.forward (modeling_vits.py:1349-1376) at three speaking rates; the fixture's flow input is
    m_f + exp(logs_f) (randn_like patched to ones, noise_scale 1)."""
    z, cfg = _load(golden_dir, "vits_tiny_path.npz")
    hop = O.hop_length(cfg)
    for i in range(3):
        logw, ls = z[f"logw{i}"], float(z[f"length_scale{i}"])
        w = np.exp(logw) * np.float32(ls)
        assert np.abs(w - np.round(w)).min() > 1e-4, "fixture has a duration on the ceil edge"
        dur = O.durations_from_logw(logw, ls)
        m_f, logs_f, Tf = O.expand_by_durations(z[f"m_p{i}"], z[f"logs_p{i}"], dur)
        assert Tf == z[f"flow_in{i}"].shape[1] and Tf * hop == int(z[f"samples{i}"])
        np.testing.assert_allclose(m_f + np.exp(logs_f), z[f"flow_in{i}"], atol=1e-6, rtol=1e-6)


@pytest.mark.parametrize("name", ["vits_tiny_e2e.npz", "vits_full_e2e.npz"])
def test_vits_e2e_torch_composition(golden_dir, name):
    """The whole `SynthesizerTrn.infer` restatement against the torch composition of transformers' modules (make_golden.e2e_case):
    embedding sum, multi-layer speaker-conditioned text encoder, dp / sdp blend, ceil, expansion, the transformer coupling flow,
    HiFi-GAN.  Case a: sdp_ratio 0, no noise; case b: sdp_ratio 0.25 with injected duration noise, length_scale 1.25."""
    z, cfg = _load(golden_dir, name)
    W = synth.make_vits_weights(cfg, int(z["seed"]))
    for tag in ("a", "b"):
        r = O.vits_forward(W, cfg, z["bert"], z["phones"], z["tones"], z["langs"], int(z["sid"]), z["style"], float(z[f"sdp_ratio_{tag}"]),
                           float(z[f"length_scale_{tag}"]), noise_w=z[f"noise_w_{tag}"], return_all=True)
        np.testing.assert_allclose(r["x"], z["x"], atol=5e-5, rtol=0)
        np.testing.assert_allclose(np.concatenate([r["m_p"], r["logs_p"]]), z["stats"], atol=5e-5, rtol=0)
        np.testing.assert_allclose(r["logw_dp"], z["logw_dp"], atol=5e-5, rtol=0)
        np.testing.assert_allclose(r["logw"], z[f"logw_{tag}"], atol=5e-4 if tag == "b" else 5e-5, rtol=0)
        w = z[f"w_{tag}"]
        safe = np.abs(w - np.round(w)) > 1.5e-3 * np.maximum(1.0, w)      # ceil() edge: the fixture keeps clear of it (make_golden searches)
        assert safe.all(), "fixture has a duration on the ceil edge"
        assert np.array_equal(r["durations"], z[f"dur_{tag}"])
        np.testing.assert_allclose(r["z_p"], z[f"z_p_{tag}"], atol=5e-5, rtol=0)
        np.testing.assert_allclose(r["z"], z[f"z_{tag}"], atol=2e-4, rtol=0)
        np.testing.assert_allclose(r["pcm"], z[f"pcm_{tag}"], atol=1e-4, rtol=0)


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
