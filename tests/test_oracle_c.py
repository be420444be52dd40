"""oracle/sbv2_ref.c (the C / OpenMP restatement that bench.py times as the CPU baseline) against the numpy oracle and against the
transformers fixtures.  CPU only; the library is built by `make -C oracle` (__graft_entry__.build())."""
import ast
import os
import subprocess

import numpy as np
import pytest

import sbv2_oracle as O
import sbv2_ref as R
from sbv2_api_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(os.path.join(ROOT, "oracle", "liboracle_ref.so")):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, capture_output=True)
    return R.load()


@pytest.mark.parametrize("cin,cout,k,dil,L", [(16, 16, 11, 5, 3000), (64, 64, 3, 1, 777), (192, 768, 5, 1, 257), (40, 72, 3, 1, 1), (7, 5, 4, 1, 33),
                                              (256, 1, 1, 1, 100)])
def test_conv_core(lib, cin, cout, k, dil, L):
    rng = np.random.default_rng(cin + cout + k)
    x = rng.standard_normal((cin, L)).astype(np.float32)
    w = (rng.standard_normal((cout, cin, k)) / np.sqrt(cin * k)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    y = np.empty((cout, L), np.float32)
    P = lambda a: a.ctypes.data_as(R._f32p)
    lib.sbv2c_conv1d_same(P(x), cin, L, P(w), P(b), cout, k, dil, 0.1, P(y))
    np.testing.assert_allclose(y, O.conv1d_same(O.leaky_relu(x, 0.1), w, b, dil), atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize("cin,cout,k,s,p,L", [(64, 32, 16, 8, 4, 301), (32, 16, 8, 2, 3, 100), (16, 8, 2, 2, 0, 99), (8, 4, 4, 4, 0, 33)])
def test_conv_transpose(lib, cin, cout, k, s, p, L):
    rng = np.random.default_rng(k * 100 + s)
    x = rng.standard_normal((cin, L)).astype(np.float32)
    w = (rng.standard_normal((cin, cout, k)) / np.sqrt(cin * k / s)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    y = np.empty((cout, L * s), np.float32)
    P = lambda a: a.ctypes.data_as(R._f32p)
    lib.sbv2c_conv_transpose1d(P(x), cin, L, P(w), P(b), cout, k, s, p, 0.1, P(y))
    np.testing.assert_allclose(y, O.conv_transpose1d(O.leaky_relu(x, 0.1), w, b, s, p), atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize("name", ["deberta_tiny_S24.npz", "deberta_tiny_conv_S24.npz", "deberta_tiny_conv_tanh_S9.npz", "deberta_tiny_conv_masked_S12.npz",
                                  "deberta_full_S64.npz"])
def test_deberta_vs_transformers_fixture(lib, golden_dir, name):
    z = np.load(os.path.join(golden_dir, name))
    cfg = ast.literal_eval(str(z["cfg"]))
    W = synth.make_deberta_weights(cfg, int(z["seed"]))
    m = R.Model(bert_blob=synth.pack_blob(synth.KIND_BERT, cfg, W), lib=lib)
    am = z["attention_mask"]
    out = m.bert(z["input_ids"], am, hidden=cfg["hidden"])
    keep = am > 0
    np.testing.assert_allclose(out[keep], z["output"][keep], atol=1e-4 if "full" in name else 2e-5, rtol=0)
    m.close()


@pytest.mark.parametrize("name", ["vits_tiny_e2e.npz", "vits_full_e2e.npz"])
def test_vits_vs_torch_composition_fixture(lib, golden_dir, name):
    z = np.load(os.path.join(golden_dir, name))
    cfg = ast.literal_eval(str(z["cfg"]))
    W = synth.make_vits_weights(cfg, int(z["seed"]))
    m = R.Model(vits_blob=synth.pack_blob(synth.KIND_VITS, cfg, W), lib=lib)
    for tag in ("a", "b"):
        r = m.vits(z["bert"], z["phones"], z["tones"], z["langs"], int(z["sid"]), z["style"], float(z[f"sdp_ratio_{tag}"]),
                   float(z[f"length_scale_{tag}"]), noise_w=z[f"noise_w_{tag}"], hidden=cfg["hidden"], return_all=True)
        np.testing.assert_allclose(r["x"], z["x"], atol=5e-5, rtol=0)
        np.testing.assert_allclose(r["logw"], z[f"logw_{tag}"], atol=5e-4 if tag == "b" else 5e-5, rtol=0)
        assert np.array_equal(r["durations"], z[f"dur_{tag}"])
        np.testing.assert_allclose(r["pcm"], z[f"pcm_{tag}"], atol=1e-4, rtol=0)
    m.close()


def test_vits_tiny_vs_numpy_oracle_forced_and_edge(lib):
    """Forced durations (the benchmark mode), a speaker id > 0, a one-symbol utterance and the all-zero-duration case."""
    cfg = O.VITS_TINY
    W = synth.make_vits_weights(cfg, 5)
    m = R.Model(vits_blob=synth.pack_blob(synth.KIND_VITS, cfg, W), lib=lib)
    for n, sid in ((7, 1), (1, 0)):
        u = synth.make_utterance(n, O.DEBERTA_TINY, cfg, seed=40 + n)
        bert = synth.hash_normal(77 + n, cfg["bert_dim"] * u["T_text"]).reshape(cfg["bert_dim"], -1)
        ref = O.vits_forward(W, cfg, bert, u["phones"], u["tones"], u["langs"], sid, u["style"], forced_durations=u["forced_durations"], return_all=True)
        got = m.vits(bert, u["phones"], u["tones"], u["langs"], sid, u["style"], forced_durations=u["forced_durations"], hidden=cfg["hidden"],
                     inter=cfg["inter"], return_all=True)
        np.testing.assert_allclose(got["pcm"], ref["pcm"], atol=5e-5, rtol=0)
        np.testing.assert_allclose(got["z"][:ref["z"].size].reshape(ref["z"].shape), ref["z"], atol=1e-4, rtol=0)
        zero = np.zeros_like(u["forced_durations"])
        ref0 = O.vits_forward(W, cfg, bert, u["phones"], u["tones"], u["langs"], sid, u["style"], forced_durations=zero)
        got0 = m.vits(bert, u["phones"], u["tones"], u["langs"], sid, u["style"], forced_durations=zero)
        assert got0.shape == ref0.shape == (O.hop_length(cfg),)
        np.testing.assert_allclose(got0, ref0, atol=5e-5, rtol=0)
    with pytest.raises(RuntimeError):
        m.vits(bert, np.array([10_000]), u["tones"][:1], u["langs"][:1], 0, u["style"])
    m.close()
    with pytest.raises(RuntimeError):
        R.Model(vits_blob=b"junk" * 10, lib=lib)
