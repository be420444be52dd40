"""BASELINE config 5 (long-form >= 2000 phonemes) as an ad-hoc parity run (too slow for the default suite: the CPU oracle needs
minutes).  usage: python tests/long_form_check.py [n_phones]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import sbv2_oracle as O
from helpers import blob, weights
from sbv2_api_amd import model, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
bc, bw = weights("bert", "full")
vc, vw = weights("vits", "full")
bs, vs = model.load_model(blob("bert", "full"), True), model.load_model(blob("vits", "full"), False)
pipe = model.Pipeline(bs, vs)
u = synth.make_utterance(n, bc, vc, seed=4242, chars=98)
b = pipe.prepare([u], forced=True)
pipe.run(b); pipe.sync()
t = time.perf_counter(); pipe.run(b); pipe.sync(); dt = time.perf_counter() - t
got = pipe.fetch(b)[0]
audio = got.shape[0] / O.SAMPLE_RATE
print(f"{n} phones -> T_text {u['T_text']}, {got.shape[0] // 512} frames, {audio:.1f} s audio in {dt * 1e3:.1f} ms (RTF {dt / audio:.5f}); finite={np.isfinite(got).all()}")
if os.environ.get("SBV2_LONG_ORACLE", "1") == "1":
    O.set_conv_backend("torch")
    t = time.perf_counter()
    h = O.deberta_forward(bw, bc, u["input_ids"])
    ref = O.vits_forward(vw, vc, O.expand_bert_features(h, u["word2ph"]), u["phones"], u["tones"], u["langs"], 0, u["style"],
                         forced_durations=u["forced_durations"])
    print(f"oracle {time.perf_counter() - t:.1f} s; max-abs error {np.abs(got - ref).max():.3e}")
