"""Builder tool (GPU box): DeBERTa forward latency at > 128 tokens, the fused key-tile attention (default) against the grouped-GEMM + softmax
path (SBV2_BERT_ATTN=nolong, read once per process: run the script twice).  Prints one JSON line.   usage: bert_long_probe.py [reps = 10]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from sbv2_api_amd import configs, model, synth

bc = configs.DEBERTA_FULL
s = model.load_model(synth.pack_blob(synth.KIND_BERT, bc, synth.make_deberta_weights(bc)), True)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rng = np.random.default_rng(5)
out = {"attn": os.environ.get("SBV2_BERT_ATTN", "fused")}
for name, lens in (("1x152", [152]), ("1x300", [300]), ("1x515", [515]), ("1x1000", [1000]), ("8x300", [300] * 8), ("mixed 60/200/515/90", [60, 200, 515, 90])):
    seqs = [np.concatenate([[1], rng.integers(3, bc["vocab_size"], n - 2), [2]]) for n in lens]
    for _ in range(3): model.predict_batch(s, seqs)
    t = []
    for _ in range(reps):
        t0 = time.perf_counter(); model.predict_batch(s, seqs); t.append(time.perf_counter() - t0)
    out[name + "_ms"] = round(1e3 * float(np.median(t)), 3)
s.close()
print(json.dumps(out))
