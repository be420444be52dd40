"""Builder tool (GPU box): batch-1 latency of one 128-phoneme utterance (BASELINE configs[1] shape), un-pipelined: wall per call."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from sbv2_api_amd import _lib, configs, model, synth
bc, vc = configs.DEBERTA_FULL, configs.VITS_FULL
bs = model.load_model(synth.pack_blob(synth.KIND_BERT, bc, synth.make_deberta_weights(bc)), True)
vs = model.load_model(synth.pack_blob(synth.KIND_VITS, vc, synth.make_vits_weights(vc)), False)
u = synth.make_utterance(128, bc, vc, seed=1)
pipe = model.Pipeline(bs, vs)
b = pipe.prepare([u], forced=True)
pin = model.PinnedArray(897 * 512)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
for _ in range(5):
    pipe.run(b); pipe.fetch(b, out=pin.array)
ts = []
for _ in range(n):
    a = time.perf_counter(); pipe.run(b); pipe.fetch(b, out=pin.array); ts.append(time.perf_counter() - a)
import json
print(f"B=1 U128: median {np.median(ts)*1e3:.2f} ms, min {min(ts)*1e3:.2f} ms per call (10.414 s audio)", flush=True)
print(json.dumps({"config": "BASELINE configs[1] shape: batch 1, 128 phonemes (T_text 257, 66 BERT tokens, 897 frames = 10.414 s), full model shapes, "
                            "forced durations, un-pipelined, PCM copied to pinned host memory inside the call", "calls": n,
                  "median_ms_per_call": round(float(np.median(ts)) * 1e3, 3), "min_ms_per_call": round(min(ts) * 1e3, 3),
                  "real_time_factor": round(float(np.median(ts)) / 10.414, 6), "bert_gemm": os.environ.get("SBV2_BERT_GEMM", "default (f16x3)")}), flush=True)
