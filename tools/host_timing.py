"""Builder tool (GPU box): where does the host spend a pipelined step?  Prints per-step wall time of pipe.run (enqueue + the duration
sync) and of the PCM collection, for a few pipeline depths."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from sbv2_api_amd import _lib, configs, model, synth
bc, vc = configs.DEBERTA_FULL, configs.VITS_FULL
bs = model.load_model(synth.pack_blob(synth.KIND_BERT, bc, synth.make_deberta_weights(bc)), True)
vs = model.load_model(synth.pack_blob(synth.KIND_VITS, vc, synth.make_vits_weights(vc)), False)
utts = [synth.make_utterance(128, bc, vc, seed=i) for i in range(32)]
pipe = model.Pipeline(bs, vs)
b = pipe.prepare(utts, forced=True)
pin = model.PinnedArray(32 * 897 * 512)
for _ in range(3):
    pipe.run(b); pipe.fetch(b, out=pin.array)
tr, tc = [], []
t0 = time.perf_counter()
prev = None
for i in range(12):
    a = time.perf_counter(); pipe.run(b); tr.append(time.perf_counter() - a)
    tk = b.ticket
    if prev is not None:
        a = time.perf_counter(); _lib.check(_lib.lib().sbv2_pipeline_fetch_pcm_ticket(pipe.h, prev, pin.ptr, pin.array.size, 0)); tc.append(time.perf_counter() - a)
    prev = tk
pipe.sync()
dt = (time.perf_counter() - t0) / 12
print(f"depth {os.environ.get('SBV2_PIPELINE_DEPTH', '2')}: step {dt*1e3:.1f} ms; run() {np.mean(tr)*1e3:.1f} ms (min {min(tr)*1e3:.1f}); collect {np.mean(tc)*1e3:.1f} ms", flush=True)
