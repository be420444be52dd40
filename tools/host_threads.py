"""Builder experiment (GPU box): one host thread driving a depth-2 pipeline vs two host threads, each driving its own depth-1 pipeline."""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from sbv2_api_amd import _lib, configs, model, synth
bc, vc = configs.DEBERTA_FULL, configs.VITS_FULL
bb = synth.pack_blob(synth.KIND_BERT, bc, synth.make_deberta_weights(bc))
vb = synth.pack_blob(synth.KIND_VITS, vc, synth.make_vits_weights(vc))
utts = [synth.make_utterance(128, bc, vc, seed=i) for i in range(32)]
N = 897 * 512 * 32
STEPS = 12

def single():
    bs, vs = model.load_model(bb, True), model.load_model(vb, False)
    pipe = model.Pipeline(bs, vs)
    b = pipe.prepare(utts, forced=True)
    pin = model.PinnedArray(N)
    prev = None
    def it():
        nonlocal prev
        pipe.run(b); tk = b.ticket
        if prev is not None:
            _lib.check(_lib.lib().sbv2_pipeline_fetch_pcm_ticket(pipe.h, prev, pin.ptr, N, 0))
        prev = tk
    for _ in range(3): it()
    pipe.sync(); t0 = time.perf_counter()
    for _ in range(STEPS): it()
    _lib.check(_lib.lib().sbv2_pipeline_fetch_pcm_ticket(pipe.h, prev, pin.ptr, N, 0)); pipe.sync()
    dt = (time.perf_counter() - t0) / STEPS
    pipe.close(); bs.close(); vs.close()
    return dt

def dual():
    os.environ["SBV2_PIPELINE_DEPTH"] = "1"
    ctxs = []
    for _ in range(2):
        bs, vs = model.load_model(bb, True), model.load_model(vb, False)
        pipe = model.Pipeline(bs, vs)
        ctxs.append((bs, vs, pipe, pipe.prepare(utts, forced=True), model.PinnedArray(N)))
    def worker(c, n):
        bs, vs, pipe, b, pin = c
        for _ in range(n):
            pipe.run(b)
            _lib.check(_lib.lib().sbv2_pipeline_fetch_pcm_ticket(pipe.h, b.ticket, pin.ptr, N, 0))
    for c in ctxs: worker(c, 2)
    ths = [threading.Thread(target=worker, args=(c, STEPS // 2)) for c in ctxs]
    t0 = time.perf_counter()
    for t in ths: t.start()
    for t in ths: t.join()
    dt = (time.perf_counter() - t0) / STEPS
    os.environ.pop("SBV2_PIPELINE_DEPTH")
    return dt

for r in range(2):
    a = single(); d = dual()
    print(f"single thread depth 2: {a*1e3:.1f} ms/step; two threads x depth 1: {d*1e3:.1f} ms/step", flush=True)
