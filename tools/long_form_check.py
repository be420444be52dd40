"""BASELINE configs[4] (streaming long-form, >= 2000 phonemes) measurement: whole-sequence synthesis vs the chunked decoder replaying its
captured hipGraph; prints one JSON line (kept under profiles/ as r02_longform*.json).
The DeBERTa input has the front end's own length (one token per character: n_phones // 2 tokens, 1000 for 2000 phonemes; the reference itself
splits on newlines, tts.rs:290-321, and its TensorRT profile caps BERT at 100 tokens, model.rs:14-16: pass a third argument to cap the characters).
usage: python tools/long_form_check.py [n_phones] [chunk_frames] [chars]         env SBV2_STREAM_GRAPH=0 -> eager chunk decode (A/B)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import blob, weights
from sbv2_api_amd import model, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 256
chars = int(sys.argv[3]) if len(sys.argv) > 3 else None
bc, bw = weights("bert", "full")
vc, vw = weights("vits", "full")
bs, vs = model.load_model(blob("bert", "full"), True), model.load_model(blob("vits", "full"), False)
pipe = model.Pipeline(bs, vs)
u = synth.make_utterance(n, bc, vc, seed=4242, chars=chars)
b = pipe.prepare([u], forced=True)
pipe.run(b); pipe.sync()
t = time.perf_counter(); pipe.run(b); whole = pipe.fetch(b)[0]; t_whole = time.perf_counter() - t
audio = whole.shape[0] / 44100.0
ws_whole = model._lib.lib().sbv2_vits_workspace_bytes(vs.handle)


def stream():
    t0 = time.perf_counter()
    st = model.StreamHandle(bs, vs, u, chunk, forced=True)
    t_begin = time.perf_counter() - t0
    parts, t_first, worst = [], None, 0.0
    while True:
        tc = time.perf_counter()
        c = st.next()
        if c is None:
            break
        if t_first is None:
            t_first = time.perf_counter() - t0
        else:
            worst = max(worst, time.perf_counter() - tc)
        parts.append(c)
    t_all = time.perf_counter() - t0
    info = (st.uses_graph, st.workspace_bytes)
    st.close()
    return np.concatenate(parts), t_begin, t_first, t_all, info, len(parts), worst


import gc
# Python's cyclic collector is frozen and switched off for the measured passes: round 3-4's "one pass of five takes 179 ms" was a generation-2 collection over
# this script's own objects (the full-size synthetic weight dictionaries) inside the second StreamHandle(), 106 ms outside every HIP call
# (profiles/r05_longform_outlier.json; LF_GC=1 measures with the collector on)
if os.environ.get("LF_GC") != "1":
    gc.collect(); gc.freeze(); gc.disable()
stream()                                  # first use of this chunk size: warm-up pass + graph capture
# five measured passes, the median by total time is reported and all totals are kept
runs = sorted((stream() for _ in range(5)), key=lambda r: r[3])
got, t_begin, t_first, t_all, (graph, ws), nchunks, _ = runs[2]
out = {"phones": n, "T_text": int(u["T_text"]), "bert_tokens": int(u["S"]), "frames": whole.shape[0] // 512, "audio_s": round(audio, 2), "chunk_frames": chunk, "chunks": nchunks,
       "whole_sequence_ms": round(t_whole * 1e3, 1), "whole_sequence_rtf": round(t_whole / audio, 6),
       "stream_text_and_flow_ms": round(t_begin * 1e3, 1), "stream_time_to_first_chunk_ms": round(t_first * 1e3, 1),
       "stream_total_ms": round(t_all * 1e3, 1), "stream_total_ms_all_passes": [round(r[3] * 1e3, 1) for r in runs],
       "stream_begin_ms_all_passes": [round(r[1] * 1e3, 1) for r in runs], "worst_chunk_ms_all_passes": [round(r[6] * 1e3, 1) for r in runs], "python_gc": gc.isenabled(), "stream_rtf": round(t_all / audio, 6),
       "stream_ms_per_chunk": round((t_all - t_begin) / nchunks * 1e3, 3), "chunk_audio_s": round(chunk * 512 / 44100.0, 3),
       "decoder_graph_replay": bool(graph), "chunk_decoder_workspace_MiB": round(ws / 2**20, 1), "whole_sequence_workspace_MiB": round(ws_whole / 2**20, 1),
       "chunked_vs_whole_max_abs": float(np.abs(got - whole).max())}
print(json.dumps(out))
