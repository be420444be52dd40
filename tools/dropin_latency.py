"""Builder tool (GPU box): latency of the path the reference's Rust shim calls, one sentence at a time (crates/sbv2_api/src/main.rs:86,104 serves one
request at a time; crates/sbv2_core/src/tts.rs:304-316 is the per-sentence sequence):

    sbv2_bert_predict(ids, mask) -> [S, 1024] on the host            (bert.rs:6-24)
    host: repeat rows by word2ph, transpose to [1024, T]             (tts_util.rs:129-154)
    sbv2_vits_synthesize(bert, x, tones, langs, sid, style, sdp_ratio 0.0, length_scale 1.0, noise 0.677 / 0.8) -> malloc'ed PCM   (model.rs:53-111)

with PREDICTED durations and noise on, at the sentence lengths the reference's TensorRT profile names (model.rs:14-16: opt 25 tokens, max 100) plus 16 and
64, next to the same sentences through sbv2_pipeline_run (device resident between the stages) with the same settings.  Prints one JSON line.
`--fp32` selects BASELINE configs[1]'s stated arithmetic (exact-f32 MFMA everywhere) and times the 128-phoneme utterance in it.

    python3 tools/dropin_latency.py [calls] [--fp32] [--tokens=25]
"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
FP32 = "--fp32" in sys.argv
if FP32:
    os.environ.update(SBV2_DECODER="f32", SBV2_GEMM="f32", SBV2_ATTN="f32")
import numpy as np

from sbv2_api_amd import _lib, configs, model, synth
from sbv2_api_amd._lib import check, f32p, i64p

TOKENS = [int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("--tokens=")]
args = [a for a in sys.argv[1:] if not a.startswith("--")]
CALLS = int(args[0]) if args else 60
bc, vc = configs.DEBERTA_FULL, configs.VITS_FULL
bs = model.load_model(synth.pack_blob(synth.KIND_BERT, bc, synth.make_deberta_weights(bc)), True)
vs = model.load_model(synth.pack_blob(synth.KIND_VITS, vc, synth.make_vits_weights(vc)), False)
l = _lib.lib()
H = l.sbv2_bert_hidden(bs.handle)
pipe = model.Pipeline(bs, vs)
SETTINGS = dict(sdp_ratio=0.0, length_scale=1.0, noise_scale=0.677, noise_scale_w=0.8)


def med(v):
    return round(float(np.median(v)) * 1e3, 3)


def one(tokens: int, n_phones: int):
    u = synth.make_utterance(n_phones, bc, vc, seed=7000 + tokens, chars=tokens - 2)
    ids = np.ascontiguousarray(u["input_ids"], np.int64)
    msk = np.ascontiguousarray(u["attention_mask"], np.int64)
    x = np.ascontiguousarray(u["phones"], np.int64)
    tn = np.ascontiguousarray(u["tones"], np.int64)
    lg = np.ascontiguousarray(u["langs"], np.int64)
    st = np.ascontiguousarray(u["style"], np.float32)
    w2p = np.asarray(u["word2ph"], np.int64)
    T = int(x.shape[0])
    feats = np.empty((ids.shape[0], H), np.float32)
    pcm, n = f32p(), C.c_int64()
    t_pred, t_host, t_syn, t_all, frames = [], [], [], [], 0
    for it in range(CALLS + 5):
        a = time.perf_counter()
        check(l.sbv2_bert_predict(bs.handle, ids.ctypes.data_as(i64p), msk.ctypes.data_as(i64p), ids.shape[0], feats.ctypes.data_as(f32p)))
        b = time.perf_counter()
        bert = np.ascontiguousarray(np.repeat(feats, w2p, axis=0).T)     # [1024, T]
        c = time.perf_counter()
        check(l.sbv2_vits_synthesize(vs.handle, bert.ctypes.data_as(f32p), x.ctypes.data_as(i64p), tn.ctypes.data_as(i64p), lg.ctypes.data_as(i64p), T, 0,
                                     st.ctypes.data_as(f32p), SETTINGS["sdp_ratio"], SETTINGS["length_scale"], SETTINGS["noise_scale"],
                                     SETTINGS["noise_scale_w"], 1234 + it, C.byref(pcm), C.byref(n)))
        d = time.perf_counter()
        frames = n.value // 512
        l.sbv2_pcm_free(pcm)
        if it >= 5:
            t_pred.append(b - a); t_host.append(c - b); t_syn.append(d - c); t_all.append(d - a)
    # the same sentence through the device-resident pipeline (ids in, PCM in pinned host memory out)
    pb = pipe.prepare([u], noise_seed=99, **SETTINGS)
    pin = model.PinnedArray(max(frames * 2, 64) * 512)
    t_pipe = []
    for it in range(CALLS + 5):
        a = time.perf_counter()
        pipe.run(pb); pipe.fetch(pb, out=pin.array)
        if it >= 5:
            t_pipe.append(time.perf_counter() - a)
    pin.close()
    return dict(tokens=tokens, phones=n_phones, t_text=T, frames=int(frames), audio_s=round(frames * 512 / 44100, 3),
                dropin_ms=med(t_all), dropin_min_ms=round(min(t_all) * 1e3, 3), predict_ms=med(t_pred), host_repeat_transpose_ms=med(t_host),
                synthesize_ms=med(t_syn), pipeline_run_ms=med(t_pipe), pipeline_min_ms=round(min(t_pipe) * 1e3, 3))


rows = []
if FP32:
    rows.append(one(64, 128))          # BASELINE configs[1]: batch 1, 128 phonemes, fp32
else:
    for tok in (TOKENS or (16, 25, 64, 100)):
        rows.append(one(tok, 2 * tok))
out = {"what": "sbv2_bert_predict -> host repeat / transpose -> sbv2_vits_synthesize (the calls of the Rust shim, INTEGRATION.md §1) vs sbv2_pipeline_run, "
               "one sentence per call, predicted durations, sdp_ratio 0.0, noise_scale 0.677, noise_scale_w 0.8, full model shapes, synthetic weights",
       "arithmetic": "fp32 everywhere (BASELINE configs[1]: SBV2_DECODER=f32 SBV2_GEMM=f32 SBV2_ATTN=f32)" if FP32 else "default (split-bf16 decoder, f16x3 DeBERTa / flow 1x1, f32 text side)",
       "calls": CALLS, "rows": rows}
print(json.dumps(out), flush=True)
