"""Builder experiment (GPU box): how many INTEGER durations change when DeBERTa's GEMMs leave the exact-f32 matrix pipe?
SBV2_BERT_GEMM = f32 (gemm_conv.hip) | bf16x6 (gemm_bfs.hip, three bf16 parts per operand: f32-grade) | f16x3 (f16 hi + scaled f16 lo: 22 bits, the library default) | bf16x3 (two parts,
2^-16 per product, opt-in) on the SAME synthetic weights and >= 2e5 symbols (N utterances of 128 phones = 257 symbols, 66 BERT tokens);
the text encoder + duration predictors stay on their exact-f32 kernels, so every difference comes from the BERT features.  The control is
the whole exact-f32 GPU path (DeBERTa + text side) against the C / OpenMP oracle, another f32 implementation with another summation order,
on the same utterances: a mode whose flip count is within 2x of the control's changes the durations no more than re-ordering an f32 sum does.
Prints one JSON line.   usage: flip_rate_bert.py [N_UTT = 800] [N_CTRL = N_UTT]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from sbv2_api_amd import configs, model, synth

vc, bc = configs.VITS_FULL, configs.DEBERTA_FULL
bb = synth.pack_blob(synth.KIND_BERT, bc, synth.make_deberta_weights(bc))
vb = synth.pack_blob(synth.KIND_VITS, vc, synth.make_vits_weights(vc))
N_UTT, PHONES, GROUP = int(sys.argv[1]) if len(sys.argv) > 1 else 800, 128, 16
N_CTRL = min(N_UTT, int(sys.argv[2]) if len(sys.argv) > 2 else N_UTT)
utts = [synth.make_utterance(PHONES, bc, vc, seed=7000 + i) for i in range(N_UTT)]
ones = [np.ones_like(u["forced_durations"]) for u in utts]      # forced 1-frame durations keep the decoder cheap; the PREDICTIONS are recorded
vs = model.load_model(vb, False)


def features(mode):
    os.environ["SBV2_BERT_GEMM"] = mode
    s = model.load_model(bb, True)
    os.environ.pop("SBV2_BERT_GEMM")
    out = []
    for i in range(0, N_UTT, GROUP):
        hs = model.predict_batch(s, [u["input_ids"] for u in utts[i:i + GROUP]])
        out += [np.repeat(h, np.asarray(u["word2ph"], np.int64), axis=0).T.copy() for h, u in zip(hs, utts[i:i + GROUP])]   # tts_util.rs:129-154
    s.close()
    return out


def durations(feats):
    d, lw = [], []
    for i in range(0, N_UTT, GROUP):
        part = [dict(u, bert=f, forced_durations=o) for u, f, o in zip(utts[i:i + GROUP], feats[i:i + GROUP], ones[i:i + GROUP])]
        model.synthesize_batch(vs, part, sdp_ratio=0.2, noise_scale_w=0.8, noise_seed=11, forced=True, fetch=False)
        a, b = model.fetch_durations(vs, sum(u["T_text"] for u in part))
        d.append(a); lw.append(b)
    return np.concatenate(d), np.concatenate(lw)


res = {}
f32 = features("f32")
d0, l0 = durations(f32)
out = {"symbols": int(d0.size), "utterances": N_UTT, "phones_per_utterance": PHONES}
for mode in ("bf16x6", "f16x3", "bf16x3"):
    f = features(mode)
    d, l = durations(f)
    out[f"flips_{mode}_vs_f32"] = int((d != d0).sum())
    out[f"median_abs_logw_diff_{mode}"] = float(np.median(np.abs(l - l0)))
    out[f"max_abs_logw_diff_{mode}"] = float(np.abs(l - l0).max())
    out[f"max_abs_feature_diff_{mode}"] = float(max(np.abs(a - b).max() for a, b in zip(f, f32)))
w = np.exp(l0.astype(np.float64))
out["symbols_within_1e-4_rel_of_ceil_edge"] = int((np.abs(w - np.round(w)) < 1e-4 * w).sum())
# control: the C oracle (f32, other summation order), DeBERTa included, same injected noise (keyed by the in-batch index)
import sbv2_ref as R
from helpers import oracle_noise_w
lib = R.load(native=True)
threads = lib.sbv2c_set_threads(R.usable_cpus())
m = R.Model(bb, vb, lib=lib)
ctrl, off, t0, dl = 0, 0, time.time(), []
for i in range(N_CTRL):
    u = utts[i]
    h = m.bert(u["input_ids"], None, hidden=bc["hidden"])
    bert = np.repeat(h, np.asarray(u["word2ph"], np.int64), axis=0).T.copy()
    r = m.vits(bert, u["phones"], u["tones"], u["langs"], 0, u["style"], sdp_ratio=0.2, noise_w=oracle_noise_w(11, i % GROUP, u["T_text"], 0.8),
               forced_durations=ones[i], return_all=True)
    ctrl += int((r["durations"] != d0[off:off + u["T_text"]]).sum())
    dl.append(np.abs(r["logw"].reshape(-1) - l0[off:off + u["T_text"]]))
    off += u["T_text"]
m.close()
dl = np.concatenate(dl)
out.update({"control_symbols": int(off), "control_flips_gpu_f32_vs_cpu_f32": ctrl, "control_median_abs_logw_diff": float(np.median(dl)),
            "control_max_abs_logw_diff": float(dl.max()), "control_cpu_threads": int(threads), "control_wall_s": round(time.time() - t0, 1)})
print(json.dumps(out))
