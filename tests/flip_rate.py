"""Builder experiment (GPU box): how many INTEGER durations change when the text side's k = 3 convolutions (text-encoder FFN, duration
predictor) run on the split-bf16 matrix cores (SBV2_TEXT_GEMM=bf16x3, ~2^-17 relative error per product) instead of the exact-f32 MFMA?
~10^5 symbols of synthetic utterances through both builds of the same weights; also the f32-vs-f32 control (two batch compositions of the
same utterances: the f32 path is batch invariant, so 0 is expected).  Prints one JSON line."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from sbv2_api_amd import configs, model, synth

vc, bc = configs.VITS_FULL, configs.DEBERTA_FULL
vb = synth.pack_blob(synth.KIND_VITS, vc, synth.make_vits_weights(vc))
N_UTT, PHONES = int(sys.argv[1]) if len(sys.argv) > 1 else 400, 128
utts = []
for i in range(N_UTT):
    u = synth.make_utterance(PHONES, bc, vc, seed=7000 + i)
    u["bert"] = synth.hash_normal(9000 + i, vc["bert_dim"] * u["T_text"]).reshape(vc["bert_dim"], -1)
    utts.append(u)


def durations(env, group):
    os.environ.update(env)
    s = model.load_model(vb, False)
    for k in env:
        os.environ.pop(k)
    out, lw = [], []
    for i in range(0, N_UTT, group):
        part = utts[i:i + group]
        model.synthesize_batch(s, [dict(u, forced_durations=np.ones_like(u["forced_durations"])) for u in part], sdp_ratio=0.2, noise_scale_w=0.8,
                               noise_seed=11, forced=True, fetch=False)      # forced 1-frame durations keep the decoder cheap; predictions are recorded
        d, l = model.fetch_durations(s, sum(u["T_text"] for u in part))
        out.append(d); lw.append(l)
    s.close()
    return np.concatenate(out), np.concatenate(lw)

d_f32, l_f32 = durations({}, 16)
d_f32b, _ = durations({}, 10)
d_x3, l_x3 = durations({"SBV2_TEXT_GEMM": "bf16x3"}, 16)
n = d_f32.size
w = np.exp(l_f32.astype(np.float64))
print(json.dumps({"symbols": int(n), "flips_f32_vs_f32_other_batching": int((d_f32 != d_f32b).sum()), "flips_split_bf16_text_convs_vs_f32": int((d_f32 != d_x3).sum()),
                  "max_abs_logw_diff": float(np.abs(l_f32 - l_x3).max()), "median_abs_logw_diff": float(np.median(np.abs(l_f32 - l_x3))),
                  "symbols_within_1e-4_rel_of_ceil_edge": int((np.abs(w - np.round(w)) < 1e-4 * w).sum())}))
