"""Builder experiment (GPU box): how many INTEGER durations change when the text side's k = 3 convolutions (text-encoder FFN, duration
predictor) run on the split-bf16 matrix cores (SBV2_TEXT_GEMM=bf16x3, ~2^-17 relative error per product) instead of the exact-f32 MFMA?
~10^5 symbols of synthetic utterances through both builds of the same weights; the control is the exact-f32 GPU path against the C / OpenMP
oracle (another f32 implementation with a different summation order) on a subset.  Prints one JSON line."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
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
d_x3, l_x3 = durations({"SBV2_TEXT_GEMM": "bf16x3"}, 16)
n = d_f32.size
w = np.exp(l_f32.astype(np.float64))
# control: the C oracle (f32, other summation order) on the first N_CTRL utterances, same injected noise (keyed by the in-batch index)
import sbv2_ref as R
from helpers import oracle_noise_w
N_CTRL = min(N_UTT, int(sys.argv[2]) if len(sys.argv) > 2 else 160)
lib = R.load(native=True)
lib.sbv2c_set_threads(R.usable_cpus())
m = R.Model(None, vb, lib=lib)
ctrl, off = 0, 0
for i in range(N_CTRL):
    u = utts[i]
    r = m.vits(u["bert"], u["phones"], u["tones"], u["langs"], 0, u["style"], sdp_ratio=0.2, noise_w=oracle_noise_w(11, i % 16, u["T_text"], 0.8),
               forced_durations=np.ones_like(u["forced_durations"]), return_all=True)
    ctrl += int((r["durations"] != d_f32[off:off + u["T_text"]]).sum())
    off += u["T_text"]
m.close()
print(json.dumps({"symbols": int(n), "flips_split_bf16_text_convs_vs_f32": int((d_f32 != d_x3).sum()),
                  "control_symbols": int(off), "control_flips_gpu_f32_vs_cpu_f32": ctrl,
                  "max_abs_logw_diff": float(np.abs(l_f32 - l_x3).max()), "median_abs_logw_diff": float(np.median(np.abs(l_f32 - l_x3))),
                  "symbols_within_1e-4_rel_of_ceil_edge": int((np.abs(w - np.round(w)) < 1e-4 * w).sum())}))
