"""Builder tool (GPU box): the per-stage precision map of the HiFi-GAN decoder.  For each assignment of an arithmetic to the five
upsampling stages (SBV2_DECODER_STAGES; stage channels 256 / 128 / 64 / 32 / 16) one full-shape 128-phoneme utterance is synthesised from weights whose
conv_post is scaled so that the waveform peaks near 0.9, and compared with the C / OpenMP f32 oracle on the same weights (max-abs and rms error).
"w16" rows round every decoder weight to an f16-representable value first (oracle and GPU alike): a single-pass f16 stage then multiplies EXACT weights
by f16-rounded activations, which is the arithmetic of a two-pass f16 scheme x_hi * (w_hi + w_lo) - its error without building its kernel.
Each assignment is also benched (batch 32 x 128 phonemes).  One JSON object per line -> profiles/r05_precision_map.json.
  python tools/precision_map.py [gain]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))

if len(sys.argv) > 3 and sys.argv[1] == "--child":
    import numpy as np
    from sbv2_api_amd import configs, model, synth
    import sbv2_ref as R
    gain, w16 = float(sys.argv[2]), int(sys.argv[3])
    bc, vc = configs.DEBERTA_FULL, configs.VITS_FULL
    bw, vw = synth.make_deberta_weights(bc), synth.make_vits_weights(vc)
    lib = R.load(native=True)
    lib.sbv2c_set_threads(R.usable_cpus())
    W = dict(vw)
    if w16:
        for k in list(W):
            if k.startswith("dec.") and k.endswith(".weight"):
                W[k] = np.asarray(W[k], np.float32).astype(np.float16).astype(np.float32)
    W["dec.conv_post.weight"] = (np.asarray(W["dec.conv_post.weight"], np.float32) * np.float32(gain)).astype(np.float32)
    bb, vb = synth.pack_blob(synth.KIND_BERT, bc, bw), synth.pack_blob(synth.KIND_VITS, vc, W)
    bs, vs = model.load_model(bb, True), model.load_model(vb, False)
    pipe = model.Pipeline(bs, vs)
    u = synth.make_utterance(128, bc, vc, seed=7)
    b = pipe.prepare([u], forced=True)
    pipe.run(b)
    got = pipe.fetch(b)[0]
    m = R.Model(bb, vb, lib=lib)
    h = m.bert(u["input_ids"], None, hidden=bc["hidden"])
    bert = np.repeat(h, np.asarray(u["word2ph"], np.int64), axis=0).T.copy()
    ref = m.vits(bert, u["phones"], u["tones"], u["langs"], 0, u["style"], forced_durations=u["forced_durations"])
    m.close(); pipe.close(); bs.close(); vs.close()
    assert got.shape == ref.shape
    print(json.dumps({"peak_abs": round(float(np.abs(ref).max()), 4), "rms": round(float(np.sqrt((ref.astype(np.float64) ** 2).mean())), 4),
                      "max_abs_err_vs_f32_oracle": float(np.abs(got - ref).max()), "rms_err": float(np.sqrt(((got - ref).astype(np.float64) ** 2).mean()))}), flush=True)
    sys.exit(0)

gain = sys.argv[1] if len(sys.argv) > 1 else "15"
X3 = "bf16x3"
ROWS = (("all stages bf16x3 (default)", [X3] * 5, 0, True),
        ("C=256 single-pass f16, the rest bf16x3", ["f16", X3, X3, X3, X3], 0, True),
        ("C=128 single-pass f16, the rest bf16x3", [X3, "f16", X3, X3, X3], 0, True),
        ("wide stages (C=256, 128) single-pass f16, narrow stages bf16x3", ["f16", "f16", X3, X3, X3], 0, True),
        ("all stages single-pass f16", ["f16"] * 5, 0, False),
        ("two-pass f16 x_hi * (w_hi + w_lo) on the wide stages (emulated: f16-exact weights), narrow stages bf16x3", ["f16", "f16", X3, X3, X3], 1, False),
        ("two-pass f16 x_hi * (w_hi + w_lo) on every stage (emulated: f16-exact weights)", ["f16"] * 5, 1, False),
        ("control: f16-exact weights, all stages bf16x3", [X3] * 5, 1, False))
for name, stages, w16, bench in ROWS:
    env = dict(os.environ, SBV2_DECODER_STAGES=",".join(stages))
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", gain, str(w16)], env=env, capture_output=True, text=True, timeout=1800)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    rec = {"assignment": name, "SBV2_DECODER_STAGES": ",".join(stages), "f16_exact_weights": bool(w16)}
    rec.update(json.loads(line[-1]) if line else {"error": r.stderr[-400:]})
    if bench:
        b = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "8", "--warmup", "2", "--no-cpu-baseline"], env=env, capture_output=True, text=True,
                           timeout=1800)
        bl = [l for l in b.stdout.splitlines() if l.startswith("{")]
        if bl:
            d = json.loads(bl[-1])
            rec["bench"] = {"audio_s_per_s": d["value"], "ms_per_step": d["ms_per_step"],
                            "decoder_buckets_ms": {k: round(v, 2) for k, v in d["roofline"]["per_config_ms"].items() if "conv_cl" in k or "respair" in k}}
    print(json.dumps(rec), flush=True)
