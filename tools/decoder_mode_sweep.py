"""Builder tool (GPU box): the single-pass decoder question with data.
For each decoder arithmetic (SBV2_DECODER = bf16x3 (default) | f16 | bf16 | f32) one full-shape 128-phoneme utterance is synthesised from weights whose
conv_post is scaled so that the waveform peaks near 0.9 (the synthetic generator's output otherwise peaks at ~0.1: an error of 1e-4 there is 1e-3
RELATIVE), and compared with the C / OpenMP f32 oracle on the same weights; the unscaled weights are measured beside it.  Each mode is also benched
(batch 32 x 128 phonemes, pipelined steps).  One JSON object per line; the whole output is kept as profiles/r04_decoder_mode_sweep.json.
  python tools/decoder_mode_sweep.py [gain]         each mode runs in a child process (SBV2_DECODER is read when the model is created)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))

if len(sys.argv) > 2 and sys.argv[1] == "--child":
    import numpy as np
    from sbv2_api_amd import configs, model, synth
    import sbv2_ref as R
    gain = float(sys.argv[2])
    bc, vc = configs.DEBERTA_FULL, configs.VITS_FULL
    bw, vw = synth.make_deberta_weights(bc), synth.make_vits_weights(vc)
    out = {"decoder": os.environ.get("SBV2_DECODER", "bf16x3")}
    lib = R.load(native=True)
    lib.sbv2c_set_threads(R.usable_cpus())
    for tag, g in (("unit_peak", gain), ("synthetic", 1.0)):
        W = dict(vw)
        W["dec.conv_post.weight"] = (np.asarray(vw["dec.conv_post.weight"], np.float32) * np.float32(g)).astype(np.float32)
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
        out[tag] = {"conv_post_gain": g, "peak_abs": round(float(np.abs(ref).max()), 4), "rms": round(float(np.sqrt((ref.astype(np.float64) ** 2).mean())), 4),
                    "max_abs_err_vs_f32_oracle": float(np.abs(got - ref).max()), "rms_err": float(np.sqrt(((got - ref).astype(np.float64) ** 2).mean()))}
    print(json.dumps(out), flush=True)
    sys.exit(0)

gain = sys.argv[1] if len(sys.argv) > 1 else "15"
for mode in ("bf16x3", "f16", "bf16", "f32"):
    env = dict(os.environ, SBV2_DECODER=mode)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", gain], env=env, capture_output=True, text=True, timeout=1200)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    rec = json.loads(line[-1]) if line else {"decoder": mode, "error": r.stderr[-400:]}
    if mode != "f32":      # (the exact-f32 decoder is a 300 ms step: parity reference only)
        b = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "2", "--no-cpu-baseline"], env=env, capture_output=True, text=True,
                           timeout=1200)
        bl = [l for l in b.stdout.splitlines() if l.startswith("{")]
        if bl:
            d = json.loads(bl[-1])
            rec["bench"] = {"audio_s_per_s": d["value"], "ms_per_step": d["ms_per_step"],
                            "decoder_buckets_ms": {k: round(v, 2) for k, v in d["roofline"]["per_config_ms"].items() if "conv_cl" in k or "respair" in k}}
    print(json.dumps(rec), flush=True)
