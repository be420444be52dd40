"""Builder tool (GPU box): race screen for the LDS-DMA kernels (gemm_skinny, its k > 1 twin, conv_cl_small).  Random shapes, many repetitions:
every result must be bit-identical to the tiled kernels' (sbv2_debug_set_skinny_max(0)); a DMA that is read before it has landed, or a
ring slot that is refilled too early, shows up as a rare mismatch.   usage: python tools/small_grid_stress.py [seconds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from sbv2_api_amd import _lib, model, synth, configs

lib = _lib.lib()
f32p = _lib.f32p
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(time.time()))


def conv(x, w, b, dil, slope):
    cout, cin, k = w.shape
    y = np.empty((cout, x.shape[1]), np.float32)
    P = lambda a: a.ctypes.data_as(f32p)
    _lib.check(lib.sbv2_debug_conv1d(0, P(x), P(w), P(b), cin, cout, k, x.shape[1], dil, slope, P(y)))
    return y


t0, n, bad = time.time(), 0, 0
while time.time() - t0 < budget * 0.6:
    cin = 16 * int(rng.integers(1, 130))
    cout = int(rng.integers(17, 1500))
    k = int(rng.choice([1, 1, 1, 3, 3, 5, 7]))
    dil = int(rng.choice([1, 1, 2, 3])) if k > 1 else 1
    L = int(rng.integers(1, 400))
    x = rng.standard_normal((cin, L)).astype(np.float32)
    w = (rng.standard_normal((cout, cin, k)) / np.sqrt(cin * k)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    slope = float(rng.choice([1.0, 0.1]))
    prev = lib.sbv2_debug_set_skinny_max(0)
    ref = conv(x, w, b, dil, slope)
    lib.sbv2_debug_set_skinny_max(1 << 30)
    for rep in range(3):
        got = conv(x, w, b, dil, slope)
        if not np.array_equal(got.view(np.uint32), ref.view(np.uint32)):
            bad += 1
            print(f"MISMATCH cin={cin} cout={cout} k={k} dil={dil} L={L} slope={slope}: {int((got != ref).sum())} of {got.size}", flush=True)
    lib.sbv2_debug_set_skinny_max(prev)
    n += 1
print(f"gemm_skinny: {n} random shapes x 3 repetitions, {bad} mismatches", flush=True)

# whole single-utterance calls (flow FFN through conv_cl_small, everything else through the skinny kernels), kernels on vs off
bc, vc = configs.DEBERTA_FULL, configs.VITS_FULL
bs = model.load_model(synth.pack_blob(synth.KIND_BERT, bc, synth.make_deberta_weights(bc)), True)
vs = model.load_model(synth.pack_blob(synth.KIND_VITS, vc, synth.make_vits_weights(vc)), False)
pipe = model.Pipeline(bs, vs)
m, bad2 = 0, 0
while time.time() - t0 < budget:
    nph = int(rng.integers(3, 200))
    u = synth.make_utterance(nph, bc, vc, seed=int(rng.integers(1 << 30)))
    kw = dict(sdp_ratio=0.2, noise_scale=0.6, noise_scale_w=0.8, noise_seed=int(rng.integers(1 << 30)))
    prev = lib.sbv2_debug_set_skinny_max(0)
    b = pipe.prepare([u], **kw); pipe.run(b); ref = pipe.fetch(b)[0]
    lib.sbv2_debug_set_skinny_max(prev)
    for rep in range(3):
        b = pipe.prepare([u], **kw); pipe.run(b); got = pipe.fetch(b)[0]
        if got.shape != ref.shape or not np.array_equal(got.view(np.uint32), ref.view(np.uint32)):
            bad2 += 1
            print(f"MISMATCH whole call, {nph} phones", flush=True)
    m += 1
print(f"single-utterance calls: {m} random utterances x 3 repetitions, {bad2} mismatches", flush=True)
sys.exit(1 if bad or bad2 else 0)
