"""Builder tool (GPU box): one k = 7 / 11 ResBlock1 branch (dilations 1, 3, 5) of the 16- / 32-channel decoder stages as three launches of the fused step
(respair_clx.hip) against ONE launch (resbranch_clx.hip), on random data at half the bench's plane size (0.47 GB): ms per branch, bits.
  python tools/resbranch_k_probe.py [iters]      (C = 32 rows: the fused branch is not instantiated there any more, the hook refuses)"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sbv2_api_amd import _lib

l = _lib.lib()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
P = lambda a: a.ctypes.data_as(_lib.f32p)
for c, k, n in ((16, 7, 459264 * 16), (16, 11, 459264 * 16)):
    rng = np.random.default_rng(c + k)
    x = rng.standard_normal((n, c), dtype=np.float32)
    w = (rng.standard_normal((6, c, c, k)) / np.sqrt(k * c)).astype(np.float32)
    b = rng.standard_normal((6, c)).astype(np.float32)
    d = np.array([1, 3, 5], np.int64)
    res, ys = {}, {}
    for variant in (0, 1, 0, 1):
        y = np.zeros((n, c), np.float32)
        ms = C.c_float(0)
        _lib.check(l.sbv2_debug_resbranch(0, P(x), P(w), P(b), c, n, k, d.ctypes.data_as(_lib.i64p), None, 1, 1.0 / 3, 0, variant, iters, P(y), C.byref(ms), None, 0))
        res.setdefault(variant, []).append(round(ms.value, 4))
        ys[variant] = y
    print(json.dumps({"C": c, "k": k, "positions": n, "three_steps_ms": res[0], "one_launch_ms": res[1],
                      "same_bits": bool(np.array_equal(ys[0], ys[1])),
                      "alg_tflops_one_launch": round(3 * 4.0 * c * c * k * n / min(res[1]) / 1e9, 1)}), flush=True)
