"""Builder tool (GPU box): the three fused steps of a k = 7 / 11 ResBlock1 branch (respair_clx.hip, dilations 1, 3, 5) at C = 64 / 32 on random data at half the
bench's plane size, ms per branch.  Run from the repo root (the library as shipped) and from build/probe16 (respair_clx.hip built with -DRPX_PROBE16=1: every
32x32x16 MFMA issued as two 16x16x32 from the same registers; results wrong, timing only):
  python3 tools/respair_shape_probe.py [iters]"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from sbv2_api_amd import _lib

l = _lib.lib()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
P = lambda a: a.ctypes.data_as(_lib.f32p)
for c, k, n in ((64, 7, 114816 * 16), (64, 11, 114816 * 16), (32, 7, 229632 * 16), (32, 11, 229632 * 16)):
    rng = np.random.default_rng(c + k)
    x = rng.standard_normal((n, c), dtype=np.float32)
    w = (rng.standard_normal((6, c, c, k)) / np.sqrt(k * c)).astype(np.float32)
    b = rng.standard_normal((6, c)).astype(np.float32)
    d = np.array([1, 3, 5], np.int64)
    res = []
    for rep in range(3):
        y = np.zeros((n, c), np.float32)
        ms = C.c_float(0)
        _lib.check(l.sbv2_debug_resbranch(0, P(x), P(w), P(b), c, n, k, d.ctypes.data_as(_lib.i64p), None, 1, 1.0 / 3, 0, 0, iters, P(y), C.byref(ms), None, 0))
        res.append(round(ms.value, 4))
    print(json.dumps({"lib": os.path.relpath(_lib.__file__), "C": c, "k": k, "positions": n, "three_steps_ms": res,
                      "alg_tflops": round(3 * 4.0 * c * c * k * n / min(res) / 1e9, 1)}), flush=True)
