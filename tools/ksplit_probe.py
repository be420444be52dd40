"""Builder tool (GPU box): a single utterance's DeBERTa products (f16x3, 68 columns) on gemm_bfs with the K loop split over groups of waves (default) and
unsplit (sbv2_debug_set_ksplit(0)): us per launch and the max-abs difference of the two results."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sbv2_api_amd import _lib
l = _lib.lib()
f32p = C.POINTER(C.c_float)
P = lambda a: None if a is None else a.ctypes.data_as(f32p)
rng = np.random.default_rng(3)
for (K, M, N) in ((1024, 1024, 68), (1024, 3072, 68), (1024, 4096, 68), (4096, 1024, 68), (192, 192, 900), (192, 384, 900)):
    x = rng.standard_normal((K, N)).astype(np.float32); w = (rng.standard_normal((M, K)) / np.sqrt(K)).astype(np.float32); b = rng.standard_normal(M).astype(np.float32)
    out = {}
    for ks in (1, 0):
        prev = l.sbv2_debug_set_ksplit(ks)
        y = np.empty((M, N), np.float32); ms = C.c_float()
        _lib.check(l.sbv2_debug_gemm_bfs(0, P(x), P(w), P(b), None, M, N, K, 4, 0, 0, 200, P(y), C.byref(ms)))
        l.sbv2_debug_set_ksplit(prev)
        out[ks] = (y, ms.value)
    ref = w.astype(np.float64) @ x.astype(np.float64) + b[:, None]
    print(json.dumps({"K": K, "M": M, "N": N, "split_us": round(out[1][1] * 1e3, 2), "unsplit_us": round(out[0][1] * 1e3, 2),
                      "split_vs_unsplit_max_abs": float(np.abs(out[1][0] - out[0][0]).max()), "split_vs_f64": float(np.abs(out[1][0] - ref).max()),
                      "unsplit_vs_f64": float(np.abs(out[0][0] - ref).max())}), flush=True)
