"""Builder tool (GPU box): 1x1 products at DeBERTa-large batch shapes, exact-f32 tiled GEMM vs the channels-last split-bf16 kernel."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sbv2_api_amd import _lib
l = _lib.lib()
f32p = _lib.f32p
rng = np.random.default_rng(0)
for (kk, m, n) in ((1024, 1024, 2112), (1024, 3072, 2112), (1024, 4096, 2112), (4096, 1024, 2112), (192, 768, 28704)):
    ms = C.c_float()
    _lib.check(l.sbv2_debug_time_conv1d(0, kk, m, 1, n, 1, 20, C.byref(ms)))
    t32 = ms.value
    x = rng.standard_normal((kk, n)).astype(np.float32)
    w = (rng.standard_normal((m, kk, 1)) / np.sqrt(kk)).astype(np.float32)
    b = rng.standard_normal(m).astype(np.float32)
    y = np.empty((m, n), np.float32)
    ms2 = C.c_float()
    P = lambda a: a.ctypes.data_as(f32p)
    _lib.check(l.sbv2_debug_conv1d_cl(0, P(x), P(w), P(b), kk, m, 1, n, 1, 1.0, 1, 20, P(y), C.byref(ms2)))
    ref = w[:, :, 0].astype(np.float64) @ x.astype(np.float64) + b[:, None]
    err = float(np.abs(y - ref).max())
    fl = 2.0 * m * n * kk
    print(f"K={kk:5d} M={m:5d} N={n:6d}: f32 {t32*1e3:8.1f} us ({fl/t32/1e9:6.1f} TF)   cl split-bf16 {ms2.value*1e3:8.1f} us ({fl/ms2.value/1e9:6.1f} TF alg)   max err {err:.2e}", flush=True)
