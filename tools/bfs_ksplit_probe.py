"""Builder tool (GPU box): gemm_bfs<f16x3> at DeBERTa's batch shapes (2 176 tokens) with and without the round-6 batch K split (two workgroups per 64 x 128 tile
for K >= SBV2_BFS_KSPLIT_MIN chunks): us per launch.  Run twice: default, and SBV2_BFS_KSPLIT_MIN=0 (never) / =64 (also K = 1024).
  python tools/bfs_ksplit_probe.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sbv2_api_amd import _lib
l = _lib.lib()
P = lambda a: a.ctypes.data_as(_lib.f32p)
rng = np.random.default_rng(0)
out = []
for (kk, m, n) in ((1024, 1024, 2176), (1024, 3072, 2176), (1024, 4096, 2176), (4096, 1024, 2176)):
    x = rng.standard_normal((kk, n)).astype(np.float32)
    w = (rng.standard_normal((m, kk)) / np.sqrt(kk)).astype(np.float32)
    b = rng.standard_normal(m).astype(np.float32)
    r = rng.standard_normal((m, n)).astype(np.float32)
    y = np.empty((m, n), np.float32)
    ms = C.c_float()
    _lib.check(l.sbv2_debug_gemm_bfs(0, P(x), P(w), P(b), P(r), m, n, kk, 4, 0, 0, 50, P(y), C.byref(ms)))
    out.append(f"K={kk} M={m} N={n}: {ms.value * 1e3:.1f} us ({2.0 * m * n * kk / ms.value / 1e9:.0f} TF alg)")
print(f"SBV2_BFS_KSPLIT_MIN={os.environ.get('SBV2_BFS_KSPLIT_MIN', '(default 128)')}: " + " | ".join(out), flush=True)
