"""Builder tool (GPU box): the split-bf16 1x1 GEMM of gemm_bfs.hip (pre-split operands, LDS-DMA ring, transposing LDS reads) at the
DeBERTa-large / flow batch shapes: error against an f64 reference and time per launch next to the exact-f32 tiled GEMM.
  python tools/bfs_probe.py [quick]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sbv2_api_amd import _lib

l = _lib.lib()
f32p = _lib.f32p
P = lambda a: a.ctypes.data_as(f32p)
rng = np.random.default_rng(0)
quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
shapes = [(1024, 1024, 2112), (1024, 3072, 2112), (1024, 4096, 2112), (4096, 1024, 2112), (192, 768, 28704), (192, 192, 28704), (96, 192, 28704),
          (1024, 1024, 68), (1024, 4096, 68), (4096, 1024, 68), (192, 576, 900)]
if quick:
    shapes = [(64, 96, 132), (1024, 1024, 2112), (192, 576, 900)]
for (kk, m, n) in shapes:
    x = rng.standard_normal((kk, n)).astype(np.float32)
    w = (rng.standard_normal((m, kk)) / np.sqrt(kk)).astype(np.float32)
    b = rng.standard_normal(m).astype(np.float32)
    r = rng.standard_normal((m, n)).astype(np.float32)
    ref = w.astype(np.float64) @ x.astype(np.float64) + b[:, None] + r
    fl = 2.0 * m * n * kk
    ms32 = C.c_float()
    _lib.check(l.sbv2_debug_time_conv1d(0, kk, m, 1, n, 1, 20, C.byref(ms32)))
    line = f"K={kk:5d} M={m:5d} N={n:6d}: f32 {ms32.value*1e3:7.1f} us ({fl/ms32.value/1e9:6.1f} TF)"
    for parts in (2, 3, 4):
        y = np.empty((m, n), np.float32)
        ms = C.c_float()
        _lib.check(l.sbv2_debug_gemm_bfs(0, P(x), P(w), P(b), P(r), m, n, kk, parts, 0, 0, 20, P(y), C.byref(ms)))
        err = float(np.abs(y - ref).max())
        tag = {2: "x3", 3: "x6", 4: "h3"}[parts]
        line += f"   {tag} {ms.value*1e3:7.1f} us ({fl/ms.value/1e9:6.1f} TF alg) err {err:.1e}"
    # f32 reference error for scale
    y32 = (w @ x + b[:, None] + r).astype(np.float32)
    line += f"   [numpy f32 err {float(np.abs(y32 - ref).max()):.1e}]"
    print(line, flush=True)
