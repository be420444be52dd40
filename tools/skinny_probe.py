"""Builder tool (GPU box): launch time of the small-grid 1x1 shapes of a single-utterance call with the tiled f32 GEMM vs the one-wave
16 x 16 kernel (gemm_skinny.hip)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sbv2_api_amd import _lib
l = _lib.lib()
shapes = [(1024, 1024, 66), (1024, 3072, 66), (1024, 4096, 66), (4096, 1024, 66), (1024, 1024, 130), (4096, 1024, 130), (1024, 2048, 512),
          (192, 576, 897), (192, 192, 897), (192, 768, 897), (768, 192, 897), (192, 384, 257), (1024, 192, 257)]
for (kk, m, n) in shapes:
    out = []
    for thr in (0, 1 << 30):
        prev = l.sbv2_debug_set_skinny_max(thr)
        ms = C.c_float()
        _lib.check(l.sbv2_debug_time_conv1d(0, kk, m, 1, n, 1, 50, C.byref(ms)))
        l.sbv2_debug_set_skinny_max(prev)
        out.append(ms.value * 1e3)
    print(f"K={kk:5d} M={m:5d} N={n:4d}: tiled {out[0]:7.1f} us   skinny {out[1]:7.1f} us   ({2.0 * m * n * kk / out[1] / 1e6:6.1f} TFLOP/s)", flush=True)
