"""Times the production conv kernel on decoder-shaped problems (run on the GPU box; not a pytest file)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sbv2_api_amd import _lib

l = _lib.lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
for (c, L) in ((256, 7176), (128, 57408), (64, 114816), (32, 229632), (16, 459264)):
    for k, d in ((3, 1), (7, 3), (11, 5)):
        ms = C.c_float()
        Lt = L * B
        _lib.check(l.sbv2_debug_time_conv1d(0, c, c, k, Lt, d, 5, C.byref(ms)))
        fl = 2.0 * c * c * k * Lt
        by = 2.0 * c * Lt * 4
        print(f"C={c:4d} k={k:2d} d={d} L={Lt:9d}: {ms.value:8.3f} ms  {fl / ms.value / 1e9:8.1f} TFLOP/s  {by / ms.value / 1e6:8.1f} GB/s", flush=True)
for (m, n, kk) in ((1024, 2048, 1024), (4096, 2048, 1024), (1024, 2048, 4096), (768, 28704, 192), (192, 28704, 768)):
    ms = C.c_float()
    _lib.check(l.sbv2_debug_time_conv1d(0, kk, m, 1, n, 1, 10, C.byref(ms)))
    print(f"GEMM M={m} N={n} K={kk}: {ms.value:8.3f} ms  {2.0 * m * n * kk / ms.value / 1e9:8.1f} TFLOP/s", flush=True)
