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

import numpy as np
f32p = _lib.f32p
P = lambda a: a.ctypes.data_as(f32p)
print("channels-last bf16 MFMA kernel (mode 1 = split-bf16, 2 = bf16)")
rng = np.random.default_rng(0)
for (c, L) in ((256, 7176), (128, 57408), (64, 114816), (32, 229632), (16, 459264)):
    Lt = L * 4   # quarter of the batch: the host transposes in the debug entry
    x = rng.standard_normal((c, Lt)).astype(np.float32)
    y = np.empty((c, Lt), np.float32)
    for k, d in ((3, 1), (11, 5)):
        w = (rng.standard_normal((c, c, k)) / np.sqrt(c * k)).astype(np.float32)
        b = np.zeros(c, np.float32)
        for mode in (1, 2):
            ms = np.zeros(1, np.float32)
            _lib.check(l.sbv2_debug_conv1d_cl(0, P(x), P(w), P(b), c, c, k, Lt, d, 0.1, mode, 5, P(y), P(ms)))
            fl = 2.0 * c * c * k * Lt
            print(f"CL mode={mode} C={c:4d} k={k:2d} d={d} L={Lt:9d}: {ms[0]:8.3f} ms  {fl / ms[0] / 1e9:8.1f} TFLOP/s  {2.0 * c * Lt * 4 / ms[0] / 1e6:8.1f} GB/s", flush=True)
