"""Builder tool (GPU box): conv_clx.hip (pre-split operands, LDS-DMA rings) against conv_cl.hip on the wide decoder stages' ResBlock shapes:
time per launch and bit equality.   python tools/clx_probe.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sbv2_api_amd import _lib

l = _lib.lib()
f32p = _lib.f32p
P = lambda a: None if a is None else a.ctypes.data_as(f32p)
rng = np.random.default_rng(0)
for (c, k, d, L) in ((128, 7, 3, 262144 * 2), (128, 11, 5, 262144 * 2), (128, 3, 1, 262144 * 2), (256, 7, 1, 131072), (256, 11, 5, 131072), (256, 3, 3, 131072)):
    x = rng.standard_normal((c, L)).astype(np.float32)
    w = (rng.standard_normal((c, c, k)) / np.sqrt(c * k)).astype(np.float32)
    b = rng.standard_normal(c).astype(np.float32)
    y0, y1, ys = (np.empty((c, L), np.float32) for _ in range(3))
    m0, m1, m2 = C.c_float(), C.c_float(), C.c_float()
    _lib.check(l.sbv2_debug_conv1d_cl(0, P(x), P(w), P(b), c, c, k, L, d, 0.1, 1, 20, P(y0), C.byref(m0)))
    _lib.check(l.sbv2_debug_conv1d_clx(0, P(x), P(w), P(b), None, c, c, k, L, d, 0.1, 1.0, 20, P(y1), None, C.byref(m1)))
    _lib.check(l.sbv2_debug_conv1d_clx(0, P(x), P(w), P(b), None, c, c, k, L, d, 0.1, 1.0, 20, P(y1), P(ys), C.byref(m2)))
    fl = 2.0 * c * c * k * L
    print(f"C={c:4d} k={k:2d} d={d} L={L}: conv_cl {m0.value*1e3:8.1f} us ({fl/m0.value/1e9:6.1f} TF alg)   conv_clx {m1.value*1e3:8.1f} us ({fl/m1.value/1e9:6.1f} TF)"
          f"   + parts out {m2.value*1e3:8.1f} us ({fl/m2.value/1e9:6.1f} TF)   bit-equal {bool(np.array_equal(y0, y1))}", flush=True)
