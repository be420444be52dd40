"""One channels-last conv shape in a loop (for rocprofv3 --pmc runs). usage: perf_cl_one.py C k d mode L iters"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sbv2_api_amd import _lib
l = _lib.lib()
c, k, d, mode, L, iters = (int(a) for a in sys.argv[1:7])
P = lambda a: a.ctypes.data_as(_lib.f32p)
rng = np.random.default_rng(0)
x = rng.standard_normal((c, L)).astype(np.float32)
w = (rng.standard_normal((c, c, k)) / np.sqrt(c * k)).astype(np.float32)
b = np.zeros(c, np.float32)
y = np.empty((c, L), np.float32)
ms = np.zeros(1, np.float32)
_lib.check(l.sbv2_debug_conv1d_cl(0, P(x), P(w), P(b), c, c, k, L, d, 0.1, mode, iters, P(y), P(ms)))
print(f"C={c} k={k} d={d} mode={mode} L={L}: {ms[0]:.3f} ms  {2.0 * c * c * k * L / ms[0] / 1e9:.1f} TFLOP/s")
