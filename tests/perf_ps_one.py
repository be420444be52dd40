"""One pre-split-operand conv shape in a loop (conv_ps.hip), beside conv_cl.  usage: perf_ps_one.py C k d split L iters [Cout]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sbv2_api_amd import _lib
l = _lib.lib()
c, k, d, split, L, iters = (int(a) for a in sys.argv[1:7])
co = int(sys.argv[7]) if len(sys.argv) > 7 else c
P = lambda a: a.ctypes.data_as(_lib.f32p)
rng = np.random.default_rng(0)
x = rng.standard_normal((c, L)).astype(np.float32)
if os.environ.get('ZERO_X'):
    x[:] = 0
if os.environ.get('ZERO_W'):
    pass
w = (rng.standard_normal((co, c, k)) / np.sqrt(c * k)).astype(np.float32)
if os.environ.get('ZERO_W'):
    w[:] = 0
b = np.zeros(co, np.float32)
y = np.empty((co, L), np.float32)
ms = np.zeros(1, np.float32)
_lib.check(l.sbv2_debug_conv1d_ps(0, P(x), P(w), P(b), c, co, k, L, d, 0.1, 0.1, split, 0, iters, P(y), None, P(ms)))
ps = float(ms[0])
_lib.check(l.sbv2_debug_conv1d_cl(0, P(x), P(w), P(b), c, co, k, L, d, 0.1, 1 if split else 2, iters, P(y), P(ms)))
f = 2.0 * c * co * k * L / 1e9
print(f"C={c}->{co} k={k} d={d} split={split} L={L}: conv_ps {ps:.3f} ms {f / ps:.1f} TFLOP/s | conv_cl {ms[0]:.3f} ms {f / ms[0]:.1f} TFLOP/s")
