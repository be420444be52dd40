"""Builder tool (GPU box, under rocprofv3 --pmc): a few launches of the product fused-ResBlock kernel per shape (no stamps)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sbv2_api_amd import _lib
l = _lib.lib()
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 1
for (c, k, d, L) in ((32, 7, 3, 229632 * 16), (16, 7, 3, 459264 * 16), (64, 7, 3, 114816 * 16)):
    out = (C.c_double * 20)()
    _lib.check(l.sbv2_debug_respair_clock(0, c, k, d, L, variant, 0, 0.02, out, 20))
    print(c, k, d, out[1], flush=True)
