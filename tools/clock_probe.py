"""Builder tool (GPU box): the in-kernel clock check of MI355X_MICROARCH.md ("DVFS give-back", item 6) on the dominant decoder convolution
(conv_cl, 128-row workgroups, split-bf16): the kernel, the kernel without its MFMAs, its MFMAs alone, its staging alone.  If the three
variants hold the same clock, the kernel's MFMA and staging halves adding up instead of overlapping is a scheduling problem, not power.
  python tools/clock_probe.py [seconds]"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sbv2_api_amd import _lib

l = _lib.lib()
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 2.5
rows = []
for (c, k, d, L) in ((128, 7, 3, 57408 * 32), (256, 7, 1, 7176 * 32), (128, 11, 5, 57408 * 32), (128, 3, 1, 57408 * 32)):
    for abl, name in ((0, "full kernel"), (1, "no MFMA (fragment reads + staging + epilogue)"), (2, "MFMA only"), (3, "staging + barriers only"),
                      (10, "conv_clx (pre-split operands, LDS-DMA rings)")):
        out = (C.c_double * 4)()
        _lib.check(l.sbv2_debug_conv_cl_clock(0, c, k, d, L, abl, secs, out))
        fl = 2.0 * c * c * k * L
        r = {"C": c, "k": k, "dil": d, "positions": L, "variant": name, "clock_mhz": round(out[0], 1), "ms_per_launch": round(out[1], 4),
             "loop_cycles_per_workgroup": int(out[2]), "alg_tflops": round(fl / out[1] / 1e9, 1) if abl in (0, 2, 10) else None}
        rows.append(r)
        print(json.dumps(r), flush=True)
