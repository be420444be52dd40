"""Builder tool (GPU box): phase timeline + in-kernel clock + ablations of the fused ResBlock step (respair_cl.hip) on random data.
  python tools/respair_probe.py [seconds]"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sbv2_api_amd import _lib

l = _lib.lib()
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
quick = len(sys.argv) > 2
NAMES = {1: "staged", 7: "c1_chunk0_mfma", 8: "c1_chunk0_bar", 9: "c1_chunk1_staged", 10: "c1_chunk1_bar", 2: "conv1_done", 3: "mid_written", 4: "mid_bar",
         5: "conv2_done", 6: "stores_issued"}
ABL = ((0, "full"), (3, "no global traffic"), (4, "no MFMA"), (8, "no window conversion"), (16, "no intermediate epilogue"), (28, "no MFMA/convert/mid"),
       (31, "barriers + fragment reads only"))
for (c, k, d, L) in ((32, 7, 3, 229632 * 16), (16, 7, 3, 459264 * 16), (64, 7, 3, 114816 * 16), (32, 11, 5, 229632 * 16), (32, 3, 1, 229632 * 16), (16, 11, 1, 459264 * 16),
                     (64, 11, 5, 114816 * 16), (64, 3, 1, 114816 * 16)):
    for abl, name in ABL:
        if abl and (k != 7 or quick):
            continue
        out = (C.c_double * 20)()
        _lib.check(l.sbv2_debug_respair_clock(0, c, k, d, L, 0, abl, secs, out, 20))
        fl = 4.0 * c * c * k * L
        r = {"C": c, "k": k, "dil": d, "positions": L, "variant": name, "clock_mhz": round(out[0], 1), "ms_per_launch": round(out[1], 4),
             "alg_tflops": round(fl / out[1] / 1e9, 1), "phases_cyc": {NAMES[i]: int(out[2 + i]) for i in (1, 7, 8, 9, 10, 2, 3, 4, 5, 6)}}
        print(json.dumps(r), flush=True)
    for var, name in ((1, "respair_cl product"), (3, "respair_clx product"), (2, "respair_clx stamped")):
        out = (C.c_double * 20)()
        _lib.check(l.sbv2_debug_respair_clock(0, c, k, d, L, var, 0, secs, out, 20))
        r = {"C": c, "k": k, "dil": d, "variant": name, "ms_per_launch": round(out[1], 4), "alg_tflops": round(4.0 * c * c * k * L / out[1] / 1e9, 1)}
        if var == 2:
            r["clock_mhz"] = round(out[0], 1)
            r["phases_cyc"] = {NAMES[i]: int(out[2 + i]) for i in (1, 7, 8, 2, 3, 4, 5, 6)}
        print(json.dumps(r), flush=True)
