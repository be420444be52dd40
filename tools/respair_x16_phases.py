"""Builder tool (GPU box): phase stamps (cycles from entry, medians over workgroups) of the fused step at C = 64 / 32, k = 7 / 11: respair_clx.hip (32x32x16) against
respair_x16.hip (16x16x32 operand scheme).   python3 tools/respair_x16_phases.py"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.getcwd())
from sbv2_api_amd import _lib
l = _lib.lib()
NAMES = {1: "staged", 9: "x16_first_pair_done", 7: "x16_chunk_pair0_done", 8: "x16_chunk_pair1_converted", 2: "conv1_done", 3: "mid_written", 4: "mid_bar", 5: "conv2_done", 6: "stores_issued"}
for (c, k, d, L) in ((64, 11, 5, 114816 * 16), (64, 11, 1, 114816 * 16), (64, 7, 3, 114816 * 16), (32, 11, 5, 229632 * 16), (32, 7, 3, 229632 * 16)):
    for var, name, abl in ((2, "respair_clx stamped", 0), (4, "respair_x16 stamped", 0), (4, "respair_x16 stamped, nobody waits for weights (wrong results)", 32)):
        out = (C.c_double * 20)()
        _lib.check(l.sbv2_debug_respair_clock(0, c, k, d, L, var, abl, 0.6, out, 20))
        r = {"C": c, "k": k, "variant": name, "ms": round(out[1], 4), "alg_tflops": round(4.0 * c * c * k * L / out[1] / 1e9, 1), "clock_mhz": round(out[0], 1),
             "dil": d, "phases_cyc": {NAMES[i]: int(out[2 + i]) for i in (1, 9, 7, 8, 2, 3, 4, 5, 6)}}
        print(json.dumps(r), flush=True)
