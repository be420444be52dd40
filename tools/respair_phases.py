import ctypes as C, json, os, sys
sys.path.insert(0, "/root/repo")
sys.path.insert(0, os.getcwd())
from sbv2_api_amd import _lib
l = _lib.lib()
NAMES = {1: "staged", 7: "c1_chunk0_mfma", 8: "c1_chunk0_bar", 2: "conv1_done", 3: "mid_written", 4: "mid_bar", 5: "conv2_done", 6: "stores_issued"}
for (c, k, d, L) in ((64, 11, 5, 114816 * 16), (64, 7, 3, 114816 * 16), (64, 3, 1, 114816 * 16), (32, 11, 5, 229632 * 16), (32, 7, 3, 229632 * 16), (32, 3, 1, 229632 * 16), (16, 11, 1, 459264 * 16), (16, 7, 3, 459264 * 16), (16, 3, 1, 459264 * 16)):
    for var, name in ((3, "product"), (2, "stamped")):
        out = (C.c_double * 20)()
        _lib.check(l.sbv2_debug_respair_clock(0, c, k, d, L, var, 0, 0.6, out, 20))
        r = {"C": c, "k": k, "variant": name, "ms": round(out[1], 4), "alg_tflops": round(4.0 * c * c * k * L / out[1] / 1e9, 1)}
        if var == 2:
            r["clock_mhz"] = round(out[0], 1)
            r["phases_cyc"] = {NAMES[i]: int(out[2 + i]) for i in (1, 7, 8, 2, 3, 4, 5, 6)}
        print(json.dumps(r), flush=True)
