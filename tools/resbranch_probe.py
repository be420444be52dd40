"""Builder tool (GPU box): one k = 3 ResBlock1 branch (dilations 1, 3, 5) of the <= 64-channel decoder stages as three launches of the fused step
(respair_clx.hip) against ONE launch (resbranch_clx.hip), on random data at half the bench's plane size (0.47 GB): ms per branch, plane passes, bits.
  python tools/resbranch_probe.py [iters]"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sbv2_api_amd import _lib

l = _lib.lib()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
P = lambda a: a.ctypes.data_as(_lib.f32p)
for c, n in ((16, 459264 * 16), (32, 229632 * 16), (64, 114816 * 16), (128, 57408 * 16)):
    rng = np.random.default_rng(c)
    x = rng.standard_normal((n, c), dtype=np.float32)
    w = (rng.standard_normal((6, c, c, 3)) / np.sqrt(3 * c)).astype(np.float32)
    b = rng.standard_normal((6, c)).astype(np.float32)
    d = np.array([1, 3, 5], np.int64)
    res = {}
    for variant in ((2, 1, 2, 1) if c > 64 else (0, 1, 0, 1)):
        y = np.zeros((n, c), np.float32)
        ms = C.c_float(0)
        NST = 16 * 40000
        st = np.zeros(NST, np.uint64)
        _lib.check(l.sbv2_debug_resbranch(0, P(x), P(w), P(b), c, n, 3, d.ctypes.data_as(_lib.i64p), None, 1, 1.0 / 3, 0, variant, iters, P(y), C.byref(ms),
                                          st.ctypes.data if variant == 1 else None, NST))
        res.setdefault(variant, []).append(round(ms.value, 4))
        res[f"y{variant}"] = y
        if variant == 1:
            s16 = st.reshape(-1, 16)
            s16 = s16[s16[:, 15] != 0]
            lo = lambda a: (a & np.uint64(0xFFFFFFFF)).astype(np.int64)
            dt = (lo(s16[:, 6]) - lo(s16[:, 0])) % (1 << 32)
            dr = (lo(s16[:, 15]) - lo(s16[:, 14])) % (1 << 32)
            res["timeline"] = {"workgroups": int(s16.shape[0]), "clock_mhz": round(float(np.median(dt / np.maximum(dr, 1)) * 100.0), 1),
                               "cycles_from_entry": {name: int(np.median((lo(s16[:, i]) - lo(s16[:, 0])) % (1 << 32)))
                                                     for i, name in ((1, "window_converted"), (2, "step1_done"), (3, "step2_done"), (4, "step3_done"), (6, "stores_issued"))}}
    plane_gb = n * c * 4 / 1e9
    ref = 2 if c > 64 else 0
    res[0], res["y0"] = res[ref], res[f"y{ref}"]
    print(json.dumps({"C": c, "positions": n, "plane_GB": round(plane_gb, 3), "reference": "six conv_cl launches" if c > 64 else "three respair_clx launches",
                      "three_steps_ms": res[0], "one_launch_ms": res[1],
                      "timeline": res.get("timeline"), "same_bits": bool(np.array_equal(res["y0"], res["y1"])),
                      "three_steps_TBps": round(6 * plane_gb / min(res[0]), 2), "one_launch_TBps_of_2_passes": round(2 * plane_gb / min(res[1]), 2),
                      "alg_tflops_one_launch": round(3 * 4.0 * c * c * 3 * n / min(res[1]) / 1e9, 1)}), flush=True)
