"""Builder tool (GPU box): where a conv_clx workgroup's life goes (sbv2_debug_clx_timeline): per shape and launch form the un-stamped launch time and,
from the stamps of one launch, the medians of a workgroup's prologue / step loop / epilogue / store drain, the gap between one workgroup leaving a CU and
the next entering it, and the average number of workgroups per CU that are inside their step loop.
  python tools/clx_timeline.py [variant ...]"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sbv2_api_amd import _lib

l = _lib.lib()
variants = [int(v) for v in sys.argv[1:]] or [0]
SHAPES = ((128, 3, 1, 57408 * 32), (128, 7, 3, 57408 * 32), (128, 11, 1, 57408 * 32), (256, 3, 1, 7176 * 32), (256, 7, 3, 7176 * 32), (256, 11, 1, 7176 * 32))
if os.environ.get("CLX_TL_SHAPES"):
    SHAPES = tuple(SHAPES[int(i)] for i in os.environ["CLX_TL_SHAPES"].split(","))
KINDS = [int(v) for v in os.environ.get("CLX_TL_KINDS", "1,2").split(",")]
for (c, k, d, L) in SHAPES:
    for kind in KINDS:
        for var in variants:
            cap = 12 * 40000
            buf = (C.c_uint64 * cap)()
            nwg, ms = C.c_int64(), C.c_double()
            _lib.check(l.sbv2_debug_clx_timeline(0, c, k, d, L, kind, var, 0.6, buf, cap, C.byref(nwg), C.byref(ms)))
            a = np.frombuffer(buf, dtype=np.uint64, count=12 * nwg.value).reshape(-1, 12).astype(np.int64)
            a = a[a[:, 4] > 0]                      # workgroups that ran (padding tiles return early)
            ent, l0, l1, ex, dr = a[:, 4], a[:, 1], a[:, 3], a[:, 5], a[:, 6]
            hw = a[:, 7]
            cu = ((hw >> 32) & 0xF) * 4096 + ((hw >> 13) & 3) * 512 + ((hw >> 12) & 1) * 256 + ((hw >> 8) & 0xF)
            us = lambda v: float(np.median(v)) / 100.0
            clock = np.median((a[:, 2] - a[:, 0]) / np.maximum(l1 - l0, 1)) * 100.0
            gaps, inloop, resident = [], [], []
            for u in np.unique(cu):
                m = cu == u
                e, dd = np.sort(ent[m]), np.sort(dr[m])
                nres = 3
                if len(e) > 2 * nres:
                    g = e[nres:] - dd[:-nres]
                    gaps.extend(g[1:-1].tolist())
                span = dd.max() - e.min()
                inloop.append((l1[m] - l0[m]).sum() / max(span, 1))
                resident.append((dr[m] - ent[m]).sum() / max(span, 1))
            r = {"C": c, "k": k, "dil": d, "kind": {1: "conv1", 2: "conv2", 3: "conv2-last"}[kind], "variant": var, "ms_per_launch": round(ms.value, 4),
                 "workgroups": int(len(a)), "cus_seen": int(len(np.unique(cu))), "loop_clock_mhz": round(float(clock), 0),
                 "prologue_us": round(us(l0 - ent), 2), "loop_us": round(us(l1 - l0), 2), "epilogue_issue_us": round(us(ex - l1), 2),
                 "epi_barrier_us": round(us(a[:, 8] - l1), 2), "epi_reads_us": round(us(a[:, 9] - a[:, 8]), 2), "epi_half0_us": round(us(a[:, 10] - a[:, 9]), 2),
                 "epi_half1_us": round(us(ex - a[:, 10]), 2), "store_drain_us": round(us(dr - ex), 2), "life_us": round(us(dr - ent), 2),
                 "gap_exit_to_next_entry_us": round(float(np.median(gaps)) / 100.0, 2) if gaps else None,
                 "avg_workgroups_in_loop_per_cu": round(float(np.mean(inloop)), 2), "avg_workgroups_resident_per_cu": round(float(np.mean(resident)), 2),
                 "launch_span_us": round(float(dr.max() - ent.min()) / 100.0, 1)}
            print(json.dumps(r), flush=True)
