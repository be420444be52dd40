"""Builder tool: how busy the GPU is inside a latency loop.  From a rocprofv3 --kernel-trace csv: over the LAST `window_ms` of the trace, the union of the
kernel intervals (busy), the idle time between consecutive kernels split into gaps < 5 us / 5-20 us / > 20 us, launches, and the kernels with the largest total
time.   python3 tools/trace_gaps.py <trace dir> [window_ms = 500]"""
import csv, glob, json, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
win = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 500e6
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
t1 = rows[-1][1]
rows = [r for r in rows if r[0] >= t1 - win]
t0 = rows[0][0]
busy, cur_e, gaps = 0, t0, []
for s, e, _ in rows:
    if s > cur_e:
        gaps.append(s - cur_e)
        busy += e - s
        cur_e = e
    elif e > cur_e:
        busy += e - cur_e
        cur_e = e
wall = t1 - t0
by = {}
for s, e, n in rows:
    n = n.replace("void sbv2::", "").replace("sbv2::", "").replace("(anonymous namespace)::", "").split("(")[0][:70]
    a = by.setdefault(n, [0, 0])
    a[0] += e - s
    a[1] += 1
top = sorted(by.items(), key=lambda kv: -kv[1][0])[:14]
print(json.dumps({"window_ms": round(wall / 1e6, 1), "launches": len(rows), "gpu_busy_fraction": round(busy / wall, 4),
                  "idle_ms": round((wall - busy) / 1e6, 2), "gaps": {"n": len(gaps), "<5us_ms": round(sum(g for g in gaps if g < 5000) / 1e6, 2),
                                                                      "5-20us_ms": round(sum(g for g in gaps if 5000 <= g < 20000) / 1e6, 2),
                                                                      ">20us_ms": round(sum(g for g in gaps if g >= 20000) / 1e6, 2),
                                                                      "n>20us": sum(1 for g in gaps if g >= 20000)},
                  "mean_kernel_us": round(busy / max(len(rows), 1) / 1e3, 2),
                  "top_kernels_ms": {k: [round(v[0] / 1e6, 2), v[1]] for k, v in top}}))
