"""Builder tool: GPU busy time of a rocprofv3 kernel trace (union of the kernel intervals), idle gaps, and how much of it two or more kernels
share.  usage: trace_union.py <..._kernel_trace.csv> [skip_fraction = 0.3]   (the first part of the trace, warm-up, is skipped)"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
t_lo = ev[0][0] + skip * (ev[-1][1] - ev[0][0])
ev = [e for e in ev if e[0] >= t_lo]
span = max(e[1] for e in ev) - ev[0][0]
pts = sorted([(s, 1) for s, e, _ in ev] + [(e, -1) for s, e, _ in ev])
busy = over = 0
depth = 0
last = pts[0][0]
gaps = []
for t, d in pts:
    if depth >= 1: busy += t - last
    if depth >= 2: over += t - last
    if depth == 0 and t > last: gaps.append(t - last)
    depth += d
    last = t
gaps.sort(reverse=True)
print(f"span {span/1e6:.1f} ms, busy {busy/1e6:.1f} ms ({100*busy/span:.1f} %), >=2 kernels in flight {over/1e6:.1f} ms ({100*over/span:.1f} %), "
      f"idle {sum(gaps)/1e6:.2f} ms in {len(gaps)} gaps, largest {[round(g/1e3) for g in gaps[:8]]} us, sum of kernel durations {sum(e-s for s,e,_ in ev)/1e6:.1f} ms")
