"""Builder tool (GPU box): names the HIP API call behind the one slow pass of tools/long_form_check.py.  Reads the hip_api_trace csv of
  rocprofv3 --hip-trace --output-format csv -d gpurun_out/lf_trace -- python3 tools/long_form_check.py 2000 256
and prints the 25 longest API calls (name, ms, start offset in s, thread) and, per API name, calls / total ms / max ms."""
import collections, csv, glob, json, os, sys
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/lf_trace"
f = max(glob.glob(os.path.join(root, "**", "*hip_api_trace.csv"), recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
t0 = min(int(r["Start_Timestamp"]) for r in rows)
calls = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Function"], int(r["Start_Timestamp"]) - t0, r.get("Thread_Id", "")) for r in rows]
calls.sort(reverse=True)
print(json.dumps({"trace": os.path.basename(f), "api_calls": len(calls)}))
for d, name, st, th in calls[:25]:
    print(json.dumps({"api": name, "ms": round(d / 1e6, 3), "start_s": round(st / 1e9, 4), "thread": th}))
print(json.dumps({"section": "every API call >= 1 ms in time order (start_s relative to the first call)"}))
for d, name, st, th in sorted((c for c in calls if c[0] >= 1e6), key=lambda c: c[2]):
    print(json.dumps({"t_s": round(st / 1e9, 4), "api": name, "ms": round(d / 1e6, 2)}))
# time the host spends OUTSIDE the HIP API: the ten longest gaps between the end of one call and the start of the next (same thread), with the calls on either side
by_start = sorted(((int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0, r["Function"]) for r in rows))
gaps = sorted(((by_start[i + 1][0] - by_start[i][1], i) for i in range(len(by_start) - 1)), reverse=True)[:10]
print(json.dumps({"section": "longest gaps between API calls (host time outside HIP)"}))
for g, i in gaps:
    print(json.dumps({"gap_ms": round(g / 1e6, 2), "after": by_start[i][2], "after_end_s": round(by_start[i][1] / 1e9, 4), "before": by_start[i + 1][2]}))
# ... and what ran inside the 150 ms before each stream pass' first long event wait (the stall of the slow pass sits there)
agg = collections.defaultdict(lambda: [0, 0, 0])
for d, name, st, th in calls:
    a = agg[name]; a[0] += 1; a[1] += d; a[2] = max(a[2], d)
for name, (n, tot, mx) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]:
    print(json.dumps({"api_total": name, "calls": n, "total_ms": round(tot / 1e6, 2), "max_ms": round(mx / 1e6, 3)}))
