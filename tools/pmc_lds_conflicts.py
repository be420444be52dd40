"""Builder tool: LDS bank-conflict ratio and average duration per kernel from a rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE pass.
  python3 tools/pmc_lds_conflicts.py <dir> [name filter]"""
import collections, csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True))[-1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
agg, dur, seen = collections.defaultdict(lambda: collections.defaultdict(float)), collections.defaultdict(list), set()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].replace("void sbv2::", "").replace("(anonymous namespace)::", "").split("(")[0]
    if flt not in k:
        continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Dispatch_Id"] not in seen:
        seen.add(r["Dispatch_Id"])
        dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, c in sorted(agg.items(), key=lambda kv: -sum(dur[kv[0]])):
    print(f"{k[:80]:80s} launches {len(dur[k]):4d} avg {sum(dur[k]) / len(dur[k]):8.1f} us  conflict/idx_active {c['SQ_LDS_BANK_CONFLICT'] / max(c['SQ_LDS_IDX_ACTIVE'], 1):.3f}")
