"""Builder tool: average PMC counters per (kernel, grid) of the rocprofv3 --pmc passes under a directory (p1, p2, ...)."""
import csv, glob, collections, sys
d = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "respair"
for p in sorted(glob.glob(d + "/p*/")):
    for f in glob.glob(p + "*/*counter_collection.csv"):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            if pat not in r["Kernel_Name"]:
                continue
            agg[(r["Kernel_Name"].split("(")[0][-40:], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for key, dd in agg.items():
            print(key, " ".join(f"{c}={sum(v) / len(v):.4g}" for c, v in sorted(dd.items())))
