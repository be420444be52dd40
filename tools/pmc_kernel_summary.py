"""Builder tool: per-kernel means of the PMC counters of a rocprofv3 --pmc run (counter_collection.csv), optionally for one grid size only.
usage: pmc_kernel_summary.py <counter_collection.csv> <kernel name substring> [grid_size]"""
import csv, sys, collections
rows = csv.DictReader(open(sys.argv[1]))
sub = sys.argv[2]
grid = int(sys.argv[3]) if len(sys.argv) > 3 else None
acc = collections.defaultdict(lambda: [0.0, 0])
for r in rows:
    if sub not in r["Kernel_Name"]: continue
    if grid is not None and int(r["Grid_Size"]) != grid: continue
    a = acc[r["Counter_Name"]]
    a[0] += float(r["Counter_Value"]); a[1] += 1
for k in sorted(acc): print(f"{k:32s} {acc[k][0] / acc[k][1]:16.1f}  (n = {acc[k][1]})")
