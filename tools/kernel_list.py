"""Builder tool: per-dispatch durations of one traced bench step (rocprofv3 --kernel-trace csv), kernels whose name contains any of the patterns, in launch
order with their grid sizes.   python3 tools/kernel_list.py <trace dir> pat1 [pat2 ...]"""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
pats = sys.argv[2:]
# the last step = the last third of the launches, roughly: print the last occurrence block
sel = [r for r in rows if any(p in r["Kernel_Name"] for p in pats)]
n = len(sel)
for r in sel[-(n // int(__import__("os").environ.get("STEPS", "3"))):]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    name = r["Kernel_Name"].replace("void sbv2::", "").split("(")[0]
    print(f"{d:9.1f} us  grid {int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1):7d} x {r['Workgroup_Size_X']:>4}  lds {r.get('LDS_Block_Size', '?'):>6}  {name}")
