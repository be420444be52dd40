"""Turns the passes of tools/pmc_mem_pass.sh (gpurun_out/mem_pmc_*) into profiles/<tag>_pmc_memory_side.csv: per kernel (>= 0.5 % of the pass) launches, us, the L2's
hit rate, the fabric's read / write requests per launch and their DRAM credit stalls, and what the waves wait for.   python tools/pmc_mem_summarise.py <tag>"""
import collections, csv, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
tag = sys.argv[1]


def per_kernel(name):
    f = glob.glob(os.path.join(G, f"mem_pmc_{name}", "*", "*counter_collection.csv"))
    if not f:
        return {}
    agg, seen, cnt, dur = collections.defaultdict(lambda: collections.defaultdict(float)), set(), collections.Counter(), collections.defaultdict(float)
    for r in csv.DictReader(open(max(f, key=os.path.getmtime))):
        k = r["Kernel_Name"]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"]); cnt[k] += 1; dur[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return {k: (cnt[k], dur[k] / cnt[k] / 1e3, {c: v / cnt[k] for c, v in agg[k].items()}) for k in cnt}


H, E, S, Q = per_kernel("hit"), per_kernel("ea"), per_kernel("stall"), per_kernel("sq")
tot = sum(n * us for n, us, _ in H.values())
rows = ["kernel,launches,avg_us,TCC_hit_rate,EA_read_requests_per_launch,EA_write_requests_per_launch,EA_RDREQ_DRAM_CREDIT_STALL_per_launch,EA_WRREQ_STALL_per_launch,"
        "SQ_WAIT_INST_ANY/SQ_WAVE_CYCLES,SQ_ACTIVE_INST_VMEM/SQ_WAVE_CYCLES,VMEM_reads_per_launch,VMEM_writes_per_launch"]
g = lambda D, k, c: D.get(k, (0, 0, {}))[2].get(c, 0.0)
for k in sorted(H, key=lambda k: -H[k][0] * H[k][1]):
    n, us, c = H[k]
    if n * us < 0.005 * tot:
        continue
    hit, miss = c.get("TCC_HIT_sum", 0), c.get("TCC_MISS_sum", 0)
    wc = max(g(Q, k, "SQ_WAVE_CYCLES"), 1.0)
    rows.append(f"\"{k[:100]}\",{n},{us:.1f},{hit / max(hit + miss, 1):.3f},{g(E, k, 'TCC_EA0_RDREQ_sum'):.3e},{g(E, k, 'TCC_EA0_WRREQ_sum'):.3e},"
                f"{g(S, k, 'TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum'):.3e},{g(S, k, 'TCC_EA0_WRREQ_STALL_sum'):.3e},{g(Q, k, 'SQ_WAIT_INST_ANY') / wc:.3f},"
                f"{g(Q, k, 'SQ_ACTIVE_INST_VMEM') / wc:.3f},{g(Q, k, 'SQ_INSTS_VMEM_RD'):.3e},{g(Q, k, 'SQ_INSTS_VMEM_WR'):.3e}")
out = os.path.join(ROOT, "profiles", f"{tag}_pmc_memory_side.csv")
open(out, "w").write("\n".join(rows) + "\n")
print(out)
