"""Turns the raw rocprofv3 outputs of tools/prof_final.sh (gpurun_out/final_*) into the tracked summaries under profiles/.
usage: python tools/prof_summarise.py <tag>      (e.g. r01i)"""
import collections
import csv
import glob
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
tag = sys.argv[1]
P = lambda name: os.path.join(ROOT, "profiles", f"{tag}_{name}")


def one(pattern):
    f = glob.glob(os.path.join(G, pattern))
    return max(f, key=os.path.getmtime) if f else None   # gpurun_out/ accumulates: take the newest run


def per_kernel(path, counters):
    """-> {kernel: (launches, avg_us, {counter: mean per launch})} from a counter_collection.csv"""
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    seen, cnt, dur = set(), collections.Counter(), collections.defaultdict(float)
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if r["Counter_Name"] in counters:
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            cnt[k] += 1
            dur[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return {k: (cnt[k], dur[k] / cnt[k] / 1e3, {c: agg[k][c] / cnt[k] for c in counters}) for k in cnt}


shutil.copy(os.path.join(G, "final_bench.json"), P("bench.json"))
for extra in ("mixed256", "b1_latency", "longform_c256", "longform_c1024", "dropin_latency", "dropin_latency_fp32", "dropin25_gaps"):
    src = os.path.join(G, f"final_{extra}.json")
    if os.path.exists(src) and os.path.getsize(src) > 2:
        shutil.copy(src, P(f"{extra}.json"))
st = one("final_stats/*/*kernel_stats.csv")
if st:
    shutil.copy(st, P("bench_kernel_stats.csv"))
b1 = one("final_b1_stats/*/*kernel_stats.csv")
if b1:
    shutil.copy(b1, P("b1_kernel_stats.csv"))   # 45 single-utterance calls (5 warm-up + 40 timed) under the tracer
fe, wr = one("final_pmc_fetch/*/*counter_collection.csv"), one("final_pmc_write/*/*counter_collection.csv")
if fe and wr:
    F, W = per_kernel(fe, ["FETCH_SIZE"]), per_kernel(wr, ["WRITE_SIZE"])
    rows = ["kernel,launches,avg_us,FETCH_SIZE_KB_per_launch(raw),fetch_bytes_per_launch(x2 gfx950 correction),WRITE_SIZE_KB_per_launch,"
            "write_bytes_per_launch,hbm_GBps"]
    # every kernel that takes >= 0.5 % of the pass's kernel time (the PMC pass runs the steps un-pipelined: launches / steps = launches per step), then the
    # HBM bytes per step of the decoder's kernels (the channels-last family) and of everything
    total_us = sum(F[k][0] * F[k][1] for k in F)
    dec = lambda k: any(t in k for t in ("conv_clx_kernel", "respair_clx_kernel", "respair_x16_kernel", "respair_cl_kernel", "resbranch_clx_kernel", "conv_cl_kernel", "conv_cl_small", "k_conv_post_tanh",
                                         "k_split_cl(", "k_clx_zero_halo", "k_add_segvec_cl", "k_transpose_out"))
    ffn = lambda k: "conv_clx_kernel<5," in k
    steps = int(os.environ.get("PMC_STEPS", "3"))   # bench.py --steps 1 --warmup 1 + its instrumented roofline step
    sums = {"decoder": 0.0, "all": 0.0}
    for k in sorted(F, key=lambda k: -F[k][0] * F[k][1]):
        n, us, c = F[k]
        fkb = c["FETCH_SIZE"]
        wkb = W.get(k, (0, 0, {"WRITE_SIZE": 0.0}))[2]["WRITE_SIZE"]
        fb, wb = fkb * 1024 * 2, wkb * 1024
        sums["all"] += (fb + wb) * n / steps
        if dec(k) and not ffn(k):
            sums["decoder"] += (fb + wb) * n / steps
        if n * us >= 0.005 * total_us:
            rows.append(f"\"{k[:110]}\",{n},{us:.1f},{fkb:.1f},{fb:.3e},{wkb:.1f},{wb:.3e},{(fb + wb) / (us * 1e-6) / 1e9:.0f}")
    rows.append(f"\"TOTAL HBM bytes per step: HiFi-GAN decoder kernels\",,,,{sums['decoder']:.4e},,,")
    rows.append(f"\"TOTAL HBM bytes per step: all kernels\",,,,{sums['all']:.4e},,,")
    open(P("pmc_hbm_traffic.csv"), "w").write("\n".join(rows) + "\n")
mf = one("final_pmc_mfma/*/*counter_collection.csv")
if mf:
    names = ["SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"]
    M = per_kernel(mf, names)
    rows = ["kernel,launches,avg_us,MFMA_busy_cycles_per_launch,GRBM_GUI_ACTIVE_per_launch,mfma_util(busy/(1024 SIMD * GUI_ACTIVE/8)),"
            "SQ_WAIT_ANY/SQ_WAVE_CYCLES,LDS_bank_conflict/LDS_idx_active,lds_util(LDS_idx_active/(256 CU * GUI_ACTIVE/8))"]
    total_us = sum(M[k][0] * M[k][1] for k in M)
    for k in sorted(M, key=lambda k: -M[k][0] * M[k][1]):
        n, us, c = M[k]
        if n * us < 0.005 * total_us:
            continue
        cyc = c["GRBM_GUI_ACTIVE"] / 8
        rows.append(f"\"{k[:110]}\",{n},{us:.1f},{c['SQ_VALU_MFMA_BUSY_CYCLES']:.3e},{c['GRBM_GUI_ACTIVE']:.3e},"
                    f"{c['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * cyc) if cyc else 0:.3f},{c['SQ_WAIT_ANY'] / max(c['SQ_WAVE_CYCLES'], 1):.3f},"
                    f"{c['SQ_LDS_BANK_CONFLICT'] / max(c['SQ_LDS_IDX_ACTIVE'], 1):.3f},{c['SQ_LDS_IDX_ACTIVE'] / (256 * cyc) if cyc else 0:.3f}")
    open(P("pmc_mfma_util.csv"), "w").write("\n".join(rows) + "\n")
print("wrote", sorted(os.path.basename(f) for f in glob.glob(P("*"))))
