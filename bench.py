"""Benchmark of the sbv2_core hot path on MI355X (contract: see the task statement; one JSON line on rank 0).

Workload (BASELINE.json `metric`, configs[2]): a batch of 32 synthetic utterances of 128 phone symbols per GPU
(T_text 257, BERT S 64, teacher-forced durations -> 897 frames = 10.414 s of 44.1 kHz audio each),
full ku-nlp/deberta-v2-large + Style-Bert-VITS2 JP-Extra shapes with procedurally generated weights.
One step = DeBERTa -> word2ph feature repeat -> text encoder + both duration predictors -> flow -> HiFi-GAN for the
whole batch, PCM left in HBM; with N > 1 ranks every rank synthesises its own 32 utterances (weak scaling, no
data-path collective) and the PCM is gathered to rank 0 over RCCL.  Steps are pipelined one deep (the library runs
consecutive batches on alternating execution contexts): step n's PCM is collected / gathered right after step n+1 has been
enqueued and the final fence drains the pipeline, so all K steps' work lies inside the timed region.

Extra objects on the JSON line:
  roofline     — the dominant kernel (the implicit-GEMM conv tile configuration with the most time), algorithmic FLOP
                 over HIP-event durations of one extra instrumented step run right after the timed ones.
  cpu_baseline — oracle/sbv2_oracle.py (a port, NOT onnxruntime) on the host cores for ONE utterance of the same
                 workload (rank 0, N = 1 only), torch CPU convolutions, all host threads.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak (not the 2:1-sparsity headline)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32, help="utterances per GPU")
    ap.add_argument("--phones", type=int, default=128)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--tiny", action="store_true", help="tiny model shapes (plumbing check only; not a valid bench number)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    torch = None
    use_dist = world > 1 or os.environ.get("SBV2_FORCE_DIST") == "1"   # the latter exercises the RCCL path on one GPU
    if use_dist:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from sbv2_api_amd import _lib, model, synth
    l = _lib.lib()
    if l.sbv2_device_count() < 1:
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists)")

    from sbv2_api_amd import configs   # the oracle is imported by the cpu_baseline leg only
    bc, vc = (configs.DEBERTA_TINY, configs.VITS_TINY) if args.tiny else (configs.DEBERTA_FULL, configs.VITS_FULL)
    bw = synth.make_deberta_weights(bc)
    vw = synth.make_vits_weights(vc)
    bs = model.load_model(synth.pack_blob(synth.KIND_BERT, bc, bw), True, device=local_rank)
    vs = model.load_model(synth.pack_blob(synth.KIND_VITS, vc, vw), False, device=local_rank)
    pipe = model.Pipeline(bs, vs)
    utts = [synth.make_utterance(args.phones, bc, vc, seed=rank * 1000 + i) for i in range(args.batch)]
    b = pipe.prepare(utts, forced=True)   # benchmark mode: blank 1 frame, phone 6 frames

    hop = l.sbv2_vits_hop(vs.handle)
    dmode = l.sbv2_vits_decoder_mode(vs.handle)
    dtype = {0: "f32", 1: "bf16x3-split (decoder convs: bf16 hi/lo MFMA, f32 accumulate/storage) + f32", 2: "bf16 (decoder convs) + f32",
             3: "f16 (decoder convs: fp16 MFMA operands, f32 accumulate/storage) + f32"}[dmode]
    send = recv = None

    # Steps are pipelined one deep: step n's PCM is collected (and, with N > 1, gathered to rank 0 over RCCL) right after step n+1
    # has been enqueued, so the latency-bound DeBERTa / text / flow part of a batch overlaps the decoder of the previous one.
    # fence() drains the pipeline: every step's work, gather included, lies inside the timed region.
    pending = []

    def collect(ticket):
        if use_dist:
            nonlocal send, recv
            n = int(b.lens.sum())
            if send is None:
                send = torch.empty(n, dtype=torch.float32, device="cuda")
                recv = [torch.empty(n, dtype=torch.float32, device="cuda") for _ in range(world)] if rank == 0 else None
            pipe.fetch_ticket_to_device(ticket, send.data_ptr(), send.numel())
            dist.gather(send, recv, dst=0)
        else:
            pipe.wait(ticket)

    def step():
        pipe.run(b)
        pending.append(b.ticket)
        if len(pending) > 1:
            collect(pending.pop(0))

    def fence():
        while pending:
            collect(pending.pop(0))
        pipe.sync()
        if use_dist:
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    if use_dist:   # create the RCCL communicators outside the measured steps, whatever --warmup is
        dist.barrier()
        dummy = torch.zeros(1, device="cuda")
        dist.gather(dummy, [torch.zeros(1, device="cuda") for _ in range(world)] if rank == 0 else None, dst=0)
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    audio_s_per_rank = float(b.lens.sum()) / configs.SAMPLE_RATE
    total_audio = audio_s_per_rank * world * args.steps
    value = total_audio / dt

    # ---- roofline leg: one extra instrumented step, HIP events around every implicit-GEMM launch ----------------
    _lib.check(l.sbv2_prof_begin())
    pipe.run(b)
    pipe.sync()
    buf = C.create_string_buffer(1 << 16)
    _lib.check(l.sbv2_prof_end(buf, len(buf)))
    prof = json.loads(buf.value.decode())
    dom = max(prof, key=lambda r: r["ms"]) if prof else None
    roofline = None
    if dom:
        ach = dom["flop"] / (dom["ms"] * 1e-3) / 1e12
        is_cl = dom["kernel"].startswith("conv_cl") or dom["kernel"].startswith("conv_ps")
        peak = PEAK_BF16_MFMA_TFLOPS if is_cl else PEAK_F32_MFMA_TFLOPS
        roofline = {"bound": "mfma", "kernel": dom["kernel"], "achieved": round(ach, 2), "peak": peak,
                    "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": None,
                    "note": ("algorithmic FLOP; split-bf16 issues 3 bf16 MFMAs per algorithmic product (hi*hi + hi*lo + lo*hi), so "
                             "executed MFMA FLOP/s = 3x achieved") if "split" in dom["kernel"] else "algorithmic FLOP",
                    "launches_per_step": dom["launches"], "avg_launch_ms": round(dom["ms"] / dom["launches"], 4),
                    "per_config_ms": {r["kernel"]: round(r["ms"], 3) for r in prof},
                    "all_conv_gemm_ms_per_step": round(sum(r["ms"] for r in prof), 3),
                    "all_conv_gemm_tflops": round(sum(r["flop"] for r in prof) / (sum(r["ms"] for r in prof) * 1e-3) / 1e12, 2)}

    # HBM traffic of the dominant kernel: PMC counters cannot be read from inside this process; the per-launch FETCH_SIZE (x2, the
    # gfx950 correction of MI355X_MICROARCH.md) + WRITE_SIZE of the SAME command line under `rocprofv3 --pmc` is kept in profiles/.
    if roofline:
        try:
            import csv, glob
            pm = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_hbm_traffic.csv")))[-1]
            # kernel names as rocprofv3 prints them (the precision template argument was a bool before the fp16 mode: both spellings)
            want = {"conv_cl<2,split-bf16>": ("conv_cl_kernel<2, 1, false, false>", "conv_cl_kernel<2, true, false, false>"),
                    "conv_cl<1,split-bf16>": ("conv_cl_kernel<1, 1, false, false>", "conv_cl_kernel<1, true, false, false>"),
                    "conv_cl<2,bf16>": ("conv_cl_kernel<2, 0, false, false>", "conv_cl_kernel<2, false, false, false>"),
                    "conv_cl<2,f16>": ("conv_cl_kernel<2, 2, false, false>",),
                    "conv_gemm<32,2,2,1,4,16>": ("conv_gemm_kernel<32, 2, 2, 1, 4, 16>",),
                    "conv_gemm<32,2,4,2,2,16>": ("conv_gemm_kernel<32, 2, 4, 2, 2, 16>",)}.get(dom["kernel"], ("\0",))
            for r in csv.DictReader(open(pm)):
                if any(w in r["kernel"] for w in want):
                    roofline["traffic"] = round(float(r["fetch_bytes_per_launch(x2 gfx950 correction)"]) + float(r["write_bytes_per_launch"]))
                    roofline["traffic_unit"] = "bytes per launch (HBM, PMC)"
                    roofline["traffic_source"] = os.path.relpath(pm, ROOT)
        except Exception:
            pass

    # ---- CPU baseline leg (rank 0, N = 1): the oracle on one utterance of the same workload ----------------------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import torch as _t
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import sbv2_oracle as O   # the checker / CPU baseline: never on the measured path
        O.set_conv_backend("torch")
        u = utts[0]
        t1 = time.perf_counter()
        h = O.deberta_forward(bw, bc, u["input_ids"])
        ref = O.vits_forward(vw, vc, O.expand_bert_features(h, u["word2ph"]), u["phones"], u["tones"], u["langs"], 0, u["style"],
                             forced_durations=u["forced_durations"])
        tc = time.perf_counter() - t1
        O.set_conv_backend("numpy")
        got = pipe.fetch(b)[0]
        err = float(np.abs(got - ref).max()) if got.shape == ref.shape else float("nan")
        cpu = {"value": round(ref.shape[0] / O.SAMPLE_RATE / tc, 4), "unit": "audio-s/s", "cores": _t.get_num_threads(), "kind": "port",
               "sample": f"1 utterance of the same workload ({args.phones} phones, {ref.shape[0] / O.SAMPLE_RATE:.3f} s audio) through "
                         f"oracle/sbv2_oracle.py (numpy + torch CPU conv), {tc:.1f} s wall; not onnxruntime",
               "gpu_vs_oracle_max_abs": err}

    if rank == 0:
        out = {
            "metric": "audio_seconds_per_second", "value": round(value, 2), "unit": "audio-s/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "real_time_factor": round(dt / total_audio, 6),
            "config": {"workload": f"batch={args.batch} x {args.phones}-phoneme utterances per GPU, DeBERTa-v2-large(22L) + "
                                   f"Style-Bert-VITS2 JP-Extra + HiFi-GAN, forced durations -> {int(b.lens[0]) // hop} frames/utt",
                       "global_batch": args.batch * world, "audio_seconds_per_step": round(audio_s_per_rank * world, 3),
                       "parallelism": f"utterance-sharded x{world}" + (", RCCL gather of PCM to rank 0" if world > 1 else ""),
                       "shapes": "tiny (plumbing check)" if args.tiny else "full"},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
