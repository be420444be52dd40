"""Benchmark of the sbv2_core hot path on MI355X (contract: see the task statement; ONE JSON line on rank 0).

Workload (BASELINE.json `metric`, configs[2]): a batch of 32 synthetic utterances of 128 phone symbols per GPU
(T_text 257, BERT S 64, teacher-forced durations -> 897 frames = 10.414 s of 44.1 kHz audio each),
full ku-nlp/deberta-v2-large + Style-Bert-VITS2 JP-Extra shapes with procedurally generated weights.
One step = DeBERTa -> word2ph feature repeat -> text encoder + both duration predictors -> flow -> HiFi-GAN for the whole
batch AND the PCM on the host of rank 0 ("host inputs resident -> PCM on host of rank 0"): at N = 1 a device -> host
copy into pinned memory, at N > 1 every rank synthesises its own 32 utterances (weak scaling, no data-path collective) and the PCM
is gathered to rank 0 over RCCL inside the library (sbv2_comm_gather_pcm) and copied to rank 0's host memory.
Steps are pipelined one deep (the library runs consecutive batches on alternating execution contexts): step n's PCM is collected right
after step n+1 has been enqueued and the final fence drains the pipeline, so all K steps' work, copies included, lies inside the timed region.

`--config mixed256` is BASELINE configs[3]: ONE global batch of 256 utterances of 32..512 phonemes, dealt to the ranks by the library's
cost-sorted deal (strong scaling: the total work is fixed).

Launch: `python bench.py --gpus N` spawns N ranks itself (before anything touches the GPU); under
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` the ranks are already there (RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_PORT from the env).  No torch is imported either way: the collective is RCCL, called by libsbv2_hip.so.

Extra objects on the JSON line:
  roofline     — the dominant kernel (the implicit-GEMM conv tile configuration with the most time), algorithmic FLOP
                 over HIP-event durations of one extra instrumented step run right after the timed ones.
  cpu_baseline — oracle/sbv2_ref.c (C + OpenMP fp32 restatement, NOT onnxruntime), batch 1 looped over utterances of the same workload on
                 all host cores for a bounded time (rank 0, N = 1 only).
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak (not the 2:1-sparsity headline)


def spawn_ranks(n: int) -> int:
    """N fresh child processes, one per GPU, started BEFORE this process has made any HIP call (a process that has initialised the GPU must
    never be replaced or forked into another GPU user).  Rank 0's stdout is the JSON line; the exit code is the first non-zero one."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   SBV2_BENCH_LAUNCH=str(os.getpid()), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for p in procs:
        p.wait()
        rc = rc or p.returncode
    return rc


def uid_path() -> str:
    tag = os.environ.get("SBV2_BENCH_LAUNCH") or f"{os.getppid()}"
    return os.path.join(os.environ.get("TMPDIR", "/tmp"), f"sbv2_bench_uid_{tag}_{os.environ.get('MASTER_PORT', '0')}")


def exchange_unique_id(model, rank: int) -> bytes:
    """Rank 0 creates the ncclUniqueId; the other ranks of this node read it from a file named after the launch (same parent process,
    same MASTER_PORT).  128 bytes, written atomically."""
    path = uid_path()
    if rank == 0:
        uid = model.Comm.unique_id()
        with open(path + ".tmp", "wb") as f:
            f.write(uid)
        os.replace(path + ".tmp", path)
        return uid
    t0 = time.time()
    while time.time() - t0 < 300:
        try:
            if time.time() - os.path.getmtime(path) < 600:
                b = open(path, "rb").read()
                if len(b) == 128:
                    return b
        except OSError:
            pass
        time.sleep(0.05)
    raise SystemExit("rank 0 never published the communicator id")


class HostGroup:
    """Barrier / max / sum over the ranks of this node through TCP on 127.0.0.1 (rank 0 serves).  Always set up for N > 1: it lets all ranks
    AGREE on whether the RCCL communicator came up, and carries the bench's two scalars if it did not (the PCM then stays on each rank's
    host: the line says so).  Never on the data path."""

    def __init__(self, rank: int, world: int):
        import socket
        self.rank, self.world = rank, world
        port = (int(os.environ.get("MASTER_PORT", "29500")) + 1 + sum(map(ord, uid_path())) % 997) % 64000 + 1024
        if rank == 0:
            srv = socket.socket()
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind(("127.0.0.1", port))
            srv.listen(world)
            srv.settimeout(300)
            self.peers = []
            for _ in range(world - 1):
                c, _a = srv.accept()
                c.settimeout(600)
                self.peers.append(c)
            srv.close()
        else:
            t0 = time.time()
            while True:
                try:
                    self.sock = socket.create_connection(("127.0.0.1", port), timeout=5)
                    self.sock.settimeout(600)
                    break
                except OSError:
                    if time.time() - t0 > 300:
                        raise SystemExit("rank 0's host group never came up")
                    time.sleep(0.05)

    @staticmethod
    def _recv(c, n=8):
        buf = b""
        while len(buf) < n:
            part = c.recv(n - len(buf))
            if not part:
                raise SystemExit("a rank of the host group went away")
            buf += part
        return buf

    def reduce(self, x: float, op) -> float:
        import struct
        if self.rank == 0:
            vals = [x] + [struct.unpack("<d", self._recv(c))[0] for c in self.peers]
            r = op(vals)
            for c in self.peers:
                c.sendall(struct.pack("<d", r))
            return r
        self.sock.sendall(struct.pack("<d", x))
        return struct.unpack("<d", self._recv(self.sock))[0]

    def max(self, x): return self.reduce(float(x), max)
    def min(self, x): return self.reduce(float(x), min)
    def sum(self, x): return self.reduce(float(x), sum)
    def barrier(self): self.reduce(0.0, max)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32, help="utterances per GPU (config u128)")
    ap.add_argument("--phones", type=int, default=128)
    ap.add_argument("--config", choices=["u128", "mixed256"], default="u128",
                    help="u128 = BASELINE configs[2] (the metric's workload); mixed256 = configs[3]: one global batch of 256 x 32..512 phonemes")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="time budget of the CPU baseline sample")
    ap.add_argument("--tiny", action="store_true", help="tiny model shapes (plumbing check only; not a valid bench number)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))     # nothing above has touched the GPU

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: reporting n_gpus={world}", file=sys.stderr)

    from sbv2_api_amd import _lib, configs, model, synth
    l = _lib.lib()
    ndev = l.sbv2_device_count()
    if ndev < 1:
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists)")
    device = local_rank % ndev     # (several ranks on one GPU only happens in plumbing checks on a one-GPU box)

    bc, vc = (configs.DEBERTA_TINY, configs.VITS_TINY) if args.tiny else (configs.DEBERTA_FULL, configs.VITS_FULL)
    bw = synth.make_deberta_weights(bc)
    vw = synth.make_vits_weights(vc)
    bs = model.load_model(synth.pack_blob(synth.KIND_BERT, bc, bw), True, device=device)
    vs = model.load_model(synth.pack_blob(synth.KIND_VITS, vc, vw), False, device=device)
    pipe = model.Pipeline(bs, vs)
    if args.config == "u128":
        utts = [synth.make_utterance(args.phones, bc, vc, seed=rank * 1000 + i) for i in range(args.batch)]
        global_batch = args.batch * world
        scaling = "weak"
    else:
        rng = np.random.default_rng(256)
        ns = [int(v) for v in rng.integers(32, 513, 256)]
        every = [synth.make_utterance(n, bc, vc, seed=3000 + i, chars=min(98, max(1, n // 2 - 2))) for i, n in enumerate(ns)]
        rank_of = model.deal([7 * n + 1 for n in ns], world)      # the library's cost-sorted deal (csrc/node.cpp)
        utts = [u for u, r in zip(every, rank_of) if r == rank]
        global_batch = 256
        scaling = "strong"
    hop = l.sbv2_vits_hop(vs.handle)
    frames = lambda us: int(sum(int(u["forced_durations"].sum()) for u in us))
    # A rank's shard is run as sub-batches of <= ~64k frames (2.2x the u128 batch): the decoder workspace is ~0.3 MB per frame, so the
    # 488k frames of mixed256 on ONE GPU would need ~150 GB per execution context in a single call.  Every rank uses the same number of
    # sub-batches (the gather is collective).
    nsub = 1
    if args.config == "mixed256":
        per_rank = [sum(7 * n + 1 for n, r in zip(ns, rank_of) if r == q) for q in range(world)]
        nsub = max(1, -(-max(per_rank) // 65536))
    order = sorted(range(len(utts)), key=lambda i: -frames([utts[i]]))
    groups = [[utts[i] for i in order[g::nsub]] for g in range(nsub)]
    # decided identically on EVERY rank (from the deal / the batch size, not from this rank's own shard): a single rank leaving before the rendezvous
    # would leave the others waiting in the collectives
    empty = [q for q in range(world) if not any(r == q for r in rank_of)] if args.config != "u128" else (list(range(world)) if args.batch <= 0 else [])
    if empty:
        raise SystemExit(f"bench.py: ranks {empty} would be dealt no utterance (world {world}): use fewer ranks for this config")
    batches = [pipe.prepare(g, forced=True) for g in groups if g]   # benchmark mode: blank 1 frame, phone 6 frames
    sub_samples = [frames(g) * hop for g in groups if g]
    # a rank with fewer sub-batches than the others repeats its last one so that every rank makes the same number of collective calls; the
    # repeats are fillers: their audio is NOT counted (weight 0)
    sub_weight = [1] * len(batches)
    while len(batches) < nsub:
        batches.append(batches[-1])
        sub_samples.append(sub_samples[-1])
        sub_weight.append(0)
    b = batches[0]
    my_samples = max(sub_samples)

    dmode = l.sbv2_vits_decoder_mode(vs.handle)
    dtype = {0: "f32", 1: "bf16x3-split (decoder + flow convs: bf16 hi/lo MFMA, f32 accumulate/storage)", 2: "bf16 (decoder convs)",
             3: "f16 (decoder convs: fp16 MFMA operands, f32 accumulate/storage)"}[dmode]
    dtype += {0: " + f32 (DeBERTa, text side)", 2: " + bf16x3 (DeBERTa GEMMs) + f32 (text side)",
              3: " + bf16x6 (DeBERTa GEMMs: three bf16 parts per operand, f32-grade) + f32 (text side)",
              4: " + f16x3 (DeBERTa GEMMs, flow 1x1: f16 hi + scaled f16 lo per operand, 22 mantissa bits, f32 accumulate) + f32 (text side)"}.get(l.sbv2_bert_gemm_parts(bs.handle), "")

    comm, host, rccl_error = None, None, ""
    if world > 1:
        host = HostGroup(rank, world)
        try:
            comm = model.Comm(exchange_unique_id(model, rank), rank, world, device)
        except Exception as e:      # e.g. two ranks on one GPU in a plumbing check: RCCL refuses ("invalid usage")
            rccl_error = str(e).splitlines()[0][:200]
        if host.min(1.0 if comm is not None else 0.0) < 1.0:    # every rank takes the same path
            if comm is not None:
                comm.close()
            comm = None
            if rank == 0:
                print(f"bench.py: no RCCL communicator ({rccl_error or 'another rank failed'}): PCM stays on each rank's host", file=sys.stderr)
    if comm is not None:
        all_samples = int(comm.max(float(my_samples)))      # capacity bound for the root's buffer
        pin = model.PinnedArray(all_samples * world) if rank == 0 else None
    else:
        pin = model.PinnedArray(my_samples)

    pending = []
    counts_seen = []

    def collect(ticket):
        """PCM of one step -> host of rank 0 (inside the timed region)."""
        if comm is not None:
            counts_seen.append(comm.gather_pcm(pipe, ticket, pin.array if rank == 0 else None))
        else:
            _lib.check(l.sbv2_pipeline_fetch_pcm_ticket(pipe.h, ticket, C.c_void_p(pin.ptr), pin.array.size, 0))

    def step():
        for bb in batches:
            pipe.run(bb)
            pending.append(bb.ticket)
            if len(pending) > 1:
                collect(pending.pop(0))

    def fence():
        while pending:
            collect(pending.pop(0))
        pipe.sync()
        if comm is not None:
            comm.barrier()
        elif host is not None:
            host.barrier()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if comm is None and host is not None:
        dt = host.max(dt)
        total_samples_per_step = int(host.sum(float(sum(s * w for s, w in zip(sub_samples, sub_weight)))))
    elif comm is not None:
        dt = comm.max(dt)
        # real sub-batches only: every rank's own sum, added over the ranks through the communicator's host-visible reduction
        mine_real = float(sum(s * w for s, w in zip(sub_samples, sub_weight)))
        total_samples_per_step = int(round(host.sum(mine_real))) if host is not None else int(mine_real)
    else:
        total_samples_per_step = sum(s * w for s, w in zip(sub_samples, sub_weight))
    total_audio = total_samples_per_step / configs.SAMPLE_RATE * args.steps
    value = total_audio / dt

    # ---- roofline leg: one extra instrumented step, HIP events around every implicit-GEMM launch (rank 0) ----------------
    roofline = None
    if rank == 0:
        _lib.check(l.sbv2_prof_begin())
        for bb in batches:
            pipe.run(bb)
            pipe.sync()     # one sub-batch at a time: consecutive runs alternate between two execution contexts, and kernels of two sub-batches in
                            # flight together would stretch each other's event durations (mixed256: 8 sub-batches; u128 has one)
        buf = C.create_string_buffer(1 << 16)
        _lib.check(l.sbv2_prof_end(buf, len(buf)))
        prof = json.loads(buf.value.decode())
        dom = max(prof, key=lambda r: r["ms"]) if prof else None
        if dom:
            ach = dom["flop"] / (dom["ms"] * 1e-3) / 1e12
            is_cl = dom["kernel"].startswith(("conv_cl", "resblock", "respair"))
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
            # gfx950 correction of MI355X_MICROARCH.md) + WRITE_SIZE of the SAME command line under `rocprofv3 --pmc` is kept in profiles/:
            # an OFFLINE measurement (of the build the file name says), replayed here.
            # The file is chosen by name (profiles/<round tag>_pmc_hbm_traffic.csv, the lexicographically last = the newest round) and the row
            # by the kernel's mangled-name prefix below; a dominant kernel that file has no row for reports traffic null and says so: it
            # never falls back to an older round's file.
            roofline["traffic_source"] = None
            try:
                import csv
                import glob
                pm = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_hbm_traffic.csv")))[-1]
                # sbv2_prof_end's kernel name -> prefixes of the rocprofv3 kernel names it covers (the CSV is sorted by total time: the
                # first match is the variant that dominates)
                want = {"conv_clx<split-bf16>": ("conv_clx_kernel<11,", "conv_clx_kernel<7,", "conv_clx_kernel<3,"),
                        "conv_clx_ffn<split-bf16>": ("conv_clx_kernel<5,",),
                        "conv_cl<2,split-bf16>": ("conv_cl_kernel<2, 1, false, false",),
                        "conv_cl<1,split-bf16>": ("conv_cl_kernel<1, 1, false, false",),
                        "conv_cl<2,bf16>": ("conv_cl_kernel<2, 0, false, false",),
                        "conv_cl<1,bf16>": ("conv_cl_kernel<1, 0, false, false",),
                        "conv_cl<2,f16>": ("conv_cl_kernel<2, 2, false, false",),
                        "conv_cl<1,f16>": ("conv_cl_kernel<1, 2, false, false",),
                        "conv_cl_km<2,split-bf16>": ("conv_cl_kernel<2, 1, true", "conv_cl_kernel<2, 1, false, true"),
                        "conv_cl_km<1,split-bf16>": ("conv_cl_kernel<1, 1, true", "conv_cl_kernel<1, 1, false, true"),
                        "respair_cl<C<=32>": ("respair_clx_kernel<32,", "respair_clx_kernel<16,", "respair_cl_kernel<1, false, 1", "respair_cl_kernel<1, true, 1"),
                        "respair_cl<C=64>": ("respair_clx_kernel<64,", "respair_cl_kernel<1, false, 2"),
                        "gemm_bfs<bf16x3>": ("gemm_bfs_kernel<2,",),
                        "gemm_bfs<bf16x6>": ("gemm_bfs_kernel<3,",),
                        "gemm_bfs<f16x3>": ("true, false>(sbv2::BfsKernelParams)", "true, true>(sbv2::BfsKernelParams)", "true>(sbv2::BfsKernelParams)"),
                        "gemm_skinny<16x16x4>": ("gemm_skinny",)}.get(dom["kernel"])
                if want is None and dom["kernel"].startswith("conv_gemm<"):
                    want = ("conv_gemm_kernel<" + dom["kernel"][len("conv_gemm<"):-1].replace(",", ", "),)
                for r in csv.DictReader(open(pm)):
                    if want and any(w in r["kernel"] for w in want):
                        roofline["traffic"] = round(float(r["fetch_bytes_per_launch(x2 gfx950 correction)"]) + float(r["write_bytes_per_launch"]))
                        roofline["traffic_unit"] = "bytes per launch (HBM, PMC)"
                        roofline["traffic_measured"] = "offline"
                        roofline["traffic_source"] = os.path.relpath(pm, ROOT)
                        break
                else:
                    roofline["traffic_source"] = f"no row for {dom['kernel']} in {os.path.relpath(pm, ROOT)}"
            except Exception as e:
                roofline["traffic_source"] = f"unavailable ({type(e).__name__})"

    # ---- CPU baseline leg (rank 0, N = 1): the C / OpenMP restatement, batch 1 looped, bounded time ----------------------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.config == "u128":
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import sbv2_ref as R   # the checker / CPU baseline: never on the measured path
        lib = R.load(native=True)
        threads = lib.sbv2c_set_threads(R.usable_cpus())    # min(affinity, cgroup quota): what the box really grants this container
        # load (incl. the input-independent relative-position projections) is outside the timed loop, as on the GPU path
        m = R.Model(synth.pack_blob(synth.KIND_BERT, bc, bw), synth.pack_blob(synth.KIND_VITS, vc, vw), lib=lib)
        got = np.split(pin.array[:my_samples].copy(), np.cumsum(b.lens)[:-1])      # (u128: one sub-batch, utterance order = `order`)
        utts = [utts[i] for i in order]
        done, audio, err, tc = 0, 0.0, 0.0, 0.0
        t1 = time.perf_counter()
        while done < len(utts) and (done == 0 or (time.perf_counter() - t1) * (done + 1) / done < args.cpu_seconds):
            u = utts[done]
            h = m.bert(u["input_ids"], None, hidden=bc["hidden"])
            bert = np.repeat(h, np.asarray(u["word2ph"], np.int64), axis=0).T.copy()       # tts_util.rs:129-154
            ref = m.vits(bert, u["phones"], u["tones"], u["langs"], 0, u["style"], forced_durations=u["forced_durations"])
            tc = time.perf_counter() - t1
            audio += ref.shape[0] / configs.SAMPLE_RATE
            err = max(err, float(np.abs(got[done] - ref).max())) if got[done].shape == ref.shape else float("nan")
            done += 1
        m.close()
        cpu = {"value": round(audio / tc, 3), "unit": "audio-s/s", "cores": threads, "kind": "port",
               "sample": f"{done} utterance(s) of the same workload ({args.phones} phones, {audio / done:.3f} s audio each), batch 1 looped like the "
                         f"reference, through oracle/sbv2_ref.c (C + OpenMP fp32, {os.path.basename(lib.path)}), {tc:.1f} s wall; a port, not onnxruntime",
               "wall_s": round(tc, 2), "host_hardware_threads": os.cpu_count(), "gpu_vs_oracle_max_abs": err}

    if rank == 0:
        out = {
            "metric": "audio_seconds_per_second", "value": round(value, 2), "unit": "audio-s/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "real_time_factor": round(dt / total_audio, 6),
            "config": {"workload": (f"batch={args.batch} x {args.phones}-phoneme utterances per GPU" if args.config == "u128" else
                                    "one global batch of 256 utterances of 32..512 phonemes (BASELINE configs[3]), cost-sorted deal over the ranks")
                                   + ", DeBERTa-v2-large(22L + ConvLayer) + Style-Bert-VITS2 JP-Extra + HiFi-GAN, forced durations"
                                   + (f" -> {int(b.lens[0]) // hop} frames/utt" if args.config == "u128" else "")
                                   + "; timed region = host ids -> PCM on the host of rank 0",
                       "global_batch": global_batch, "audio_seconds_per_step": round(total_samples_per_step / configs.SAMPLE_RATE, 3),
                       "parallelism": f"utterance-sharded x{world}" + ((", RCCL gather of PCM to rank 0 (in-library, no torch)" if comm is not None else
                                                                           f", NO RCCL communicator ({rccl_error or 'a rank failed'}): PCM left on each rank's host") if world > 1 else ""),
                       "rccl_ranks": (comm.world if comm is not None else (0 if world > 1 else 1)),
                       "shapes": "tiny (plumbing check)" if args.tiny else "full"},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if comm is not None:
        comm.barrier()
        comm.close()
    if host is not None:
        host.barrier()
        if rank == 0:
            try:
                os.remove(uid_path())
            except OSError:
                pass


if __name__ == "__main__":
    main()
