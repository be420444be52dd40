"""Run by tests/test_gpu_parity.py::test_comm_two_ranks_gather_over_rccl as TWO fresh interpreters (needs >= 2 GPUs): rank r drives GPU r,
the ranks share an ncclUniqueId through a file, every rank synthesises ITS shard of one dealt batch (the library's deal) and
sbv2_comm_gather_pcm brings the PCM to rank 0 over RCCL (ncclSend / ncclRecv between distinct devices), which checks every utterance
against a single-GPU run of the whole batch, bit for bit.
  usage: rccl_world2_check.py RANK WORLD UID_FILE"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from helpers import blob, make_utts, weights  # noqa: E402
from sbv2_api_amd import model  # noqa: E402

rank, world, uid_file = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
bc, bw = weights("bert", "tiny", 3)
vc, vw = weights("vits", "tiny", 5)
bs = model.load_model(blob("bert", "tiny", 3), True, device=rank)
vs = model.load_model(blob("vits", "tiny", 5), False, device=rank)
pipe = model.Pipeline(bs, vs)
if rank == 0:
    uid = model.Comm.unique_id()
    with open(uid_file + ".tmp", "wb") as f:
        f.write(uid)
    os.replace(uid_file + ".tmp", uid_file)
else:
    t0 = time.time()
    while not os.path.exists(uid_file):
        if time.time() - t0 > 120:
            raise SystemExit("rank 0 never published the communicator id")
        time.sleep(0.05)
    uid = open(uid_file, "rb").read()
comm = model.Comm(uid, rank, world, rank)
assert comm.max(float(rank)) == float(world - 1)
utts = make_utts([7, 15, 4, 9, 21, 5, 12], bc, vc, seed0=161, with_bert=False)
kw = dict(sdp_ratio=0.25)     # predicted durations, noise off: a shard-local call then equals the rows of a whole-batch call
costs = [3 * u["T_text"] for u in utts]
rank_of = model.deal(costs, world)
mine = [i for i in range(len(utts)) if rank_of[i] == rank]
b = pipe.prepare([utts[i] for i in mine], **kw)
pipe.run(b)
if rank == 0:
    pin = model.PinnedArray(1 << 22)
    counts = comm.gather_pcm(pipe, b.ticket, pin.array)
    # single-GPU run of the whole batch on this rank's GPU
    bw_ = pipe.prepare(utts, **kw)
    pipe.run(bw_)
    whole = pipe.fetch(bw_)
    off = 0
    for r in range(world):
        for i in [i for i in range(len(utts)) if rank_of[i] == r]:
            got = pin.array[off:off + whole[i].size]
            np.testing.assert_array_equal(got, whole[i])
            off += whole[i].size
        assert off == int(counts[:r + 1].sum())
    pin.close()
else:
    comm.gather_pcm(pipe, b.ticket, None)
comm.barrier()
comm.close(); pipe.close(); bs.close(); vs.close()
print(f"RCCL_WORLD{world}_RANK{rank}_OK")
