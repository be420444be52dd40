"""Run by tests/test_gpu_parity.py::test_comm_world1_gather_through_rccl in a fresh interpreter (GPU box): the library's RCCL communicator
with world size 1: dlopen, ncclGetUniqueId, ncclCommInitRank, all-reduce, all-gather, self send/recv-free gather, capacity check."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from helpers import blob, make_utts, weights  # noqa: E402
from sbv2_api_amd import model  # noqa: E402

bc, bw = weights("bert", "tiny", 3)
vc, vw = weights("vits", "tiny", 5)
bs, vs = model.load_model(blob("bert", "tiny", 3), True), model.load_model(blob("vits", "tiny", 5), False)
pipe = model.Pipeline(bs, vs)
comm = model.Comm(model.Comm.unique_id(), 0, 1, 0)
assert comm.max(3.5) == 3.5
comm.barrier()
utts = make_utts([7, 15, 4], bc, vc, seed0=161, with_bert=False)
b = pipe.prepare(utts, forced=True)
pipe.run(b)
ref = np.concatenate(pipe.fetch(b))
pin = model.PinnedArray(ref.size)
counts = comm.gather_pcm(pipe, b.ticket, pin.array)
assert counts.tolist() == [ref.size], counts
np.testing.assert_array_equal(pin.array, ref)
try:
    comm.gather_pcm(pipe, b.ticket, np.empty(5, np.float32))
    raise SystemExit("capacity check did not fire")
except model.Sbv2Error as e:
    assert "too small" in str(e)
pin.close(); comm.close(); pipe.close(); bs.close(); vs.close()
print("RCCL_WORLD1_OK")
