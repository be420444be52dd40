"""CPU-only checks of the host side: the C-ABI library loads and exports every symbol include/sbv2_hip.h declares,
host-only entry points work without a GPU, product code never imports the oracle, and the N > 1 sharding + PCM gather
works with world_size 2 over gloo."""
import ctypes as C
import os
import re
import socket

import numpy as np
import pytest

import gather_twin as shard
from sbv2_api_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "sbv2_hip.h")).read()
    declared = set(re.findall(r"\b(sbv2_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    l = _lib.lib()
    for name in declared:
        assert hasattr(l, name), f"{name} declared in sbv2_hip.h but not exported"
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)


def test_bucket_table_matches_transformers(golden_dir):
    z = np.load(os.path.join(golden_dir, "deberta_buckets.npz"))
    for key in z.files:
        S, b, m = (int(t[1:]) for t in key.split("_"))
        out = np.zeros(2 * S - 1, np.int32)
        _lib.check(_lib.lib().sbv2_debug_bucket_table(S, b, m, out.ctypes.data_as(C.POINTER(C.c_int32))))
        g = z[key]   # g[i, j] = bucket(i - j)
        ref = np.array([g[0, S - 1 - r] if r < S else g[r - S + 1, 0] for r in range(2 * S - 1)])  # rel = r - (S-1)
        assert np.array_equal(out, ref), key


def test_errors_are_reported_not_thrown():
    l = _lib.lib()
    h = C.c_void_p()
    junk = b"definitely not a model"
    rc = l.sbv2_vits_create(C.cast(C.create_string_buffer(junk), C.c_void_p), len(junk), 0, C.byref(h))
    assert rc != 0 and b"SBV2W001" in l.sbv2_last_error()


def test_product_code_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "sbv2-api_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(dp, f), errors="replace").read()
                assert "import sbv2_oracle" not in src and "from oracle" not in src, f


def test_deal_is_balanced_and_complete():
    costs = [7 * n + 1 for n in (512, 32, 300, 128, 64, 480, 33, 256, 100, 77, 400, 50)]
    for world in (1, 2, 4, 8):
        sh = shard.deal(costs, world)
        assert sorted(i for s in sh for i in s) == list(range(len(costs)))
        loads = [sum(costs[i] for i in s) for s in sh]
        assert max(loads) - min(loads) <= max(costs)


def test_library_deal_matches_host_deal():
    """sbv2_deal (C ABI, csrc/node.cpp) == shard.deal: complete, deterministic, balanced to one utterance's cost, skewed counts allowed."""
    from sbv2_api_amd import model
    rng = np.random.default_rng(0)
    cases = [[7 * n + 1 for n in rng.integers(32, 513, 256)], [512, 32, 32, 32, 32, 40], [5], [3, 3, 3, 3], []]
    for costs in cases:
        for world in (1, 2, 8):
            r = model.deal(costs, world)
            ref = shard.deal(costs, world)
            got = [[i for i in range(len(costs)) if r[i] == q] for q in range(world)]
            assert [sorted(s) for s in ref] == got
            if costs:
                loads = [sum(costs[i] for i in s) for s in got]
                assert max(loads) - min(loads) <= max(costs)


_SKEWED = [512, 32, 32, 32, 32, 40]     # deals 1 + 5 utterances on two ranks: more than ceil(6 / 2) on one of them


def _worker(rank, world, port, q, lens=(5, 17, 1, 9, 12)):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sbv2_api_amd import model
    lens = list(lens)
    out = shard.gather_by_plan(model, lens, world, rank, dist, lambda i: np.arange(lens[i], dtype=np.float32) + 100 * i)
    if rank == 0:
        q.put([o.tolist() for o in out])
    dist.destroy_process_group()


def test_gather_pcm_world2_gloo():
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    got = q.get(timeout=120)
    [p.join(60) for p in ps]
    lens = [5, 17, 1, 9, 12]
    for i, n in enumerate(lens):
        assert got[i] == (np.arange(n, dtype=np.float32) + 100 * i).tolist()


def test_gather_pcm_world2_gloo_skewed_costs():
    """A deal whose largest shard exceeds ceil(n / world) utterances (the mixed 32..512 case of BASELINE configs[3])."""
    import torch.multiprocessing as mp
    assert sorted(len(s) for s in shard.deal(_SKEWED, 2)) == [1, 5]
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q, tuple(_SKEWED))) for r in range(2)]
    [p.start() for p in ps]
    got = q.get(timeout=120)
    [p.join(60) for p in ps]
    for i, n in enumerate(_SKEWED):
        assert got[i] == (np.arange(n, dtype=np.float32) + 100 * i).tolist()


def test_package_configs_match_oracle():
    import sbv2_oracle as O
    from sbv2_api_amd import configs
    for name in ("DEBERTA_FULL", "DEBERTA_TINY", "DEBERTA_TINY_CONV", "VITS_FULL", "VITS_TINY", "SAMPLE_RATE"):
        assert getattr(configs, name) == getattr(O, name), name


def test_style_vector_blend_and_json():
    """style.rs:11-28 mirror vs the oracle restatement."""
    import json
    import orchestrator_oracle as OO
    from sbv2_api_amd import orchestrator as orch, model
    rng = np.random.default_rng(3)
    sv = rng.standard_normal((4, 256)).astype(np.float32)
    blob = json.dumps({"shape": [4, 256], "data": [[float(v) for v in row] for row in sv]}).encode()
    got = orch.load_style(blob)
    assert got.dtype == np.float32 and np.array_equal(got, sv)
    for sid, w in ((0, 1.0), (2, 1.0), (3, 0.35), (1, 2.5)):
        np.testing.assert_array_equal(orch.get_style_vector(got, sid, w), OO.get_style_vector(sv, sid, w))
    np.testing.assert_array_equal(orch.get_style_vector(got, 0, 7.0), sv[0])
    with pytest.raises(model.Sbv2Error):
        orch.load_style(json.dumps({"shape": [4, 256], "data": [[0.0] * 255] * 4}).encode())
    with pytest.raises(IndexError):
        orch.get_style_vector(got, 4, 1.0)


def test_wav_encoding_reads_back():
    """tts_util.rs:163-180 mirror: same bytes as the oracle restatement; an independent reader (scipy) sees 44.1 kHz mono f32
    with the samples of all batch rows back to back."""
    import io
    import orchestrator_oracle as OO
    from scipy.io import wavfile
    from sbv2_api_amd import orchestrator as orch
    rng = np.random.default_rng(4)
    audio = np.tanh(rng.standard_normal((2, 1, 1000))).astype(np.float32)
    wav = orch.array_to_wav(audio)
    assert wav == OO.array_to_wav(audio)
    assert len(wav) == 68 + 4 * 2000 and wav[:4] == b"RIFF" and int.from_bytes(wav[4:8], "little") == len(wav) - 8
    rate, data = wavfile.read(io.BytesIO(wav))
    assert rate == 44100 and data.dtype == np.float32 and data.ndim == 1
    np.testing.assert_array_equal(data, audio.reshape(-1))
    assert orch.array_to_wav(np.zeros((1, 1, 0), np.float32))[-4:] == (0).to_bytes(4, "little")


def test_cpp_host_mirror_builds_and_reports_errors(tmp_path):
    """The C++ mirror of the reference interface (include/sbv2_core.hpp) compiles with plain g++ against the C ABI; without model
    files its driver fails cleanly (exit code 1 = exception caught and printed, after the bad-container check passed), never crashes."""
    import subprocess
    exe = os.path.join(ROOT, "tests", "cpp", "sbv2_core_demo")
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", os.path.join(ROOT, "sbv2-api_amd", "csrc")], check=True, capture_output=True)
    r = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "cannot open" in r.stderr, (r.returncode, r.stderr)


class _FakeSession:
    live = 0

    def __init__(self, data, is_bert):
        self.data, self.is_bert, self.closed = data, is_bert, False
        _FakeSession.live += 0 if is_bert else 1

    def close(self):
        if not self.closed and not self.is_bert:
            _FakeSession.live -= 1
        self.closed = True


class _FakePipe:
    def __init__(self, bert, vits):
        self.vits = vits

    def close(self):
        pass


def _holder(max_loaded):
    from sbv2_api_amd import holder
    _FakeSession.live = 0
    return holder.TTSModelHolder(b"bert", max_loaded_models=max_loaded, load_session=_FakeSession, make_pipeline=_FakePipe)


def _style(n=2, dim=4):
    import json
    return json.dumps({"shape": [n, dim], "data": [[float(i + j) for j in range(dim)] for i in range(n)]}).encode()


def test_model_holder_cache_semantics():
    """tts.rs:149-258 mirrored: duplicate idents ignored, sessions created only below max_loaded_models, raw bytes kept iff the limit is set,
    unload removes the entry, a non-resident model is rebuilt on demand and, when the cache is full, the FIRST entry of the list is dropped
    from the holder (the reference's behaviour, not an LRU).  'Resident' here = weights in HBM (fake sessions count themselves)."""
    from sbv2_api_amd import holder
    h = _holder(None)
    h.load("a", _style(), b"A"); h.load("b", _style(), b"B"); h.load("a", _style(), b"A2")
    assert h.models() == ["a", "b"] and _FakeSession.live == 2 and all(m.bytes is None for m in h.models_)
    assert h.unload("a") and not h.unload("a") and h.models() == ["b"] and _FakeSession.live == 1
    with pytest.raises(holder.ModelNotFoundError):
        h.find_and_load_model("zzz")
    h = _holder(2)
    for k in "abc":
        h.load(k, _style(), k.encode())
    assert h.models() == ["a", "b", "c"] and [m.vits2 is not None for m in h.models_] == [True, True, False] and _FakeSession.live == 2
    assert all(m.bytes is not None for m in h.models_)
    assert h.find_and_load_model("a") and h.models() == ["a", "b", "c"]          # already resident: nothing moves
    assert h.find_and_load_model("c")
    # c was rebuilt from its kept bytes; the cache was full, so the first entry ("a") left the holder altogether
    assert h.models() == ["b", "c"] and _FakeSession.live == 2 and h._find("c").vits2.data == b"c"
    np.testing.assert_array_equal(h.get_style_vector("b", 1, 0.5), np.array([0.5, 1.5, 2.5, 3.5], np.float32))
    h.close()
    assert _FakeSession.live == 0


def test_model_holder_load_aivmx():
    """tts.rs:77-130: an .aivmx entry gets its style table from the ONNX metadata and follows the same cache rules as load()."""
    import onnx_writer as OW
    sv = np.arange(12, dtype=np.float32).reshape(3, 4)
    aivmx = OW.model_proto([], [], {"aivm_style_vectors": OW.aivm_style_vectors(sv, fortran=True)})
    h = _holder(1)
    h.load_aivmx("x", aivmx); h.load_aivmx("y", aivmx); h.load_aivmx("x", b"ignored: the ident exists")
    assert h.models() == ["x", "y"] and [m.vits2 is not None for m in h.models_] == [True, False] and _FakeSession.live == 1
    assert h._find("x").vits2.data == aivmx and h._find("y").bytes == aivmx
    np.testing.assert_array_equal(h.get_style_vector("y", 2, 1.0), sv[2])
    np.testing.assert_array_equal(h.get_style_vector("y", 2, 0.5), sv[0] + (sv[2] - sv[0]) * 0.5)
    h.close()


def test_bench_host_group_world3(tmp_path):
    """bench.py's fallback group (TCP on 127.0.0.1: the ranks agree on whether RCCL came up, and exchange the two scalars of the bench line
    if it did not): three processes, max / min / sum / barrier."""
    import subprocess, sys, os, socket
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    code = ("import sys, os; sys.path.insert(0, %r); import bench; r = int(os.environ['RANK']); g = bench.HostGroup(r, 3); "
            "print(g.max(r + 1.5), g.min(r + 1.5), g.sum(r + 1.0)); g.barrier(); print('done')") % root
    procs = [subprocess.Popen([sys.executable, "-c", code], env=dict(os.environ, RANK=str(r), WORLD_SIZE="3", MASTER_PORT=str(port),
                                                                     SBV2_BENCH_LAUNCH="cputest", TMPDIR=str(tmp_path)),
                              stdout=subprocess.PIPE, text=True) for r in range(3)]
    outs = [p.communicate(timeout=120)[0].split() for p in procs]
    assert all(p.returncode == 0 for p in procs)
    for o in outs:
        assert o == ["3.5", "1.5", "6.0", "done"], o


def test_rest_contract():
    """main.rs:24-100,192-196 + error.rs:10-18: routes, JSON defaults, audio/wav, 'Something went wrong: ...' with status 500."""
    from fastapi.testclient import TestClient
    from sbv2_api_amd import orchestrator, rest

    class H:
        def __init__(self):
            self.calls = []

        def models(self):
            return ["tsukuyomi"]

        def easy_synthesize(self, ident, text, style_id, speaker_id, options):
            if ident != "tsukuyomi":
                raise RuntimeError(f"model not found: {ident}")
            self.calls.append((text, style_id, speaker_id, options.sdp_ratio, options.length_scale))
            return orchestrator.array_to_wav(np.zeros((1, 1, 10), np.float32))

    h = H()
    c = TestClient(rest.make_app(h), raise_server_exceptions=False)
    assert c.get("/").text == "Hello, World!"
    assert c.get("/models").json() == ["tsukuyomi"]
    r = c.post("/synthesize", json={"text": "こんにちは", "ident": "tsukuyomi"})
    assert r.status_code == 200 and r.headers["content-type"] == "audio/wav" and r.content[:4] == b"RIFF" and len(r.content) == 68 + 40
    assert h.calls[-1] == ("こんにちは", 0, 0, 0.0, 1.0)                              # the reference's serde defaults
    c.post("/synthesize", json={"text": "a\nb", "ident": "tsukuyomi", "sdp_ratio": 0.4, "length_scale": 1.2, "style_id": 3, "speaker_id": 1})
    assert h.calls[-1] == ("a\nb", 3, 1, 0.4, 1.2)
    r = c.post("/synthesize", json={"text": "x", "ident": "nope"})
    assert r.status_code == 500 and r.text == "Something went wrong: model not found: nope"
    assert c.post("/synthesize", json={"ident": "tsukuyomi"}).status_code == 422        # missing field (axum answers 422 as well)
