"""Host mirror of crates/sbv2_core/src/model.rs (`load_model`, `synthesize`) and bert.rs (`predict`) over the C ABI.

Same names, argument order and meaning as the reference; numpy arrays stand in for ndarray.  The batched /
pipeline entry points are the new capabilities (the reference is strictly batch 1).
"""
import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import Sbv2Batch, Sbv2Error, check, f32p, i64p

__all__ = ["Session", "load_model", "predict", "synthesize", "predict_batch", "synthesize_batch", "Pipeline", "Node", "Comm", "deal", "Sbv2Error"]


def _i64(a):
    a = np.ascontiguousarray(a, dtype=np.int64)
    return a, a.ctypes.data_as(i64p)


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(f32p)


class Session:
    """Opaque model handle; stands in for `ort::session::Session` (model.rs:6, tts.rs:32-46)."""

    def __init__(self, handle, bert):
        self.handle, self.bert = handle, bert

    def close(self):
        if self.handle:
            (_lib.lib().sbv2_bert_destroy if self.bert else _lib.lib().sbv2_vits_destroy)(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def load_model(model_file: bytes, bert: bool, device: int = 0) -> Session:
    """model.rs:6-50 `load_model(model_file, bert)`; the EP list / thread options have no meaning here."""
    l = _lib.lib()
    h = C.c_void_p()
    buf = (C.c_char * len(model_file)).from_buffer_copy(model_file) if not isinstance(model_file, (bytearray, memoryview)) else \
        (C.c_char * len(model_file)).from_buffer(model_file)
    fn = l.sbv2_bert_create if bert else l.sbv2_vits_create
    check(fn(C.cast(buf, C.c_void_p), len(model_file), device, C.byref(h)))
    return Session(h, bert)


def predict(session: Session, token_ids, attention_masks) -> np.ndarray:
    """bert.rs:6-24 `predict(session, token_ids, attention_masks) -> Array2<f32> [S, 1024]`."""
    ids, pi = _i64(token_ids)
    msk, pm = _i64(attention_masks)
    if ids.shape != msk.shape or ids.ndim != 1:
        raise Sbv2Error("token_ids and attention_masks must be 1-D and of equal length")
    l = _lib.lib()
    out = np.empty((ids.shape[0], l.sbv2_bert_hidden(session.handle)), np.float32)
    check(l.sbv2_bert_predict(session.handle, pi, pm, ids.shape[0], out.ctypes.data_as(f32p)))
    return out


def predict_batch(session: Session, token_ids_list, attention_masks_list=None):
    l = _lib.lib()
    lens = np.array([len(t) for t in token_ids_list], np.int64)
    ids, pi = _i64(np.concatenate([np.asarray(t, np.int64) for t in token_ids_list]))
    pm = None
    if attention_masks_list is not None:
        msk, pm = _i64(np.concatenate([np.asarray(t, np.int64) for t in attention_masks_list]))
    out = np.empty((int(lens.sum()), l.sbv2_bert_hidden(session.handle)), np.float32)
    check(l.sbv2_bert_predict_batch(session.handle, len(lens), pi, pm, lens.ctypes.data_as(i64p), out.ctypes.data_as(f32p)))
    return np.split(out, np.cumsum(lens)[:-1], axis=0)


def fresh_noise_seed() -> int:
    """A new 64-bit seed per call: the reference's graph draws fresh RandomNormalLike noise on every run (tts.rs:313-314 passes
    noise_scale 0.677 / noise_scale_w 0.8), so an unseeded request must not be bit-reproducible here either."""
    return int.from_bytes(os.urandom(8), "little")


def synthesize(session: Session, bert_ori, x_tst, spk_ids, tones, lang_ids, style_vector, sdp_ratio, length_scale, noise_scale,
               noise_scale_w, noise_seed: int | None = None) -> np.ndarray:
    """model.rs:53-111 `synthesize(...) -> Array3<f32> [1, 1, L]` (same argument order).  noise_seed None = fresh noise per call."""
    l = _lib.lib()
    if noise_seed is None:
        noise_seed = fresh_noise_seed()
    b, pb = _f32(bert_ori)
    x, px = _i64(x_tst)
    t, pt = _i64(tones)
    g, pg = _i64(lang_ids)
    s, ps = _f32(style_vector)
    T = x.shape[0]
    if b.shape != (l.sbv2_vits_bert_dim(session.handle), T) or t.shape != (T,) or g.shape != (T,):
        raise Sbv2Error("input shapes do not agree (bert [1024, T], x_tst / tones / lang_ids [T])")
    sid = int(np.asarray(spk_ids).reshape(-1)[0])
    pcm = f32p()
    n = C.c_int64()
    check(l.sbv2_vits_synthesize(session.handle, pb, px, pt, pg, T, sid, ps, sdp_ratio, length_scale, noise_scale, noise_scale_w,
                                 noise_seed, C.byref(pcm), C.byref(n)))
    try:
        out = np.ctypeslib.as_array(pcm, shape=(n.value,)).copy()
    finally:
        l.sbv2_pcm_free(pcm)
    return out.reshape(1, 1, -1)


class _Batch:
    """Keeps the numpy buffers of one sbv2_batch alive."""

    def __init__(self, utts, sdp_ratio, length_scale, noise_scale, noise_scale_w, noise_seed, forced, with_bert):
        self.keep = []
        cat = lambda key, dt: np.ascontiguousarray(np.concatenate([np.asarray(u[key]).reshape(-1) for u in utts]), dtype=dt)
        self.t_lens = np.array([len(u["phones"]) for u in utts], np.int64)
        self.x, self.tones, self.langs = cat("phones", np.int64), cat("tones", np.int64), cat("langs", np.int64)
        self.sids = np.array([int(u.get("sid", 0)) for u in utts], np.int64)
        self.styles = np.ascontiguousarray(np.stack([np.asarray(u["style"], np.float32) for u in utts]))
        self.bert = cat("bert", np.float32) if with_bert else None
        self.forced = cat("forced_durations", np.int64) if forced else None
        p = lambda a, t: a.ctypes.data_as(t) if a is not None else None
        self.c = Sbv2Batch(len(utts), p(self.t_lens, i64p), p(self.x, i64p), p(self.tones, i64p), p(self.langs, i64p), p(self.sids, i64p),
                           p(self.styles, f32p), p(self.bert, f32p), sdp_ratio, length_scale, noise_scale, noise_scale_w, noise_seed,
                           p(self.forced, i64p))


def synthesize_batch(session: Session, utts, sdp_ratio=0.0, length_scale=1.0, noise_scale=0.0, noise_scale_w=0.0, noise_seed=0,
                     forced=False, fetch=True):
    """New capability: a list of utterance dicts (bert [1024,T], phones, tones, langs, style, sid[, forced_durations])
    in one call.  Returns a list of PCM arrays (or the lengths when fetch=False: PCM stays in HBM)."""
    l = _lib.lib()
    b = _Batch(utts, sdp_ratio, length_scale, noise_scale, noise_scale_w, noise_seed, forced, True)
    lens = np.zeros(len(utts), np.int64)
    check(l.sbv2_vits_synthesize_batch(session.handle, C.byref(b.c), lens.ctypes.data_as(i64p)))
    if not fetch:
        return lens
    pcm = np.empty(int(lens.sum()), np.float32)
    check(l.sbv2_vits_fetch_pcm(session.handle, pcm.ctypes.data_as(f32p), pcm.size))
    return np.split(pcm, np.cumsum(lens)[:-1])


class PinnedArray:
    """float32 numpy view of page-locked host memory from the library (sbv2_host_alloc): the destination of overlapped D2H copies."""

    def __init__(self, n: int):
        self.ptr = _lib.lib().sbv2_host_alloc(4 * max(n, 1))
        if not self.ptr:
            raise Sbv2Error(_lib.lib().sbv2_last_error().decode(errors="replace"))
        self.array = np.ctypeslib.as_array(C.cast(self.ptr, f32p), shape=(n,))

    def close(self):
        if self.ptr:
            self.array = None
            _lib.lib().sbv2_host_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def fetch_durations(session: Session, total_t: int):
    d = np.zeros(total_t, np.int64)
    lw = np.zeros(total_t, np.float32)
    check(_lib.lib().sbv2_vits_fetch_durations(session.handle, d.ctypes.data_as(i64p), lw.ctypes.data_as(f32p), total_t))
    return d, lw


def set_trace(session: Session, on: bool):
    check(_lib.lib().sbv2_vits_set_trace(session.handle, int(on)))


def get_trace(session: Session, name: str, utt: int = 0) -> np.ndarray:
    l = _lib.lib()
    r, c = C.c_int64(), C.c_int64()
    check(l.sbv2_vits_get_trace(session.handle, name.encode(), utt, None, 0, C.byref(r), C.byref(c)))
    out = np.empty((r.value, c.value), np.float32)
    check(l.sbv2_vits_get_trace(session.handle, name.encode(), utt, out.ctypes.data_as(f32p), out.size, C.byref(r), C.byref(c)))
    return out


class Pipeline:
    """New capability: bert::predict -> word2ph feature repeat (tts_util.rs:129-154) -> model::synthesize for a batch,
    device resident between the stages."""

    def __init__(self, bert: Session, vits: Session):
        self.bert, self.vits = bert, vits
        self.h = C.c_void_p()
        check(_lib.lib().sbv2_pipeline_create(bert.handle, vits.handle, C.byref(self.h)))

    def prepare(self, utts, sdp_ratio=0.0, length_scale=1.0, noise_scale=0.0, noise_scale_w=0.0, noise_seed=0, forced=False):
        """Pack the host-side inputs once (outside any timed region)."""
        b = _Batch(utts, sdp_ratio, length_scale, noise_scale, noise_scale_w, noise_seed, forced, False)
        b.ids = np.ascontiguousarray(np.concatenate([np.asarray(u["input_ids"], np.int64) for u in utts]))
        b.s_lens = np.array([len(u["input_ids"]) for u in utts], np.int64)
        b.w2p = np.ascontiguousarray(np.concatenate([np.asarray(u["word2ph"], np.int64) for u in utts]))
        b.lens = np.zeros(len(utts), np.int64)
        return b

    def run(self, b):
        check(_lib.lib().sbv2_pipeline_run(self.h, C.byref(b.c), b.ids.ctypes.data_as(i64p), b.s_lens.ctypes.data_as(i64p),
                                           b.w2p.ctypes.data_as(i64p), b.lens.ctypes.data_as(i64p)))
        b.ticket = _lib.lib().sbv2_pipeline_last_ticket(self.h)   # identifies this run's results until `depth` further runs
        return b.lens

    def wait(self, ticket: int):
        check(_lib.lib().sbv2_pipeline_wait(self.h, ticket))

    def fetch_ticket_to_device(self, ticket: int, device_ptr: int, capacity: int):
        check(_lib.lib().sbv2_pipeline_fetch_pcm_ticket(self.h, ticket, C.c_void_p(device_ptr), capacity, 1))

    def sync(self):
        check(_lib.lib().sbv2_pipeline_sync(self.h))

    def fetch(self, b, out=None):
        """PCM of the run that `b` was last submitted as (by ticket: a later run of another batch does not change what this returns;
        once the pipeline has reused that run's context the ticket is stale and the library raises).  `out`: optional preallocated
        float32 array (e.g. a view of pinned memory from `pinned_array`) of at least sum(b.lens) samples."""
        n = int(b.lens.sum())
        pcm = np.empty(n, np.float32) if out is None else out
        check(_lib.lib().sbv2_pipeline_fetch_pcm_ticket(self.h, b.ticket, pcm.ctypes.data_as(C.c_void_p), pcm.size, 0))
        return np.split(pcm[:n], np.cumsum(b.lens)[:-1])

    def close(self):
        if self.h:
            _lib.lib().sbv2_pipeline_destroy(self.h)
            self.h = None


def stream_synthesize(bert: Session, vits: Session, utt, chunk_frames=256, **kw):
    """Generator over the PCM chunks of ONE long utterance (BASELINE configs[4]): whole-sequence DeBERTa / text / flow, then the HiFi-GAN
    decoder chunk by chunk through a captured hipGraph.  Yields float32 arrays; `.info` of the generator's first item is not needed:
    use stream_open for the handle-level interface."""
    st = StreamHandle(bert, vits, utt, chunk_frames, **kw)
    try:
        while True:
            c = st.next()
            if c is None:
                return
            yield c
    finally:
        st.close()


class StreamHandle:
    def __init__(self, bert: Session, vits: Session, utt, chunk_frames=256, **kw):
        l = _lib.lib()
        self.b = Pipeline.prepare(None, [utt], **kw)
        self.h = C.c_void_p()
        tot = C.c_int64()
        check(l.sbv2_stream_begin(bert.handle, vits.handle, C.byref(self.b.c), self.b.ids.ctypes.data_as(i64p), self.b.s_lens.ctypes.data_as(i64p),
                                  self.b.w2p.ctypes.data_as(i64p), chunk_frames, C.byref(self.h), C.byref(tot)))
        self.total_samples = tot.value
        self.buf = np.empty(chunk_frames * l.sbv2_vits_hop(vits.handle), np.float32)
        self.uses_graph = bool(l.sbv2_stream_uses_graph(self.h))
        self.workspace_bytes = l.sbv2_stream_workspace_bytes(self.h)

    def next(self):
        n = C.c_int64()
        check(_lib.lib().sbv2_stream_next(self.h, self.buf.ctypes.data_as(C.c_void_p), self.buf.size, C.byref(n)))
        return None if n.value == 0 else self.buf[:n.value].copy()

    def close(self):
        if self.h:
            _lib.lib().sbv2_stream_end(self.h)
            self.h = None


def deal(costs, world: int) -> np.ndarray:
    """rank_of[i] for every utterance: the library's longest-processing-time-first deal (csrc/node.cpp; host only)."""
    c = np.ascontiguousarray(costs, np.int64)
    out = np.zeros(len(c), np.int32)
    check(_lib.lib().sbv2_deal(len(c), c.ctypes.data_as(i64p), world, out.ctypes.data_as(C.POINTER(C.c_int32))))
    return out


class Node:
    """One process, N devices: utterance-sharded synthesis with the PCM gathered to device 0 inside the library."""

    def __init__(self, bert_bytes: bytes, vits_bytes: bytes, devices):
        self.h = C.c_void_p()
        dv = (C.c_int * len(devices))(*devices)
        bb = (C.c_char * len(bert_bytes)).from_buffer_copy(bert_bytes)
        vb = (C.c_char * len(vits_bytes)).from_buffer_copy(vits_bytes)
        check(_lib.lib().sbv2_node_create(C.cast(bb, C.c_void_p), len(bert_bytes), C.cast(vb, C.c_void_p), len(vits_bytes), dv, len(devices),
                                          C.byref(self.h)))

    def prepare(self, utts, **kw):
        return Pipeline.prepare(self, utts, **kw)

    def synthesize(self, b, out=None):
        """Runs the prepared batch; returns the list of PCM arrays in the caller's utterance order."""
        l = _lib.lib()
        grow = out is None
        if grow:
            # capacity: exact when durations are forced (hop = 512 for every JP-Extra checkpoint; a larger hop shows up as a capacity error and
            # is retried below); with predicted durations ~8 frames per text symbol, grown on demand (the library refuses, it never overflows)
            cap = int(b.forced.sum()) * 512 if b.forced is not None else int(b.t_lens.sum()) * 512 * 8
            out = np.empty(max(cap, 1), np.float32)
        for _ in range(4):
            rc = l.sbv2_node_synthesize(self.h, C.byref(b.c), b.ids.ctypes.data_as(i64p), b.s_lens.ctypes.data_as(i64p), b.w2p.ctypes.data_as(i64p),
                                        b.lens.ctypes.data_as(i64p), out.ctypes.data_as(C.c_void_p), out.size)
            if rc == 0 or not grow or b"too small" not in l.sbv2_last_error():
                break
            out = np.empty(out.size * 4, np.float32)     # the shards are synthesised again: a rare path (unusually slow speech)
        check(rc)
        n = int(b.lens.sum())
        return np.split(out[:n], np.cumsum(b.lens)[:-1])

    def last_deal(self, n: int) -> np.ndarray:
        r = np.zeros(n, np.int32)
        check(_lib.lib().sbv2_node_last_deal(self.h, r.ctypes.data_as(C.POINTER(C.c_int32)), n))
        return r

    @property
    def uses_rccl(self) -> bool:
        return bool(_lib.lib().sbv2_node_uses_rccl(self.h))

    def close(self):
        if self.h:
            _lib.lib().sbv2_node_destroy(self.h)
            self.h = None


class Comm:
    """One process per GPU: the RCCL communicator of the library (no torch).  Rank 0 creates the id, the launcher distributes it."""

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        check(_lib.lib().sbv2_comm_unique_id(buf))
        return buf.raw

    def __init__(self, uid: bytes, rank: int, world: int, device: int):
        self.h = C.c_void_p()
        self.rank, self.world = rank, world
        check(_lib.lib().sbv2_comm_create(uid, rank, world, device, C.byref(self.h)))

    def barrier(self):
        check(_lib.lib().sbv2_comm_barrier(self.h))

    def max(self, v: float) -> float:
        d = C.c_double(v)
        check(_lib.lib().sbv2_comm_max_f64(self.h, C.byref(d)))
        return d.value

    def gather_pcm(self, pipe: Pipeline, ticket: int, dst: np.ndarray | None, root: int = 0) -> np.ndarray:
        """counts[world]; on the root `dst` (float32, possibly a PinnedArray view) receives the ranks' PCM in rank order."""
        counts = np.zeros(self.world, np.int64)
        check(_lib.lib().sbv2_comm_gather_pcm(self.h, pipe.h, ticket, root, dst.ctypes.data_as(C.c_void_p) if dst is not None else None,
                                              dst.size if dst is not None else 0, counts.ctypes.data_as(i64p)))
        return counts

    def close(self):
        if self.h:
            _lib.lib().sbv2_comm_destroy(self.h)
            self.h = None
