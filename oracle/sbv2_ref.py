"""ctypes binding of oracle/sbv2_ref.c (TEST INFRASTRUCTURE / CPU baseline: never imported by the product package).

`load()` returns the prebuilt oracle/liboracle_ref.so (built by `make -C oracle`, i.e. __graft_entry__.build()); `load(native=True)` first
tries to rebuild the same source with -march=native into a temp dir (bench.py's cpu_baseline leg on the GPU box: the host there need not be
the build container's CPU) and falls back to the prebuilt library."""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# libgomp's default is to spin at barriers; on a box whose cgroup grants fewer CPUs than it shows that turns into a livelock
os.environ.setdefault("OMP_WAIT_POLICY", "passive")


def usable_cpus() -> int:
    """CPUs this process may really use: min(scheduler affinity, cgroup CPU quota).  (The GPU boxes of the pool show 256 hardware threads
    but grant a container 16 CPUs of time through cgroup v2 cpu.max; 256 OpenMP threads on that are ~100x slower than 16.)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, -(-int(q) // int(per))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, -(-q // per)))
        except (OSError, ValueError):
            pass
    return n
_f32p, _i64p = C.POINTER(C.c_float), C.POINTER(C.c_int64)
_libs = {}


def _bind(path):
    l = C.CDLL(path)
    l.sbv2c_last_error.restype = C.c_char_p
    l.sbv2c_load.restype = C.c_void_p
    l.sbv2c_load.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    l.sbv2c_free.argtypes = [C.c_void_p]
    l.sbv2c_set_threads.argtypes = [C.c_int]
    l.sbv2c_bert.argtypes = [C.c_void_p, _i64p, _i64p, C.c_int, _f32p]
    l.sbv2c_vits.argtypes = [C.c_void_p, _f32p, _i64p, _i64p, _i64p, C.c_int, C.c_int, _f32p, C.c_float, C.c_float, _f32p, _i64p,
                             C.POINTER(_f32p), _i64p, _i64p, _f32p, _f32p, _f32p, C.c_int64]
    l.sbv2c_free_pcm.argtypes = [_f32p]
    l.sbv2c_conv1d_same.argtypes = [_f32p, C.c_int, C.c_int64, _f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_float, _f32p]
    l.sbv2c_conv_transpose1d.argtypes = [_f32p, C.c_int, C.c_int64, _f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _f32p]
    return l


def load(native=False, path=None):
    key = path or ("native" if native else "prebuilt")
    if key in _libs:
        return _libs[key]
    if path is None:
        path = os.path.join(HERE, "liboracle_ref.so")
        if native:
            try:
                out = os.path.join(tempfile.mkdtemp(prefix="sbv2ref_"), "liboracle_ref_native.so")
                subprocess.run(["gcc", "-O3", "-march=native", "-fopenmp", "-fPIC", "-std=gnu11", "-fno-math-errno", "-shared", "-o", out,
                                os.path.join(HERE, "sbv2_ref.c"), "-lm"], check=True, capture_output=True, timeout=300)
                path = out
            except Exception:
                pass
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: run `make -C oracle` (__graft_entry__.build() does)")
    _libs[key] = _bind(path)
    _libs[key].path = path
    return _libs[key]


def _p(a, t):
    return None if a is None else a.ctypes.data_as(t)


class Model:
    def __init__(self, bert_blob=None, vits_blob=None, lib=None):
        self.l = lib or load()
        bb = (C.c_char * len(bert_blob)).from_buffer_copy(bert_blob) if bert_blob else None
        vb = (C.c_char * len(vits_blob)).from_buffer_copy(vits_blob) if vits_blob else None
        self.h = self.l.sbv2c_load(C.cast(bb, C.c_void_p) if bb else None, len(bert_blob) if bert_blob else 0,
                                   C.cast(vb, C.c_void_p) if vb else None, len(vits_blob) if vits_blob else 0)
        if not self.h:
            raise RuntimeError(self.l.sbv2c_last_error().decode())

    def close(self):
        if self.h:
            self.l.sbv2c_free(self.h)
            self.h = None

    def bert(self, ids, mask=None, hidden=None):
        ids = np.ascontiguousarray(ids, np.int64)
        m = None if mask is None else np.ascontiguousarray(mask, np.int64)
        out = np.empty((len(ids), hidden), np.float32)
        if self.l.sbv2c_bert(self.h, _p(ids, _i64p), _p(m, _i64p), len(ids), _p(out, _f32p)):
            raise RuntimeError(self.l.sbv2c_last_error().decode())
        return out

    def vits(self, bert, phones, tones, langs, sid, style, sdp_ratio=0.0, length_scale=1.0, noise_w=None, forced_durations=None,
             hidden=None, inter=None, return_all=False):
        bert = np.ascontiguousarray(bert, np.float32)
        ph, tn, lg = (np.ascontiguousarray(a, np.int64) for a in (phones, tones, langs))
        T = len(ph)
        st = np.ascontiguousarray(style, np.float32)
        nw = None if noise_w is None else np.ascontiguousarray(noise_w, np.float32)
        fd = None if forced_durations is None else np.ascontiguousarray(forced_durations, np.int64)
        pcm, n = _f32p(), C.c_int64()
        dur, lw = np.zeros(T, np.int64), np.zeros(T, np.float32)
        x = np.zeros((hidden, T), np.float32) if (return_all and hidden) else None
        zcap = 0
        z = None
        if return_all and inter:
            tot = int(fd.sum()) if fd is not None else 1 << 16
            zcap = inter * max(tot, 1)
            z = np.zeros(zcap, np.float32)
        rc = self.l.sbv2c_vits(self.h, _p(bert, _f32p), _p(ph, _i64p), _p(tn, _i64p), _p(lg, _i64p), T, int(sid), _p(st, _f32p), sdp_ratio,
                               length_scale, _p(nw, _f32p), _p(fd, _i64p), C.byref(pcm), C.byref(n), _p(dur, _i64p), _p(lw, _f32p),
                               _p(x, _f32p), _p(z, _f32p), zcap)
        if rc:
            raise RuntimeError(self.l.sbv2c_last_error().decode())
        try:
            out = np.ctypeslib.as_array(pcm, shape=(n.value,)).copy()
        finally:
            self.l.sbv2c_free_pcm(pcm)
        if return_all:
            return dict(pcm=out, durations=dur, logw=lw, x=x, z=z)
        return out
