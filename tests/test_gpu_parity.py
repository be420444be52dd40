"""Parity of the HIP path (through the C ABI) against the CPU oracle and the transformers golden fixtures.
Run on the GPU box:  python -m pytest tests -m gpu -x -q

Tolerances: north_star asks for 1e-3 max-abs on the fp32 waveform given identical integer durations; the f32 MFMA
path is held to much tighter bounds here (stated per assertion)."""
import ast
import ctypes as C
import os

import numpy as np
import pytest

import sbv2_oracle as O
from helpers import blob, make_utts, oracle_noise_w, oracle_noise_z, weights
from sbv2_api_amd import _lib, model, synth

pytestmark = pytest.mark.gpu
f32p = _lib.f32p


def _conv_dev(x, w, b, dil, slope=1.0):
    cout, cin, k = w.shape
    y = np.empty((cout, x.shape[1]), np.float32)
    P = lambda a: np.ascontiguousarray(a, np.float32).ctypes.data_as(f32p)
    xs, ws, bs = np.ascontiguousarray(x, np.float32), np.ascontiguousarray(w, np.float32), np.ascontiguousarray(b, np.float32)
    _lib.check(_lib.lib().sbv2_debug_conv1d(0, P(xs), P(ws), P(bs), cin, cout, k, x.shape[1], dil, slope, y.ctypes.data_as(f32p)))
    return y


@pytest.mark.parametrize("cin,cout,k,dil,L", [
    (16, 16, 11, 5, 3000), (32, 32, 7, 3, 1500), (64, 64, 3, 1, 777), (128, 128, 11, 1, 1030), (256, 256, 7, 5, 515),
    (192, 768, 3, 1, 257), (768, 192, 5, 1, 897), (192, 29, 1, 1, 300), (256, 1, 1, 1, 100), (1, 24, 1, 1, 50),
    (96, 192, 1, 1, 4), (40, 72, 3, 1, 1), (1024, 192, 1, 1, 130), (16, 16, 3, 3, 70000),
])
def test_conv1d_kernel(cin, cout, k, dil, L):
    rng = np.random.default_rng(cin * 1000 + cout + k)
    x = rng.standard_normal((cin, L)).astype(np.float32)
    w = (rng.standard_normal((cout, cin, k)) / np.sqrt(cin * k)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    for slope in (1.0, 0.1):
        ref = O.conv1d_same(O.leaky_relu(x, slope), w, b, dil)
        got = _conv_dev(x, w, b, dil, slope)
        np.testing.assert_allclose(got, ref, atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize("C,k,dil,L", [(128, 3, 1, 700), (128, 7, 3, 1500), (128, 11, 5, 1030), (256, 7, 1, 515), (256, 11, 3, 300), (256, 3, 5, 257),
                                       (128, 7, 5, 256), (128, 11, 1, 8200)])
def test_conv1d_clx_kernel_agrees_with_conv_cl(C, k, dil, L):
    """conv_clx.hip (pre-split operands, LDS-DMA rings, v_mfma_f32_16x16x32_bf16 with both cross terms of the split in one instruction) against conv_cl.hip's
    split-bf16 path (32x32x16, lo*hi, hi*lo, hi*hi per tap): the same products in another summation order, so f32-rounding apart (1e-5), not bit for bit
    (rounds 3-4 ran both on one MFMA shape and held them bit-equal; round 5 moved conv_clx to the shape the chip sustains at a higher clock).  Also with a
    residual and beta, and the bf16 parts of lrelu(result) it emits for the next convolution are the split of its own f32 result."""
    rng = np.random.default_rng(C + k + dil + L)
    x = rng.standard_normal((C, L)).astype(np.float32)
    w = (rng.standard_normal((C, C, k)) / np.sqrt(C * k)).astype(np.float32)
    b = rng.standard_normal(C).astype(np.float32)
    r = rng.standard_normal((C, L)).astype(np.float32)
    P = lambda a: None if a is None else a.ctypes.data_as(f32p)
    lib = _lib.lib()
    ref = np.empty((C, L), np.float32)
    _lib.check(lib.sbv2_debug_conv1d_cl(0, P(x), P(w), P(b), C, C, k, L, dil, 0.1, 1, 0, P(ref), None))
    got, ys = np.empty((C, L), np.float32), np.empty((C, L), np.float32)
    _lib.check(lib.sbv2_debug_conv1d_clx(0, P(x), P(w), P(b), None, C, C, k, L, dil, 0.1, 1.0, 0, P(got), P(ys), None))
    np.testing.assert_allclose(got, ref, atol=1e-5, rtol=1e-5)
    np.testing.assert_allclose(got, O.conv1d_same(O.leaky_relu(x, 0.1), w, b, dil), atol=3e-5, rtol=1e-5)
    lr = np.where(got >= 0, got, got * np.float32(0.1)).astype(np.float32)
    assert float(np.abs(ys - lr).max()) <= 2.0 ** -16 * float(np.abs(lr).max())
    got2 = np.empty((C, L), np.float32)
    _lib.check(lib.sbv2_debug_conv1d_clx(0, P(x), P(w), P(b), P(r), C, C, k, L, dil, 0.1, 1.0 / 3, 0, P(got2), None, None))
    np.testing.assert_array_equal(got2, ((got + r) * np.float32(1.0 / 3)).astype(np.float32))   # (the epilogue's arithmetic on the kernel's own sum: exact)


def _respair(x, w1, w2, b1, b2, k, dil, mask, mask_div, beta, prev, variant):
    N, C = x.shape
    y = np.ascontiguousarray(prev, np.float32).copy() if prev is not None else np.zeros((N, C), np.float32)
    P = lambda a: a.ctypes.data_as(f32p)
    m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
    _lib.check(_lib.lib().sbv2_debug_respair(0, P(x), P(w1), P(w2), P(b1), P(b2), C, N, k, dil, None if m is None else m.ctypes.data, mask_div,
                                             float(beta), 1 if prev is not None else 0, variant, P(y)))
    return y


@pytest.mark.parametrize("C,k,dil,N", [(16, 3, 1, 700), (16, 7, 3, 1024), (16, 11, 5, 300), (32, 3, 5, 257), (32, 7, 1, 2000), (32, 11, 3, 740), (64, 3, 3, 130),
                                       (64, 7, 5, 1111), (64, 11, 1, 512), (32, 11, 5, 31), (16, 7, 1, 246 * 9), (64, 11, 5, 118 * 17 + 3), (64, 7, 3, 122 * 9 + 1),
                                       (32, 7, 5, 250 * 3), (64, 11, 3, 50), (32, 11, 1, 246 * 5 + 7)])
def test_respair_clx_kernel_same_bits_as_respair_cl(C, k, dil, N):
    """respair_clx.hip (round 4: templated taps / channels, LDS-DMA weight groups, 128-position tiles at C = 64, XCD-contiguous tile order) gives the
    SAME bits as respair_cl.hip for one fused ResBlock1 step at C = 32 / 64: plain, with a column mask (edges of packed utterances: mask_div 4), and with
    beta + accumulate (the last step of a branch); and both agree with the numpy oracle's resblock step.  C = 16 runs two taps per 32-deep MFMA
    (v_mfma_f32_16x16x32_bf16): another summation order, so it is held to f32-grade closeness (1e-5) instead of bit equality; so is the default dispatch at
    C = 32 / 64, k = 7 / 11 since round 6 (respair_x16.hip: cross terms in one 16x16x32 instruction, the hi x hi terms of two steps in another), while
    respair_clx.hip itself (variant 2) keeps respair_cl's bits there."""
    x16 = C >= 32 and k >= 7
    close = lambda a, b: np.testing.assert_allclose(a, b, atol=1e-5, rtol=1e-5)
    same = close if (C == 16 or x16) else np.testing.assert_array_equal
    rng = np.random.default_rng(C * 1000 + k * 10 + dil + N)
    x = rng.standard_normal((N, C)).astype(np.float32)
    w1 = (rng.standard_normal((C, C, k)) / np.sqrt(C * k)).astype(np.float32)
    w2 = (rng.standard_normal((C, C, k)) / np.sqrt(C * k)).astype(np.float32)
    b1, b2 = rng.standard_normal(C).astype(np.float32), rng.standard_normal(C).astype(np.float32)
    ref = _respair(x, w1, w2, b1, b2, k, dil, None, 1, 1.0, None, 0)
    got = _respair(x, w1, w2, b1, b2, k, dil, None, 1, 1.0, None, 1)
    same(got, ref)
    xt = x.T.copy()
    t = O.conv1d_same(O.leaky_relu(xt, 0.1), w1, b1, dil)
    want = O.conv1d_same(O.leaky_relu(t, 0.1), w2, b2, 1) + xt
    np.testing.assert_allclose(got.T, want, atol=1e-4, rtol=1e-5)
    mask = (rng.random((N + 3) // 4) > 0.15).astype(np.uint8)
    xm = x * np.repeat(mask, 4)[:N, None]                      # (a masked column holds zeros in the real planes)
    prev = rng.standard_normal((N, C)).astype(np.float32)
    ref = _respair(xm, w1, w2, b1, b2, k, dil, mask, 4, 1.0 / 3, prev, 0)
    got = _respair(xm, w1, w2, b1, b2, k, dil, mask, 4, 1.0 / 3, prev, 1)
    same(got, ref)
    ref = _respair(xm, w1, w2, b1, b2, k, dil, mask, 4, 1.0, None, 0)
    got = _respair(xm, w1, w2, b1, b2, k, dil, mask, 4, 1.0, None, 1)
    same(got, ref)
    assert not np.any(got[np.repeat(mask, 4)[:N] == 0])
    if x16:   # respair_clx.hip at the shapes the default dispatch gives to respair_x16.hip: still respair_cl's bits
        np.testing.assert_array_equal(_respair(xm, w1, w2, b1, b2, k, dil, mask, 4, 1.0, None, 2), ref)
        ref3 = _respair(xm, w1, w2, b1, b2, k, dil, mask, 4, 1.0 / 3, prev, 0)
        np.testing.assert_array_equal(_respair(xm, w1, w2, b1, b2, k, dil, mask, 4, 1.0 / 3, prev, 2), ref3)


def _resbranch(x, w, b, k, dils, mask, mask_div, beta, prev, variant):
    N, C = x.shape
    y = np.ascontiguousarray(prev, np.float32).copy() if prev is not None else np.zeros((N, C), np.float32)
    P = lambda a: a.ctypes.data_as(f32p)
    m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
    d = np.asarray(dils, np.int64)
    _lib.check(_lib.lib().sbv2_debug_resbranch(0, P(x), P(w), P(b), C, N, k, d.ctypes.data_as(_lib.i64p), None if m is None else m.ctypes.data, mask_div,
                                               float(beta), 1 if prev is not None else 0, variant, 0, P(y), None, None, 0))
    return y


@pytest.mark.parametrize("C,N,k", [(16, 700, 3), (16, 233, 3), (16, 232 * 9 + 5, 3), (32, 257, 3), (32, 2000, 3), (32, 232, 3), (32, 31, 3), (64, 130, 3), (64, 1111, 3),
                                   (64, 104 * 17 + 3, 3), (64, 105, 3), (128, 300, 3), (128, 104 * 5 + 1, 3),
                                   # the wide kernels of the 16-channel stage: 512-row windows, 440 (k = 7) / 392 (k = 11) outputs per workgroup
                                   (16, 440 * 3 + 1, 7), (16, 441, 7), (16, 90, 7), (16, 392 * 2, 11), (16, 393, 11), (16, 3000, 11)])
def test_resbranch_kernel_same_bits_as_three_respair_steps(C, N, k):
    """resbranch_clx.hip (round 6): the three steps of a k = 3 ResBlock1 branch (dilations 1, 3, 5) in ONE launch, the residual stream in registers and the
    operand windows in LDS, gives the SAME bits as three launches of the fused step (respair_clx.hip): plain, with a column mask (edges of packed utterances:
    mask_div 4) and with beta + accumulate (the branch's last step), at lengths around the tile sizes (232 / 104 outputs per workgroup); and both agree with
    the numpy oracle's three resblock steps (O.conv1d_same: the checker).  The 16-channel stage's k = 7 / 11 branches run the same kernel on 512-row windows."""
    dils = (1, 3, 5)
    RV = 0 if C <= 64 else 2      # the reference chain: three fused steps (respair_clx, C <= 64) or six conv_cl launches (any C; same bits as the fused step)
    rng = np.random.default_rng(C * 1000 + N + k)
    x = rng.standard_normal((N, C)).astype(np.float32)
    w = (rng.standard_normal((6, C, C, k)) / np.sqrt(C * k)).astype(np.float32)
    b = rng.standard_normal((6, C)).astype(np.float32)
    ref = _resbranch(x, w, b, k, dils, None, 1, 1.0, None, RV)
    got = _resbranch(x, w, b, k, dils, None, 1, 1.0, None, 1)
    np.testing.assert_array_equal(got, ref)
    y = x.T.copy()
    for q in range(3):
        t = O.conv1d_same(O.leaky_relu(y, 0.1), w[2 * q], b[2 * q], dils[q])
        y = O.conv1d_same(O.leaky_relu(t, 0.1), w[2 * q + 1], b[2 * q + 1], 1) + y
    np.testing.assert_allclose(got.T, y, atol=3e-4, rtol=1e-5)
    mask = (rng.random((N + 3) // 4) > 0.15).astype(np.uint8)
    xm = x * np.repeat(mask, 4)[:N, None]                      # (a masked column holds zeros in the real planes)
    prev = rng.standard_normal((N, C)).astype(np.float32) * np.repeat(mask, 4)[:N, None]
    ref = _resbranch(xm, w, b, k, dils, mask, 4, 1.0 / 3, prev, RV)
    got = _resbranch(xm, w, b, k, dils, mask, 4, 1.0 / 3, prev, 1)
    np.testing.assert_array_equal(got, ref)
    ref = _resbranch(xm, w, b, k, dils, mask, 4, 1.0, None, RV)
    got = _resbranch(xm, w, b, k, dils, mask, 4, 1.0, None, 1)
    np.testing.assert_array_equal(got, ref)
    assert not np.any(got[np.repeat(mask, 4)[:N] == 0])
    got = _resbranch(x, w, b, k, (5, 1, 3), None, 1, 1.0, None, 1)      # (another order of the dilations: same halo, other per-step reach)
    np.testing.assert_array_equal(got, _resbranch(x, w, b, k, (5, 1, 3), None, 1, 1.0, None, RV))
    if C >= 32:   # ... and the six-launch conv_cl path (same fragments, same per-accumulator order)
        np.testing.assert_array_equal(got, _resbranch(x, w, b, k, (5, 1, 3), None, 1, 1.0, None, 2))


import contextlib


@contextlib.contextmanager
def _size_independent_dispatch():
    """The dispatch on which a batch row equals its single-utterance call bit for bit: no small-grid launch shapes (gemm_bfs K splits, LayerNorm's few-column
    workgroups: sbv2_debug_set_ksplit(0)) and the wide decoder stages on conv_cl at every launch size (sbv2_debug_set_clx(0)).  The default dispatch picks
    kernels by launch size, whose summation orders differ: f32-rounding agreement, asserted with tolerances beside these blocks."""
    lib = _lib.lib()
    prev_clx, prev_ks = lib.sbv2_debug_set_clx(0), lib.sbv2_debug_set_ksplit(0)
    try:
        yield
    finally:
        lib.sbv2_debug_set_clx(prev_clx)
        lib.sbv2_debug_set_ksplit(prev_ks)


def _gemm_bfs(x, w, b, r, parts, act=0, split_out=0, iters=0):
    m, k = w.shape
    n = x.shape[1]
    y = np.empty((m, n), np.float32)
    P = lambda a: None if a is None else np.ascontiguousarray(a, np.float32).ctypes.data_as(f32p)
    xs, ws = np.ascontiguousarray(x, np.float32), np.ascontiguousarray(w, np.float32)
    bs = None if b is None else np.ascontiguousarray(b, np.float32)
    rs = None if r is None else np.ascontiguousarray(r, np.float32)
    ms = C.c_float()
    _lib.check(_lib.lib().sbv2_debug_gemm_bfs(0, P(xs), P(ws), P(bs), P(rs), m, n, k, parts, act, split_out, iters, y.ctypes.data_as(f32p), C.byref(ms)))
    return y


@pytest.mark.parametrize("K,M,N", [(16, 1, 4), (32, 29, 68), (48, 50, 132), (96, 192, 900), (192, 576, 2052), (1024, 1024, 68), (1024, 3072, 260),
                                   (4096, 1024, 132), (1024, 4096, 2112), (64, 130, 388), (1024, 1024, 2112), (192, 192, 28704)])
def test_gemm_bfs_kernel(K, M, N):
    """The split-bf16 1x1 GEMM on k-major planes (gemm_bfs.hip: pre-split operands, LDS-DMA ring, transposing LDS reads), every tile / ring
    configuration the launcher can pick, against an f64 product of the SAME f32 inputs.  Tolerances: bf16x3 drops the lo*lo term
    (2^-16 relative per product: ~2e-5 on O(1) sums), bf16x6 only 2^-24 terms (f32-grade: compared with numpy's own f32 product)."""
    rng = np.random.default_rng(K + M + N)
    x = rng.standard_normal((K, N)).astype(np.float32)
    w = (rng.standard_normal((M, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(M).astype(np.float32)
    r = rng.standard_normal((M, N)).astype(np.float32)
    ref = w.astype(np.float64) @ x.astype(np.float64) + b[:, None] + r
    err32 = float(np.abs((w @ x + b[:, None] + r).astype(np.float32) - ref).max())
    y3 = _gemm_bfs(x, w, b, r, 2)
    y6 = _gemm_bfs(x, w, b, r, 3)
    assert float(np.abs(y3 - ref).max()) < 1e-4
    assert float(np.abs(y6 - ref).max()) < max(4 * err32, 2e-5)
    # f16x3 (parts code 4): f16 hi + scaled f16 lo, 22 mantissa bits per operand, dropped term 2^-22: f32-grade within a small factor
    yh = _gemm_bfs(x, w, b, r, 4)
    assert float(np.abs(yh - ref).max()) < max(8 * err32, 2e-5)
    # GELU epilogue without bias / residual; the result re-emitted as bf16 parts (what the next product reads) loses nothing at 3 parts
    g = O.gelu(w.astype(np.float64) @ x.astype(np.float64)) if hasattr(O, "gelu") else None
    if g is not None:
        y6g = _gemm_bfs(x, w, None, None, 3, act=2)
        assert float(np.abs(y6g - g).max()) < max(4 * err32, 2e-5)
        y6s = _gemm_bfs(x, w, None, None, 3, act=2, split_out=3)
        assert float(np.abs(y6s - y6g).max()) <= 1e-7 * max(1.0, float(np.abs(y6g).max()))
        y3s = _gemm_bfs(x, w, None, None, 2, act=2, split_out=2)
        assert float(np.abs(y3s - g).max()) < 1e-4
        yhs = _gemm_bfs(x, w, None, None, 4, act=2, split_out=4)     # ... and as the f16 pair: 2^-22 relative
        assert float(np.abs(yhs - g).max()) < max(8 * err32, 2e-5)


@pytest.mark.parametrize("K,M,N", [(1024, 1024, 68), (1024, 3072, 68), (1024, 4096, 68), (4096, 1024, 68), (4096, 1024, 132), (2048, 192, 260), (2048, 512, 1100)])
def test_gemm_bfs_small_grid_k_split(K, M, N):
    """gemm_bfs' small-grid K split (SK: several workgroups per output tile, the last one to arrive adds the partial sums in group order): against the f64
    product and the unsplit kernel (sbv2_debug_set_ksplit(0)), the same bits on every repetition (the order of the sum does not depend on which workgroup
    arrives last) and arrival counters that are zero again after every launch (otherwise the second launch would add garbage or never finish its tiles)."""
    lib = _lib.lib()
    rng = np.random.default_rng(K + M + N)
    x = rng.standard_normal((K, N)).astype(np.float32)
    w = (rng.standard_normal((M, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(M).astype(np.float32)
    r = rng.standard_normal((M, N)).astype(np.float32)
    ref = w.astype(np.float64) @ x.astype(np.float64) + b[:, None] + r
    err32 = float(np.abs((w @ x + b[:, None] + r).astype(np.float32) - ref).max())
    prev = lib.sbv2_debug_set_ksplit(1)
    try:
        first = _gemm_bfs(x, w, b, r, 4, iters=25)        # 26 launches on the same scratch, the last one's result
        assert float(np.abs(first - ref).max()) < max(8 * err32, 2e-5)
        for _ in range(3):
            np.testing.assert_array_equal(_gemm_bfs(x, w, b, r, 4), first)
        lib.sbv2_debug_set_ksplit(0)
        unsplit = _gemm_bfs(x, w, b, r, 4)
        assert float(np.abs(unsplit - ref).max()) < max(8 * err32, 2e-5)
        d = float(np.abs(first - unsplit).max())
        print(f"gemm_bfs K split vs unsplit ({K} x {M} x {N}): max-abs {d:.2e}")
        assert d < 2e-5
        # stale partial sums: two DIFFERENT inputs alternate on one scratch buffer that is never cleared (a stale L2 line of the previous launch would hold the
        # other input's sums); each must give the bits of its own single launch
        lib.sbv2_debug_set_ksplit(1)
        x2 = rng.standard_normal((K, N)).astype(np.float32)
        ya, yb = np.empty((M, N), np.float32), np.empty((M, N), np.float32)
        P = lambda a: a.ctypes.data_as(f32p)
        _lib.check(lib.sbv2_debug_gemm_bfs_alt(0, P(x), P(x2), P(w), P(b), P(r), M, N, K, 4, 24, P(ya), P(yb)))
        np.testing.assert_array_equal(ya, first)
        np.testing.assert_array_equal(yb, _gemm_bfs(x2, w, b, r, 4))
    finally:
        lib.sbv2_debug_set_ksplit(prev)


def test_gemm_bfs_f16x3_range():
    """The f16 pair saturates instead of overflowing (common.h split_store*: values beyond +-65504 are clamped before the split, so nothing
    turns into inf / NaN), keeps 22 bits down to f16's subnormals (the scaled lo part), and bf16x6 carries the full f32 exponent range."""
    rng = np.random.default_rng(5)
    K, M, N = 64, 32, 8
    x = rng.standard_normal((K, N)).astype(np.float32)
    w = (rng.standard_normal((M, K)) / 8).astype(np.float32)
    x[3, 2] = 3.0e5                      # beyond f16
    x[5, :] *= 1e-6                      # far below f16's normal range (6e-5)
    ref = w.astype(np.float64) @ x.astype(np.float64)
    y6 = _gemm_bfs(x, w, None, None, 3)
    np.testing.assert_allclose(y6, ref, rtol=0, atol=2e-6 * float(np.abs(ref).max()))
    yh = _gemm_bfs(x, w, None, None, 4)
    assert np.isfinite(yh).all()
    xs = x.copy(); xs[3, 2] = 65504.0    # what the split saturates to
    np.testing.assert_allclose(yh, w.astype(np.float64) @ xs.astype(np.float64), rtol=0, atol=2e-6 * 65504.0 / 8)
    cols = [c for c in range(N) if c != 2]      # columns the huge value does not touch: full accuracy, the tiny row included
    np.testing.assert_allclose(yh[:, cols], ref[:, cols], rtol=0, atol=1e-5)
    # the saturation is countable (sbv2_debug_f16x3_saturation), and what is not finite is not clamped: a NaN / an infinity reaches the result
    lib, cnt = _lib.lib(), C.c_uint64(0)
    _lib.check(lib.sbv2_debug_f16x3_saturation(0, 1, C.byref(cnt)))
    try:
        _gemm_bfs(x, w, None, None, 4)
        _lib.check(lib.sbv2_debug_f16x3_saturation(0, -1, C.byref(cnt)))
        assert cnt.value == 1
        xn = x.copy(); xn[3, 2] = 1.0; xn[7, 4] = np.nan; xn[9, 6] = -np.inf
        yn = _gemm_bfs(xn, w, None, None, 4)
        _lib.check(lib.sbv2_debug_f16x3_saturation(0, -1, C.byref(cnt)))
        assert cnt.value == 0
        assert np.isnan(yn[:, 4]).all() and not np.isfinite(yn[:, 6]).any()
        assert np.isfinite(yn[:, [c for c in range(N) if c not in (4, 6)]]).all()
    finally:
        _lib.check(lib.sbv2_debug_f16x3_saturation(0, 1, None))   # (counting is on by default since round 6; the call also resets the count)


def test_f16x3_clamp_warns_once_per_handle(capfd):
    """Real-checkpoint safety (round 6): counting the f16x3 split's clamps is ON by default, and a model handle prints ONE stderr warning, independent of
    SBV2_LOG, the first time the count is non-zero, naming the bf16x6 fallback.  A tiny DeBERTa whose FFN up-projection is scaled by 1e6 produces GELU
    outputs far beyond f16's +-65504 (the FFN intermediate exists only as f16x3 parts); the unscaled model stays silent."""
    lib, cnt = _lib.lib(), C.c_uint64(0)
    bc, bw = weights("bert", "tiny")
    ids = np.array([1, 5, 9, 13, 17, 21, 2], np.int64)
    msk = np.ones_like(ids)
    _lib.check(lib.sbv2_debug_f16x3_saturation(0, 1, C.byref(cnt)))    # reset
    quiet = model.load_model(synth.pack_blob(synth.KIND_BERT, bc, bw), True)
    if lib.sbv2_bert_gemm_parts(quiet.handle) != 4:
        quiet.close()
        pytest.skip("DeBERTa products are not on the f16x3 format in this environment (SBV2_BERT_GEMM)")
    model.predict(quiet, ids, msk)
    model.predict(quiet, ids, msk)
    quiet.close()
    assert "WARNING" not in capfd.readouterr().err
    W = dict(bw)
    W["deberta.encoder.layer.0.intermediate.dense.weight"] = (bw["deberta.encoder.layer.0.intermediate.dense.weight"] * np.float32(1e6)).astype(np.float32)
    loud = model.load_model(synth.pack_blob(synth.KIND_BERT, bc, W), True)
    out = model.predict(loud, ids, msk)
    err1 = capfd.readouterr().err
    assert err1.count("sbv2_hip WARNING") == 1 and "SBV2_BERT_GEMM=bf16x6" in err1 and "65504" in err1
    model.predict(loud, ids, msk)
    assert "WARNING" not in capfd.readouterr().err          # once per handle
    loud.close()
    assert np.isfinite(out).all()                            # clamped, not overflowed
    _lib.check(lib.sbv2_debug_f16x3_saturation(0, -1, C.byref(cnt)))
    assert cnt.value > 0
    # a second handle on the same (still dirty after the reset above? no: reset) device warns again when ITS input clamps
    loud2 = model.load_model(synth.pack_blob(synth.KIND_BERT, bc, W), True)
    model.predict(loud2, ids, msk)
    assert capfd.readouterr().err.count("sbv2_hip WARNING") == 1
    loud2.close()
    _lib.check(lib.sbv2_debug_f16x3_saturation(0, 1, C.byref(cnt)))    # leave the count at zero for the tests that follow


@pytest.mark.parametrize("cin,cout,k,dil,L", [(1024, 1024, 1, 1, 66), (1024, 3072, 1, 1, 130), (4096, 1024, 1, 1, 66), (1024, 4096, 1, 1, 35),
                                              (192, 576, 1, 1, 897), (768, 192, 1, 1, 897), (192, 29, 1, 1, 300), (96, 192, 1, 1, 4), (1024, 192, 1, 1, 130),
                                              (64, 50, 1, 1, 19), (192, 768, 3, 1, 257), (768, 192, 3, 1, 257), (256, 256, 3, 1, 130), (192, 192, 5, 1, 61),
                                              (64, 48, 3, 3, 100), (96, 64, 7, 5, 33), (192, 192, 3, 1, 9)])
def test_gemm_small_grid_kernel_same_bits(cin, cout, k, dil, L):
    """The one-wave 16 x 16 kernels of small grids (gemm_skinny.hip) and the tiled kernel give the SAME bits (every f32 MFMA shape is a
    sequential fma chain over k), which is what keeps a batch row bit-identical to the single call of the same utterance."""
    rng = np.random.default_rng(cin + cout + L + k)
    x = rng.standard_normal((cin, L)).astype(np.float32)
    w = (rng.standard_normal((cout, cin, k)) / np.sqrt(cin * k)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    lib = _lib.lib()
    for slope in (1.0, 0.1):
        prev = lib.sbv2_debug_set_skinny_max(0)
        try:
            tiled = _conv_dev(x, w, b, dil, slope)
            lib.sbv2_debug_set_skinny_max(1 << 30)
            skinny = _conv_dev(x, w, b, dil, slope)
        finally:
            lib.sbv2_debug_set_skinny_max(prev)
        assert np.array_equal(tiled.view(np.uint32), skinny.view(np.uint32)), f"{int((tiled != skinny).sum())} of {tiled.size} differ"
        np.testing.assert_allclose(skinny, O.conv1d_same(O.leaky_relu(x, slope), w, b, dil), atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize("cin,cout,k,s,p,L", [(64, 32, 16, 8, 4, 301), (32, 16, 8, 2, 3, 1000), (16, 8, 2, 2, 0, 999),
                                              (512, 256, 16, 8, 4, 57), (8, 4, 4, 4, 0, 33)])
def test_conv_transpose_kernel(cin, cout, k, s, p, L):
    rng = np.random.default_rng(k * 100 + s)
    x = rng.standard_normal((cin, L)).astype(np.float32)
    w = (rng.standard_normal((cin, cout, k)) / np.sqrt(cin * k / s)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    ref = O.conv_transpose1d(O.leaky_relu(x, 0.1), w, b, s, p)
    y = np.empty((cout, L * s), np.float32)
    P = lambda a: a.ctypes.data_as(f32p)
    _lib.check(_lib.lib().sbv2_debug_conv_transpose1d(0, P(x), P(w), P(b), cin, cout, k, L, s, p, 0.1, P(y)))
    np.testing.assert_allclose(y, ref, atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize("cin,cout,k,s,L", [(64, 64, 16, 8, 300), (128, 64, 8, 2, 1000), (64, 128, 16, 8, 256), (256, 128, 16, 8, 515), (128, 64, 8, 2, 4097)])
def test_conv_transpose_clx_kernel(cin, cout, k, s, L):
    """The wide stages' ConvTranspose1d as ONE phased conv_clx launch (round 6): rows = (phase, cout), taps = the union of the phases' input taps padded to an
    odd count with zero weights, output row n * stride + phase, on pre-split bf16 hi / lo operands: against the oracle's conv_transpose1d (the checker), with a
    column mask on the input positions, every output row written (the buffer starts as NaN), and the bf16 parts of lrelu(result) it leaves for the ResBlocks."""
    rng = np.random.default_rng(k * 100 + s + L)
    x = rng.standard_normal((cin, L)).astype(np.float32)
    w = (rng.standard_normal((cin, cout, k)) / np.sqrt(cin * k / s)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    P = lambda a: a.ctypes.data_as(f32p)
    p = (k - s) // 2
    ref = O.conv_transpose1d(O.leaky_relu(x, 0.1), w, b, s, p)
    y, ys = np.empty((cout, L * s), np.float32), np.empty((cout, L * s), np.float32)
    _lib.check(_lib.lib().sbv2_debug_conv_transpose1d_clx(0, P(x), P(w), P(b), cin, cout, k, L, s, 0.1, None, 1, 0, P(y), P(ys), None))
    assert np.isfinite(y).all()
    np.testing.assert_allclose(y, ref, atol=3e-5, rtol=1e-5)
    lr = np.where(y >= 0, y, y * np.float32(0.1)).astype(np.float32)
    assert float(np.abs(ys - lr).max()) <= 2.0 ** -16 * float(np.abs(lr).max())
    # a launch that writes parts packs its rows as (phase pair, 16 channels, phase in pair, channel) so that the 32-byte parts rows of two adjacent output rows
    # leave together (ConvClxParams::phase_group); the plain (phase, channel) order (sbv2_debug_set_upx(2)) gives the same bits
    y0, ys0, y1, ys1 = np.empty_like(y), np.empty_like(ys), np.empty_like(y), np.empty_like(ys)
    prev = _lib.lib().sbv2_debug_set_upx(1)
    try:
        _lib.check(_lib.lib().sbv2_debug_conv_transpose1d_clx(0, P(x), P(w), P(b), cin, cout, k, L, s, 0.1, None, 1, 0, P(y0), P(ys0), None))
        _lib.lib().sbv2_debug_set_upx(2)
        _lib.check(_lib.lib().sbv2_debug_conv_transpose1d_clx(0, P(x), P(w), P(b), cin, cout, k, L, s, 0.1, None, 1, 0, P(y1), P(ys1), None))
        _lib.lib().sbv2_debug_set_upx(3)    # the union of all phases' taps (a zero tap per phase): another pairing of the steps, f32 rounding apart
        y3 = np.empty_like(y)
        _lib.check(_lib.lib().sbv2_debug_conv_transpose1d_clx(0, P(x), P(w), P(b), cin, cout, k, L, s, 0.1, None, 1, 0, P(y3), P(ys1.copy()), None))
    finally:
        _lib.lib().sbv2_debug_set_upx(prev)
    np.testing.assert_array_equal(y1, y0)
    np.testing.assert_array_equal(ys1, ys0)
    np.testing.assert_allclose(y3, y0, atol=1e-5, rtol=1e-5)
    mask = (rng.random((L + 3) // 4) > 0.2).astype(np.uint8)
    keep = np.repeat(mask, 4)[:L]
    xm = x * keep[None, :]
    refm = O.conv_transpose1d(O.leaky_relu(xm, 0.1), w, b, s, p) * np.repeat(keep, s)[None, :]
    _lib.check(_lib.lib().sbv2_debug_conv_transpose1d_clx(0, P(xm), P(w), P(b), cin, cout, k, L, s, 0.1, mask.ctypes.data, 4, 0, P(y), None, None))
    np.testing.assert_allclose(y, refm, atol=3e-5, rtol=1e-5)
    assert not np.any(y[:, np.repeat(keep, s) == 0])


def _conv_cl_dev(x, w, b, dil, slope, mode):
    cout, cin, k = w.shape
    y = np.empty((cout, x.shape[1]), np.float32)
    P = lambda a: a.ctypes.data_as(f32p)
    xs, ws, bs = (np.ascontiguousarray(a, np.float32) for a in (x, w, b))
    ms = np.zeros(1, np.float32)
    _lib.check(_lib.lib().sbv2_debug_conv1d_cl(0, P(xs), P(ws), P(bs), cin, cout, k, x.shape[1], dil, slope, mode, 0, P(y), P(ms)))
    return y


@pytest.mark.parametrize("cin,cout,k,dil,L", [(16, 16, 11, 5, 3000), (32, 32, 7, 3, 1500), (64, 64, 3, 1, 777), (128, 128, 11, 1, 1030),
                                              (256, 256, 7, 5, 515), (192, 512, 7, 1, 130), (16, 16, 3, 3, 70000), (48, 20, 5, 1, 9)])
def test_conv1d_cl_kernel(cin, cout, k, dil, L):
    """Channels-last bf16 MFMA kernel: split-bf16 must be f32-grade, plain bf16 within bf16 rounding of the operands."""
    rng = np.random.default_rng(cin * 1000 + cout + k)
    x = rng.standard_normal((cin, L)).astype(np.float32)
    w = (rng.standard_normal((cout, cin, k)) / np.sqrt(cin * k)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    ref = O.conv1d_same(O.leaky_relu(x, 0.1), w, b, dil)
    got = _conv_cl_dev(x, w, b, dil, 0.1, 1)
    np.testing.assert_allclose(got, ref, atol=5e-5, rtol=1e-5)
    got = _conv_cl_dev(x, w, b, dil, 0.1, 2)
    np.testing.assert_allclose(got, ref, atol=3e-2, rtol=0)
    got = _conv_cl_dev(x, w, b, dil, 0.1, 3)       # fp16 operands: 8x finer than bf16
    np.testing.assert_allclose(got, ref, atol=4e-3, rtol=0)


def _golden(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name))
    return z, ast.literal_eval(str(z["cfg"]))


@pytest.fixture(scope="module")
def bert_tiny():
    s = model.load_model(blob("bert", "tiny", 3), True)
    yield s
    s.close()


def test_deberta_tiny_golden(golden_dir):
    """HIP output vs transformers' hidden_states[-3][0] directly; the *_conv_* fixtures have DebertaV2Encoder.conv (the ConvLayer after
    layer 0: gelu k3, tanh k5, and a masked tail)."""
    for name in ("deberta_tiny_S24.npz", "deberta_tiny_S5.npz", "deberta_tiny_conv_S24.npz", "deberta_tiny_conv_tanh_S9.npz",
                 "deberta_tiny_conv_masked_S12.npz"):
        z, cfg = _golden(golden_dir, name)
        s = model.load_model(synth.pack_blob(synth.KIND_BERT, cfg, synth.make_deberta_weights(cfg, int(z["seed"]))), True)
        ids, am = z["input_ids"], z["attention_mask"]
        out = model.predict(s, ids, am)
        keep = am > 0
        np.testing.assert_allclose(out[keep], z["output"][keep], atol=5e-5, rtol=0)
        s.close()


def test_deberta_conv_layer_batch_and_guard():
    """ConvLayer in a packed batch (neighbouring utterances must not leak through the k = 3 window: the layout gap is its zero
    padding) and the load-time guard: a container that carries encoder.conv.* but does not declare conv_kernel_size is refused."""
    cfg = O.DEBERTA_TINY_CONV
    W = synth.make_deberta_weights(cfg, 3)
    s = model.load_model(synth.pack_blob(synth.KIND_BERT, cfg, W), True)
    rng = np.random.default_rng(1)
    seqs = [np.concatenate([[1], rng.integers(3, cfg["vocab_size"], n), [2]]) for n in (2, 6, 30, 70, 11, 62)]   # lengths 4, 8, 32, 72: no alignment gaps
    batch = model.predict_batch(s, seqs)
    for ids, got in zip(seqs, batch):
        np.testing.assert_allclose(got, O.deberta_forward(W, cfg, ids), atol=5e-5, rtol=0)
        np.testing.assert_array_equal(got, model.predict(s, ids, np.ones_like(ids)))
    s.close()
    with pytest.raises(model.Sbv2Error, match="conv"):
        model.load_model(synth.pack_blob(synth.KIND_BERT, dict(cfg, conv_kernel_size=0), W), True)
    bad = dict(W)
    bad["deberta.encoder.conv.LayerNorm.weight"] = np.ones(cfg["hidden"] + 1, np.float32)
    with pytest.raises(model.Sbv2Error, match="shape"):
        model.load_model(synth.pack_blob(synth.KIND_BERT, cfg, bad), True)


def test_deberta_batch_equals_single(bert_tiny):
    cfg, W = weights("bert", "tiny", 3)
    rng = np.random.default_rng(0)
    # 72 / 100 / 128 tokens take the tiled fused kernel (65 .. 128), 130 and 200 its key-tile loop (> 128): the batch mixes all three
    seqs = [np.concatenate([[1], rng.integers(3, cfg["vocab_size"], n), [2]]) for n in (1, 7, 30, 70, 18, 3, 62, 128, 98, 126, 63, 198)]
    batch = model.predict_batch(bert_tiny, seqs)
    for ids, got in zip(seqs, batch):
        single = model.predict(bert_tiny, ids, np.ones_like(ids))
        ref = O.deberta_forward(W, cfg, ids)
        np.testing.assert_allclose(single, ref, atol=5e-5, rtol=0)
        np.testing.assert_array_equal(got, single)   # packing must not change a single bit


def test_deberta_full_shape_mid_lengths_vs_oracle():
    """Full ku-nlp-large shape (16 heads x 64, 256 log buckets, ConvLayer) at 65 .. 128 tokens: the tiled fused attention (the reference's
    TensorRT profile allows 100 tokens, model.rs:15) against the oracle, alone and inside a batch with short and long neighbours."""
    cfg, W = weights("bert", "full")
    s = model.load_model(blob("bert", "full"), True)
    rng = np.random.default_rng(7)
    seqs = [np.concatenate([[1], rng.integers(3, cfg["vocab_size"], n), [2]]) for n in (98, 20, 126, 63, 150)]
    batch = model.predict_batch(s, seqs)
    prev = _lib.lib().sbv2_debug_set_ksplit(0)      # the unsplit dispatch: one summation order whatever the grid
    try:
        batch0 = model.predict_batch(s, seqs)
        single0 = {k: model.predict(s, seqs[k], np.ones_like(seqs[k])) for k in (0, 2, 4)}
    finally:
        _lib.lib().sbv2_debug_set_ksplit(prev)
    for k in (0, 2, 4):     # 100 and 128 tokens: the tiled fused attention; 152 tokens: the key-tile loop (attn_deberta.hip, > 128) at the full shape
        ref = O.deberta_forward(W, cfg, seqs[k])
        np.testing.assert_allclose(batch[k], ref, atol=2e-4, rtol=0)
        np.testing.assert_array_equal(single0[k], batch0[k])
        # default dispatch: small grids split K over several workgroups (gemm_bfs.hip SK, round 5; this 470-token batch is small enough to split two ways,
        # the single call up to eight ways): f32-rounding agreement between the single call and its batch row
        single = model.predict(s, seqs[k], np.ones_like(seqs[k]))
        np.testing.assert_allclose(single, ref, atol=2e-4, rtol=0)
        d = float(np.abs(single - batch[k]).max())
        print(f"DeBERTa full shape, {len(seqs[k])} tokens: single call vs batch row under the default (K split) dispatch max-abs {d:.2e}")
        assert d < 5e-5
    s.close()

def test_deberta_long_attention_key_tile_loop(bert_tiny):
    """> 128 tokens (long-form text, BASELINE configs[4]): the fused attention's key-tile loop with an online softmax, against the oracle:
    lengths that end inside a tile / on a tile edge / with idle waves (fewer than four key tiles per query tile is impossible here, more
    than four per wave at 515), a masked tail, packing next to short and mid neighbours without changing a bit, and the full shape
    (16 heads x 64, 256 log buckets: relative distances beyond the exact range of the bucket function) at 300 tokens."""
    cfg, W = weights("bert", "tiny", 3)
    rng = np.random.default_rng(21)
    seqs = [np.concatenate([[1], rng.integers(3, cfg["vocab_size"], n), [2]]) for n in (127, 158, 513, 40, 254, 100)]   # 129, 160, 515, 42, 256, 102
    batch = model.predict_batch(bert_tiny, seqs)
    with _size_independent_dispatch():      # (the single calls' small-grid launch shapes sum in another order than the 1200-column batch's: round 5)
        batch0 = model.predict_batch(bert_tiny, seqs)
        for ids, got in zip(seqs, batch0):
            np.testing.assert_array_equal(got, model.predict(bert_tiny, ids, np.ones_like(ids)))
    for ids, got in zip(seqs, batch):
        single = model.predict(bert_tiny, ids, np.ones_like(ids))
        ref = O.deberta_forward(W, cfg, ids)
        np.testing.assert_allclose(single, ref, atol=5e-5, rtol=0)
        np.testing.assert_allclose(got, ref, atol=5e-5, rtol=0)
        np.testing.assert_allclose(got, single, atol=2e-5, rtol=0)
    ids = seqs[2]
    mask = np.ones_like(ids); mask[-70:] = 0     # two whole key tiles and a part of a third masked
    got = model.predict(bert_tiny, ids, mask)
    ref = O.deberta_forward(W, cfg, ids, mask)
    np.testing.assert_allclose(got[:-70], ref[:-70], atol=5e-5, rtol=0)
    cfgf, Wf = weights("bert", "full")
    s = model.load_model(blob("bert", "full"), True)
    ids = np.concatenate([[1], np.random.default_rng(22).integers(3, cfgf["vocab_size"], 298), [2]])
    np.testing.assert_allclose(model.predict(s, ids, np.ones_like(ids)), O.deberta_forward(Wf, cfgf, ids), atol=2e-4, rtol=0)
    s.close()


def test_deberta_attention_mask(bert_tiny):
    cfg, W = weights("bert", "tiny", 3)
    ids = np.array([1, 5, 9, 33, 70, 2, 0, 0], np.int64)
    mask = np.array([1, 1, 1, 1, 1, 1, 0, 0], np.int64)
    got = model.predict(bert_tiny, ids, mask)
    ref = O.deberta_forward(W, cfg, ids, mask)
    np.testing.assert_allclose(got[:6], ref[:6], atol=5e-5, rtol=0)
    # the same through the tiled kernel (90 tokens, the last 7 masked)
    rng = np.random.default_rng(3)
    ids = np.concatenate([[1], rng.integers(3, cfg["vocab_size"], 88), [2]])
    mask = np.ones_like(ids); mask[-7:] = 0
    got = model.predict(bert_tiny, ids, mask)
    ref = O.deberta_forward(W, cfg, ids, mask)
    np.testing.assert_allclose(got[:-7], ref[:-7], atol=5e-5, rtol=0)


@pytest.mark.parametrize("name", ["deberta_full_S64.npz", "deberta_full_S64_noconv.npz"])
def test_deberta_full_golden(golden_dir, name):
    """Full ku-nlp-large shape with and without the ConvLayer vs transformers."""
    z, cfg = _golden(golden_dir, name)
    s = model.load_model(synth.pack_blob(synth.KIND_BERT, cfg, synth.make_deberta_weights(cfg, int(z["seed"]))), True)
    ids = z["input_ids"]
    out = model.predict(s, ids, np.ones_like(ids))
    np.testing.assert_allclose(out, z["output"], atol=2e-4, rtol=0)
    s.close()


@pytest.fixture(scope="module")
def vits_tiny():
    s = model.load_model(blob("vits", "tiny", 5), False)
    yield s
    s.close()


def _oracle_utt(W, cfg, u, i, sdp_ratio, length_scale, ns, nsw, seed, forced):
    return O.vits_forward(W, cfg, u["bert"], u["phones"], u["tones"], u["langs"], u["sid"], u["style"], sdp_ratio, length_scale,
                          noise_w=oracle_noise_w(seed, i, u["T_text"], nsw) if nsw else None,
                          noise_z=oracle_noise_z(seed, i, cfg["inter"], ns) if ns else None,
                          forced_durations=u["forced_durations"] if forced else None, return_all=True)


def test_vits_tiny_stages_single(vits_tiny):
    """One utterance, no noise: every traced stage against the oracle."""
    cfg, W = weights("vits", "tiny", 5)
    u = make_utts([9], O.DEBERTA_TINY, cfg, seed0=11)[0]
    model.set_trace(vits_tiny, True)
    pcm = model.synthesize(vits_tiny, u["bert"], u["phones"], [0], u["tones"], u["langs"], u["style"], 0.0, 1.0, 0.0, 0.0)
    r = _oracle_utt(W, cfg, u, 0, 0.0, 1.0, 0.0, 0.0, 0, False)
    np.testing.assert_allclose(model.get_trace(vits_tiny, "x"), r["x"], atol=5e-5, rtol=0)
    np.testing.assert_allclose(model.get_trace(vits_tiny, "stats"), np.concatenate([r["m_p"], r["logs_p"]]), atol=5e-5, rtol=0)
    d, lw = model.fetch_durations(vits_tiny, u["T_text"])
    np.testing.assert_allclose(lw, r["logw"], atol=5e-5, rtol=0)
    assert np.array_equal(d, r["durations"])
    np.testing.assert_allclose(model.get_trace(vits_tiny, "z_p"), r["z_p"], atol=5e-5, rtol=0)
    np.testing.assert_allclose(model.get_trace(vits_tiny, "z"), r["z"], atol=1e-4, rtol=0)
    model.set_trace(vits_tiny, False)
    assert pcm.shape == (1, 1, r["pcm"].shape[0])
    np.testing.assert_allclose(pcm[0, 0], r["pcm"], atol=1e-4, rtol=0)


@pytest.mark.parametrize("sdp_ratio,ns,nsw,length_scale", [(0.0, 0.0, 0.0, 1.0), (0.5, 0.0, 0.8, 1.0), (1.0, 0.667, 0.8, 1.3),
                                                            (0.0, 0.667, 0.8, 0.7)])
def test_vits_tiny_batch_mixed(vits_tiny, sdp_ratio, ns, nsw, length_scale):
    """Mixed-length batch with injected counter-based noise: each utterance equals the oracle's batch-1 result."""
    cfg, W = weights("vits", "tiny", 5)
    utts = make_utts([5, 17, 2, 11], O.DEBERTA_TINY, cfg, seed0=21)
    utts[1]["sid"] = 1
    seed = 77
    pcms = model.synthesize_batch(vits_tiny, utts, sdp_ratio, length_scale, ns, nsw, seed)
    tot = sum(u["T_text"] for u in utts)
    d, lw = model.fetch_durations(vits_tiny, tot)
    off = 0
    refs, flipped = [], []
    for i, (u, got) in enumerate(zip(utts, pcms)):
        r = _oracle_utt(W, cfg, u, i, sdp_ratio, length_scale, ns, nsw, seed, False)
        T = u["T_text"]
        # the noisy SDP path goes through the spline's quadratic root: looser bound on log-durations there
        np.testing.assert_allclose(lw[off:off + T], r["logw"], atol=1e-3 if nsw else 2e-4, rtol=0)
        # ceil() is discontinuous: only compare durations whose pre-ceil value is not within 1e-3 of an integer
        w = np.exp(r["logw"]) * length_scale
        safe = np.abs(w - np.round(w)) > 1e-3
        assert np.array_equal(d[off:off + T][safe], r["durations"][safe])
        refs.append(r)
        if np.array_equal(d[off:off + T], r["durations"]):
            assert got.shape == r["pcm"].shape
            np.testing.assert_allclose(got, r["pcm"], atol=2e-4, rtol=0)
        else:
            flipped.append(i)      # only durations inside the 1e-3 ceil band may differ (asserted above)
        off += T
    if flipped:
        # a duration on the ceil edge came out differently: the waveform is then compared with the oracle's durations teacher-forced
        # (same noise streams: they are keyed by seed and utterance index), never skipped
        forced = [dict(u, forced_durations=r["durations"]) for u, r in zip(utts, refs)]
        pcms2 = model.synthesize_batch(vits_tiny, forced, sdp_ratio, length_scale, ns, nsw, seed, forced=True)
        for i in flipped:
            assert pcms2[i].shape == refs[i]["pcm"].shape
            np.testing.assert_allclose(pcms2[i], refs[i]["pcm"], atol=2e-4, rtol=0)


def test_vits_tiny_forced_durations_and_single_equals_batch(vits_tiny):
    cfg, W = weights("vits", "tiny", 5)
    utts = make_utts([6, 13, 8], O.DEBERTA_TINY, cfg, seed0=31)
    pcms = model.synthesize_batch(vits_tiny, utts, forced=True)
    for i, (u, got) in enumerate(zip(utts, pcms)):
        r = _oracle_utt(W, cfg, u, i, 0.0, 1.0, 0.0, 0.0, 0, True)
        assert got.shape[0] == O.hop_length(cfg) * (7 * ((u["T_text"] - 1) // 2) + 1)
        np.testing.assert_allclose(got, r["pcm"], atol=1e-4, rtol=0)
        alone = model.synthesize_batch(vits_tiny, [u], forced=True)[0]
        np.testing.assert_array_equal(alone, got)     # batching must not change a single bit


def test_vits_full_small_utterance():
    """Full JP-Extra shape, 12 phone symbols (T_text 25), predicted durations."""
    cfg, W = weights("vits", "full")
    s = model.load_model(blob("vits", "full"), False)
    u = make_utts([12], O.DEBERTA_FULL, cfg, seed0=41)[0]
    model.set_trace(s, True)
    pcm = model.synthesize(s, u["bert"], u["phones"], [0], u["tones"], u["langs"], u["style"], 0.0, 1.0, 0.0, 0.0)
    r = _oracle_utt(W, cfg, u, 0, 0.0, 1.0, 0.0, 0.0, 0, False)
    d, lw = model.fetch_durations(s, u["T_text"])
    np.testing.assert_allclose(lw, r["logw"], atol=2e-4, rtol=0)
    assert np.array_equal(d, r["durations"])
    np.testing.assert_allclose(model.get_trace(s, "z"), r["z"], atol=5e-4, rtol=0)
    np.testing.assert_allclose(pcm[0, 0], r["pcm"], atol=1e-3, rtol=0)      # north_star tolerance
    err = float(np.abs(pcm[0, 0] - r["pcm"]).max())
    print("full-config waveform max-abs error:", err)
    assert err < 2e-4
    s.close()


def test_decoder_clx_path_agrees_with_conv_cl_path():
    """Full JP-Extra shape: the wide decoder stages on conv_clx.hip (pre-split operands written by the producing epilogues, LDS-DMA rings, 16x16x32 MFMAs:
    the default for large launches) against the same stages on conv_cl.hip (sbv2_debug_set_clx(0); what a single utterance and a streaming window run): the
    same products in another summation order.  The waveforms (peak ~0.1) agree to 2e-6; the oracle tolerance of the decoder tests is 5e-5."""
    cfg, W = weights("vits", "full")
    s = model.load_model(blob("vits", "full"), False)
    utts = make_utts([12, 31, 5], O.DEBERTA_FULL, cfg, seed0=77)
    lib = _lib.lib()
    prev = lib.sbv2_debug_set_clx(2)      # 2 = conv_clx at every size (1, the default, leaves small launches to conv_cl)
    try:
        a = model.synthesize_batch(s, utts, forced=True)
        lib.sbv2_debug_set_clx(0)
        b = model.synthesize_batch(s, utts, forced=True)
    finally:
        lib.sbv2_debug_set_clx(prev)
    for x, y in zip(a, b):
        np.testing.assert_allclose(x, y, atol=2e-6, rtol=0)
    s.close()


def test_decoder_respair_clx_path_same_bits_as_respair_cl_path():
    """Full JP-Extra shape: the fused ResBlock steps of the 64- / 32- / 16-channel decoder stages on respair_clx.hip (the default) against the same
    stages on respair_cl.hip (sbv2_debug_set_respair_clx(0)).  The 64- and 32-channel kernels give the same bits; the 16-channel kernel sums two taps per
    MFMA (another order): the waveforms agree to 2e-6 (peak ~0.1; the oracle tolerance of the decoder tests is 5e-5)."""
    cfg, W = weights("vits", "full")
    s = model.load_model(blob("vits", "full"), False)
    utts = make_utts([12, 31, 5], O.DEBERTA_FULL, cfg, seed0=78)
    lib = _lib.lib()
    prev = lib.sbv2_debug_set_respair_clx(1)
    try:
        a = model.synthesize_batch(s, utts, forced=True)
        lib.sbv2_debug_set_respair_clx(0)
        b = model.synthesize_batch(s, utts, forced=True)
    finally:
        lib.sbv2_debug_set_respair_clx(prev)
    for x, y in zip(a, b):
        np.testing.assert_allclose(x, y, atol=2e-6, rtol=0)
    s.close()


def test_decoder_fused_branch_and_phased_upsampler_paths_agree_with_the_unfused_ones():
    """Full JP-Extra shape, a batch large enough for the wide stages' big-launch dispatch (conv_clx, the phased transposed convolutions, the fused branch
    at 128 channels): the k = 3 branches fused into one launch per stage (resbranch_clx.hip, the default) against three fused steps / six conv_clx launches
    (sbv2_debug_set_resbranch(0)), and the phased conv_clx transposed convolutions against conv_cl's phase groups (sbv2_debug_set_upx(0)).  The narrow
    stages' fused branch gives the bits of the steps it replaces; the 128-channel one and the upsamplers replace launches that sum in another order: the
    waveforms agree to 2e-6 (peak ~0.1; the oracle tolerance of the decoder tests is 5e-5)."""
    cfg, W = weights("vits", "full")
    s = model.load_model(blob("vits", "full"), False)
    utts = make_utts([40, 64, 25, 51], O.DEBERTA_FULL, cfg, seed0=91)      # 1 264 frames: 316 tiles at the 256-channel stage
    lib = _lib.lib()
    prev_rb, prev_up = lib.sbv2_debug_set_resbranch(1), lib.sbv2_debug_set_upx(1)
    try:
        a = model.synthesize_batch(s, utts, forced=True)
        lib.sbv2_debug_set_resbranch(0)
        b = model.synthesize_batch(s, utts, forced=True)
        lib.sbv2_debug_set_upx(0)
        c = model.synthesize_batch(s, utts, forced=True)
    finally:
        lib.sbv2_debug_set_resbranch(prev_rb)
        lib.sbv2_debug_set_upx(prev_up)
    worst = 0.0
    for x, y, z in zip(a, b, c):
        assert np.isfinite(x).all()
        np.testing.assert_allclose(x, y, atol=2e-6, rtol=0)
        np.testing.assert_allclose(y, z, atol=2e-6, rtol=0)
        worst = max(worst, float(np.abs(x - z).max()))
    print(f"fused branches + phased upsamplers vs the unfused decoder: worst max-abs {worst:.2e}")
    s.close()


def test_flow_attention_on_presplit_keys_values_same_bits():
    """The flow's split-bf16 attention on keys / values pre-split by the q | k | v product's epilogue (attn_flash.hip k_vits_flash_x3q: LDS-DMA tiles,
    software-pipelined steps; and k_vits_flash_x3p: 64-key staged tiles; the default from 4096 frames and for launches of <= 64 workgroups) against
    the kernel that converts every key tile while staging it
    (sbv2_debug_set_flash_parts(0)): every sample identical, bit for bit, on a mixed batch whose lengths end inside a 32-key step, on a
    64-key tile edge, and below one tile, at the full shape (2 heads x 96) and the tiny one (head dimension 16: padded rows)."""
    lib = _lib.lib()
    for size, lens in (("full", [12, 31, 5, 9]), ("tiny", [40, 3, 17])):
        cfg, W = weights("vits", size)
        s = model.load_model(blob("vits", size), False)
        utts = make_utts(lens, O.DEBERTA_FULL if size == "full" else O.DEBERTA_TINY, cfg, seed0=177)
        # forced durations chosen so that the frame counts hit 64 k, 64 k + 32 and odd remainders
        for u, f in zip(utts, (128, 96, 37, 64)):
            d = np.ones_like(u["forced_durations"]); d[0] = max(1, f - (d.size - 1)); u["forced_durations"] = d
        prev = lib.sbv2_debug_set_flash_parts(2)      # 2 = at every length (1, the default, takes it from 4096 frames): k_vits_flash_x3q
        try:
            a = model.synthesize_batch(s, utts, forced=True)
            lib.sbv2_debug_set_flash_parts(3)         # ... on the un-pipelined kernel k_vits_flash_x3p
            c = model.synthesize_batch(s, utts, forced=True)
            lib.sbv2_debug_set_flash_parts(4)         # ... on the 8-wave (256-query) shape large batches take
            e = model.synthesize_batch(s, utts, forced=True)
            lib.sbv2_debug_set_flash_parts(0)
            b = model.synthesize_batch(s, utts, forced=True)
        finally:
            lib.sbv2_debug_set_flash_parts(prev)
        for x, y, z, v in zip(a, b, c, e):
            np.testing.assert_array_equal(x, y)
            np.testing.assert_array_equal(z, y)
            np.testing.assert_array_equal(v, y)
        s.close()


@pytest.mark.parametrize("name", ["vits_tiny_e2e.npz", "vits_full_e2e.npz"])
def test_vits_e2e_golden(golden_dir, name):
    """HIP traces vs the torch composition of transformers' modules (tests/golden/make_golden.py::e2e_case), NOT via the numpy oracle:
    text-encoder output, prior stats, log-durations, integer durations, expanded prior, flow output, waveform.  Case a: sdp_ratio 0, no
    noise; case b: sdp_ratio 0.25 with the fixture's duration noise (the library's counter-based generator reproduces it from the key)."""
    z, cfg = _golden(golden_dir, name)
    W = synth.make_vits_weights(cfg, int(z["seed"]))
    s = model.load_model(synth.pack_blob(synth.KIND_VITS, cfg, W), False)
    model.set_trace(s, True)
    T = len(z["phones"])
    for tag in ("a", "b"):
        nsw, key = float(z[f"noise_scale_w_{tag}"]), int(z[f"noise_key_{tag}"])
        if nsw:
            np.testing.assert_array_equal(oracle_noise_w(key, 0, T, nsw), z[f"noise_w_{tag}"])   # same noise as the fixture's
        pcm = model.synthesize(s, z["bert"], z["phones"], [int(z["sid"])], z["tones"], z["langs"], z["style"], float(z[f"sdp_ratio_{tag}"]),
                               float(z[f"length_scale_{tag}"]), 0.0, nsw, key)
        np.testing.assert_allclose(model.get_trace(s, "x"), z["x"], atol=1e-4, rtol=0)
        np.testing.assert_allclose(model.get_trace(s, "stats"), z["stats"], atol=1e-4, rtol=0)
        d, lw = model.fetch_durations(s, T)
        np.testing.assert_allclose(lw, z[f"logw_{tag}"], atol=1e-3 if nsw else 2e-4, rtol=0)
        assert np.array_equal(d, z[f"dur_{tag}"])          # the fixture keeps every duration >= 1.5e-3 (relative) off the ceil edge
        np.testing.assert_allclose(model.get_trace(s, "z_p"), z[f"z_p_{tag}"], atol=1e-4, rtol=0)
        np.testing.assert_allclose(model.get_trace(s, "z"), z[f"z_{tag}"], atol=5e-4, rtol=0)
        assert pcm.shape == (1, 1, z[f"pcm_{tag}"].shape[0])
        np.testing.assert_allclose(pcm[0, 0], z[f"pcm_{tag}"], atol=2e-4, rtol=0)
    s.close()


@pytest.mark.parametrize("mode,tol", [("f32", 5e-5), ("bf16x3", 2e-4), ("bf16", 5e-2), ("f16", 5e-4)])
def test_vits_decoder_cl_modes(mode, tol):
    """Every decoder arithmetic (SBV2_DECODER = f32 | bf16x3 | bf16 | f16) against the oracle: tiny batch + full-shape utterance."""
    os.environ["SBV2_DECODER"] = mode
    try:
        cfg = dict(O.VITS_TINY, up_initial=128)     # decoder channels 64/32/16: the bf16 MFMA needs multiples of 16
        W = synth.make_vits_weights(cfg, 5)
        s = model.load_model(synth.pack_blob(synth.KIND_VITS, cfg, W), False)
        utts = make_utts([6, 13, 8], O.DEBERTA_TINY, cfg, seed0=31)
        pcms = model.synthesize_batch(s, utts, forced=True)
        for i, (u, got) in enumerate(zip(utts, pcms)):
            r = _oracle_utt(W, cfg, u, i, 0.0, 1.0, 0.0, 0.0, 0, True)
            np.testing.assert_allclose(got, r["pcm"], atol=tol, rtol=0)
            alone = model.synthesize_batch(s, [u], forced=True)[0]
            np.testing.assert_array_equal(alone, got)
        s.close()
        cfg, W = weights("vits", "full")
        s = model.load_model(blob("vits", "full"), False)
        u = make_utts([12], O.DEBERTA_FULL, cfg, seed0=41)[0]
        pcm = model.synthesize(s, u["bert"], u["phones"], [0], u["tones"], u["langs"], u["style"], 0.0, 1.0, 0.0, 0.0)
        r = _oracle_utt(W, cfg, u, 0, 0.0, 1.0, 0.0, 0.0, 0, False)
        err = float(np.abs(pcm[0, 0] - r["pcm"]).max())
        print(f"decoder {mode}: full-shape waveform max-abs error {err:.3e} (peak |pcm| {np.abs(r['pcm']).max():.3f})")
        assert err < tol
        s.close()
    finally:
        os.environ.pop("SBV2_DECODER", None)


def test_full_shapes_mixed_batch_and_long_utterance():
    """BASELINE configs 4/5 as parity cases: a mixed-length full-shape batch (32..512 phone symbols) and one long utterance
    (600 symbols -> 4201 frames, 48.8 s of audio) through the whole pipeline; a subset is checked against the oracle (torch CPU
    convolutions), every waveform for length, finiteness and |x| < 1."""
    bc, bw = weights("bert", "full")
    vc, vw = weights("vits", "full")
    bs, vs = model.load_model(blob("bert", "full"), True), model.load_model(blob("vits", "full"), False)
    pipe = model.Pipeline(bs, vs)
    ns = [32, 512, 77, 128, 300, 45, 256, 33]
    utts = [synth.make_utterance(n, bc, vc, seed=900 + i, chars=min(98, max(1, n // 2 - 2))) for i, n in enumerate(ns)]
    b = pipe.prepare(utts, forced=True)
    pipe.run(b)
    pcms = pipe.fetch(b)
    O.set_conv_backend("torch")
    try:
        for i, (u, got) in enumerate(zip(utts, pcms)):
            assert got.shape[0] == 512 * (7 * ns[i] + 1) and np.isfinite(got).all() and np.abs(got).max() < 1.0
            if i in (0, 2):
                h = O.deberta_forward(bw, bc, u["input_ids"])
                ref = O.vits_forward(vw, vc, O.expand_bert_features(h, u["word2ph"]), u["phones"], u["tones"], u["langs"], 0, u["style"],
                                     forced_durations=u["forced_durations"])
                np.testing.assert_allclose(got, ref, atol=1e-3, rtol=0)
                assert np.abs(got - ref).max() < 5e-5
        u = synth.make_utterance(600, bc, vc, seed=990, chars=98)
        b = pipe.prepare([u], forced=True)
        pipe.run(b)
        got = pipe.fetch(b)[0]
        h = O.deberta_forward(bw, bc, u["input_ids"])
        ref = O.vits_forward(vw, vc, O.expand_bert_features(h, u["word2ph"]), u["phones"], u["tones"], u["langs"], 0, u["style"],
                             forced_durations=u["forced_durations"])
        assert got.shape == ref.shape == (512 * 4201,)
        np.testing.assert_allclose(got, ref, atol=1e-3, rtol=0)
    finally:
        O.set_conv_backend("numpy")
    pipe.close(); bs.close(); vs.close()


def _oracle_pipeline(bw, bc, vw, vc, u):
    h = O.deberta_forward(bw, bc, u["input_ids"])
    return O.vits_forward(vw, vc, O.expand_bert_features(h, u["word2ph"]), u["phones"], u["tones"], u["langs"], 0, u["style"],
                          forced_durations=u["forced_durations"])


def test_config1_b1_u128_fp32_full_path():
    """BASELINE configs[1]: batch 1, 128 phonemes, fp32 everywhere (exact-f32 MFMA decoder, GEMMs and attention), full DeBERTa + VITS +
    HiFi-GAN shapes through the pipeline; waveform vs the oracle within 1e-4 (north_star: 1e-3)."""
    env = dict(SBV2_DECODER="f32", SBV2_GEMM="f32", SBV2_ATTN="f32")
    os.environ.update(env)
    try:
        bc, bw = weights("bert", "full")
        vc, vw = weights("vits", "full")
        bs, vs = model.load_model(blob("bert", "full"), True), model.load_model(blob("vits", "full"), False)
        assert _lib.lib().sbv2_vits_decoder_mode(vs.handle) == 0
        pipe = model.Pipeline(bs, vs)
        u = synth.make_utterance(128, bc, vc, seed=4100)
        b = pipe.prepare([u], forced=True)
        pipe.run(b)
        got = pipe.fetch(b)[0]
        O.set_conv_backend("torch")
        try:
            ref = _oracle_pipeline(bw, bc, vw, vc, u)
        finally:
            O.set_conv_backend("numpy")
        assert got.shape == ref.shape == (512 * 897,)
        err = float(np.abs(got - ref).max())
        print(f"configs[1] fp32 full path: waveform max-abs error {err:.3e}")
        assert err < 1e-4
        pipe.close(); bs.close(); vs.close()
    finally:
        for k in env:
            os.environ.pop(k, None)


def test_config2_b32_u128_default_path():
    """BASELINE configs[2] (the bench workload): batch 32 x 128 phonemes, default arithmetic (split-bf16 MFMA decoder).  Every one of the
    32 waveforms: length, finiteness, |x| < 1 and agreement with a batch-1 call of the same utterance; two of them vs the oracle, all 32 vs the C oracle.
    Batch row vs single call: since round 5 the wide decoder stages of a LARGE launch run on conv_clx.hip's 16x16x32 MFMAs (another summation order than
    conv_cl.hip's, which small launches keep): the two agree to f32 rounding (measured 1.9e-6 on these 0.1-peak waveforms; tolerance 5e-6), and bit for
    bit when the batch is kept on conv_cl too (sbv2_debug_set_clx(0): the rounds 1-4 invariant, still held on that path)."""
    bc, bw = weights("bert", "full")
    vc, vw = weights("vits", "full")
    bs, vs = model.load_model(blob("bert", "full"), True), model.load_model(blob("vits", "full"), False)
    pipe = model.Pipeline(bs, vs)
    utts = [synth.make_utterance(128, bc, vc, seed=i) for i in range(32)]
    b = pipe.prepare(utts, forced=True)
    pipe.run(b)
    pcms = pipe.fetch(b)
    assert len(pcms) == 32
    lib = _lib.lib()
    prev = lib.sbv2_debug_set_clx(0)     # everything on conv_cl: batch row == single call, bit for bit
    prev_ks = lib.sbv2_debug_set_ksplit(0)   # ... and the single call's DeBERTa products on the batch's summation order (no K split)
    try:
        b0 = pipe.prepare(utts, forced=True)
        pipe.run(b0)
        pcms_cl = pipe.fetch(b0)
        for i, u in enumerate(utts):
            b1 = pipe.prepare([u], forced=True)
            pipe.run(b1)
            np.testing.assert_array_equal(pipe.fetch(b1)[0], pcms_cl[i])
    finally:
        lib.sbv2_debug_set_clx(prev)
        lib.sbv2_debug_set_ksplit(prev_ks)
    worst_row = 0.0
    for i, (u, got) in enumerate(zip(utts, pcms)):   # the default dispatch (a single call's 256-channel stage is too small for conv_clx, the batch's is not)
        assert got.shape == (512 * 897,) and np.isfinite(got).all() and np.abs(got).max() < 1.0
        b1 = pipe.prepare([u], forced=True)
        pipe.run(b1)
        worst_row = max(worst_row, float(np.abs(pipe.fetch(b1)[0] - got).max()))
    print(f"configs[2]: batch row vs single call under the default dispatch, worst max-abs {worst_row:.3e}")
    assert worst_row < 5e-6
    O.set_conv_backend("torch")
    try:
        for i in (0, 17):
            ref = _oracle_pipeline(bw, bc, vw, vc, utts[i])
            err = float(np.abs(pcms[i] - ref).max())
            print(f"configs[2] utterance {i}: waveform max-abs error {err:.3e}")
            assert err < 5e-5
    finally:
        O.set_conv_backend("numpy")
    # ... and ALL 32 against the C / OpenMP restatement (oracle/sbv2_ref.c, itself pinned to the same golden vectors: tests/test_oracle_c.py)
    import sbv2_ref as R
    lib = R.load()
    lib.sbv2c_set_threads(R.usable_cpus())
    m = R.Model(blob("bert", "full"), blob("vits", "full"), lib=lib)
    worst = 0.0
    for u, got in zip(utts, pcms):
        h = m.bert(u["input_ids"], None, hidden=bc["hidden"])
        bert = np.repeat(h, np.asarray(u["word2ph"], np.int64), axis=0).T.copy()
        ref = m.vits(bert, u["phones"], u["tones"], u["langs"], 0, u["style"], forced_durations=u["forced_durations"])
        assert ref.shape == got.shape
        worst = max(worst, float(np.abs(got - ref).max()))
    m.close()
    print(f"configs[2]: worst waveform max-abs error of the 32 utterances vs the C oracle {worst:.3e}")
    assert worst < 5e-5
    pipe.close(); bs.close(); vs.close()


def test_batch_row_vs_single_call_predicted_durations():
    """Under the DEFAULT dispatch a single call and its batch row run different summation orders in DeBERTa (small-grid K splits, LayerNorm launch shapes),
    and DeBERTa's features feed ceil(exp(logw) * length_scale): with PREDICTED durations (every other single-vs-batch test forces them) an utterance may get
    another integer frame count when it is co-batched only where exp(logw) sits on a ceil edge.  32 full-shape utterances (8 224 symbols): features agree to
    f32 rounding, log-durations to 1e-5 in the median (1e-3 at worst), and every symbol whose integer duration differs is within 1e-4 relative of an integer (at most 2 of them: the
    f32-vs-f32 control of tools/flip_rate_bert.py is 2 per 205 600); on the size-independent dispatch the integers are equal.  INTEGRATION.md states it."""
    bc, bw = weights("bert", "full")
    vc, vw = weights("vits", "full")
    bs, vs = model.load_model(blob("bert", "full"), True), model.load_model(blob("vits", "full"), False)
    utts = [synth.make_utterance(128, bc, vc, seed=7300 + i) for i in range(32)]
    ones = [np.ones_like(u["forced_durations"]) for u in utts]      # forced 1-frame durations keep the decoder cheap; the PREDICTIONS are compared

    def run(batched):
        if batched:
            hs = model.predict_batch(bs, [u["input_ids"] for u in utts])
        else:
            hs = [model.predict(bs, u["input_ids"], u["attention_mask"]) for u in utts]
        feats = [np.repeat(h, np.asarray(u["word2ph"], np.int64), axis=0).T.copy() for h, u in zip(hs, utts)]
        part = [dict(u, bert=f, forced_durations=o) for u, f, o in zip(utts, feats, ones)]
        d, lw = [], []
        if batched:
            model.synthesize_batch(vs, part, sdp_ratio=0.5, forced=True, fetch=False)
            return feats, *model.fetch_durations(vs, sum(u["T_text"] for u in utts))
        for q in part:
            model.synthesize_batch(vs, [q], sdp_ratio=0.5, forced=True, fetch=False)
            a, b = model.fetch_durations(vs, q["T_text"])
            d.append(a); lw.append(b)
        return feats, np.concatenate(d), np.concatenate(lw)

    fb, db, lb = run(True)
    f1, d1, l1 = run(False)
    dfeat = max(float(np.abs(a - b).max()) for a, b in zip(fb, f1))
    flips = np.nonzero(db != d1)[0]
    print(f"single call vs batch row, predicted durations: features max-abs {dfeat:.2e}, logw max-abs {float(np.abs(lb - l1).max()):.2e}, "
          f"{flips.size} of {db.size} integer durations differ")
    # (the spline of the stochastic predictor amplifies locally: tools/flip_rate_bert.py's f32-vs-f32 control has median 1e-6 / max 7e-4 on log-durations)
    assert dfeat < 2e-5 and float(np.median(np.abs(lb - l1))) < 1e-5 and float(np.abs(lb - l1).max()) < 1e-3
    w = np.exp(lb.astype(np.float64))
    for i in flips:
        assert abs(w[i] - np.round(w[i])) < 1e-4 * w[i], f"symbol {i}: duration {db[i]} vs {d1[i]} away from a ceil edge (w = {w[i]})"
    assert flips.size <= 2
    with _size_independent_dispatch():
        _, db0, lb0 = run(True)
        _, d10, l10 = run(False)
    np.testing.assert_array_equal(db0, d10)
    np.testing.assert_array_equal(lb0, l10)
    bs.close(); vs.close()


def test_small_grid_dispatch_reaches_the_small_grid_kernels():
    """Dispatch guard (a tile-policy edit once put the ring GEMM in front of the small-grid check: same bits, so no parity test noticed, and
    single-utterance latency went from 13.4 to 15.7 ms): at the default threshold a DeBERTa Linear of a 66-token call must run on gemm_skinny,
    the same product at 2112 tokens on the tiled kernel."""
    import json
    lib = _lib.lib()
    rng = np.random.default_rng(5)
    w = (rng.standard_normal((1024, 1024, 1)) / 32).astype(np.float32)
    b = np.zeros(1024, np.float32)

    def kernels_of(L):
        x = rng.standard_normal((1024, L)).astype(np.float32)
        _lib.check(lib.sbv2_prof_begin())
        _conv_dev(x, w, b, 1, 1.0)
        buf = C.create_string_buffer(1 << 14)
        _lib.check(lib.sbv2_prof_end(buf, len(buf)))
        return {r["kernel"] for r in json.loads(buf.value.decode()) if r["launches"] > 0}

    small, big = kernels_of(66), kernels_of(2112)
    assert any(k.startswith("gemm_skinny") for k in small), small
    assert not any(k.startswith("gemm_skinny") for k in big) and any(k.startswith("conv_gemm") for k in big), big


def test_small_grid_kernels_keep_the_bits():
    """A single-utterance call runs its small grids on gemm_skinny / conv_cl_small; switched off (threshold 0) the same call goes through
    the tiled kernels of a large batch.  Both must give the same waveform bit for bit (predicted durations + noise, 100 phonemes)."""
    bc, bw = weights("bert", "full")
    vc, vw = weights("vits", "full")
    bs, vs = model.load_model(blob("bert", "full"), True), model.load_model(blob("vits", "full"), False)
    pipe = model.Pipeline(bs, vs)
    u = synth.make_utterance(100, bc, vc, seed=77)
    lib = _lib.lib()
    outs = []
    for thr in (0, None):
        prev = lib.sbv2_debug_set_skinny_max(0) if thr == 0 else None
        try:
            b = pipe.prepare([u], sdp_ratio=0.2, noise_scale=0.6, noise_scale_w=0.8, noise_seed=5)
            pipe.run(b)
            outs.append(pipe.fetch(b)[0])
        finally:
            if prev is not None:
                lib.sbv2_debug_set_skinny_max(prev)
    assert outs[0].shape == outs[1].shape and outs[0].size > 0
    np.testing.assert_array_equal(outs[0], outs[1])
    pipe.close(); bs.close(); vs.close()


def test_node_shards_equal_single_gpu_call():
    """sbv2_node_synthesize with four shards on one GPU (the same ordinal four times: peers exchange by device-to-device copies) ==
    one pipeline call of the whole batch, bit for bit, with predicted durations AND noise (streams are keyed by the caller's utterance
    index, not by the position inside a shard); the deal is the library's LPT deal."""
    bc, bw = weights("bert", "tiny", 3)
    vc, vw = weights("vits", "tiny", 5)
    bb, vb = blob("bert", "tiny", 3), blob("vits", "tiny", 5)
    utts = make_utts([7, 15, 4, 22, 9, 3, 11, 6, 18], bc, vc, seed0=151, with_bert=False)
    utts[3]["sid"] = 1
    bs, vs = model.load_model(bb, True), model.load_model(vb, False)
    pipe = model.Pipeline(bs, vs)
    node = model.Node(bb, vb, [0, 0, 0, 0])
    assert not node.uses_rccl
    for kw in (dict(forced=True), dict(sdp_ratio=0.3, length_scale=1.1, noise_scale=0.667, noise_scale_w=0.8, noise_seed=123)):
        b = pipe.prepare(utts, **kw)
        pipe.run(b)
        ref = pipe.fetch(b)
        nb = node.prepare(utts, **kw)
        cap = np.empty(sum(len(r) for r in ref) + 7, np.float32)
        got = node.synthesize(nb, out=cap)
        assert len(got) == len(ref)
        for g, r in zip(got, ref):
            np.testing.assert_array_equal(g, r)
        deal = node.last_deal(len(utts))
        assert set(deal) == {0, 1, 2, 3}
        if kw.get("forced"):
            np.testing.assert_array_equal(deal, model.deal([int(u["forced_durations"].sum()) for u in utts], 4))
        with pytest.raises(model.Sbv2Error, match="too small"):
            node.synthesize(nb, out=np.empty(10, np.float32))
    node.close(); pipe.close(); bs.close(); vs.close()


def test_comm_world1_gather_through_rccl():
    """The one-process-per-GPU communicator with world size 1 on the box's GPU: RCCL is dlopen'ed, ncclCommInitRank / all-reduce / all-gather
    run for real, and the gather returns the run's PCM (by ticket) with the count table.  Runs in a FRESH interpreter (tests/rccl_world1_check.py):
    this pytest process has usually imported torch by now (the oracle's conv backend), whose bundled ROCm runtime libraries make the system
    RCCL fail to find the GPU when it is loaded afterwards; the product (bench.py, a server) never has torch in the process."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "rccl_world1_check.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_WORLD1_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def _ngpus():
    return _lib.lib().sbv2_device_count()


@pytest.mark.skipif(_lib.lib().sbv2_device_count() < 2, reason="needs >= 2 GPUs (RCCL between distinct devices)")
def test_node_two_devices_equal_single_gpu_call():
    """sbv2_node_synthesize on devices [0, 1] (ncclCommInitAll, grouped ncclSend / ncclRecv over xGMI) == one pipeline call of the whole
    batch on one GPU, bit for bit, with predicted durations AND noise.  Skips on a one-GPU box (the same logic runs there with both shards
    on one GPU: test_node_shards_equal_single_gpu_call)."""
    bc, vc = O.DEBERTA_TINY, O.VITS_TINY
    bb, vb = blob("bert", "tiny", 3), blob("vits", "tiny", 5)
    utts = make_utts([7, 15, 4, 22, 9, 3, 11, 6, 18], bc, vc, seed0=151, with_bert=False)
    bs, vs = model.load_model(bb, True), model.load_model(vb, False)
    pipe = model.Pipeline(bs, vs)
    node = model.Node(bb, vb, list(range(min(_ngpus(), 8))))
    assert node.uses_rccl
    for kw in (dict(forced=True), dict(sdp_ratio=0.3, length_scale=1.1, noise_scale=0.667, noise_scale_w=0.8, noise_seed=123)):
        b = pipe.prepare(utts, **kw)
        pipe.run(b)
        ref = pipe.fetch(b)
        got = node.synthesize(node.prepare(utts, **kw), out=np.empty(sum(len(r) for r in ref) + 7, np.float32))
        for g, r in zip(got, ref):
            np.testing.assert_array_equal(g, r)
    node.close(); pipe.close(); bs.close(); vs.close()


@pytest.mark.skipif(_lib.lib().sbv2_device_count() < 2, reason="needs >= 2 GPUs (RCCL between distinct devices)")
def test_comm_two_ranks_gather_over_rccl(tmp_path):
    """One process per GPU, world size 2 (tests/rccl_world2_check.py, two FRESH interpreters): ncclCommInitRank from a shared id,
    the library's deal, every rank synthesises its shard, sbv2_comm_gather_pcm (all-gather of counts + ncclSend / ncclRecv + the chunked
    device -> host copies on rank 0) == the rows of a single-GPU call of the whole batch, bit for bit."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    uid = str(tmp_path / "uid")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    ps = [subprocess.Popen([sys.executable, os.path.join(here, "rccl_world2_check.py"), str(r), "2", uid], stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, text=True, env=env) for r in range(2)]
    outs = [p.communicate(timeout=900)[0] for p in ps]
    for r, (p, o) in enumerate(zip(ps, outs)):
        assert p.returncode == 0 and f"RCCL_WORLD2_RANK{r}_OK" in o, o[-3000:]


def test_config3_b256_mixed_lengths_sharded_8_ways():
    """BASELINE configs[3]: batch 256 of mixed 32..512-phoneme utterances, utterance-sharded 8 ways.  One GPU is all a test box has, so the
    eight shards run on eight execution contexts of that GPU (the library's deal, threads, gather and permutation are the ones an 8-GPU
    node uses; only ncclSend / ncclRecv are replaced by device-to-device copies).  Every utterance: length and finiteness; 6 of them
    against a batch-1 call (5e-6); 2 of them vs the oracle."""
    bc, bw = weights("bert", "full")
    vc, vw = weights("vits", "full")
    bb, vb = blob("bert", "full"), blob("vits", "full")
    rng = np.random.default_rng(256)
    ns = [int(v) for v in rng.integers(32, 513, 256)]
    ns[5], ns[77] = 32, 512
    utts = [synth.make_utterance(n, bc, vc, seed=3000 + i, chars=min(98, max(1, n // 2 - 2))) for i, n in enumerate(ns)]
    node = model.Node(bb, vb, [0] * 8)
    b = node.prepare(utts, forced=True)
    total = sum(512 * (7 * n + 1) for n in ns)
    pin = model.PinnedArray(total)
    got = node.synthesize(b, out=pin.array)
    assert [len(g) for g in got] == [512 * (7 * n + 1) for n in ns]
    deal = node.last_deal(256)
    loads = [sum(7 * ns[i] + 1 for i in range(256) if deal[i] == r) for r in range(8)]
    assert max(loads) - min(loads) <= 7 * 512 + 1
    for g in got:
        assert np.isfinite(g).all() and np.abs(g).max() < 1.0
    bs, vs = model.load_model(bb, True), model.load_model(vb, False)
    pipe = model.Pipeline(bs, vs)
    for i in (0, 5, 77, 100, 200, 255):   # (shard rows on conv_clx's 16x16x32 MFMAs vs single calls on conv_cl: f32 rounding apart, see test_config2)
        b1 = pipe.prepare([utts[i]], forced=True)
        pipe.run(b1)
        np.testing.assert_allclose(pipe.fetch(b1)[0], got[i], atol=5e-6, rtol=0)
    O.set_conv_backend("torch")
    try:
        for i in (5, int(np.argmin(np.abs(np.array(ns) - 128)))):
            ref = _oracle_pipeline(bw, bc, vw, vc, utts[i])
            err = float(np.abs(got[i] - ref).max())
            print(f"configs[3] utterance {i} ({ns[i]} phones): waveform max-abs error {err:.3e}")
            assert err < 5e-5
    finally:
        O.set_conv_backend("numpy")
    pin.close(); node.close(); pipe.close(); bs.close(); vs.close()


def _stream_all(bs, vs, u, chunk, **kw):
    st = model.StreamHandle(bs, vs, u, chunk, **kw)
    parts = []
    while True:
        c = st.next()
        if c is None:
            break
        parts.append(c)
    info = (st.total_samples, st.uses_graph, st.workspace_bytes)
    st.close()
    return np.concatenate(parts), info, [len(p) for p in parts]


def test_streaming_tiny_chunked_equals_whole():
    """Streaming decode (fixed window + halo, one captured hipGraph replayed per chunk) == whole-sequence decode, bit for bit, for chunk
    sizes that do and do not divide the utterance, with predicted durations and noise; a second utterance reuses the captured graph."""
    bc, bw = weights("bert", "tiny", 3)
    vc, vw = weights("vits", "tiny", 5)
    bs, vs = model.load_model(blob("bert", "tiny", 3), True), model.load_model(blob("vits", "tiny", 5), False)
    pipe = model.Pipeline(bs, vs)
    for n, kw in ((40, dict(forced=True)), (23, dict(sdp_ratio=0.2, noise_scale=0.667, noise_scale_w=0.8, noise_seed=5))):
        u = make_utts([n], bc, vc, seed0=171 + n, with_bert=False)[0]
        b = pipe.prepare([u], **kw)
        pipe.run(b)
        whole = pipe.fetch(b)[0]
        for chunk in (16, 50, 64):
            got, (total, graph, ws), sizes = _stream_all(bs, vs, u, chunk, **kw)
            assert graph and total == whole.size == got.size and ws > 0
            assert all(s == chunk * O.hop_length(vc) for s in sizes[:-1])
            np.testing.assert_array_equal(got, whole)
    pipe.close(); bs.close(); vs.close()


def test_config4_streaming_long_form_full_shapes():
    """BASELINE configs[4]: >= 2000 phonemes (T_text 4001, 14001 frames, 162.6 s of audio) streamed in 256-frame chunks through the
    captured decoder graph == the whole-sequence decode within 1e-5 (same arithmetic, same summation order: expected bit-equal);
    a 600-symbol utterance streamed vs the oracle within north_star's 1e-3; the chunk decoder's workspace does not grow with the utterance.
    The DeBERTa inputs have the front end's own length (one token per character): 1000 tokens for the 2000 phonemes, 300 for the 600 symbols, so the
    long-sequence attention (k_deberta_attn_long) is inside every one of these runs (the reference caps at 100 tokens per sentence, model.rs:14-16)."""
    bc, bw = weights("bert", "full")
    vc, vw = weights("vits", "full")
    bs, vs = model.load_model(blob("bert", "full"), True), model.load_model(blob("vits", "full"), False)
    pipe = model.Pipeline(bs, vs)
    u = synth.make_utterance(2000, bc, vc, seed=991)
    assert u["S"] == 1000
    b = pipe.prepare([u], forced=True)
    pipe.run(b)
    whole = pipe.fetch(b)[0]
    assert whole.shape == (512 * 14001,)
    got, (total, graph, ws_long), sizes = _stream_all(bs, vs, u, 256, forced=True)
    assert graph and total == whole.size and len(sizes) == 55
    err = float(np.abs(got - whole).max())
    print(f"configs[4]: 2000 phonemes, 55 chunks of 256 frames, chunked vs whole-sequence max-abs {err:.3e}, chunk workspace {ws_long / 2**20:.0f} MiB")
    assert err <= 1e-5
    u6 = synth.make_utterance(600, bc, vc, seed=990)
    assert u6["S"] == 300
    got6, (_, graph6, ws_short), _ = _stream_all(bs, vs, u6, 256, forced=True)
    assert graph6 and ws_short == ws_long          # bounded by the window, not by the utterance
    O.set_conv_backend("torch")
    try:
        ref = _oracle_pipeline(bw, bc, vw, vc, u6)
    finally:
        O.set_conv_backend("numpy")
    assert got6.shape == ref.shape
    e6 = float(np.abs(got6 - ref).max())
    print(f"configs[4]: 600 symbols streamed vs oracle max-abs {e6:.3e}")
    assert e6 < 1e-3 and e6 < 5e-5
    # predicted durations + both noise streams (seeded) at the full shape: the streamed decode equals the whole-sequence call of the same seed
    kw = dict(sdp_ratio=0.2, noise_scale=0.667, noise_scale_w=0.8, noise_seed=1234)
    un = synth.make_utterance(300, bc, vc, seed=992)
    bn = pipe.prepare([un], **kw)
    pipe.run(bn)
    whole_n = pipe.fetch(bn)[0]
    got_n, (total_n, graph_n, _), _ = _stream_all(bs, vs, un, 256, **kw)
    assert graph_n and total_n == whole_n.size == got_n.size
    en = float(np.abs(got_n - whole_n).max())
    print(f"configs[4]: 300 phonemes, predicted durations + noise, streamed vs whole-sequence max-abs {en:.3e}")
    assert en <= 1e-5
    pipe.close(); bs.close(); vs.close()


def test_pipeline_tiny():
    """DeBERTa -> word2ph repeat -> VITS on the device equals predict + expand + synthesize through the host."""
    bc, bw = weights("bert", "tiny", 3)
    vc, vw = weights("vits", "tiny", 5)
    vc2 = dict(vc)
    bs, vs = model.load_model(blob("bert", "tiny", 3), True), model.load_model(blob("vits", "tiny", 5), False)
    assert bc["hidden"] == vc2["bert_dim"]
    utts = make_utts([7, 15, 4], bc, vc, seed0=51, with_bert=False)
    pipe = model.Pipeline(bs, vs)
    b = pipe.prepare(utts, forced=True)
    pipe.run(b)
    pcms = pipe.fetch(b)
    for u, got in zip(utts, pcms):
        h = O.deberta_forward(bw, bc, u["input_ids"])
        bert = O.expand_bert_features(h, u["word2ph"])
        ref = O.vits_forward(vw, vc, bert, u["phones"], u["tones"], u["langs"], 0, u["style"], forced_durations=u["forced_durations"])
        np.testing.assert_allclose(got, ref, atol=2e-4, rtol=0)
    pipe.close(); bs.close(); vs.close()


def test_pipeline_contexts_do_not_share_events():
    """Execution contexts cloned from sessions that have already run (an earlier Pipeline, a StreamHandle) own their stream-ordering
    event: Pipeline -> run -> close -> Pipeline on the SAME sessions, and a stream followed by a Pipeline, keep working and keep their bits
    (a cloned context used to copy the lazily created event and destroy it with the first pipeline)."""
    bc, vc = O.DEBERTA_TINY, O.VITS_TINY
    bs, vs = model.load_model(blob("bert", "tiny", 3), True), model.load_model(blob("vits", "tiny", 5), False)
    utts = make_utts([7, 15, 4], bc, vc, seed0=51, with_bert=False)
    outs = []
    for _ in range(3):
        pipe = model.Pipeline(bs, vs)
        for _ in range(3):      # both execution contexts of the pipeline get used
            b = pipe.prepare(utts, forced=True)
            pipe.run(b)
            outs.append(pipe.fetch(b))
        pipe.close()
    streamed = _stream_all(bs, vs, utts[1], 16, forced=True)[0]
    pipe = model.Pipeline(bs, vs)
    b = pipe.prepare(utts, forced=True)
    pipe.run(b)
    outs.append(pipe.fetch(b))
    pipe.close()
    for o in outs[1:]:
        for x, y in zip(o, outs[0]):
            np.testing.assert_array_equal(x, y)
    np.testing.assert_allclose(streamed, outs[0][1], atol=1e-5, rtol=0)
    bs.close(); vs.close()


def test_pipeline_shortest_and_ragged_inputs():
    """Edge shapes: the shortest utterances the front end can produce (1 and 2 phones -> T_text 3 and 5, one BERT character -> S = 3), alone
    and batched next to a 40-phone neighbour, with forced and with predicted durations (noise off): every waveform vs the oracle, the
    short ones bit-equal between the ragged batch and their single calls; an empty batch is an error, not a crash."""
    bc, bw = weights("bert", "tiny", 3)
    vc, vw = weights("vits", "tiny", 5)
    bs, vs = model.load_model(blob("bert", "tiny", 3), True), model.load_model(blob("vits", "tiny", 5), False)
    pipe = model.Pipeline(bs, vs)
    utts = [synth.make_utterance(n, bc, vc, seed=400 + n, chars=c) for n, c in ((1, 1), (40, 18), (2, 1), (1, 1))]
    for kw in (dict(forced=True), dict(sdp_ratio=0.0, length_scale=1.0, noise_scale=0.0, noise_scale_w=0.0)):
        b = pipe.prepare(utts, **kw)
        pipe.run(b)
        pcms = pipe.fetch(b)
        assert len(pcms) == 4
        for i, (u, got) in enumerate(zip(utts, pcms)):
            bert = O.expand_bert_features(O.deberta_forward(bw, bc, u["input_ids"]), u["word2ph"])
            fd = u["forced_durations"] if "forced" in kw else None
            ref = O.vits_forward(vw, vc, bert, u["phones"], u["tones"], u["langs"], 0, u["style"], forced_durations=fd, sdp_ratio=0.0)
            assert got.shape == (ref.shape[-1],) and got.size > 0 and np.isfinite(got).all(), (i, got.shape, ref.shape)
            np.testing.assert_allclose(got, ref.reshape(-1), atol=2e-4, rtol=0)
            if u["T_text"] <= 5:
                b1 = pipe.prepare([u], **kw)
                pipe.run(b1)
                np.testing.assert_array_equal(pipe.fetch(b1)[0], got)
    with pytest.raises((model.Sbv2Error, ValueError)):
        pipe.run(pipe.prepare([], forced=True))
    pipe.close(); bs.close(); vs.close()


def test_orchestrator_request_tiny():
    """tts.rs:280-349 through ONE batched pipeline call == the reference's per-sentence loop (oracle), including the rule that
    22050 zero samples follow every sentence that is not the request's last LINE (empty lines count), and the WAV framing."""
    import io
    import orchestrator_oracle as OO
    from scipy.io import wavfile
    from sbv2_api_amd import orchestrator as orch
    bc, bw = weights("bert", "tiny", 3)
    vc, vw = weights("vits", "tiny", 5)
    bs, vs = model.load_model(blob("bert", "tiny", 3), True), model.load_model(blob("vits", "tiny", 5), False)
    pipe = model.Pipeline(bs, vs)
    us = make_utts([6, 11, 3], bc, vc, seed0=91, with_bert=False)
    keys = ("input_ids", "word2ph", "phones", "tones", "langs")
    sentences = [{k: us[0][k] for k in keys}, None, {k: us[1][k] for k in keys}, {k: us[2][k] for k in keys}, None]
    sv = np.random.default_rng(9).standard_normal((3, vc["style_dim"])).astype(np.float32) * 0.1
    opts = orch.SynthesizeOptions(style_weight=0.6, length_scale=1.1)
    wav = orch.easy_synthesize(pipe, sentences, sv, style_id=2, speaker_id=0, options=opts, noise_scale=0.0, noise_scale_w=0.0)
    style = OO.get_style_vector(sv, 2, 0.6)

    def synth_one(s):
        bert = O.expand_bert_features(O.deberta_forward(bw, bc, s["input_ids"]), s["word2ph"])
        return O.vits_forward(vw, vc, bert, s["phones"], s["tones"], s["langs"], 0, style, sdp_ratio=0.0, length_scale=1.1)

    ref = OO.easy_synthesize(sentences, synth_one)
    rate, got = wavfile.read(io.BytesIO(wav))
    assert rate == 44100 and got.shape[0] == ref.shape[2]      # same durations, same three gaps (trailing one included)
    np.testing.assert_allclose(got, ref[0, 0], atol=2e-4, rtol=0)
    assert np.all(got[-22050:] == 0.0)
    assert wav[:68] == OO.array_to_wav(ref)[:68]
    with pytest.raises(model.Sbv2Error):
        orch.easy_synthesize(pipe, [None, None], sv)
    pipe.close(); bs.close(); vs.close()


def test_cpp_host_mirror(tmp_path):
    """include/sbv2_core.hpp (C++ mirror of model.rs / bert.rs: load_model, predict, synthesize, Error) through a plain g++ program
    linked against the C ABI only: same results as the oracle, errors surface as exceptions with the library's message."""
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cpp", "sbv2_core_demo")
    assert os.path.exists(exe), "tests/cpp/sbv2_core_demo is built by sbv2-api_amd/csrc/Makefile (__graft_entry__.build())"
    bc, bw = weights("bert", "tiny", 3)
    vc, vw = weights("vits", "tiny", 5)
    u = make_utts([8], bc, vc, seed0=77, with_bert=False)[0]
    h = O.deberta_forward(bw, bc, u["input_ids"])
    bert_ori = O.expand_bert_features(h, u["word2ph"])
    d = str(tmp_path)
    open(os.path.join(d, "bert.blob"), "wb").write(blob("bert", "tiny", 3))
    open(os.path.join(d, "vits.blob"), "wb").write(blob("vits", "tiny", 5))
    for name, arr, dt in (("ids.i64", u["input_ids"], "<i8"), ("mask.i64", np.ones_like(u["input_ids"]), "<i8"), ("x.i64", u["phones"], "<i8"),
                          ("tones.i64", u["tones"], "<i8"), ("langs.i64", u["langs"], "<i8"), ("style.f32", u["style"], "<f4"),
                          ("bertori.f32", bert_ori, "<f4")):
        np.ascontiguousarray(arr).astype(dt).tofile(os.path.join(d, name))
    r = subprocess.run([exe, d], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    got_h = np.fromfile(os.path.join(d, "predict.f32"), "<f4").reshape(len(u["input_ids"]), bc["hidden"])
    np.testing.assert_allclose(got_h, h, atol=5e-5, rtol=0)
    ref = O.vits_forward(vw, vc, bert_ori, u["phones"], u["tones"], u["langs"], 0, u["style"])
    got = np.fromfile(os.path.join(d, "pcm.f32"), "<f4")
    assert got.shape == ref.shape
    np.testing.assert_allclose(got, ref, atol=2e-4, rtol=0)


def test_workspace_does_not_accumulate():
    """A handle that has served one large batch gives the workspace back after a run of small ones (Arena::reset trims when the
    capacity is far above what recent passes used), and results are unaffected by the re-allocation."""
    cfg, W = weights("vits", "full")
    s = model.load_model(blob("vits", "full"), False)
    l = _lib.lib()
    small = make_utts([6], O.DEBERTA_FULL, cfg, seed0=301)
    big = make_utts([200, 180, 150, 120], O.DEBERTA_FULL, cfg, seed0=311)
    first = model.synthesize_batch(s, small, forced=True)[0]
    ws_small = l.sbv2_vits_workspace_bytes(s.handle)
    model.synthesize_batch(s, big, forced=True, fetch=False)
    model.synthesize_batch(s, big, forced=True, fetch=False)
    ws_big = l.sbv2_vits_workspace_bytes(s.handle)
    assert ws_big > 4 * ws_small and ws_big >= (2 << 30)
    for _ in range(40):
        got = model.synthesize_batch(s, small, forced=True)[0]
    ws_after = l.sbv2_vits_workspace_bytes(s.handle)
    assert ws_after < ws_big / 2, (ws_small, ws_big, ws_after)
    np.testing.assert_array_equal(got, first)
    s.close()


def test_edge_cases_tiny(vits_tiny, bert_tiny):
    """Smallest inputs the front end can produce and the degenerate duration case (sum(w_ceil) == 0 -> clamp_min(1) frame)."""
    cfg, W = weights("vits", "tiny", 5)
    bc, bw = weights("bert", "tiny", 3)
    # one BERT token, and the 2-token [CLS][SEP] sequence
    for ids in (np.array([1]), np.array([1, 2])):
        np.testing.assert_allclose(model.predict(bert_tiny, ids, np.ones_like(ids)), O.deberta_forward(bw, bc, ids), atol=5e-5, rtol=0)
    # a single text position
    rng = np.random.default_rng(5)
    u = dict(bert=rng.standard_normal((cfg["bert_dim"], 1)).astype(np.float32), phones=np.array([7]), tones=np.array([6]), langs=np.array([1]),
             style=rng.standard_normal(cfg["style_dim"]).astype(np.float32) * 0.1, sid=0, forced_durations=np.array([3]), T_text=1)
    got = model.synthesize_batch(vits_tiny, [u], forced=True)[0]
    ref = O.vits_forward(W, cfg, u["bert"], u["phones"], u["tones"], u["langs"], 0, u["style"], forced_durations=u["forced_durations"])
    np.testing.assert_allclose(got, ref, atol=1e-4, rtol=0)
    # all durations zero in one utterance of a batch: exactly one frame (512 // hop samples here) of "silence through the decoder"
    utts = make_utts([4, 6], O.DEBERTA_TINY, cfg, seed0=71)
    utts[0]["forced_durations"] = np.zeros_like(utts[0]["forced_durations"])
    pcms = model.synthesize_batch(vits_tiny, utts, forced=True)
    assert pcms[0].shape[0] == O.hop_length(cfg)
    for i, (u, got) in enumerate(zip(utts, pcms)):
        r = _oracle_utt(W, cfg, u, i, 0.0, 1.0, 0.0, 0.0, 0, True)
        np.testing.assert_allclose(got, r["pcm"], atol=1e-4, rtol=0)


def test_error_paths(vits_tiny):
    cfg, _ = weights("vits", "tiny", 5)
    u = make_utts([4], O.DEBERTA_TINY, cfg, seed0=61)[0]
    bad = u["phones"].copy(); bad[1] = 10_000
    with pytest.raises(model.Sbv2Error):
        model.synthesize(vits_tiny, u["bert"], bad, [0], u["tones"], u["langs"], u["style"], 0.0, 1.0, 0.0, 0.0)
    with pytest.raises(model.Sbv2Error):
        model.load_model(b"not a model", False)
    # a container whose tensors do not have the shapes its config implies is refused at load time, before any kernel indexes them
    W = dict(weights("vits", "tiny", 5)[1])
    for name, bad_shape in (("dp.norm_1.gamma", (cfg["dp_filter"] + 1,)), ("enc_p.emb.weight", (cfg["n_vocab"], cfg["hidden"] - 4)),
                            ("dec.resblocks.0.convs1.0.weight", (cfg["up_initial"] // 2, cfg["up_initial"] // 2 + 16, cfg["res_kernels"][0]))):
        Wb = dict(W)
        Wb[name] = np.zeros(bad_shape, np.float32)
        with pytest.raises(model.Sbv2Error, match="config implies"):
            model.load_model(synth.pack_blob(synth.KIND_VITS, cfg, Wb), False)
