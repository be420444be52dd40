// 1x1 products whose grid is too small for the tiled f32 GEMM of gemm_conv.hip (single-utterance calls: the DeBERTa Linear layers at
// 64-130 tokens are 32-64 workgroups on 256 CUs, 47 us per launch whatever the size; the flow's Linear layers at 897 frames 45).
//
//   C[m][n] (+)= epilogue( sum_k A[k][m] * pre(B[k][n]) )                same operands, same epilogue as launch_conv
//
// What makes a second kernel legal here: on gfx950 every f32 MFMA shape (32x32x2, 16x16x4, 4x4x1) accumulates one k at a time with a fused
// multiply-add, i.e. a K-long product is bit-for-bit the scalar chain fmaf(a[K-1], b[K-1], ... fmaf(a[0], b[0], 0)) whatever the tile
// (measured: experiments/mfma_order.hip, 0 differing results of 1024 at K = 64 / 1024 / 4096).  So the tile may follow the problem size
// without changing one output bit, and a batch row stays bit-identical to the single call of the same utterance.
//
// Machine mapping: one WAVE per workgroup computes a 16 x (16 * TN) tile with v_mfma_f32_16x16x4_f32: 320 independent waves for a
// 1024 x 66 output instead of 32 workgroups, each a chain of K / 4 MFMAs (32 cycles each: 3.4 us at K = 1024).  A lone wave per SIMD hides
// no latency by occupancy, so the operands stream through a 16-stage LDS ring filled by global_load_lds_dwordx4 (no staging registers,
// up to 14 chunks = 224 k in flight per wave, counted vmcnt; no barrier anywhere: the wave that issued a DMA is the only reader).
// LDS image of a 16-k chunk: A [16 k][16 m] then TN x B [16 k][16 n], 1 KB each, row-major = lane-linear for the DMA (lane l carries
// row l >> 2, floats 4 * (l & 3) ..) AND for the MFMA fragments (k-step s of lane l is dword 64 s + l): conflict-free ds_read_b32.
#include "common.h"

namespace sbv2 {

namespace {

typedef float f32x4s __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;


struct SkinnyParams {
    ConvParams p;
    int mask_shift;
};

template <int N>
__device__ __forceinline__ void wait_vm() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// ---- epilogue: the arithmetic of gemm_conv.hip's, element for element; every global read before the first store -------------------------
template <int TN>
__device__ __forceinline__ void skinny_epilogue(const SkinnyParams& kp, const f32x4s (&acc)[TN], int m0, int n0, int lane) {
    const ConvParams& p = kp.p;
    const int M = p.M, N = p.N;
    const float* Rg = p.R;
    float rr[TN][4], old[TN][4], bcol[TN], brow[4];
    unsigned char keep[TN];
    const int mbase = m0 + (lane >> 4) * 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) brow[r] = (p.bias_mode == BIAS_ROW) ? p.bias[min(mbase + r, M - 1)] : 0.f;
#pragma unroll
    for (int t = 0; t < TN; ++t) {
        const int n = min(n0 + t * 16 + (lane & 15), N - 1);
        bcol[t] = (p.bias_mode == BIAS_COL) ? p.bias[n] : 0.f;
        keep[t] = 1;
        if (p.mask) keep[t] = p.mask[kp.mask_shift >= 0 ? (n >> kp.mask_shift) : (n / p.mask_div)];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = min(mbase + r, M - 1);
            rr[t][r] = Rg ? Rg[(int64_t)m * p.ldr + n] : 0.f;
            old[t][r] = p.accumulate ? p.C[(int64_t)m * p.ldc + n] : 0.f;
        }
    }
#pragma unroll
    for (int t = 0; t < TN; ++t) {
        const int n = n0 + t * 16 + (lane & 15);
        if (n >= N) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = mbase + r;
            if (m >= M) continue;
            float v = acc[t][r] + brow[r];
            if (p.bias_mode == BIAS_COL) v += bcol[t];
            if (p.act == ACT_RELU) v = fmaxf(v, 0.f);
            else if (p.act == ACT_GELU) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
            else if (p.act == ACT_TANH) v = tanhf(v);
            v *= p.alpha;
            if (Rg) v += rr[t][r];
            v *= p.beta;
            if (p.accumulate) v += old[t][r];
            if (!keep[t]) v = 0.f;
            p.C[(int64_t)m * p.ldc + n] = v;
        }
    }
}

template <int TN, int kStages>
__global__ __launch_bounds__(64) void gemm_skinny_kernel(const SkinnyParams kp) {
    const ConvParams& p = kp.p;
    constexpr int SB = (1 + TN) * 1024;   // bytes per ring slot
    constexpr int PER = 1 + TN;           // DMA instructions per chunk
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x;
    const int m0 = blockIdx.y * 16, n0 = blockIdx.x * (16 * TN);
    const int M = p.M, N = p.N;
    const int nch = p.K >> 4;

    // per-lane source of a chunk: row (lane >> 2) of its 16 k, four floats at column 4 * (lane & 3).  Rows / columns outside the problem
    // are redirected to column 0 (always readable): they only feed output rows / columns that are never stored.
    const int kr = lane >> 2, c4 = (lane & 3) * 4;
    const float* asrc = p.A + (int64_t)kr * p.lda + (m0 + c4 < M ? m0 + c4 : 0);
    const float* bsrc[TN];
#pragma unroll
    for (int t = 0; t < TN; ++t) {
        const int col = n0 + t * 16 + c4;
        bsrc[t] = p.B + (int64_t)kr * p.ldb + (col < N ? col : 0);
    }
    const int64_t astep = (int64_t)16 * p.lda, bstep = (int64_t)16 * p.ldb;
    auto stage = [&](int c) {
        char* dst = smem + (c & (kStages - 1)) * SB;
        __builtin_amdgcn_global_load_lds((gbl_void_t*)(asrc + c * astep), (lds_void_t*)dst, 16, 0, 0);
#pragma unroll
        for (int t = 0; t < TN; ++t)
            __builtin_amdgcn_global_load_lds((gbl_void_t*)(bsrc[t] + c * bstep), (lds_void_t*)(dst + 1024 * (1 + t)), 16, 0, 0);
    };

    f32x4s acc[TN];
#pragma unroll
    for (int t = 0; t < TN; ++t) acc[t] = f32x4s{0.f, 0.f, 0.f, 0.f};
    const float slope = p.pre_slope;
    // the fragments of chunk c + 1 are read from LDS before the MFMAs of chunk c are issued (a lone wave has nobody to hide the ds_read
    // latency behind)
    struct Frags {
        float a[4], b[TN][4];
    };
    auto load_frags = [&](Frags& f, int c) {
        const float* sl = reinterpret_cast<const float*>(smem + (c & (kStages - 1)) * SB) + lane;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            f.a[s] = sl[64 * s];
#pragma unroll
            for (int t = 0; t < TN; ++t) f.b[t][s] = sl[256 * (1 + t) + 64 * s];
        }
    };
    auto mma = [&](const Frags& f) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int t = 0; t < TN; ++t) {
                float b = f.b[t][s];
                if (slope != 1.0f) b = b >= 0.f ? b : b * slope;
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[s], b, acc[t], 0, 0, 0);
            }
    };

    const int pre = min(kStages - 1, nch);
    for (int c = 0; c < pre; ++c) stage(c);
    // loads retire in order: with chunks 0 .. c + 15 issued, "at most 14 chunks outstanding" means chunk c + 1 has landed
    if (nch >= kStages - 1) wait_vm<(kStages - 2) * PER>();
    else wait_vm<0>();
    Frags cur, nxt;
    load_frags(cur, 0);
    int c = 0;
    for (; c + kStages - 1 < nch; ++c) {
        stage(c + kStages - 1);              // into the slot of chunk c - 1, whose fragments were consumed by the MFMAs of the last pass
        wait_vm<(kStages - 2) * PER>();
        load_frags(nxt, c + 1);
        __builtin_amdgcn_sched_barrier(0);   // keep the reads ahead of the MFMA chain (hipcc sinks them behind it otherwise)
        mma(cur);
        cur = nxt;
    }
    wait_vm<0>();
    for (; c < nch; ++c) {
        if (c + 1 < nch) load_frags(nxt, c + 1);
        __builtin_amdgcn_sched_barrier(0);
        mma(cur);
        cur = nxt;
    }

    skinny_epilogue<TN>(kp, acc, m0, n0, lane);
}

// ---- k > 1 (the k = 3 convolutions of the text encoder's FFN and of the duration predictors: 15-60 workgroups tiled, 40-180 us) ----------
// Same chain order as the tiled kernel: chunk (16 k) outermost, then tap, then k.  Per chunk the ring slot holds ntaps weight tiles
// [16 k][16 m] and ONE window [16 k][W columns] that starts at the 16-byte aligned column w0 <= n0 + min shift (W = 32 or 64: 8 or 16
// lanes x 16 bytes per row, so a DMA covers 8 or 4 rows); tap t reads it at column offset shift_t - (w0 - n0).  Columns outside [0, nb)
// are zero by contract: their DMA source is redirected to a readable address and the operand is masked at the read.
struct SkinnyTapParams {
    ConvParams p;
    int mask_shift;
    int W;           // window pitch in floats (32 or 64)
    int w0_rel;      // w0 - n0 (<= min shift, multiple of 4)
    int slot_bytes;  // ntaps KB + W / 16 KB
    int ahead;       // chunks in flight
};
constexpr int kTapSlots = 8;

__device__ __forceinline__ void wait_vm_dyn(int n) {
    switch (n) {   // s_waitcnt takes an immediate; n is wave-uniform
#define W_(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
        W_(0) W_(1) W_(2) W_(3) W_(4) W_(5) W_(6) W_(7) W_(8) W_(9) W_(10) W_(11) W_(12) W_(13) W_(14) W_(15)
        W_(16) W_(17) W_(18) W_(19) W_(20) W_(21) W_(22) W_(23) W_(24) W_(25) W_(26) W_(27) W_(28) W_(29) W_(30) W_(31)
        W_(32) W_(33) W_(34) W_(35) W_(36) W_(37) W_(38) W_(39) W_(40) W_(41) W_(42) W_(43) W_(44) W_(45) W_(46) W_(47)
        W_(48) W_(49) W_(50) W_(51) W_(52) W_(53) W_(54) W_(55) W_(56) W_(57) W_(58) W_(59) W_(60) W_(61) W_(62)
#undef W_
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

__global__ __launch_bounds__(64) void gemm_skinny_taps_kernel(const SkinnyTapParams kp) {
    const ConvParams& p = kp.p;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x;
    const int m0 = blockIdx.y * 16, n0 = blockIdx.x * 16;
    const int M = p.M, nb = p.nb, ntaps = p.ntaps, W = kp.W;
    const int nch = p.K >> 4;
    const int wbytes = ntaps * 1024;
    const int xg = W >> 4;                      // DMAs per window
    const int per = ntaps + xg;
    // weights: row (lane >> 2) of the chunk, floats 4 * (lane & 3) ..; window: row g * (256 / W) + lane / (W / 4), floats 4 * (lane % (W / 4)) ..
    const float* asrc = p.A + (int64_t)(lane >> 2) * p.lda + (m0 + (lane & 3) * 4 < M ? m0 + (lane & 3) * 4 : 0);
    const int lpr = W >> 2, rpg = 64 / lpr;     // lanes per row, rows per DMA
    const int xrow = lane / lpr;
    int xcol = n0 + kp.w0_rel + (lane % lpr) * 4;
    if (xcol < 0 || xcol + 3 >= p.ldb) xcol = 0;
    const float* bsrc = p.B + (int64_t)xrow * p.ldb + xcol;
    auto stage = [&](int c) {
        char* dst = smem + (c & (kTapSlots - 1)) * kp.slot_bytes;
        for (int t = 0; t < ntaps; ++t)
            __builtin_amdgcn_global_load_lds((gbl_void_t*)(asrc + (int64_t)t * p.a_tap_stride + (int64_t)c * 16 * p.lda), (lds_void_t*)(dst + t * 1024), 16, 0, 0);
        for (int g = 0; g < xg; ++g)
            __builtin_amdgcn_global_load_lds((gbl_void_t*)(bsrc + ((int64_t)c * 16 + g * rpg) * p.ldb), (lds_void_t*)(dst + wbytes + g * 1024), 16, 0, 0);
    };
    f32x4s acc[1];
    acc[0] = f32x4s{0.f, 0.f, 0.f, 0.f};
    const float slope = p.pre_slope;
    const int krow = lane >> 4, lcol = lane & 15;
    auto mma_chunk = [&](int c) {
        const float* sl = reinterpret_cast<const float*>(smem + (c & (kTapSlots - 1)) * kp.slot_bytes);
        const float* xw = sl + (wbytes >> 2);
        for (int t = 0; t < ntaps; ++t) {
            const int off = p.shift[t] - kp.w0_rel + lcol;          // column inside the window
            const int g = n0 + kp.w0_rel + off;                       // global column
            const bool ok = g >= 0 && g < nb;
            float a[4], b[4];
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                a[s4] = sl[t * 256 + 64 * s4 + lane];
                b[s4] = xw[(4 * s4 + krow) * W + off];
            }
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                float bv = ok ? b[s4] : 0.f;
                if (slope != 1.0f) bv = bv >= 0.f ? bv : bv * slope;
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s4], bv, acc[0], 0, 0, 0);
            }
        }
    };
    const int ahead = kp.ahead;
    int issued = 0;
    for (; issued < min(ahead, nch); ++issued) stage(issued);
    for (int c = 0; c < nch; ++c) {
        if (issued < nch) {
            stage(issued);        // slot of chunk issued - 8 < c: consumed by the MFMAs of an earlier pass
            ++issued;
        }
        wait_vm_dyn((issued - c - 1) * per);    // loads retire in order: chunk c has landed
        mma_chunk(c);
    }
    skinny_epilogue<1>(SkinnyParams{kp.p, kp.mask_shift}, acc, m0, n0, lane);
}

template <int TN, int kStages>
void launch_skinny(const SkinnyParams& kp, hipStream_t stream) {
    const ConvParams& p = kp.p;
    dim3 grid((p.N + 16 * TN - 1) / (16 * TN), (p.M + 15) / 16);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const bool prof = conv_prof_active();
    if (prof) {
        HIP_CHECK(hipEventCreate(&e0));
        HIP_CHECK(hipEventCreate(&e1));
        HIP_CHECK(hipEventRecord(e0, stream));
    }
    auto kern = gemm_skinny_kernel<TN, kStages>;
    static std::atomic<uint64_t> lds_allowed{0};   // per (kernel instantiation, device)
    allow_full_lds(reinterpret_cast<const void*>(kern), lds_allowed);
    hipLaunchKernelGGL(kern, grid, dim3(64), kStages * (1 + TN) * 1024, stream, kp);
    HIP_CHECK(hipGetLastError());
    if (prof) {
        HIP_CHECK(hipEventRecord(e1, stream));
        conv_prof_add(23, 2.0 * p.M * (double)p.N * p.K, e0, e1);
    }
}

}  // namespace

// true = launched.  The caller (launch_conv) has already decided that the grid of the tiled kernel would be small.
bool launch_gemm_skinny(const ConvParams& p, int mask_shift, hipStream_t stream) {
    if (p.ntaps != 1 || p.groups || p.shift[0] != 0 || (p.K & 15) != 0 || p.K < 16 || p.phase_rows < (1 << 30)) return false;
    if ((p.lda & 3) != 0 || (p.ldb & 3) != 0 || p.N > p.nb) return false;
    SkinnyParams kp;
    kp.p = p;
    kp.mask_shift = mask_shift;
    // one wave per 16 x 16 tile while that is at most one wave per SIMD of the chip; 16 x 32 tiles beyond
    const int64_t waves = (int64_t)((p.M + 15) / 16) * ((p.N + 15) / 16);
    if (waves <= 1024) {
        launch_skinny<1, 16>(kp, stream);
    } else {
        launch_skinny<2, 16>(kp, stream);
    }
    return true;
}

// k > 1 twin; same contract
bool launch_gemm_skinny_taps(const ConvParams& p, int mask_shift, hipStream_t stream) {
    if (p.ntaps < 2 || p.groups || (p.K & 15) != 0 || p.K < 16 || p.phase_rows < (1 << 30)) return false;
    if ((p.lda & 3) != 0 || (p.ldb & 3) != 0 || (p.a_tap_stride & 3) != 0) return false;
    int smin = p.shift[0], smax = p.shift[0];
    for (int t = 1; t < p.ntaps; ++t) {
        smin = std::min(smin, p.shift[t]);
        smax = std::max(smax, p.shift[t]);
    }
    SkinnyTapParams kp;
    kp.p = p;
    kp.mask_shift = mask_shift;
    kp.w0_rel = (smin >= 0) ? (smin / 4) * 4 : -(((-smin) + 3) / 4) * 4;
    const int need = 16 + smax - kp.w0_rel;            // columns w0 .. n0 + 15 + max shift
    if (need > 64) return false;
    kp.W = need <= 32 ? 32 : 64;
    kp.slot_bytes = p.ntaps * 1024 + (kp.W / 16) * 1024;
    const int per = p.ntaps + kp.W / 16;
    kp.ahead = std::max(1, std::min(kTapSlots - 1, 62 / per));
    dim3 grid((p.N + 15) / 16, (p.M + 15) / 16);
    if ((int64_t)grid.x * grid.y > 2048) return false;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const bool prof = conv_prof_active();
    if (prof) {
        HIP_CHECK(hipEventCreate(&e0));
        HIP_CHECK(hipEventCreate(&e1));
        HIP_CHECK(hipEventRecord(e0, stream));
    }
    static std::atomic<uint64_t> lds_allowed{0};   // per (kernel instantiation, device)
    allow_full_lds(reinterpret_cast<const void*>(gemm_skinny_taps_kernel), lds_allowed);
    hipLaunchKernelGGL(gemm_skinny_taps_kernel, grid, dim3(64), kTapSlots * kp.slot_bytes, stream, kp);
    HIP_CHECK(hipGetLastError());
    if (prof) {
        HIP_CHECK(hipEventRecord(e1, stream));
        conv_prof_add(23, 2.0 * p.M * (double)p.N * p.K * p.ntaps, e0, e1);
    }
    return true;
}

}  // namespace sbv2
