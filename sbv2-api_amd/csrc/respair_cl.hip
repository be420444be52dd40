// One HiFi-GAN ResBlock1 step fused into one launch for the decoder stages with C <= 64 channels (channels-last planes):
//
//     y' = beta * ( conv2( lrelu( conv1( lrelu(y), dilation d ) + b1 ) ) + b2 + y )  [+ previous contents]  (then column mask)
//
// (HifiGanResidualBlock.forward, transformers modeling_vits.py:455-463 = modules.ResBlock1 upstream.)
// Unfused, this is two launches that move 5 planes through HBM (read y, write t, read t, read y again, write y'); the PMC pass
// showed the 16/32-channel launches at 3.2 TB/s, i.e. bandwidth bound.  Fused, the intermediate t never leaves the CU: a
// workgroup computes t for 256 positions into LDS (already leaky-ReLU'ed, masked and split into bf16 hi/lo: exactly the B-operand
// window conv2 needs), then conv2 for the 256 - (k-1) positions whose taps it covers.  Arithmetic, operand split and summation
// order are those of conv_cl.hip, so results are bit-identical to the two-launch path.
#include <type_traits>

#include "common.h"

namespace sbv2 {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

enum { PREC_BF16 = 0, PREC_BF16X3 = 1, PREC_F16 = 2 };   // as in conv_cl.hip
__device__ __forceinline__ f32x16 rp_mfma(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x16 rp_mfma(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

constexpr int kRpThreads = 256;
constexpr int kRpNT = 256;        // positions of the intermediate per workgroup
constexpr int kRpWin2Rows = 272;  // 256 + tap overrun of the last wave (read for discarded columns only)

template <int I, int N, class F>
__device__ __forceinline__ void rp_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        rp_static_for<I + 1, N>(f);
    }
}

// WM = 32-row tiles of the output channels, one group of four waves each: 1 for C = 16 / 32 (256 threads), 2 for C = 64 (512 threads: the
// 134 KB of LDS at k = 11 allow one workgroup per CU, so the second wave per SIMD has to come from inside the workgroup)
template <int PREC, bool PERSIST, int WM, int DG = -1>   // DG >= 0: the diagnostic instantiation with the compile-time ablation mask DG
__global__ __launch_bounds__(kRpThreads * WM) __attribute__((amdgpu_waves_per_eu(PERSIST ? 1 : (WM > 1 ? 2 : 3)))) void respair_cl_kernel(const ResPairParams p) {
    constexpr bool DIAG = DG >= 0;
    constexpr int ABL = DIAG ? DG : 0;
    // DIAG (sbv2_debug_respair_clock only): the phase boundaries of the workgroup's first tile are stamped with s_memtime into scalar registers and
    // written to p.stamps[16 per workgroup] by thread 0 at the very end (a store in the middle would hold back every later load: one in-order vmcnt)
    unsigned st_[16];   // low words (a workgroup lives < 2^32 cycles); 64-bit stamps pushed the kernel over its SGPR budget into scratch
    auto stamp = [&](auto ic) {
        if constexpr (DIAG) {
            constexpr int i = decltype(ic)::value;
            st_[i] = (unsigned)(i >= 14 ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime());
        }
    };
#define RP_STAMP(i) stamp(std::integral_constant<int, i>{})
    if constexpr (DIAG) {
#pragma unroll
        for (int i = 0; i < 16; ++i) st_[i] = 0;
    }
    RP_STAMP(0);
    RP_STAMP(14);
    constexpr int T = kRpThreads * WM;
    constexpr bool SPLIT = PREC == PREC_BF16X3;
    using elem_t = std::conditional_t<PREC == PREC_F16, _Float16, __bf16>;
    using ex8 = std::conditional_t<PREC == PREC_F16, f16x8, bf16x8>;
    using ex4 = std::conditional_t<PREC == PREC_F16, f16x4, bf16x4>;
    constexpr int PARTS = SPLIT ? 2 : 1;
    constexpr int TN = 2;
    constexpr int MAXW = (kMaxTaps * WM * PARTS * 64 + T - 1) / T;
    constexpr int NX = ((kRpNT + 64) * 4 + T - 1) / T;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int ntaps = p.k, C = p.C, nchunks = C >> 4;
    const int lc = C == 64 ? 6 : (C == 32 ? 5 : 4);   // C is 16, 32 or 64: row addressing by shift (a 64-bit multiply is a quarter-rate VALU op)
    const int h2 = (p.k - 1) / 2, h1 = p.dil * (p.k - 1) / 2;
    const int rows1 = kRpNT + 2 * h1;
    const int wbytes = ntaps * WM * PARTS * 1024;
    char* wsm = smem;
    char* x1_hi = smem + wbytes;
    char* x1_lo = x1_hi + rows1 * 32;
    // The intermediate window ALIASES the conv1 window: it is written after the barrier that ends conv1's last MFMA block (every wave is done
    // reading x1 by then), and the next tile's conv1 window is staged after conv2's last barrier.  LDS per workgroup: weights + max(x1, x2)
    // instead of weights + x1 + x2, i.e. 49 instead of 67 KB at C = 32, k = 7: three workgroups per CU instead of two (the kernel is
    // bound by its staging / barrier phases, MFMA busy 0.34: a third workgroup fills them).  The persistent variant prefetches the next
    // tile's window into registers only, which does not touch LDS before the tile's last barrier either.
    char* x2_hi = p.alias_x2 ? x1_hi : x1_hi + rows1 * 32 * PARTS;   // [chunk][kRpWin2Rows][32 B]
    char* x2_lo = x2_hi + nchunks * kRpWin2Rows * 32;
    // biases and the column-mask bytes of this tile, staged once: read from LDS by the two epilogues instead of three dependent
    // global round trips (~0.7-1 us each on a ~12 us workgroup in the clock-stamp timeline).  Placed behind everything the epilogue's
    // transpose tiles overlay.
    const int main_bytes = p.alias_x2 ? wbytes + max(rows1 * 32 * PARTS, nchunks * kRpWin2Rows * 32 * PARTS)
                                      : wbytes + rows1 * 32 * PARTS + nchunks * kRpWin2Rows * 32 * PARTS;
    float* bias_s = reinterpret_cast<float*>(smem + max(main_bytes, WM * 4 * 64 * 36 * 4));   // [0, 64): b1, [64, 128): b2
    unsigned char* mask_s = reinterpret_cast<unsigned char*>(bias_s + 128);                      // [kRpWin2Rows + 16]: rows of the intermediate

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lcol = lane & 31, lh = lane >> 5;
    const int wm = WM > 1 ? wave >> 2 : 0;     // this wave's 32-row tile of the output channels
    const int wn0 = (wave & 3) * 64;           // ... and its 64 positions
    const int nto = kRpNT - 2 * h2;                 // outputs per workgroup
    const int ntiles = (p.N + nto - 1) / nto;
    int n0 = blockIdx.x * nto;                      // first output position (persistent: tile += gridDim.x)
    int t0 = n0 - h2;                               // first position of the intermediate
    int wstart = t0 - h1;                           // first row of the conv1 window
    const int NB = p.N;
    const int nwf4 = ntaps * WM * PARTS * 64;
    const int nxf4 = rows1 * 4;

    f32x16 acc[TN];
    auto zero_acc = [&]() {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    };
    zero_acc();

    f32x4v rw[MAXW];
    f32x4v rx[NX], rx1[NX];
    auto load_w = [&](const void* W, int chunk) {
        const f32x4v* src = reinterpret_cast<const f32x4v*>(W) + (int64_t)chunk * ntaps * WM * PARTS * 64;   // nmt == WM
        rp_static_for<0, MAXW>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            rw[i] = src[min(tid + i * T, nwf4 - 1)];
        });
    };
    auto store_w = [&]() {
        rp_static_for<0, MAXW>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const int idx = tid + i * T;
            if (idx < nwf4) reinterpret_cast<f32x4v*>(wsm)[idx] = rw[i];
        });
    };
    // both 16-channel chunks of a 32-channel group of the conv1 window are requested at once (one 128-byte line per position)
    auto load_x = [&](int pair) {
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int idx = min(tid + i * T, nxf4 - 1);
            const int pos = ((ABL & 1) != 0) ? ((idx >> 2) & 63) : min(max(wstart + (idx >> 2), 0), NB - 1);
            const float* src = p.X + ((int64_t)pos << lc) + (WM == 1 ? 0 : pair * 32) + (idx & 3) * 4;
            rx[i] = *reinterpret_cast<const f32x4v*>(src);
            rx1[i] = *reinterpret_cast<const f32x4v*>(nchunks > 1 ? src + 16 : src);
        }
    };
    auto store_x = [&](int chunk) {
        if ((ABL & 8) != 0) return;   // no conversion / LDS stores of the conv1 window (stale LDS)
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int idx = tid + i * T;
            if (idx < nxf4) {
                const int row = idx >> 2, q = idx & 3;
                const int pos = wstart + row;
                f32x4v v = (chunk & 1) ? rx1[i] : rx[i];
                if (pos < 0 || pos >= NB) v = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], v[e] * p.slope);   // leaky ReLU for 0 <= slope <= 1
                const int off = row * 32 + ((((q >> 1) ^ (row >> 3)) & 1) << 4) + ((q & 1) << 3);
                ex4 h;
#pragma unroll
                for (int e = 0; e < 4; ++e) h[e] = (elem_t)v[e];
                *reinterpret_cast<ex4*>(x1_hi + off) = h;
                if (SPLIT) {
                    ex4 l;
#pragma unroll
                    for (int e = 0; e < 4; ++e) l[e] = (elem_t)(v[e] - (float)h[e]);
                    *reinterpret_cast<ex4*>(x1_lo + off) = l;
                }
            }
        }
    };
    // software-pipelined over the taps (two fragment sets, static ping-pong as in conv_cl.hip): the LDS reads of tap t+1 are in
    // flight while the MFMAs of tap t issue, instead of one exposed LDS round trip per tap
    struct Frags {
        ex8 bh[TN], bl[TN], ah, al;
    };
    auto mfma_chunk = [&](const char* bhi, const char* blo, int tap_stride) {
        auto load_frags = [&](Frags& f, int tap) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int row = wn0 + j * 32 + lcol + tap * tap_stride;
                const int off = row * 32 + (((lh ^ (row >> 3)) & 1) << 4);
                f.bh[j] = *reinterpret_cast<const ex8*>(bhi + off);
                if (SPLIT) f.bl[j] = *reinterpret_cast<const ex8*>(blo + off);
            }
            const char* blk = wsm + ((wm * ntaps + tap) * PARTS) * 1024 + lane * 16;
            f.ah = *reinterpret_cast<const ex8*>(blk);
            if (SPLIT) f.al = *reinterpret_cast<const ex8*>(blk + 1024);
        };
        auto mfma_frags = [&](const Frags& f) {
            if ((ABL & 4) != 0) {   // no MFMAs: the fragments are consumed by one VALU op each so that their reads stay
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[j][0] += (float)f.bh[j][0] + (float)f.ah[0] + (SPLIT ? (float)f.bl[j][0] + (float)f.al[0] : 0.f);
                return;
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if (SPLIT) {
                    acc[j] = rp_mfma(f.al, f.bh[j], acc[j]);
                    acc[j] = rp_mfma(f.ah, f.bl[j], acc[j]);
                }
                acc[j] = rp_mfma(f.ah, f.bh[j], acc[j]);
            }
        };
        Frags fa, fb;
        load_frags(fa, 0);
        int tap = 0;
        for (; tap + 2 <= ntaps; tap += 2) {
            load_frags(fb, tap + 1);
            mfma_frags(fa);
            load_frags(fa, min(tap + 2, ntaps - 1));
            mfma_frags(fb);
        }
        if (tap < ntaps) mfma_frags(fa);
    };

    // ---- phase 1: t = lrelu(conv1(lrelu(y)) + b1) on positions [t0, t0 + 256) -> LDS ----------------------------------------
    load_w(p.W1, 0);
    load_x(0);
  for (int tile = blockIdx.x;; tile += gridDim.x) {
    const bool next_tile = PERSIST && tile + (int)gridDim.x < ntiles;
    {
        const float bval = tid < 128 ? (tid < 64 ? p.b1[min(tid, C - 1)] : p.b2[min(tid - 64, C - 1)]) : 0.f;
        constexpr int NMV = (kRpWin2Rows + 16 + T - 1) / T;
        unsigned char mval[NMV];
#pragma unroll
        for (int h = 0; h < NMV; ++h) {
            const int pos = min(max(t0 + tid + h * T, 0), NB - 1);
            mval[h] = p.mask ? p.mask[pos >> p.mask_shift] : (unsigned char)1;
        }
        store_w();
        store_x(0);
        if (tid < 128) bias_s[tid] = bval;
#pragma unroll
        for (int h = 0; h < NMV; ++h)
            if (tid + h * T < kRpWin2Rows + 16) mask_s[tid + h * T] = mval[h];
    }
    __syncthreads();
    RP_STAMP(1);
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const bool more = chunk + 1 < nchunks;
        load_w(more ? p.W1 : p.W2, more ? chunk + 1 : 0);   // the next weights: conv1's next chunk, then conv2's first
        if (WM > 1 && (chunk & 1) && more) load_x((chunk + 1) >> 1);   // C = 64: the second 32-channel group (its registers are free now)
        mfma_chunk(x1_hi, x1_lo, p.dil);
        if (chunk == 0) RP_STAMP(7);
        __syncthreads();
        if (chunk == 0) RP_STAMP(8);
        if (more) {
            store_w();
            store_x(chunk + 1);
            if (chunk == 0) RP_STAMP(9);
            __syncthreads();
            if (chunk == 0) RP_STAMP(10);
        }
    }
    RP_STAMP(2);
    {
        bool keepj[TN];   // the column mask depends on the position only: once per column, not once per channel quad
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int pos = t0 + wn0 + j * 32 + lcol;
            keepj[j] = pos >= 0 && pos < NB && mask_s[wn0 + j * 32 + lcol] != 0;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int co = wm * 32 + 8 * q + 4 * lh;
            if (co >= C || ((ABL & 16) != 0)) continue;
            const f32x4v b4 = *reinterpret_cast<const f32x4v*>(bias_s + co);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int row = wn0 + j * 32 + lcol;
                const bool keep = keepj[j];
                f32x4v v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float t = acc[j][4 * q + e] + b4[e];
                    t = fmaxf(t, t * p.slope);
                    v[e] = keep ? t : 0.f;
                }
                const int off = (co >> 4) * (kRpWin2Rows * 32) + row * 32 + (((((co >> 3) & 1) ^ (row >> 3)) & 1) << 4) + ((co & 7) << 1);
                ex4 h;
#pragma unroll
                for (int e = 0; e < 4; ++e) h[e] = (elem_t)v[e];
                *reinterpret_cast<ex4*>(x2_hi + off) = h;
                if (SPLIT) {
                    ex4 l;
#pragma unroll
                    for (int e = 0; e < 4; ++e) l[e] = (elem_t)(v[e] - (float)h[e]);
                    *reinterpret_cast<ex4*>(x2_lo + off) = l;
                }
            }
        }
    }
    RP_STAMP(3);
    store_w();   // conv2 chunk 0 (requested before the last conv1 MFMA block)
    zero_acc();
    const int n0_cur = n0;
    if (next_tile) {
        // cross-tile prefetch: the next tile's conv1 window is requested now and lands during conv2 + the epilogue of this tile
        n0 += gridDim.x * nto;
        t0 = n0 - h2;
        wstart = t0 - h1;
        load_x(0);
    }
    __syncthreads();
    RP_STAMP(4);

    // the residual rows of the epilogue are requested now (branch-free, clamped) and land behind conv2's MFMAs; they were fetched
    // for the conv1 window a moment ago, so these are L2 hits, but still a dependent ~1 us round trip if left to the epilogue
    f32x4v rres[8];
    {
        const int c4r = min(wm * 32 + (lane & 7) * 4, C - 4);
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int64_t posr = ((ABL & 1) != 0) ? (it * 8 + (lane >> 3)) : min((int64_t)n0_cur + wn0 + it * 8 + (lane >> 3), (int64_t)NB - 1);
            rres[it] = *reinterpret_cast<const f32x4v*>(p.X + (posr << lc) + c4r);
        }
    }

    // ---- phase 2: conv2 over the LDS-resident intermediate --------------------------------------------------------------------
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const bool more = chunk + 1 < nchunks;
        if (more) load_w(p.W2, chunk + 1);
        else if (next_tile) load_w(p.W1, 0);
        mfma_chunk(x2_hi + chunk * (kRpWin2Rows * 32), x2_lo + chunk * (kRpWin2Rows * 32), 1);
        __syncthreads();
        if (more) {
            store_w();
            __syncthreads();
        }
    }

    RP_STAMP(5);
    // ---- epilogue: + b2 + y, beta, accumulate, mask; full 128-byte lines through a per-wave LDS transpose ---------------------
    float* ttile = reinterpret_cast<float*>(smem) + wave * (64 * 36);
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4v v = {acc[j][4 * q], acc[j][4 * q + 1], acc[j][4 * q + 2], acc[j][4 * q + 3]};
            *reinterpret_cast<f32x4v*>(ttile + (j * 32 + lcol) * 36 + 8 * q + 4 * lh) = v;
        }
    const int c4 = wm * 32 + (lane & 7) * 4;
    const f32x4v b4 = *reinterpret_cast<const f32x4v*>(bias_s + 64 + c4);   // (channels >= C are discarded below)
    // accumulate: all eight rows' previous contents are requested back to back (read inside the row loop, each load waited behind the
    // previous row's store: 8-9 us of epilogue instead of 3 in the clock-stamp timeline of this kernel)
    f32x4v rold[8];
    if (p.accumulate) {
        const int c4o = min(c4, C - 4);
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int64_t po = ((ABL & 1) != 0) ? (it * 8 + (lane >> 3)) : min((int64_t)n0_cur + wn0 + it * 8 + (lane >> 3), (int64_t)NB - 1);
            rold[it] = *reinterpret_cast<const f32x4v*>(p.Y + (po << lc) + c4o);
        }
    }
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int row = it * 8 + (lane >> 3);
        const int o = wn0 + row;          // output index inside the workgroup's range
        const int pos = n0_cur + o;                 // < 2^31 (checked by the caller)
        const f32x4v a = *reinterpret_cast<const f32x4v*>(ttile + row * 36 + (lane & 7) * 4);
        if (o >= nto || pos >= NB || c4 >= C) continue;
        const f32x4v r = rres[it];
        f32x4v v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (a[e] + b4[e] + r[e]) * p.beta;
        f32x4v* dst = reinterpret_cast<f32x4v*>(p.Y + ((int64_t)pos << lc) + c4);
        if (p.accumulate) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += rold[it][e];
        }
        if (!mask_s[o + h2]) v = f32x4v{0.f, 0.f, 0.f, 0.f};   // position n0 + o = intermediate row o + h2
        if ((ABL & 2) != 0) continue;
        *dst = v;
    }
    RP_STAMP(6);
    RP_STAMP(15);
    if constexpr (DIAG) {
        if (threadIdx.x == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) p.stamps[(size_t)blockIdx.x * 16 + i] = st_[i];
        }
    }
    if (!next_tile) break;
    zero_acc();
    __syncthreads();   // the transpose tiles overlap the staging regions
  }
}

template <int PREC, bool PERSIST, int WM>
static void launch_rp(const ResPairParams& p, hipStream_t stream) {
    constexpr int T = kRpThreads * WM;
    constexpr int PARTS = PREC == PREC_BF16X3 ? 2 : 1;
    const int h1 = p.dil * (p.k - 1) / 2, h2 = (p.k - 1) / 2;
    const int rows1 = kRpNT + 2 * h1;
    size_t lds = (size_t)p.k * WM * PARTS * 1024 + (p.alias_x2 ? std::max((size_t)rows1 * 32 * PARTS, (size_t)(p.C >> 4) * kRpWin2Rows * 32 * PARTS)
                                                               : (size_t)rows1 * 32 * PARTS + (size_t)(p.C >> 4) * kRpWin2Rows * 32 * PARTS);
    lds = std::max<size_t>(lds, (size_t)WM * 4 * 64 * 36 * sizeof(float));
    lds += 128 * sizeof(float) + kRpWin2Rows + 16;   // biases + mask bytes (kernel: bias_s, mask_s)
    lds = (lds + 15) / 16 * 16;
    SBV2_REQUIRE(lds <= 160 * 1024, "respair: LDS budget exceeded");
    auto kern = respair_cl_kernel<PREC, PERSIST, WM>;
    static std::atomic<uint64_t> lds_allowed{0};   // per (kernel instantiation, device)
    allow_full_lds(reinterpret_cast<const void*>(kern), lds_allowed);
    const int nto = kRpNT - 2 * h2;
    int per_cu = 1;
    HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kern), T, lds));
    per_cu = std::max(1, std::min(per_cu, 4));
    const int ntiles = (p.N + nto - 1) / nto;
    dim3 grid(PERSIST ? std::min(ntiles, 256 * per_cu) : ntiles);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const bool prof = conv_prof_active();
    if (prof) {
        HIP_CHECK(hipEventCreate(&e0));
        HIP_CHECK(hipEventCreate(&e1));
        HIP_CHECK(hipEventRecord(e0, stream));
    }
    hipLaunchKernelGGL(kern, grid, dim3(T), lds, stream, p);
    HIP_CHECK(hipGetLastError());
    if (prof) {
        HIP_CHECK(hipEventRecord(e1, stream));
        conv_prof_add(WM > 1 ? 17 : 16, 2.0 * 2.0 * p.C * (double)p.N * p.C * p.k, e0, e1);
    }
}

// diagnostics (sbv2_debug_respair_clock): the split-bf16 kernel with phase stamps and the ablations of p.abl
void launch_respair_cl_diag(const ResPairParams& p0, hipStream_t stream) {
    ResPairParams p = p0;
    p.mask_shift = 0;
    while (p.mask && (1 << p.mask_shift) < p.mask_div) ++p.mask_shift;
    p.alias_x2 = 1;
    SBV2_REQUIRE(p.split && p.stamps && (p.C == 16 || p.C == 32 || p.C == 64), "respair diag: split-bf16 only");
    constexpr int PARTS = 2;
    const int WM = p.C == 64 ? 2 : 1;
    const int h1 = p.dil * (p.k - 1) / 2, h2 = (p.k - 1) / 2;
    const int rows1 = kRpNT + 2 * h1;
    size_t lds = (size_t)p.k * WM * PARTS * 1024 + std::max((size_t)rows1 * 32 * PARTS, (size_t)(p.C >> 4) * kRpWin2Rows * 32 * PARTS);
    lds = std::max<size_t>(lds, (size_t)WM * 4 * 64 * 36 * sizeof(float));
    lds += 128 * sizeof(float) + kRpWin2Rows + 16;
    lds = (lds + 15) / 16 * 16;
    const int nto = kRpNT - 2 * h2;
    const int ntiles = (p.N + nto - 1) / nto;
    auto go = [&](auto ablc) {
        constexpr int A = decltype(ablc)::value;
        if (WM == 2) {
            auto kern = respair_cl_kernel<PREC_BF16X3, false, 2, A>;
            static std::atomic<uint64_t> a2{0};
            allow_full_lds(reinterpret_cast<const void*>(kern), a2);
            hipLaunchKernelGGL(kern, dim3(ntiles), dim3(512), lds, stream, p);
        } else {
            auto kern = respair_cl_kernel<PREC_BF16X3, false, 1, A>;
            static std::atomic<uint64_t> a1{0};
            allow_full_lds(reinterpret_cast<const void*>(kern), a1);
            hipLaunchKernelGGL(kern, dim3(ntiles), dim3(256), lds, stream, p);
        }
    };
    switch (p.abl) {
        case 0: go(std::integral_constant<int, 0>{}); break;
        case 3: go(std::integral_constant<int, 3>{}); break;
        case 4: go(std::integral_constant<int, 4>{}); break;
        case 8: go(std::integral_constant<int, 8>{}); break;
        case 16: go(std::integral_constant<int, 16>{}); break;
        case 28: go(std::integral_constant<int, 28>{}); break;
        case 31: go(std::integral_constant<int, 31>{}); break;
        default: SBV2_REQUIRE(false, "respair diag: ablation mask not instantiated");
    }
    HIP_CHECK(hipGetLastError());
}

// 0: respair_cl.hip; 1 (default): respair_clx.hip, and respair_x16.hip where it exists (C = 32 / 64, k = 7 / 11); 2 (SBV2_RESPAIR_X16=0): respair_clx.hip only
static std::atomic<int> g_rpx{getenv("SBV2_RESPAIR_X16") && atoi(getenv("SBV2_RESPAIR_X16")) == 0 ? 2 : 1};   // sbv2_debug_set_respair_clx
int set_respair_clx(int on) { return g_rpx.exchange(on); }

void launch_respair_cl(const ResPairParams& p0, hipStream_t stream) {
    ResPairParams p = p0;
    // the column mask is indexed by position / upsampling factor; only powers of two are supported here (a shift: the 64-bit integer
    // division this replaced was ~100 VALU instructions per output row, 8 rows per thread, in a kernel that is VALU-issue bound)
    SBV2_REQUIRE(!p.mask || (p.mask_div > 0 && (p.mask_div & (p.mask_div - 1)) == 0), "respair: mask_div must be a power of two");
    p.mask_shift = 0;
    while (p.mask && (1 << p.mask_shift) < p.mask_div) ++p.mask_shift;
    p.alias_x2 = 1;
    p.abl = 0;   // (ablations exist in the diagnostic instantiation only: launch_respair_cl_diag)
    SBV2_REQUIRE(p.C == 16 || p.C == 32 || p.C == 64, "respair: only the 16-, 32- and 64-channel stages are fused");
    SBV2_REQUIRE(p.k >= 1 && p.k <= kMaxTaps && (p.k & 1) == 1, "respair: odd kernel sizes only");
    SBV2_REQUIRE(p.dil * (p.k - 1) <= 64, "respair: tap span too large");
    if (p.N <= 0) return;
    if (g_rpx.load(std::memory_order_relaxed) == 1 && respair_x16_usable(p)) return launch_respair_x16(p, stream);   // round 6: the 16x16x32 operand scheme
    if (g_rpx.load(std::memory_order_relaxed) && respair_clx_usable(p)) return launch_respair_clx(p, stream);   // the round-4 kernel (same bits at C = 32 / 64; C = 16 sums two taps per MFMA: f32 rounding apart)
    SBV2_REQUIRE(!(p.split && p.f16), "respair: split and f16 are exclusive");
    if (p.C == 64) {
        if (p.split) launch_rp<PREC_BF16X3, false, 2>(p, stream);
        else if (p.f16) launch_rp<PREC_F16, false, 2>(p, stream);
        else launch_rp<PREC_BF16, false, 2>(p, stream);
    } else if (p.split) {
        launch_rp<PREC_BF16X3, false, 1>(p, stream);
    } else if (p.f16) {
        launch_rp<PREC_F16, false, 1>(p, stream);
    } else {
        launch_rp<PREC_BF16, false, 1>(p, stream);
    }
}

}  // namespace sbv2
