// Channels-last implicit-GEMM convolution on the gfx950 bf16 matrix cores (v_mfma_f32_32x32x16_bf16), f32 storage.
//
//   Y[pos][m] (+)= epi( sum_tap sum_k W_tap[m][k] * pre(X[n + shift_tap][k]) ),   pos = n * out_stride + phase_off[m / phase_rows]
//
// Why a second conv kernel: the f32 MFMA of gemm_conv.hip tops out at 157 TFLOP/s; the bf16 MFMA is 16x faster per clock but
// wants 8 CONSECUTIVE k per lane for both operands (A[row][k = 8h + j], B[k = 8h + j][col]), i.e. k-contiguous ("channels-last")
// tiles.  The HiFi-GAN decoder (82 % of the path's FLOPs: SURVEY.md §8a a8) therefore runs on planes X[position][channel].
//
// Precision modes (template SPLIT):
//   SPLIT = true : every f32 operand is split into bf16 hi + bf16 lo (x = hi + lo + O(2^-17 x)) and the product is taken as
//                  hi*hi + hi*lo + lo*hi with f32 accumulation: 3 MFMAs per tile step, relative error ~1e-5 per product,
//                  i.e. f32-grade results (the path stays inside the 1e-3 waveform tolerance) at 3/16 the cost of the f32 MFMA.
//   SPLIT = false: plain bf16 operands (BASELINE.json configs[2] "bf16 MFMA for the decoder GEMMs"), error reported by bench.py.
// Storage stays f32 in HBM in both modes, so rounding does not compound through the residual chain.
//
// Structure: one workgroup = 4 waves = (TM*32 rows) x 256 positions, wave tile (TM*32) x 64.  Per 16-channel chunk the
// activation window [256 + tap span][16] is converted (leaky-ReLU, hi/lo split) ONCE into LDS and ALL taps' weight fragments of
// that chunk are copied into LDS in fragment order (1 KB blocks, lane-linear -> conflict-free ds_read_b128), so the tap loop
// contains no global memory operation and no barrier: B fragments are plain shifted rows of the window.  Global loads for chunk
// c+1 are issued before the MFMA loop of chunk c and written to LDS after it (two barriers per chunk).
// Channels-last to channels-last launches whose row tiles pair up (Cout >= 128) run 8 waves = 128 rows x 256 positions per workgroup
// (template WM = 2): both 64-row wave groups read the one staged window (measured in round 2: -4.5 ms of a 112 ms step).
#include <type_traits>

#include "common.h"

namespace sbv2 {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

// operand precision of the matrix-core path: bf16 (1 MFMA per product), split bf16 hi/lo (3 MFMAs, f32-grade) or fp16 (1 MFMA,
// 11-bit significands: 8x finer than bf16 at the same rate; f32 accumulation and f32 storage in every mode)
enum { PREC_BF16 = 0, PREC_BF16X3 = 1, PREC_F16 = 2 };
__device__ __forceinline__ f32x16 mfma_32x32x16(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x16 mfma_32x32x16(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

// compile-time loop: indices are constants before SROA runs, so per-thread staging arrays stay in registers (a late-unrolled
// `for` over a 12-entry float4 array was left in scratch by hipcc)
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

constexpr int kClThreads = 256;
constexpr int kClNT = 256;
constexpr int kClMaxSpan = 64;

struct ClKernelParams {
    ConvClParams p;
    int xrows;    // window rows staged per chunk (NT + tap span)
    int wshift0;  // min shift
    int wbytes;   // LDS bytes of the weight region
    int nmt;      // 32-row tiles in the packed weights (padded to a multiple of TM)
    int sh0, sh_step;   // tap t reads window row offset sh0 + t * sh_step (tap shifts are an arithmetic progression)
    int mask_shift;
    int aux_off;      // LDS offset of the staged bias / mask bytes (channels-last epilogue), 0 = not staged
    int mask_nshift;  // mask index of position n * out_stride + phase offset == n >> mask_nshift (power-of-two strides)
    unsigned long long* stamps;   // diagnostics only (sbv2_debug_conv_cl_clock): per workgroup {s_memtime, s_memrealtime} before / after the chunk loop
};

// WM = 2: eight waves; waves 0-3 and 4-7 compute two DIFFERENT row groups (TM * 32 rows each) over the SAME 256 positions, so the activation
// window of a position tile is fetched, converted and written to LDS once for 2 * TM * 32 output rows instead of once per TM * 32.
// ABL (diagnostic builds only, never launched by the product path): 0 = the kernel; 1 = no MFMAs (fragments still read); 2 = MFMAs only (no
// staging, no fragment reads inside the loop); 3 = staging + barriers only.  Used with the clock stamps to tell a scheduling bound from
// a clock (power) bound: MI355X_MICROARCH.md "DVFS give-back" item 6.
template <int TM, int PREC, bool IN_KM, bool OUT_KM, int WM, int ABL = 0>
__global__ __launch_bounds__(kClThreads * WM) void conv_cl_kernel(const ClKernelParams kp) {
    constexpr int kT = kClThreads * WM;
    constexpr bool YS = ABL == 20;   // the channels-last epilogue also writes the result's bf16 parts (ConvClParams::ys_p): the transposed convs of the wide stages
    constexpr bool SPLIT = PREC == PREC_BF16X3;
    using elem_t = std::conditional_t<PREC == PREC_F16, _Float16, __bf16>;
    using ex8 = std::conditional_t<PREC == PREC_F16, f16x8, bf16x8>;
    using ex4 = std::conditional_t<PREC == PREC_F16, f16x4, bf16x4>;
    constexpr int PARTS = SPLIT ? 2 : 1;
    constexpr int TN = 2;
    constexpr int MAXW = (kMaxTaps * TM * WM * PARTS * 64 + kT - 1) / kT;  // float4 per thread for one chunk's weights
    constexpr int NX = ((kClNT + kClMaxSpan) * 4 + kT - 1) / kT;
    const ConvClParams& p = kp.p;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* wsm = smem;
    char* xs_hi = smem + kp.wbytes;
    char* xs_lo = xs_hi + kp.xrows * 32;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn0 = (wave & 3) * 64;
    const int wm = wave >> 2;   // row group of this wave (0 when WM == 1)
    // XCD-aware tile order (1-D grid): workgroups are dealt round-robin over the 8 XCDs (id % 8), each with its own L2.  The gy row tiles
    // of one position tile read the SAME activation window, so they are given ids that differ by 8 (same XCD, back to back) and
    // consecutive position tiles of an XCD are neighbours (shared halo).  Speed only: any placement is correct.
    const int gy = kp.nmt / (TM * WM);
    const int bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;
    const int by = slot % gy;
    const int bx = (slot / gy) * 8 + xcd;
    const int m0 = (by * WM + wm) * (TM * 32);   // first output row of this WAVE's tile
    const int n0 = bx * kClNT;
    if (n0 >= p.N) return;
    const int M = p.M, N = p.N, NB = p.NB, ntaps = p.ntaps;
    const int nchunks = (p.K + 15) >> 4;
    const int wstart = n0 + kp.wshift0;
    const int nwf4 = ntaps * TM * WM * PARTS * 64;   // float4 of one chunk's weight region
    const int nxf4 = kp.xrows * 4;
    const float slope = p.pre_slope;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4v rw[MAXW];
    f32x4v rx[NX];    // channels-last input: both 16-channel halves of a 128-byte line are fetched together (rx = even chunk,
    f32x4v rx1[NX];   // rx1 = the following odd chunk)
    // branch-free loads (clamped, always-valid addresses); see gemm_conv.hip for why
    auto load_w = [&](int chunk) {
        const f32x4v* src = reinterpret_cast<const f32x4v*>(p.W) + ((int64_t)chunk * kp.nmt + by * TM * WM) * ntaps * PARTS * 64;
        static_for<0, MAXW>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            rw[i] = src[min(tid + i * kT, nwf4 - 1)];
        });
    };
    auto store_w = [&]() {
        static_for<0, MAXW>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const int idx = tid + i * kT;
            if (idx < nwf4) reinterpret_cast<f32x4v*>(wsm)[idx] = rw[i];
        });
    };
    // IN_KM = false: X is channels-last [pos][ldx]; a float4 is 4 channels of one position.
    // IN_KM = true : X is a k-major plane [k][ldx]; a float4 is 4 positions of one channel and is transposed while staging.
    const int xr4 = kp.xrows >> 2;
    auto load_x = [&](int chunk) {
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int idx = min(tid + i * kT, nxf4 - 1);
            if (IN_KM) {
                // handled by load_xk below (4 x 4 blocks)
            } else {
                // called for even chunks only: channels [16c, 16c+16) now, [16c+16, 16c+32) kept in registers for chunk c+1, so every
                // 128-byte line of X is fetched once (PMC: fetching the halves one chunk apart doubled FETCH_SIZE)
                const int pos = min(max(wstart + (idx >> 2), 0), NB - 1);
                const float* src = p.X + (int64_t)pos * p.ldx + chunk * 16 + (idx & 3) * 4;
                rx[i] = *reinterpret_cast<const f32x4v*>(src);
                rx1[i] = *reinterpret_cast<const f32x4v*>(chunk + 1 < nchunks ? src + 16 : src);
            }
        }
    };
    // k-major input: one thread owns a 4-channel x 4-position block (4 float4 loads along the position axis), transposes it in
    // registers and writes 4 channels per position with one 8-byte LDS store (instead of sixteen 2-byte stores)
    constexpr int NBK = ((kClNT + kClMaxSpan) + kT - 1) / kT;   // blocks per thread: 4 * xrows / 4 / 256
    f32x4v rk[NBK][4];
    const int nblk = 4 * xr4;
    auto load_xk = [&](int chunk) {
#pragma unroll
        for (int bi = 0; bi < NBK; ++bi) {
            const int b = min(tid + bi * kT, nblk - 1);
            const int kq = b / xr4;
            const int j = wstart + (b - kq * xr4) * 4;
            const int jj = (j >= 0 && j < NB) ? j : 0;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int k = min(chunk * 16 + kq * 4 + t, p.K - 1);
                rk[bi][t] = *reinterpret_cast<const f32x4v*>(p.X + (int64_t)k * p.ldx + jj);
            }
        }
    };
    auto lrelu4 = [&](float4& v) {
        if (slope != 1.0f) {
            v.x = v.x >= 0.f ? v.x : v.x * slope;
            v.y = v.y >= 0.f ? v.y : v.y * slope;
            v.z = v.z >= 0.f ? v.z : v.z * slope;
            v.w = v.w >= 0.f ? v.w : v.w * slope;
        }
    };
    // LDS window: row = position, 32 bytes = two 16-byte halves (k 0-7 | k 8-15); the half index is XOR-swizzled with bit 3 of
    // the row so that the 16-lane groups of ds_read_b128 hit 16 distinct 16-byte slots
    int xchunk = 0;
    auto store_x = [&]() {
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int idx = tid + i * kT;
            if (idx < nxf4) {
                f32x4v rv = rx[i];
                if (!IN_KM && (xchunk & 1)) rv = rx1[i];
                float4 v = make_float4(rv[0], rv[1], rv[2], rv[3]);
                if (IN_KM) {
                    // handled by store_xk below
                } else {
                    const int row = idx >> 2, q = idx & 3;
                    const int pos = wstart + row;
                    if (pos < 0 || pos >= NB) v = make_float4(0.f, 0.f, 0.f, 0.f);
                    lrelu4(v);
                    const int off = row * 32 + ((((q >> 1) ^ (row >> 3)) & 1) << 4) + ((q & 1) << 3);
                    ex4 h;
                    h[0] = (elem_t)v.x; h[1] = (elem_t)v.y; h[2] = (elem_t)v.z; h[3] = (elem_t)v.w;
                    *reinterpret_cast<ex4*>(xs_hi + off) = h;
                    if (SPLIT) {
                        ex4 l;
                        l[0] = (elem_t)(v.x - (float)h[0]); l[1] = (elem_t)(v.y - (float)h[1]);
                        l[2] = (elem_t)(v.z - (float)h[2]); l[3] = (elem_t)(v.w - (float)h[3]);
                        *reinterpret_cast<ex4*>(xs_lo + off) = l;
                    }
                }
            }
        }
    };

    auto store_xk = [&]() {
#pragma unroll
        for (int bi = 0; bi < NBK; ++bi) {
            const int b = tid + bi * kT;
            if (b < nblk) {
                const int kq = b / xr4;
                const int r0 = (b - kq * xr4) * 4;
                const int j = wstart + r0;
                const int kbase = xchunk * 16 + kq * 4;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bool pin = (j + e >= 0) && (j + e < NB);
                    float4 v = make_float4((pin && kbase < p.K) ? rk[bi][0][e] : 0.f, (pin && kbase + 1 < p.K) ? rk[bi][1][e] : 0.f,
                                           (pin && kbase + 2 < p.K) ? rk[bi][2][e] : 0.f, (pin && kbase + 3 < p.K) ? rk[bi][3][e] : 0.f);
                    lrelu4(v);
                    const int row = r0 + e;
                    const int off = row * 32 + ((((kq >> 1) ^ (row >> 3)) & 1) << 4) + ((kq & 1) << 3);
                    ex4 h;
                    h[0] = (elem_t)v.x; h[1] = (elem_t)v.y; h[2] = (elem_t)v.z; h[3] = (elem_t)v.w;
                    *reinterpret_cast<ex4*>(xs_hi + off) = h;
                    if (SPLIT) {
                        ex4 l;
                        l[0] = (elem_t)(v.x - (float)h[0]); l[1] = (elem_t)(v.y - (float)h[1]);
                        l[2] = (elem_t)(v.z - (float)h[2]); l[3] = (elem_t)(v.w - (float)h[3]);
                        *reinterpret_cast<ex4*>(xs_lo + off) = l;
                    }
                }
            }
        }
    };

    load_w(0);
    if (IN_KM) load_xk(0); else load_x(0);
    // The channels-last epilogue's bias (one value per row of this tile) and column-mask bytes (one per position) are fetched with the
    // first tiles and parked in LDS: read in the epilogue they are dependent global round trips (~1 us each per workgroup; the same
    // change in respair_cl.hip removed 2 ms per step).
    float* bias_s = reinterpret_cast<float*>(smem + kp.aux_off);          // [WM * TM * 32]
    unsigned char* mask_s = reinterpret_cast<unsigned char*>(bias_s + 128);  // [kClNT]
    const bool staged = !OUT_KM && kp.aux_off != 0;
    float bstage = 0.f;
    unsigned char mstage = 1;
    if (staged) {
        if (tid < WM * TM * 32) {
            const int m = by * (WM * TM * 32) + tid;
            int co = m;
            if (p.phase_rows < (1 << 30)) co = m - (m / p.phase_rows) * p.phase_rows;
            bstage = (p.bias && m < M) ? p.bias[co] : 0.f;
        }
        if (p.mask) mstage = p.mask[min(n0 + (tid & (kClNT - 1)), N - 1) >> kp.mask_nshift];
    }
    store_w();
    if (IN_KM) store_xk(); else store_x();
    if (staged) {
        if (tid < WM * TM * 32) bias_s[tid] = bstage;
        if (tid < kClNT) mask_s[tid] = mstage;
    }
    __syncthreads();

    const int lcol = lane & 31, lh = lane >> 5;
    unsigned long long st_t0 = 0, st_r0 = 0;
    if (kp.stamps) {
        st_t0 = __builtin_amdgcn_s_memtime();
        st_r0 = __builtin_amdgcn_s_memrealtime();
    }
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const bool more = chunk + 1 < nchunks;
        if (more && ABL != 2) {
            load_w(chunk + 1);
            if (IN_KM) load_xk(chunk + 1);
            else if (((chunk + 1) & 1) == 0) load_x(chunk + 1);
        }
        // Software-pipelined tap loop: the LDS fragments of tap t+1 are requested before the MFMAs of tap t are issued (two register
        // sets, static ping-pong), so a wave's LDS latency hides under its own MFMA block instead of relying on the partner wave.
        // Tap offsets are an arithmetic progression (shift0 + tap * step): no scalar load shares the lgkm counter with the ds_reads.
        struct Frags {
            ex8 bh[TN], bl[TN], ah[TM], al[TM];
        };
        auto load_frags = [&](Frags& f, int tap) {
            const int sh = kp.sh0 + tap * kp.sh_step;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int row = wn0 + j * 32 + lcol + sh;
                const int off = row * 32 + (((lh ^ (row >> 3)) & 1) << 4);
                f.bh[j] = *reinterpret_cast<const ex8*>(xs_hi + off);
                if (SPLIT) f.bl[j] = *reinterpret_cast<const ex8*>(xs_lo + off);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const char* blk = wsm + (((wm * TM + i) * ntaps + tap) * PARTS) * 1024 + lane * 16;
                f.ah[i] = *reinterpret_cast<const ex8*>(blk);
                if (SPLIT) f.al[i] = *reinterpret_cast<const ex8*>(blk + 1024);
            }
        };
        auto mfma_frags = [&](const Frags& f) {
            if (ABL == 1 || ABL == 3) {   // keep the fragments live, issue nothing
#pragma unroll
                for (int i = 0; i < TM; ++i) asm volatile("" ::"v"(f.ah[i]), "v"(f.al[i]));
#pragma unroll
                for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(f.bh[j]), "v"(f.bl[j]));
                return;
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (SPLIT) {
                        acc[i][j] = mfma_32x32x16(f.al[i], f.bh[j], acc[i][j]);
                        acc[i][j] = mfma_32x32x16(f.ah[i], f.bl[j], acc[i][j]);
                    }
                    acc[i][j] = mfma_32x32x16(f.ah[i], f.bh[j], acc[i][j]);
                }
        };
        Frags fa, fb;
        if (ABL == 2 || ABL == 3) {   // fragments read once per chunk (ABL 2: MFMA stream only; ABL 3: nothing but staging)
            if (ABL == 2 || chunk == 0) {
                load_frags(fa, 0);
                load_frags(fb, 0);
            }
            if (ABL == 2)
                for (int tap = 0; tap < ntaps; tap += 2) {
                    mfma_frags(fa);
                    if (tap + 1 < ntaps) mfma_frags(fb);
                }
        } else {
        load_frags(fa, 0);
        int tap = 0;
        for (; tap + 2 <= ntaps; tap += 2) {
            load_frags(fb, tap + 1);
            mfma_frags(fa);
            load_frags(fa, min(tap + 2, ntaps - 1));
            mfma_frags(fb);
        }
        if (tap < ntaps) mfma_frags(fa);
        }
        if (ABL != 2) __syncthreads();   // every wave is done reading this chunk's tiles
        if (more && ABL != 2) {
            xchunk = chunk + 1;
            store_w();
            if (IN_KM) store_xk(); else store_x();
        }
        if (ABL != 2) __syncthreads();
    }
    if (kp.stamps && tid == 0) {
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        unsigned long long* o = kp.stamps + (size_t)blockIdx.x * 4;
        o[0] = st_t0; o[1] = st_r0; o[2] = t1; o[3] = r1;
    }

    // ---- epilogue ------------------------------------------------------------------------------------------------------------
    const bool phased = p.phase_rows < (1 << 30);
    if (OUT_KM) {
        // k-major output plane Y[m][ldy]: for one accumulator register the 32 lanes of a half-wave hold 32 consecutive columns.
        // Everything read from global memory (bias rows, column mask, residual, previous contents) is requested BEFORE the first store:
        // on gfx9 loads and stores share the in-order vmcnt queue, so a load issued after a store is not usable before that store is
        // acknowledged; interleaved per row that was one store round trip per accumulator row (32 per tile).
        int nn[TN];
        bool nok[TN], keepn[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            nn[j] = n0 + wn0 + j * 32 + lcol;
            nok[j] = nn[j] < N;
            keepn[j] = true;
            if (p.mask) {
                const int nc = min(nn[j], N - 1);
                keepn[j] = p.mask[kp.mask_shift >= 0 ? (nc >> kp.mask_shift) : (nc / p.mask_div)] != 0;
            }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            float brow[16], rr[16][TN], old[16][TN];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int mc = min(m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, M - 1);
                brow[r] = p.bias ? p.bias[mc] : 0.f;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int nc = min(nn[j], N - 1);
                    rr[r][j] = p.R ? p.R[(int64_t)mc * p.ldr + nc] : 0.f;
                    old[r][j] = p.accumulate ? p.Y[(int64_t)mc * p.ldy + nc] : 0.f;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m >= M) continue;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (!nok[j]) continue;
                    float v = acc[i][j][r] + brow[r];
                    if (p.act == ACT_RELU) v = fmaxf(v, 0.f);
                    else if (p.act == ACT_GELU) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
                    else if (p.act == ACT_TANH) v = tanhf(v);
                    v *= p.alpha;
                    if (p.R) v += rr[r][j];
                    v *= p.beta;
                    if (p.accumulate) v += old[r][j];
                    if (!keepn[j]) v = 0.f;
                    p.Y[(int64_t)m * p.ldy + nn[j]] = v;
                }
            }
        }
        return;
    }
    // channels-last output.  In the accumulators a lane owns one position and, per register quad, 4 consecutive channels: stored
    // directly that is 16 bytes per lane scattered over 32 rows.  Instead each wave transposes its 32 x 64 sub-tile through a private
    // LDS tile [64 positions][32 channels (+4 pad)] so that 8 consecutive lanes write (and read the residual as) one full 128-byte
    // line of a row.
    // Address arithmetic is incremental (64-bit pointers advanced by wave-uniform strides, 32-bit positions, mask index by shift):
    // the straightforward form cost ~75 VALU + 10 quarter-rate integer multiplies per output row, 16 rows per thread.
    float* tile = reinterpret_cast<float*>(smem) + wave * (64 * 36);
    const float beta = p.beta;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4v v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                *reinterpret_cast<f32x4v*>(tile + (j * 32 + lcol) * 36 + 8 * q + 4 * lh) = v;
            }
        const int c4 = (lane & 7) * 4;
        const int m = m0 + i * 32 + c4;
        int co = m, po = 0, ostride = 1;
        if (phased) {
            const int ph = m / p.phase_rows;
            co = m - ph * p.phase_rows;
            ostride = p.out_stride;
#pragma unroll
            for (int t = 0; t < kMaxPhases; ++t) po = (ph == t) ? p.phase_off[t] : po;
        }
        float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (staged) b4 = *reinterpret_cast<const float4*>(bias_s + (wm * TM + i) * 32 + c4);
        else if (p.bias && m < M) b4 = *reinterpret_cast<const float4*>(p.bias + co);
        const int nfirst = n0 + wn0 + (lane >> 3);
        int pos = nfirst * ostride + po;              // < 2^31 (checked by the callers)
        const int pstep = 8 * ostride;
        const float* rp = p.R ? p.R + (int64_t)pos * p.ldr + co : nullptr;
        float* yp = p.Y + (int64_t)pos * p.ldy + co;
        const int64_t rstep = (int64_t)pstep * p.ldr, ystep = (int64_t)pstep * p.ldy;
        const float* trow = tile + (lane >> 3) * 36 + c4;
        // accumulate (the last step of a ResBlock branch adds into the stage sum): the eight previous values are requested back to back
        // before the row loop; read inside it, each load waited behind the previous row's store (eight dependent round trips per tile:
        // the clock-stamp timeline of respair_cl.hip showed 5 us of a 12 us workgroup for the same pattern)
        f32x4v rold[8], rres[8];
        if (p.accumulate || rp) {
            const int64_t last = (int64_t)(N - 1) * ostride + po;
            const int cc = m < M ? co : 0;
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int64_t pp = min((int64_t)(nfirst + it * 8) * ostride + po, last);
                if (p.accumulate) rold[it] = *reinterpret_cast<const f32x4v*>(p.Y + pp * p.ldy + cc);
                if (rp) rres[it] = *reinterpret_cast<const f32x4v*>(p.R + pp * p.ldr + cc);   // the residual rows as well
            }
        }
        // Interior sub-tiles with the mask staged in LDS (round 5): no per-lane conditions around the stores.  In the loop below every iteration's store sits
        // in a divergent branch, behind which hipcc cannot count the outstanding stores: it waits s_waitcnt vmcnt(0) at the first use of a pre-loaded
        // residual row in the NEXT iteration, i.e. one exposed store acknowledgement per row group, eight per sub-tile.  Same arithmetic per element.
        if (staged && nfirst - (lane >> 3) + 64 <= N && m0 + i * 32 + 32 <= M) {
            const bool has_r = rp != nullptr, has_acc = p.accumulate != 0;
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const f32x4v a = *reinterpret_cast<const f32x4v*>(trow + it * 8 * 36);
                float4 v = make_float4(a[0] + b4.x, a[1] + b4.y, a[2] + b4.z, a[3] + b4.w);
                if (has_r) {
                    v.x += rres[it][0]; v.y += rres[it][1]; v.z += rres[it][2]; v.w += rres[it][3];
                }
                if (beta != 1.0f) { v.x *= beta; v.y *= beta; v.z *= beta; v.w *= beta; }
                if (has_acc) {
                    v.x += rold[it][0]; v.y += rold[it][1]; v.z += rold[it][2]; v.w += rold[it][3];
                }
                if (!mask_s[nfirst + it * 8 - n0]) v = make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4*>(yp + it * ystep) = v;
                if (YS && p.ys_p) {   // bf16 parts of lrelu(result): 4 channels = 8 bytes of a 32-byte row of chunk co >> 4; the lo plane follows the hi plane
                    typedef __bf16 cl_bf16x4 __attribute__((ext_vector_type(4)));
                    const float vv[4] = {v.x, v.y, v.z, v.w};
                    cl_bf16x4 h, l;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float x = vv[e] >= 0.f ? vv[e] : vv[e] * p.ys_slope;
                        h[e] = (__bf16)x;
                        l[e] = (__bf16)(x - (float)h[e]);
                    }
                    char* qs = static_cast<char*>(p.ys_p) + ((int64_t)(co >> 4) * 2 * p.ys_rows + p.ys_front + pos + it * pstep) * 32 + (co & 15) * 2;
                    *reinterpret_cast<cl_bf16x4*>(qs) = h;
                    *reinterpret_cast<cl_bf16x4*>(qs + p.ys_rows * 32) = l;
                }
            }
            continue;
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int n = nfirst + it * 8;
            const f32x4v a = *reinterpret_cast<const f32x4v*>(trow + it * 8 * 36);
            if (n < N && m < M) {
                float4 v = make_float4(a[0] + b4.x, a[1] + b4.y, a[2] + b4.z, a[3] + b4.w);
                if (rp) {
                    v.x += rres[it][0]; v.y += rres[it][1]; v.z += rres[it][2]; v.w += rres[it][3];
                }
                if (beta != 1.0f) { v.x *= beta; v.y *= beta; v.z *= beta; v.w *= beta; }
                if (p.accumulate) {
                    v.x += rold[it][0]; v.y += rold[it][1]; v.z += rold[it][2]; v.w += rold[it][3];
                }
                if (staged) {
                    if (!mask_s[n - n0]) v = make_float4(0.f, 0.f, 0.f, 0.f);
                } else if (p.mask) {
                    const int mi = kp.mask_shift >= 0 ? (pos >> kp.mask_shift) : (pos / p.mask_div);
                    if (!p.mask[mi]) v = make_float4(0.f, 0.f, 0.f, 0.f);
                }
                *reinterpret_cast<float4*>(yp) = v;
                if (YS && p.ys_p) {   // (uniform; compiled into the ABL = 20 instantiation only: in every kernel of the family it cost 40 registers) bf16 parts of lrelu(result): 4 channels = 8 bytes of a 32-byte row of chunk co >> 4; the lo plane follows the hi plane
                    typedef __bf16 cl_bf16x4 __attribute__((ext_vector_type(4)));
                    const float vv[4] = {v.x, v.y, v.z, v.w};
                    cl_bf16x4 h, l;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float x = vv[e] >= 0.f ? vv[e] : vv[e] * p.ys_slope;
                        h[e] = (__bf16)x;
                        l[e] = (__bf16)(x - (float)h[e]);
                    }
                    char* qs = static_cast<char*>(p.ys_p) + ((int64_t)(co >> 4) * 2 * p.ys_rows + p.ys_front + pos) * 32 + (co & 15) * 2;
                    *reinterpret_cast<cl_bf16x4*>(qs) = h;
                    *reinterpret_cast<cl_bf16x4*>(qs + p.ys_rows * 32) = l;
                }
            }
            pos += pstep;
            if (rp) rp += rstep;
            yp += ystep;
        }
    }
}

template <int TM, int PREC, bool IN_KM, bool OUT_KM, int WM = 1>
static void launch_cl(ClKernelParams kp, hipStream_t stream) {
    constexpr bool SPLIT = PREC == PREC_BF16X3;
    constexpr int PARTS = SPLIT ? 2 : 1;
    const ConvClParams& p = kp.p;
    kp.wbytes = p.ntaps * TM * WM * PARTS * 1024;
    size_t lds = (size_t)kp.wbytes + (size_t)kp.xrows * 32 * PARTS;
    if (!OUT_KM) lds = std::max<size_t>(lds, 4 * WM * 64 * 36 * sizeof(float));  // the epilogue's per-wave transpose tiles
    kp.aux_off = 0;
    kp.mask_nshift = 0;
    if (!OUT_KM) {
        // bias / mask staging needs "mask index = n >> const": mask_div and out_stride powers of two, mask_div a multiple of out_stride
        bool ok = true;
        if (p.mask) {
            const int os = p.phase_rows < (1 << 30) ? p.out_stride : 1;
            ok = kp.mask_shift >= 0 && os > 0 && (os & (os - 1)) == 0 && p.mask_div % os == 0;
            int ls = 0;
            while ((1 << ls) < os) ++ls;
            kp.mask_nshift = kp.mask_shift - ls;
        }
        if (ok) {
            lds = (lds + 15) / 16 * 16;
            kp.aux_off = (int)lds;
            lds += 128 * sizeof(float) + kClNT;
        }
    }
    SBV2_REQUIRE(lds <= 160 * 1024, "conv_cl: LDS budget exceeded");
    auto kern = conv_cl_kernel<TM, PREC, IN_KM, OUT_KM, WM>;
    static std::atomic<uint64_t> lds_allowed{0};   // per (kernel instantiation, device)
    allow_full_lds(reinterpret_cast<const void*>(kern), lds_allowed);
    if constexpr (SPLIT && !IN_KM && !OUT_KM && TM == 2) {
        if (p.ys_p) {
            kern = conv_cl_kernel<TM, PREC, IN_KM, OUT_KM, WM, 20>;
            static std::atomic<uint64_t> lds_allowed_ys{0};
            allow_full_lds(reinterpret_cast<const void*>(kern), lds_allowed_ys);
        }
    } else {
        SBV2_REQUIRE(!p.ys_p, "conv_cl: the parts output exists for the split-bf16 channels-last kernel with 64-row wave tiles only");
    }
    const int ntx = round_up((p.N + kClNT - 1) / kClNT, 8);   // padded so that the (xcd, slot) <-> (tile, row tile) map is a bijection
    dim3 grid(ntx * (kp.nmt / (TM * WM)));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const bool prof = conv_prof_active();
    if (prof) {
        HIP_CHECK(hipEventCreate(&e0));
        HIP_CHECK(hipEventCreate(&e1));
        HIP_CHECK(hipEventRecord(e0, stream));
    }
    hipLaunchKernelGGL(kern, grid, dim3(kClThreads * WM), lds, stream, kp);
    HIP_CHECK(hipGetLastError());
    if (prof) {
        HIP_CHECK(hipEventRecord(e1, stream));
        conv_prof_add(PREC == PREC_F16 ? 19 + (TM == 2 ? 0 : 1) + ((IN_KM || OUT_KM) ? 2 : 0) : (SPLIT ? 8 : 10) + (TM == 2 ? 0 : 1) + ((IN_KM || OUT_KM) ? 4 : 0), 2.0 * p.M * (double)p.N * p.K * p.ntaps, e0, e1);
    }
}

// 128-row workgroups (two 64-row wave groups sharing one staged window) when the row tiles pair up and the grid still covers the chip
static bool wide_rows(const ClKernelParams& kp) {
    if ((kp.nmt & 3) != 0) return false;
    const int64_t wgs = (int64_t)round_up((kp.p.N + kClNT - 1) / kClNT, 8) * (kp.nmt / 4);
    return wgs >= 512;
}

template <int TM, int PREC>
static void launch_cl_layout(const ClKernelParams& kp, hipStream_t stream) {
    const ConvClParams& p = kp.p;
    if (p.in_km) {
        if (p.out_km) launch_cl<TM, PREC, true, true>(kp, stream);
        else launch_cl<TM, PREC, true, false>(kp, stream);
    } else {
        if (p.out_km) launch_cl<TM, PREC, false, true>(kp, stream);
        else if (TM == 2 && wide_rows(kp)) launch_cl<TM, PREC, false, false, TM == 2 ? 2 : 1>(kp, stream);
        else launch_cl<TM, PREC, false, false>(kp, stream);
    }
}

void launch_conv_cl(const ConvClParams& p, hipStream_t stream) {
    SBV2_REQUIRE(p.ntaps >= 1 && p.ntaps <= kMaxTaps, "bad tap count");
    SBV2_REQUIRE((p.ldx & 3) == 0, "conv_cl: input pitch must be a multiple of 4 floats");
    SBV2_REQUIRE(p.in_km || (p.K & 15) == 0, "conv_cl: channels-last input needs Cin % 16 == 0");
    SBV2_REQUIRE(p.out_km || ((p.ldy & 3) == 0 && (p.M & 3) == 0), "conv_cl: channels-last output needs Cout % 4 == 0");
    SBV2_REQUIRE(p.tm == 1 || p.tm == 2, "conv_cl: bad row tiling");
    if (p.N <= 0) return;
    ClKernelParams kp;
    kp.p = p;
    int smin = p.shift[0], smax = p.shift[0];
    for (int t = 1; t < p.ntaps; ++t) {
        smin = std::min(smin, p.shift[t]);
        smax = std::max(smax, p.shift[t]);
    }
    if (p.in_km) {  // float4 loads along the position axis: 16-byte aligned window
        kp.wshift0 = (smin >= 0) ? (smin / 4) * 4 : -(((-smin) + 3) / 4) * 4;
        kp.xrows = kClNT + round_up(smax - kp.wshift0, 4);
    } else {
        kp.wshift0 = smin;
        kp.xrows = kClNT + (smax - smin);
    }
    SBV2_REQUIRE(kp.xrows - kClNT <= kClMaxSpan, "conv_cl: tap span too large");
    kp.sh0 = p.shift[0] - kp.wshift0;
    kp.sh_step = p.ntaps > 1 ? p.shift[1] - p.shift[0] : 0;
    for (int t = 1; t < p.ntaps; ++t) SBV2_REQUIRE(p.shift[t] - p.shift[t - 1] == kp.sh_step, "conv_cl: tap shifts must be an arithmetic progression");
    kp.nmt = p.nmt;
    kp.stamps = nullptr;
    kp.mask_shift = -1;
    if (p.mask && p.mask_div > 0 && (p.mask_div & (p.mask_div - 1)) == 0) {
        int s = 0;
        while ((1 << s) < p.mask_div) ++s;
        kp.mask_shift = s;
    }
    SBV2_REQUIRE(!(p.split && p.f16), "conv_cl: split and f16 are exclusive");
    // small grids (single-utterance calls: the flow's FFN convs at 897 frames are 12 / 48 workgroups of 64 rows): 32-row tiles double the
    // workgroup count; the weights are packed per 32-row tile either way (nmt is a multiple of the packed tm)
    // ... and the channels-last -> k-major products of such calls run as independent waves fed by an LDS-DMA ring (conv_cl_small.hip: same bits)
    // (a launch that must also write the result's bf16 parts never takes the small-grid kernel, which has no parts epilogue: should conv_cl_parts_ok drift
    // from this dispatch, the SBV2_REQUIRE in launch_cl refuses the launch instead of leaving the parts unwritten)
    if (!p.ys_p && (int64_t)((p.N + kClNT - 1) / kClNT) * std::max(1, p.nmt / 2) < small_grid_max() && launch_conv_cl_small(p, kp.mask_shift, stream)) return;
    int tm = p.tm;
    if (tm == 2 && (int64_t)((p.N + kClNT - 1) / kClNT) * (p.nmt / 2) < 128) tm = 1;
    if (p.split) {
        if (tm == 2) launch_cl_layout<2, PREC_BF16X3>(kp, stream);
        else launch_cl_layout<1, PREC_BF16X3>(kp, stream);
    } else if (p.f16) {
        if (tm == 2) launch_cl_layout<2, PREC_F16>(kp, stream);
        else launch_cl_layout<1, PREC_F16>(kp, stream);
    } else {
        if (tm == 2) launch_cl_layout<2, PREC_BF16>(kp, stream);
        else launch_cl_layout<1, PREC_BF16>(kp, stream);
    }
}

// whether launch_conv_cl(p) would take a kernel that can also write the result's bf16 parts (ConvClParams::ys_p): the split-bf16 channels-last kernel
// with 64-row wave tiles, i.e. not a small grid (those go to 32-row tiles or conv_cl_small: same bits, no parts epilogue)
bool conv_cl_parts_ok(const ConvClParams& p) {
    if (!p.split || p.f16 || p.in_km || p.out_km || p.tm != 2 || p.N <= 0) return false;
    const int64_t tiles = (p.N + kClNT - 1) / kClNT;
    return tiles * std::max(1, p.nmt / 2) >= small_grid_max() && tiles * (p.nmt / 2) >= 128;
}

// Diagnostic launch of the dominant configuration (128-row workgroups, split-bf16, channels-last in and out) with an ablation variant and
// the clock stamps: stamps[4 * workgroup] = {s_memtime, s_memrealtime} before, {..} after the chunk loop.  Results of abl != 0 are garbage.
void launch_conv_cl_diag(const ConvClParams& p, int abl, unsigned long long* stamps, hipStream_t stream) {
    SBV2_REQUIRE(p.split && !p.in_km && !p.out_km && p.tm == 2 && (p.nmt & 3) == 0, "conv_cl diag: the 128-row split-bf16 configuration only");
    ClKernelParams kp;
    kp.p = p;
    int smin = p.shift[0], smax = p.shift[0];
    for (int t = 1; t < p.ntaps; ++t) {
        smin = std::min(smin, p.shift[t]);
        smax = std::max(smax, p.shift[t]);
    }
    kp.wshift0 = smin;
    kp.xrows = kClNT + (smax - smin);
    kp.sh0 = p.shift[0] - kp.wshift0;
    kp.sh_step = p.ntaps > 1 ? p.shift[1] - p.shift[0] : 0;
    kp.nmt = p.nmt;
    kp.mask_shift = -1;
    kp.stamps = stamps;
    kp.wbytes = p.ntaps * 2 * 2 * 2 * 1024;
    size_t lds = std::max<size_t>((size_t)kp.wbytes + (size_t)kp.xrows * 32 * 2, 4 * 2 * 64 * 36 * sizeof(float));
    lds = (lds + 15) / 16 * 16;
    kp.aux_off = (int)lds;
    kp.mask_nshift = 0;
    lds += 128 * sizeof(float) + kClNT;
    const int ntx = round_up((p.N + kClNT - 1) / kClNT, 8);
    dim3 grid(ntx * (kp.nmt / 4));
    auto go = [&](auto kern) {
        static std::atomic<uint64_t> lds_allowed{0};
        (void)lds_allowed;
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        hipLaunchKernelGGL(kern, grid, dim3(kClThreads * 2), lds, stream, kp);
        HIP_CHECK(hipGetLastError());
    };
    if (abl == 0) go(conv_cl_kernel<2, PREC_BF16X3, false, false, 2, 0>);
    else if (abl == 1) go(conv_cl_kernel<2, PREC_BF16X3, false, false, 2, 1>);
    else if (abl == 2) go(conv_cl_kernel<2, PREC_BF16X3, false, false, 2, 2>);
    else go(conv_cl_kernel<2, PREC_BF16X3, false, false, 2, 3>);
}

}  // namespace sbv2
