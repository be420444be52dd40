// The fused ResBlock1 step of respair_clx.hip for the 64- / 32-channel stages at k = 7 / 11 on v_mfma_f32_16x16x32_bf16 with conv_clx.hip's operand scheme
// (round 6):
//
//     y' = beta * ( conv2( lrelu( conv1( lrelu(y), dilation d ) + b1 ) ) + b2 + y )  [+ previous contents]  (then column mask)
//
// Why: these launches run at a power-managed clock, and a probe build of respair_clx.hip that issued every 32x32x16 MFMA as two 16x16x32 instructions from the
// same registers (wrong results, -DRPX_PROBE16=1) ran 7.5 / 12.6 % faster at C = 64, k = 7 / 11 and 6 / 9.5 % at C = 32 (profiles/r06z_respair_shape_probe.jsonl):
// // half the accumulator traffic per FLOP.  The 32-deep K dimension carries a PAIR of consecutive steps a, b (step = one tap of one 16-channel chunk), as the
// 16-channel kernel of respair_clx.hip carries two taps:
//     A_part = [W_part(a) | W_part(b)] (k groups 0, 1 | 2, 3; packed at load: pack_step_pairs),   B_part = [X_part(a) ; X_part(b)] (k groups 0, 1 read step a's
//     row of the window, 2, 3 step b's),   and per accumulator   A_lo B_hi,  A_hi B_lo,  A_hi B_hi
// i.e. 3 instructions of 16 x 16 x 32 per pair and accumulator tile where the 32x32x16 scheme issues 6 of 32 x 32 x 16 per two tiles: same FLOP, same weight
// bytes, the same 12 fragment reads per pair and wave, no operand shuffling.  (First built with conv_clx.hip's scheme - both cross terms of a step in one
// instruction, the hi x hi operands of a pair made by v_permlane32_swap + 4 more reads: correct, and no faster than respair_clx.hip: its 8 swaps and 16 reads
// per 24 MFMAs gave back what the shape gains, profiles/r06z_respair_x16_phases*.jsonl.)
//   * A wave owns 32 channels x 64 positions = 2 x 4 accumulator tiles; 4 waves: C = 64: 2 channel groups x 128 positions, C = 32: 256 positions.
//   * Steps are paired over the sequence (chunk, tap) of TWO chunks (NTAPS is odd: the middle pair spans the chunk boundary), so conv1's window holds a
//     chunk PAIR ([part][chunk of the pair][row][16 channels]: 32-byte rows, the layout conv_clx.hip's fragment reads take without bank conflicts; the
//     conversion writes 64 contiguous rows per instruction); the intermediate is [part][chunk][row][16] and aliases it.
//   * Weights: one pair (C / 8 KB) per LDS-DMA group, double buffered, one barrier per pair as respair_clx.hip at C = 64.  The hi x hi products of pair u are
//     issued BEHIND the barrier of pair u + 1, with that pair's first fragment reads in their gaps (their operands are registers): the read burst behind a
//     barrier, which respair_clx.hip waits for with an idle matrix pipe, is covered.
//   * Every global load of a tile (both chunk pairs of the window, the residual rows, bias, mask) is requested at entry.
// Summation order: 32 products per instruction, cross terms together, hi x hi of two steps together: f32 rounding apart from respair_clx.hip / conv_cl.hip
// (the tests hold 1e-5 kernel against kernel, as for C = 16 and for conv_clx.hip); the oracle tolerance is unchanged.
#include <atomic>
#include <type_traits>

#include "common.h"

#pragma clang fp contract(off)

namespace sbv2 {

typedef __bf16 x6_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 x6_bf16x4 __attribute__((ext_vector_type(4)));
typedef float x6_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned x6_u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void x6_lds_t;
typedef const __attribute__((address_space(1))) void x6_gbl_t;

template <int I, int N, class F>
__device__ __forceinline__ void x6_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        x6_for<I + 1, N>(f);
    }
}
template <int OFF>
__device__ __forceinline__ x6_bf16x8 x6_read_b128(unsigned addr) {
    static_assert(OFF >= 0 && OFF < 65536, "LDS immediate");
    x6_bf16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF));
    return v;
}
template <int OFF>
__device__ __forceinline__ x6_f32x4 x6_read_f128(unsigned addr) {
    x6_f32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF));
    return v;
}
template <int OFF>
__device__ __forceinline__ unsigned x6_read_u8(unsigned addr) {
    unsigned v;
    asm volatile("ds_read_u8 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF));
    return v;
}
template <int OFF>
__device__ __forceinline__ void x6_write_b64(unsigned addr, x6_bf16x4 v) {
    static_assert(OFF >= 0 && OFF < 65536, "LDS immediate");
    asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(addr), "v"(v), "i"(OFF) : "memory");
}
__device__ __forceinline__ void x6_write_b32(unsigned addr, float v) { asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ void x6_write_b8(unsigned addr, unsigned v) { asm volatile("ds_write_b8 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
// (inline asm: the accumulator stays in its registers; conv_clx.hip has the note)
__device__ __forceinline__ void x6_mfma(x6_f32x4& c, const x6_bf16x8& a, const x6_bf16x8& b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
// the low halves (lanes 0-31) of two fragments side by side: [W_hi(a) | W_hi(b)] out of [W_hi | W_lo](a), [W_hi | W_lo](b); one v_permlane32_swap per register
__device__ __forceinline__ x6_bf16x8 x6_lows(const x6_bf16x8& a, const x6_bf16x8& b) {
    const x6_u32x4 ua = __builtin_bit_cast(x6_u32x4, a), ub = __builtin_bit_cast(x6_u32x4, b);
    x6_u32x4 l;
#pragma unroll
    for (int r = 0; r < 4; ++r) l[r] = __builtin_amdgcn_permlane32_swap(ua[r], ub[r], false, false)[0];   // lanes 32-63 of the first <-> lanes 0-31 of the second
    return __builtin_bit_cast(x6_bf16x8, l);
}

// Global loads the COMPILER DOES NOT SEE (inline asm), waited for with counted vmcnt written by hand.  With an LDS-DMA in flight hipcc's wait-count pass puts
// vmcnt(0) in front of every use of a loaded register (it treats the DMA and the loads as out of order with each other), so a tile's first conversion waited
// for EVERY request of the tile (both chunk pairs, the residual rows): 3.5k cycles per tile.  vmcnt returns in order: the requests are issued oldest-needed
// first, and each consumer waits for exactly what is older than it.
__device__ __forceinline__ x6_f32x4 x6_gload128(const float* ptr) {
    x6_f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(ptr));
    return v;
}
// ... uniform base (scalar registers) + this lane's 32-bit byte offset + immediate: no 64-bit address per request
template <int OFF>
__device__ __forceinline__ x6_f32x4 x6_gload128s(const char* base, unsigned voff) {
    static_assert(OFF >= 0 && OFF < 4096, "global immediate offset");
    x6_f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(v) : "v"(voff), "s"(base), "n"(OFF));
    return v;
}
__device__ __forceinline__ float x6_gload32(const float* ptr) {
    float v;
    asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(ptr));
    return v;
}
__device__ __forceinline__ unsigned x6_gload8(const unsigned char* ptr) {
    unsigned v;
    asm volatile("global_load_ubyte %0, %1, off" : "=v"(v) : "v"(ptr));
    return v;
}

// (fragment addresses are formed where they are used, from an opaque copy of the lane base: left to the compiler, the per-tap sums are hoisted over the whole
// unrolled tile and held in registers; conv_clx.hip has the same note)
__device__ __forceinline__ unsigned x6_opaque(unsigned x) {
    asm volatile("" : "+v"(x));
    return x;
}

constexpr int x6_max(int a, int b) { return a > b ? a : b; }

template <int C, int NTAPS>
struct X6Cfg {
    static constexpr int NCH = C / 16;                 // 16-channel chunks
    static constexpr int NMT = C / 32;                 // 32-channel groups of the output (a wave owns one: two 16-row accumulator tiles)
    static constexpr int WN = C == 64 ? 2 : 4;         // 64-position groups
    static constexpr int NW = NMT * WN, T = 64 * NW;   // 4 waves
    static constexpr int NT = 64 * WN;                 // positions of the intermediate per workgroup
    static constexpr int PAIRB = NCH * 2048;           // a weight group = one pair of steps: C / 16 row tiles of 16 x (hi block, lo block) of 1 KB (pack_step_pairs)
    static constexpr int NPC = PAIRB / 1024;           // its 1 KB pieces
    static constexpr int WREG = 2 * PAIRB;             // double buffered
    static constexpr int NCP = NCH / 2;                // chunk pairs
    static constexpr int NP1 = NCP * NTAPS;            // pairs of conv1
    static constexpr int NP2 = NCP * NTAPS;            // ... of conv2
    static constexpr int RB = T / 4;                   // window rows one load instruction of the workgroup covers (4 threads per 64-byte row piece)
    static constexpr int NXC = (NT + 64 + RB - 1) / RB;
    static constexpr int ROWS1 = NXC * RB;             // rows of the conv1 window (NT + tap span <= NT + 64)
    static constexpr int ROWS2 = NT + NTAPS + 1;       // rows of the intermediate (NT + k - 1 are read)
    static constexpr int X1CH = ROWS1 * 32, X1PART = 2 * X1CH;
    static constexpr int X2CH = ROWS2 * 32, X2PART = NCH * X2CH;
    // k groups 0, 1 of a B fragment read the lo plane, 2, 3 the hi plane: the planes a multiple of 256 bytes apart (respair_clx.hip has the measurement)
    static_assert(X1PART % 256 == 0 && X2PART % 256 == 0, "part planes: 256-byte multiples");
    static constexpr int XREG = x6_max(2 * X1PART, 2 * X2PART);   // the intermediate ALIASES the conv1 window
    static constexpr int TPITCH = 36;
    static constexpr int TT = NW * 64 * TPITCH * 4;    // the epilogue's per-wave transpose tiles (overlay everything above)
    static constexpr int MAIN = x6_max(WREG + XREG, TT);
    static constexpr int BIAS_OFF = MAIN;              // 64 floats b1, 64 floats b2
    static constexpr int MASK_OFF = MAIN + 512;        // one byte per row of the intermediate
    static constexpr int LDS = (MASK_OFF + ROWS2 + 16 + 15) / 16 * 16;
    static_assert(T == 256 && (NTAPS & 1) == 1 && NPC % NW == 0, "4 waves, odd kernel sizes");
};

// DG >= 0: diagnostic instantiation (phase stamps of thread 0 into p.stamps[16 per workgroup]; sbv2_debug_respair_clock)
template <int C, int NTAPS, int DG>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void respair_x16_kernel(const ResPairParams p) {
    using K = X6Cfg<C, NTAPS>;
    constexpr int T = K::T, WN = K::WN, RB = K::RB, NMT = K::NMT, NT = K::NT, NXC = K::NXC, NCP = K::NCP, NP1 = K::NP1, NP2 = K::NP2;
    constexpr bool DIAG = DG >= 0;
    constexpr int h2 = (NTAPS - 1) / 2, nto = NT - 2 * h2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)smem);

    unsigned st_[16];
    auto stamp = [&](auto ic) __attribute__((always_inline)) {
        if constexpr (DIAG) {
            constexpr int i = decltype(ic)::value;
            st_[i] = (unsigned)(i >= 14 ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime());
        }
    };
#define X6_STAMP(i) stamp(std::integral_constant<int, i>{})
    if constexpr (DIAG) {
#pragma unroll
        for (int i = 0; i < 16; ++i) st_[i] = 0;
    }
    X6_STAMP(0);
    X6_STAMP(14);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = NMT == 2 ? wave / WN : 0;           // this wave's 32 output channels
    const int wn = NMT == 2 ? wave - wm * WN : wave;   // ... and its 64 positions
    const int l16 = lane & 15, lg = lane >> 4;         // column / k group of a 16x16x32 operand
    const int d = p.dil, h1 = d * h2, NB = p.N;
    const int ntiles = (NB + nto - 1) / nto;
    // tiles are dealt to the XCDs in contiguous ranges (workgroup ids go round-robin over the 8 XCDs): neighbours share their halo rows in one L2
    const int per = (ntiles + 7) >> 3;
    const int tile = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (tile >= ntiles) return;
    const int n0 = tile * nto;                          // first output position
    const int t0 = n0 - h2;                             // first position of the intermediate
    const int wstart = t0 - h1;                         // first row of the conv1 window
    const bool interior = wstart >= 0 && wstart + K::ROWS1 <= NB;   // (uniform) every row the window loads touch exists

    // ---- weight pairs by LDS-DMA.  Pair u (conv1: 0 .. NP1 - 1, then conv2) = 2 consecutive steps = PAIRB contiguous bytes of pack_clx16's blocks
    // ([chunk][tap][row tile of 16]); slot u & 1.
    auto dma_pair = [&](auto uc) __attribute__((always_inline)) {
        constexpr int u = decltype(uc)::value;
        const char* W = u < NP1 ? static_cast<const char*>(p.W1x) + (int64_t)u * K::PAIRB : static_cast<const char*>(p.W2x) + (int64_t)(u - NP1) * K::PAIRB;
#pragma unroll
        for (int i = 0; i < K::NPC / K::NW; ++i) {
            const int pc = wave + K::NW * i;            // (uniform)
            __builtin_amdgcn_global_load_lds((x6_gbl_t*)(W + pc * 1024 + lane * 16),
                                             (x6_lds_t*)(uintptr_t)__builtin_amdgcn_readfirstlane(lds0 + (u & 1) * K::PAIRB + pc * 1024), 16, 0, 0);
        }
    };

    // biases and the keep flags of the intermediate's rows (position inside the batch and not masked): REQUESTED first, parked in LDS (for both epilogues)
    // behind the window requests: vmcnt returns in order, so parking them waits for nothing younger.  (No divergent branch around the loads: with one, the
    // compiler's wait-count pass put a vmcnt(0) behind the first weight DMA.)
    float bval = x6_gload32((tid & 64 ? p.b2 : p.b1) + min(tid & 63, C - 1));
    constexpr int NMV = (K::ROWS2 + T - 1) / T;
    static_assert(NMV <= 2, "keep flags: at most two per thread");
    unsigned mval[NMV];
#pragma unroll
    for (int h = 0; h < NMV; ++h) {
        const int pos = t0 + tid + h * T;
        const int pc = min(max(pos, 0), NB - 1);
        // (without a mask the byte is read from the plane and ignored)
        mval[h] = x6_gload8(p.mask ? p.mask + (pc >> max(p.mask_shift, 0)) : reinterpret_cast<const unsigned char*>(p.X));
    }
    // ---- conv1 window: f32 rows -> registers -> lrelu, hi / lo -> LDS, a chunk PAIR (one 128-byte line of a row) at a time.  Thread: row (tid >> 2) of every
    // 64-row block, 16-byte quad (tid & 3) of a chunk's 64-byte row piece.  Interior tiles read one contiguous range.
    x6_f32x4 rx[NCP][2][NXC];
    {
        dma_pair(std::integral_constant<int, 0>{});
        __builtin_amdgcn_sched_barrier(0);   // (pair 0's weights are the OLDEST request in flight: pair_barrier(0) relies on it)
        if (interior) {   // one contiguous range: uniform row-block bases + one lane offset + immediates
            const char* xwin = reinterpret_cast<const char*>(p.X) + (int64_t)wstart * (C * 4);
            const unsigned xlane = (unsigned)((tid >> 2) * (C * 4) + (tid & 3) * 16);
            x6_for<0, NCP>([&](auto prc) __attribute__((always_inline)) {
                x6_for<0, 2>([&](auto cc) __attribute__((always_inline)) {
                    x6_for<0, NXC>([&](auto ic) __attribute__((always_inline)) {
                        constexpr int pr = decltype(prc)::value, c = decltype(cc)::value, i = decltype(ic)::value;
                        rx[pr][c][i] = x6_gload128s<pr * 128 + c * 64>(xwin + (int64_t)i * (RB * C * 4), xlane);
                    });
                });
            });
        } else {          // (the batch's first / last tiles) clamped rows; the same number of requests in the same order
#pragma unroll
            for (int pr = 0; pr < NCP; ++pr)
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int i = 0; i < NXC; ++i) {
                        const int pos = min(max(wstart + i * RB + (tid >> 2), 0), NB - 1);   // (rows outside the batch are zeroed by convert())
                        rx[pr][c][i] = x6_gload128(p.X + (int64_t)pos * C + (pr * 2 + c) * 16 + (tid & 3) * 4);
                        __builtin_amdgcn_sched_barrier(0);   // (one address at a time: hoisted in front of the burst they took 40 registers)
                    }
        }
    }
    const unsigned x1w = lds0 + K::WREG + (tid >> 2) * 32 + (tid & 3) * 8;
    const float slope = p.slope;
    auto convert_pair = [&](auto prc, auto edgec) __attribute__((always_inline)) {
        constexpr int pr = decltype(prc)::value;
        constexpr bool EDGE = decltype(edgec)::value;
        x6_for<0, 2>([&](auto cc) __attribute__((always_inline)) {
            constexpr int c = decltype(cc)::value;
            x6_for<0, NXC>([&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value;
                x6_f32x4 v = rx[pr][c][i];
                if constexpr (EDGE) {
                    const int pos = wstart + i * RB + (tid >> 2);
                    if (pos < 0 || pos >= NB) v = x6_f32x4{0.f, 0.f, 0.f, 0.f};
                }
                x6_bf16x4 h, l;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float x = fmaxf(v[e], v[e] * slope);   // leaky ReLU for 0 <= slope <= 1
                    h[e] = (__bf16)x;
                    l[e] = (__bf16)(x - (float)h[e]);
                }
                x6_write_b64<c * K::X1CH + i * RB * 32>(x1w, h);
                x6_write_b64<K::X1PART + c * K::X1CH + i * RB * 32>(x1w, l);
            });
        });
    };
    auto convert = [&](auto prc) __attribute__((always_inline)) {
        if (interior) convert_pair(prc, std::false_type{});
        else convert_pair(prc, std::true_type{});
    };

    // The residual rows of the epilogue are requested NOW, together with the conv1 window that contains them (respair_clx.hip has the measurement)
    constexpr int LPR = 8;                             // lanes per output row of the wave's transposed tile (32 channels = 128 bytes)
    constexpr int RPI = 64 / LPR, NIT = 64 / RPI;      // rows per iteration, iterations
    const int c4 = wm * 32 + (lane % LPR) * 4, rowi = lane / LPR;
    x6_f32x4 rres[NIT];
    if (interior) {
        const char* rb = reinterpret_cast<const char*>(p.X) + (int64_t)n0 * (C * 4);
        const unsigned rl = (unsigned)((wn * 64 + rowi) * (C * 4) + c4 * 4);
        x6_for<0, NIT>([&](auto ic) __attribute__((always_inline)) {
            constexpr int it = decltype(ic)::value;
            rres[it] = x6_gload128s<0>(rb + (int64_t)it * (RPI * C * 4), rl);
        });
    } else {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int64_t posr = min((int64_t)n0 + wn * 64 + it * RPI + rowi, (int64_t)NB - 1);
            rres[it] = x6_gload128(p.X + posr * C + c4);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // Requests of this wave in flight, oldest first: bias (1), keep flags (NMV), weight pair 0 (NPC / NW), window rows of chunk pair 0 (2 NXC), of chunk pair 1
    // (C = 64), residual rows (NIT).  Parking the bias / keep flags waits for the first 1 + NMV of them.
    constexpr int NRX = 2 * NXC, NYOUNG0 = (NCP - 1) * NRX + NIT;   // younger than chunk pair 0's rows
    if constexpr (NMV == 1) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(bval), "+v"(mval[0]) : "n"(K::NPC / K::NW + NRX + NYOUNG0));
    else asm volatile("s_waitcnt vmcnt(%3)" : "+v"(bval), "+v"(mval[0]), "+v"(mval[NMV - 1]) : "n"(K::NPC / K::NW + NRX + NYOUNG0));
#pragma unroll
    for (int h = 0; h < NMV; ++h) {
        const int pos = t0 + tid + h * T;
        mval[h] = (p.mask == nullptr || mval[h] != 0) && pos >= 0 && pos < NB ? 1u : 0u;
    }
    if (tid < 128) x6_write_b32(lds0 + K::BIAS_OFF + tid * 4, bval);
#pragma unroll
    for (int h = 0; h < NMV; ++h)
        if (tid + h * T < K::ROWS2) x6_write_b8(lds0 + K::MASK_OFF + tid + h * T, mval[h]);
    // chunk pair 0's rows (and, older than them, weight pair 0: pair_barrier(0) relies on it); the younger requests stay in flight
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int i = 0; i < NXC; ++i) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(rx[0][c][i]) : "n"(NYOUNG0));
    convert(std::integral_constant<int, 0>{});
    X6_STAMP(1);

    // ---- fragment addresses.  A: 1 KB block (row tile 2 wm + r, part) of the pair in slot s: immediate.  B: lane (column l16, k group lg): k groups 0, 1 read
    // step a's row, 2, 3 step b's; 16-byte half lg & 1 of the 32-byte row; the lo plane is an immediate further.  The two steps of a pair are one tap apart,
    // or (the middle pair of a chunk pair) from the first chunk's last tap to the second chunk's first.
    const unsigned abase = lds0 + lane * 16 + wm * 4096;
    const unsigned rowl = (unsigned)((wn * 64 + l16) * 32 + ((lg & 1) << 4));
    const int d32 = d * 32;
    const unsigned h1same = lds0 + K::WREG + rowl + (lg >= 2 ? d32 : 0);
    const unsigned h1wrap = lds0 + K::WREG + rowl + (lg >= 2 ? K::X1CH - (NTAPS - 1) * d32 : 0);
    const unsigned h2same = lds0 + K::WREG + rowl + (lg >= 2 ? 32 : 0);
    const unsigned h2wrap = lds0 + K::WREG + rowl + (lg >= 2 ? K::X2CH - (NTAPS - 1) * 32 : 0);
    struct Fr {
        x6_bf16x8 ah[2], al[2], bh[4], bl[4];
    };
    x6_f32x4 acc[2][4];
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[r][j] = x6_f32x4{0.f, 0.f, 0.f, 0.f};
        // (the zeros are written HERE, wait states before the first MFMA that takes them: left alone the compiler moves each v_mov in front of its first use,
        // and it inserts no wait states in front of an inline-asm MFMA)
        asm volatile("s_nop 3" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[0][2]), "+v"(acc[0][3]), "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[1][2]), "+v"(acc[1][3]));
    };
    // read rr (0 .. 5: a0, b0, b1, b2, b3, a1) of part `part` (0 = hi, 1 = lo) of A / B of the pair whose first local step is sa (of chunk pair blk of conv `conv`)
    auto read_one = [&](Fr& f, auto rrc, auto partc, auto convc, auto blkc, auto sac, auto slotc) __attribute__((always_inline)) {
        constexpr int rr = decltype(rrc)::value, part = decltype(partc)::value, conv = decltype(convc)::value, blk = decltype(blkc)::value, sa = decltype(sac)::value;
        constexpr int aoff = decltype(slotc)::value * K::PAIRB + part * 1024;
        constexpr int cpa = sa / NTAPS, tapa = sa % NTAPS;
        constexpr bool WRAP = tapa == NTAPS - 1;     // step b is the next chunk's first tap
        if constexpr (rr == 0) (part ? f.al[0] : f.ah[0]) = x6_read_b128<aoff>(abase);
        else if constexpr (rr == 5) (part ? f.al[1] : f.ah[1]) = x6_read_b128<aoff + 2048>(abase);
        else {
            constexpr int j = rr - 1;
            x6_bf16x8& dst = part ? f.bl[j] : f.bh[j];
            if constexpr (conv == 0) dst = x6_read_b128<part * K::X1PART + cpa * K::X1CH + j * 512>(x6_opaque(WRAP ? h1wrap : h1same) + (unsigned)(tapa * d32));
            else dst = x6_read_b128<part * K::X2PART + (blk * 2 + cpa) * K::X2CH + tapa * 32 + j * 512>(WRAP ? h2wrap : h2same);
        }
    };
    // the waits are tied to the registers they cover
    auto wait_lo_a_hi_b = [&](Fr& f) __attribute__((always_inline)) {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f.al[0]), "+v"(f.al[1]), "+v"(f.bh[0]), "+v"(f.bh[1]), "+v"(f.bh[2]), "+v"(f.bh[3]));
    };
    auto wait_hi_a_lo_b = [&](Fr& f) __attribute__((always_inline)) {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f.ah[0]), "+v"(f.ah[1]), "+v"(f.bl[0]), "+v"(f.bl[1]), "+v"(f.bl[2]), "+v"(f.bl[3]));
    };
    auto mfma_n = [&](const x6_bf16x8 (&a)[2], const x6_bf16x8 (&b)[4], auto nc) __attribute__((always_inline)) {
        constexpr int n = decltype(nc)::value;
        x6_mfma(acc[n >> 2][n & 3], a[n >> 2], b[n & 3]);
    };
    Fr fr[2];                    // the fragments of pair u live in fr[u & 1]: a pair's hi x hi product is issued behind the NEXT barrier
    // the barrier in front of pair u: its weights have landed (every wave's DMA pieces: vmcnt(0) then barrier), everybody is done with pair u - 1 (its slot
    // takes pair u + 1), and every LDS write issued so far is visible
    auto pair_barrier = [&](auto uc) __attribute__((always_inline)) {
        constexpr int u = decltype(uc)::value;
        // (pair 0: its weights were requested before the first chunk pair's window rows, which convert() has consumed: vmcnt returns in order, so they have
        // landed, while the second chunk pair's rows and the residual rows may still be in flight)
        if constexpr (u == 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        if (u == 1) {   // (vmcnt(0) above: the rest of the tile's requests have landed too; the compiler learns it here)
#pragma unroll
            for (int c = 0; c < (NCP > 1 ? 2 : 0); ++c)
#pragma unroll
                for (int i = 0; i < NXC; ++i) asm volatile("" : "+v"(rx[NCP - 1][c][i]));
#pragma unroll
            for (int it = 0; it < NIT; ++it) asm volatile("" : "+v"(rres[it]));
        }
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (u + 1 < NP1 + NP2) dma_pair(std::integral_constant<int, u + 1>{});
    };
    // The operands of a set of MFMAs stay allocated until the set has been issued: the fragment reads dealt between the MFMAs return asynchronously, and
    // the compiler (which sees neither the MFMAs nor the loads: inline asm) would otherwise give a load the registers of an operand whose last MFMA in
    // program order is still queued in front of the matrix pipe: the rows of the set's second half then saw the NEXT pair's data (run- and pair-dependent).
    auto keep_a = [&](const x6_bf16x8 (&a)[2]) __attribute__((always_inline)) { asm volatile("" ::"v"(a[0]), "v"(a[1])); };
    auto keep_b = [&](const x6_bf16x8 (&b)[4]) __attribute__((always_inline)) { asm volatile("" ::"v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3])); };
    // pair u behind its barrier: [the previous pair's hi x hi products, covering this pair's first reads] -> lo x hi -> hi x lo; its hi x hi stays pending
    auto pair_body = [&](auto uc, auto pendc) __attribute__((always_inline)) {
        constexpr int u = decltype(uc)::value;
        constexpr bool PEND = decltype(pendc)::value;
        constexpr int conv = u >= NP1 ? 1 : 0, v = conv ? u - NP1 : u;
        constexpr int blk = v / NTAPS, pi = v % NTAPS, sa = 2 * pi;
        using CV = std::integral_constant<int, conv>;
        using BK = std::integral_constant<int, blk>;
        using SA = std::integral_constant<int, sa>;
        using SL = std::integral_constant<int, u & 1>;
        using HI = std::integral_constant<int, 0>;
        using LO = std::integral_constant<int, 1>;
        Fr& f = fr[u & 1];
        Fr& g = fr[(u & 1) ^ 1];
        // ---- W_lo (A) and X_hi (B) of this pair
        if constexpr (PEND) {
            x6_for<0, 8>([&](auto nc) __attribute__((always_inline)) {
                constexpr int n = decltype(nc)::value;
                mfma_n(g.ah, g.bh, nc);
                if constexpr (n == 0 || n == 5) read_one(f, nc, LO{}, CV{}, BK{}, SA{}, SL{});   // al[0], al[1]
                else if constexpr (n < 5) read_one(f, nc, HI{}, CV{}, BK{}, SA{}, SL{});         // bh[0 .. 3]
                __builtin_amdgcn_sched_barrier(0);
            });
            keep_a(g.ah);
            keep_b(g.bh);
        } else {
            read_one(f, std::integral_constant<int, 0>{}, LO{}, CV{}, BK{}, SA{}, SL{});
            x6_for<1, 5>([&](auto rc) __attribute__((always_inline)) { read_one(f, rc, HI{}, CV{}, BK{}, SA{}, SL{}); });
            read_one(f, std::integral_constant<int, 5>{}, LO{}, CV{}, BK{}, SA{}, SL{});
        }
        wait_lo_a_hi_b(f);
        __builtin_amdgcn_sched_barrier(0);
        // ---- lo x hi; W_hi and X_lo
        x6_for<0, 8>([&](auto nc) __attribute__((always_inline)) {
            constexpr int n = decltype(nc)::value;
            mfma_n(f.al, f.bh, nc);
            if constexpr (n == 0 || n == 5) read_one(f, nc, HI{}, CV{}, BK{}, SA{}, SL{});   // ah[0], ah[1]
            else if constexpr (n < 5) read_one(f, nc, LO{}, CV{}, BK{}, SA{}, SL{});         // bl[0 .. 3]
            __builtin_amdgcn_sched_barrier(0);
        });
        keep_a(f.al);
        wait_hi_a_lo_b(f);
        __builtin_amdgcn_sched_barrier(0);
        // ---- hi x lo
        x6_for<0, 8>([&](auto nc) __attribute__((always_inline)) {
            mfma_n(f.ah, f.bl, nc);
            __builtin_amdgcn_sched_barrier(0);
        });
        keep_b(f.bl);
    };
    // the pending hi x hi products of pair u without a following pair (the end of a convolution / of a chunk pair)
    auto flush_m2 = [&](auto uc) __attribute__((always_inline)) {
        Fr& g = fr[decltype(uc)::value & 1];
        x6_for<0, 8>([&](auto nc) __attribute__((always_inline)) { mfma_n(g.ah, g.bh, nc); });
        keep_a(g.ah);
        keep_b(g.bh);
        // the accumulators are read by VALU / LDS instructions next: the compiler does not see these MFMAs (inline asm) and inserts no wait states
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };

    // ================================================================================================================================
    // ---- phase 1: t = lrelu(conv1(lrelu(y)) + b1) on positions [t0, t0 + NT) ------------------------------------------------------------
    zero_acc();
    x6_for<0, NCP>([&](auto bc) __attribute__((always_inline)) {
        constexpr int blk = decltype(bc)::value;
        constexpr int u0 = blk * NTAPS;
        pair_barrier(std::integral_constant<int, u0>{});
        if constexpr (blk > 0) {
            // the window is free once everybody has passed this barrier (all fragment reads of the previous chunk pair were waited for)
            flush_m2(std::integral_constant<int, u0 - 1>{});
            convert(bc);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (blk == 1) X6_STAMP(8);
        pair_body(std::integral_constant<int, u0>{}, std::false_type{});
        if constexpr (blk == 0) X6_STAMP(9);
        x6_for<1, NTAPS>([&](auto pc) __attribute__((always_inline)) {
            constexpr int u = u0 + decltype(pc)::value;
            pair_barrier(std::integral_constant<int, u>{});
            pair_body(std::integral_constant<int, u>{}, std::true_type{});
        });
        if constexpr (blk == 0) X6_STAMP(7);
    });
    X6_STAMP(2);

    // ---- intermediate: + b1, lrelu, keep flag, hi / lo -> the window conv2 reads (aliases the conv1 window: behind a barrier) -------------
    pair_barrier(std::integral_constant<int, NP1>{});
    flush_m2(std::integral_constant<int, NP1 - 1>{});
    {
        // accumulator tile [r][j]: lane (column l16, row group lg) holds channels wm * 32 + 16 r + 4 lg .. + 3 of position wn * 64 + 16 j + l16
        x6_f32x4 bq[2];
        bq[0] = x6_read_f128<0>(lds0 + K::BIAS_OFF + (wm * 32 + 4 * lg) * 4);
        bq[1] = x6_read_f128<64>(lds0 + K::BIAS_OFF + (wm * 32 + 4 * lg) * 4);
        unsigned mk[4];
        x6_for<0, 4>([&](auto jc) __attribute__((always_inline)) { mk[decltype(jc)::value] = x6_read_u8<decltype(jc)::value * 16>(lds0 + K::MASK_OFF + wn * 64 + l16); });
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bq[0]), "+v"(bq[1]), "+v"(mk[0]), "+v"(mk[1]), "+v"(mk[2]), "+v"(mk[3]));
        const bool allkeep = __builtin_amdgcn_ballot_w64(mk[0] != 0 && mk[1] != 0 && mk[2] != 0 && mk[3] != 0) == ~0ull;
        const unsigned x2w = lds0 + K::WREG + wm * 2 * K::X2CH + (wn * 64 + l16) * 32 + lg * 8;
        auto mid = [&](auto keepc) __attribute__((always_inline)) {
            constexpr bool ALL = decltype(keepc)::value;
            x6_for<0, 2>([&](auto rc) __attribute__((always_inline)) {
                constexpr int r = decltype(rc)::value;
                x6_for<0, 4>([&](auto jc) __attribute__((always_inline)) {
                    constexpr int j = decltype(jc)::value;
                    x6_bf16x4 h, l;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float t = acc[r][j][e] + bq[r][e];
                        t = fmaxf(t, t * slope);
                        if constexpr (!ALL) t = mk[j] != 0 ? t : 0.f;
                        h[e] = (__bf16)t;
                        l[e] = (__bf16)(t - (float)h[e]);
                    }
                    x6_write_b64<r * K::X2CH + j * 512>(x2w, h);
                    x6_write_b64<K::X2PART + r * K::X2CH + j * 512>(x2w, l);
                });
            });
        };
        if (allkeep) mid(std::true_type{});
        else mid(std::false_type{});
    }
    X6_STAMP(3);
    zero_acc();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    X6_STAMP(4);

    // ---- phase 2: conv2 over the LDS-resident intermediate -----------------------------------------------------------------------------------
    pair_body(std::integral_constant<int, NP1>{}, std::false_type{});
    x6_for<NP1 + 1, NP1 + NP2>([&](auto uc) __attribute__((always_inline)) {
        pair_barrier(uc);
        pair_body(uc, std::true_type{});
    });
    X6_STAMP(5);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();   // the transpose tiles overlay the weight buffers and the window
    flush_m2(std::integral_constant<int, NP1 + NP2 - 1>{});

    // ---- epilogue: + b2 + y, beta, accumulate, mask; full lines through a per-wave LDS transpose (no LDS-DMA is pending: plain LDS accesses) ---
    constexpr int TP = K::TPITCH;
    float* ttile = reinterpret_cast<float*>(smem) + wave * (64 * TP);
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<x6_f32x4*>(ttile + (j * 16 + l16) * TP + 16 * r + 4 * lg) = acc[r][j];
    const float* bias_s = reinterpret_cast<const float*>(smem + K::BIAS_OFF);
    const unsigned char* mask_s = reinterpret_cast<const unsigned char*>(smem + K::MASK_OFF);
    const x6_f32x4 b4 = *reinterpret_cast<const x6_f32x4*>(bias_s + 64 + c4);
    x6_f32x4 rold[NIT];
    if (p.accumulate) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int64_t po = min((int64_t)n0 + wn * 64 + it * RPI + rowi, (int64_t)NB - 1);
            rold[it] = *reinterpret_cast<const x6_f32x4*>(p.Y + po * C + c4);
        }
    }
    const float beta = p.beta;
    auto store_rows = [&](auto ntc) __attribute__((always_inline)) {
        constexpr bool NTS = decltype(ntc)::value;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int row = it * RPI + rowi;
            const int o = wn * 64 + row;                    // output index inside the workgroup's range
            const int pos = n0 + o;                         // < 2^31 (checked by the caller)
            const x6_f32x4 a = *reinterpret_cast<const x6_f32x4*>(ttile + row * TP + (lane % LPR) * 4);
            if (o >= nto || pos >= NB) continue;
            // (contraction is off in this file; respair_clx.hip is compiled with hipcc's default, which fuses `x * beta + old` into one fma: written out)
            x6_f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float s = a[e] + b4[e] + rres[it][e];
                v[e] = p.accumulate ? __builtin_fmaf(s, beta, rold[it][e]) : s * beta;
            }
            if (!mask_s[o + h2]) v = x6_f32x4{0.f, 0.f, 0.f, 0.f};   // position n0 + o = intermediate row o + h2
            if constexpr (NTS) __builtin_nontemporal_store(v, reinterpret_cast<x6_f32x4*>(p.Y + (int64_t)pos * C + c4));
            else *reinterpret_cast<x6_f32x4*>(p.Y + (int64_t)pos * C + c4) = v;
        }
    };
    if (p.nt_store) store_rows(std::true_type{});
    else store_rows(std::false_type{});
    X6_STAMP(6);
    X6_STAMP(15);
    if constexpr (DIAG) {
        if (threadIdx.x == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) p.stamps[(size_t)blockIdx.x * 16 + i] = st_[i];
        }
    }
#undef X6_STAMP
}

template <int C, int NTAPS, int DG>
static void launch_x6(const ResPairParams& p, hipStream_t stream) {
    using K = X6Cfg<C, NTAPS>;
    static_assert(K::LDS * 3 <= 160 * 1024, "three workgroups per CU");
    auto kern = respair_x16_kernel<C, NTAPS, DG>;
    static std::atomic<uint64_t> lds_allowed{0};   // per (kernel instantiation, device)
    allow_full_lds(reinterpret_cast<const void*>(kern), lds_allowed);
    constexpr int nto = K::NT - (NTAPS - 1);
    const int ntiles = (p.N + nto - 1) / nto;
    const int grid = ((ntiles + 7) >> 3) * 8;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const bool prof = DG < 0 && conv_prof_active();
    if (prof) {
        HIP_CHECK(hipEventCreate(&e0));
        HIP_CHECK(hipEventCreate(&e1));
        HIP_CHECK(hipEventRecord(e0, stream));
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(K::T), K::LDS, stream, p);
    HIP_CHECK(hipGetLastError());
    if (prof) {
        HIP_CHECK(hipEventRecord(e1, stream));
        conv_prof_add(C == 64 ? 17 : 16, 2.0 * 2.0 * p.C * (double)p.N * p.C * p.k, e0, e1);
    }
}

// p.mask_shift set (launch_respair_cl does it)
bool respair_x16_usable(const ResPairParams& p) {
    return p.W1x && p.W2x && p.split && !p.f16 && (p.C == 32 || p.C == 64) && (p.k == 7 || p.k == 11) && p.dil >= 1 && p.dil * (p.k - 1) <= 64 && p.slope >= 0.f &&
           p.slope <= 1.f && p.N >= 1 && (!p.mask || p.mask_shift >= 0);
}

template <int DG>
static void launch_x6_any(const ResPairParams& p, hipStream_t stream) {
    if (p.C == 64 && p.k == 7) return launch_x6<64, 7, DG>(p, stream);
    if (p.C == 64 && p.k == 11) return launch_x6<64, 11, DG>(p, stream);
    if (p.C == 32 && p.k == 7) return launch_x6<32, 7, DG>(p, stream);
    if (p.C == 32 && p.k == 11) return launch_x6<32, 11, DG>(p, stream);
    SBV2_REQUIRE(false, "respair_x16: shape not instantiated");
}

void launch_respair_x16(const ResPairParams& p0, hipStream_t stream) {
    SBV2_REQUIRE(respair_x16_usable(p0), "respair_x16: operands do not fit the kernel");
    ResPairParams p = p0;
    p.nt_store = (int64_t)p.N * p.C * 4 >= ((int64_t)128 << 20);
    launch_x6_any<-1>(p, stream);
}
void launch_respair_x16_diag(const ResPairParams& p, hipStream_t stream) {
    SBV2_REQUIRE(respair_x16_usable(p) && p.stamps, "respair_x16 diag: operands do not fit the kernel");
    launch_x6_any<0>(p, stream);
}

}  // namespace sbv2
