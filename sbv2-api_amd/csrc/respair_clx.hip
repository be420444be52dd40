// One HiFi-GAN ResBlock1 step fused into one launch for the decoder stages with C <= 64 channels, split-bf16: the round-4 rebuild of respair_cl.hip.
//
//     y' = beta * ( conv2( lrelu( conv1( lrelu(y), dilation d ) + b1 ) ) + b2 + y )  [+ previous contents]  (then column mask)
//
// (HifiGanResidualBlock.forward, transformers modeling_vits.py:455-463 = modules.ResBlock1 upstream; the graph scripts/convert/convert_model.py:97-110
// exports.)  Same arithmetic, operand split and per-accumulator summation order (chunk, tap, lo*hi, hi*lo, hi*hi) as respair_cl.hip / conv_cl.hip:
// bit-identical results at C = 32 / 64 (C = 16 multiplies two taps per 16x16x32 MFMA: another summation order, f32-grade; the tests hold it to 1e-5).
// What changed is everything around the MFMAs, after the measurements in profiles/r04a_respair_cl_*:
//   * respair_cl issues 1700 vector instructions per wave and 256-position tile beside its 168 MFMAs; VALU issue (4 cycles each) + MFMA busy time fill
//     the SIMD completely at C <= 32 (428k + 309k of 726k cycles per SIMD), i.e. the kernel is bound by its own instruction count, and at C = 64 its
//     one 8-wave workgroup per CU (134 KB of LDS) runs its phases one after the other (SIMD 77 % busy, 37 % of wave time parked).
//   * Here the tap count and channel count are template parameters and every loop is unrolled: LDS addresses are one lane base + immediates (the
//     windows are stored as [part][chunk][channel half][row] 16-byte cells, conflict-free for ds_read_b128 without an address swizzle), the conv1
//     window of an interior tile is one contiguous byte range read with base + immediate offsets, no integer division / 64-bit VALU address math.
//   * Weights go L2 -> LDS by LDS-DMA (global_load_lds, no staging registers, no ds_write) in groups of <= 4 taps, double buffered: the next group
//     lands while the current one is multiplied; one barrier per group.
//   * C = 64 runs 128-position tiles on 4 waves (52 KB of LDS: three workgroups per CU whose phases interleave) instead of 256 positions on 8.
//   * C = 16 multiplies two taps per MFMA (v_mfma_f32_16x16x32_bf16, weights packed as tap pairs at load): no half-empty A tiles.
//   * The fragment reads of the next tap are dealt one per gap between the current tap's MFMAs (sched_barrier pins them).
//   * Tiles are dealt to the XCDs in contiguous ranges (neighbouring tiles share their halo rows through one L2).
// Measured and NOT kept (profiles/r04d_respair_persistent_probe.txt): persistent workgroups that prefetch the next tile's window / mask bytes / first
// weight group (the per-tile dependency chain fell from 15.2k to 7.5k cycles at C = 16, 28.6k to 19.3k at C = 32) ran 5-10 % SLOWER: the loop keeps 40-60
// more registers live (one wave per SIMD less at C = 16), and these launches are not latency bound any more: C = 16 and every k = 3 launch move their
// 0.94 GB per half-size launch at 3.5-3.7 TB/s, C >= 32 at k >= 7 run 0.85-1.1 PFLOP/s of executed MFMA at a power-managed 1.65-1.75 GHz.
// Fragment reads, window writes and the waits that cover them are inline asm: with an LDS-DMA pending hipcc puts s_waitcnt vmcnt(0) in front of every
// LDS access it can see, which would serialise the weight stream with the MFMAs (conv_clx.hip has the same note).
#include <atomic>
#include <type_traits>

#include "common.h"

// RPX_PROBE16 (diagnostic builds only, tools/respair_shape_probe.sh; results WRONG): every v_mfma_f32_32x32x16_bf16 of the C >= 32 kernels issued as two
// v_mfma_f32_16x16x32_bf16 from the same fragment registers: what the instruction shape alone is worth at the managed clock
#ifndef RPX_PROBE16
#define RPX_PROBE16 0
#endif

namespace sbv2 {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void rpx_lds_t;
typedef const __attribute__((address_space(1))) void rpx_gbl_t;

template <int I, int N, class F>
__device__ __forceinline__ void rpx_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        rpx_for<I + 1, N>(f);
    }
}
template <int OFF>
__device__ __forceinline__ bf16x8 rpx_read_b128(unsigned addr) {
    bf16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF));
    return v;
}
template <int OFF>
__device__ __forceinline__ f32x4v rpx_read_f128(unsigned addr) {
    f32x4v v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF));
    return v;
}
template <int OFF>
__device__ __forceinline__ unsigned rpx_read_u8(unsigned addr) {
    unsigned v;
    asm volatile("ds_read_u8 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF));
    return v;
}
template <int OFF>
__device__ __forceinline__ void rpx_write_b64(unsigned addr, bf16x4 v) {
    asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(addr), "v"(v), "i"(OFF) : "memory");
}
__device__ __forceinline__ void rpx_write_b32(unsigned addr, float v) { asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ void rpx_write_b8(unsigned addr, unsigned v) { asm volatile("ds_write_b8 %0, %1" ::"v"(addr), "v"(v) : "memory"); }

constexpr int rpx_max(int a, int b) { return a > b ? a : b; }

// WNP = position groups of 64 per workgroup (0: the default of the channel count: 2 at C = 64, 4 below); a workgroup has NMT * WN waves
template <int C, int NTAPS, int GT, int WNP = 0>
struct RpxCfg {
    static constexpr int NCH = C / 16;                 // 16-channel chunks (the K dimension of one MFMA)
    static constexpr int NMT = C == 64 ? 2 : 1;        // 32-row tiles of the output channels (C = 16: half of the one tile is zero rows)
    static constexpr int WN = WNP > 0 ? WNP : (C == 64 ? 2 : 4);   // 64-position groups
    static constexpr int NW = NMT * WN;                // waves
    static constexpr int T = 64 * NW;                  // threads
    static constexpr int NT = 64 * WN;                 // positions of the intermediate per workgroup
    static constexpr bool TWOTAP = C == 16;            // C = 16: two taps per 32-deep MFMA (v_mfma_f32_16x16x32_bf16), see the kernel
    static constexpr int NTW = TWOTAP ? (NTAPS + 1) / 2 : NTAPS;   // weight steps of a chunk: taps, or tap pairs
    static constexpr int G = NTW < GT ? NTW : GT;      // weight steps per group
    static constexpr int NG = (NTW + G - 1) / G;
    static constexpr int WSLOT = G * NMT * 2048;       // one weight buffer: [row tile][tap of the group][part][1 KB fragment block]
    static constexpr int WREG = 2 * WSLOT;
    static constexpr int RB = T / 4;                   // window rows one load instruction of the workgroup covers (4 threads per 64-byte row piece)
    static constexpr int NXC = (NT + 64 + RB - 1) / RB;   // such blocks in the conv1 window (NT + tap span <= NT + 64)
    static constexpr int ROWS1 = NXC * RB;
    static constexpr int ROWS2 = NT + NTAPS + 1;       // rows of the intermediate window (NT + k - 1 are read)
    // conv1 window, one chunk: [part][channel half][row] 16-byte cells.  32x32x16 kernels: + 64 puts the two halves a conversion store touches on
    // different bank halves (a fragment read touches one half only).  16x16x32 (C = 16): a fragment read touches BOTH halves (k groups 0 / 2 and 1 / 3):
    // the half planes must be a multiple of 256 bytes apart or rows 4-11 of one collide with rows 0-3 / 12-15 of the other (PMC: conflict ratio 0.41)
    static constexpr int X1HALF = TWOTAP ? (ROWS1 * 16 + 255) / 256 * 256 : ROWS1 * 16 + 64;
    static constexpr int X1PART = 2 * X1HALF;
    static constexpr int X2HALF = TWOTAP ? (ROWS2 * 16 + 255) / 256 * 256 : ROWS2 * 16;   // intermediate: [part][chunk][channel half][row]
    static constexpr int X2CH = 2 * X2HALF;
    static constexpr int X2PART = NCH * X2CH;
    static constexpr int XREG = rpx_max(2 * X1PART, 2 * X2PART);   // the intermediate ALIASES the conv1 window
    static constexpr int TPITCH = TWOTAP ? 20 : 36;    // floats per row of the epilogue's transpose tiles
    static constexpr int TT = NW * 64 * TPITCH * 4;    // the epilogue's per-wave transpose tiles (overlay everything above)
    static constexpr int MAIN = rpx_max(WREG + XREG, TT);
    static constexpr int BIAS_OFF = MAIN;              // 64 floats b1, 64 floats b2
    static constexpr int MASK_OFF = MAIN + 512;        // one byte per row of the intermediate
    static constexpr int LDS = (MASK_OFF + ROWS2 + 16 + 15) / 16 * 16;
    static constexpr int NSEQ = 2 * NCH * NG;          // weight groups of a tile, in order: (conv, chunk, group)
};

// DG >= 0: diagnostic instantiation (phase stamps of thread 0 into p.stamps[16 per workgroup]; sbv2_debug_respair_clock)
template <int C, int NTAPS, int DG, int GT, int WNP = 0>
__global__ __launch_bounds__((RpxCfg<C, NTAPS, GT, WNP>::T)) __attribute__((amdgpu_waves_per_eu(3))) void respair_clx_kernel(const ResPairParams p) {
    using K = RpxCfg<C, NTAPS, GT, WNP>;
    constexpr int T = K::T, NW = K::NW, WN = K::WN, RB = K::RB;
    constexpr bool DIAG = DG >= 0;
    constexpr int NCH = K::NCH, NMT = K::NMT, NT = K::NT, G = K::G, NG = K::NG, NXC = K::NXC, NTW = K::NTW;
    constexpr bool TWOTAP = K::TWOTAP;
    constexpr int PAIR = NCH >= 2 ? 2 : 1;             // chunks whose 64-byte row pieces are requested together (one 128-byte line)
    constexpr int NQ = 4;                              // 4-row groups of a 32 x 32 accumulator tile
    constexpr int h2 = (NTAPS - 1) / 2, nto = NT - 2 * h2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)smem);

    unsigned st_[16];
    auto stamp = [&](auto ic) {
        if constexpr (DIAG) {
            constexpr int i = decltype(ic)::value;
            st_[i] = (unsigned)(i >= 14 ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime());
        }
    };
#define RPX_STAMP(i) stamp(std::integral_constant<int, i>{})
    if constexpr (DIAG) {
#pragma unroll
        for (int i = 0; i < 16; ++i) st_[i] = 0;
    }
    RPX_STAMP(0);
    RPX_STAMP(14);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = NMT == 2 ? wave / WN : 0;           // this wave's 32-row tile of the output channels
    const int wn = NMT == 2 ? wave - wm * WN : wave;   // ... and its 64 positions
    const int lcol = lane & 31, lh = lane >> 5;
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

    // ---- weight groups by LDS-DMA.  Sequence index s = (conv * NCH + chunk) * NG + g, buffer s & 1.
    auto dma_group = [&](auto sc) {
        constexpr int s = decltype(sc)::value;
        constexpr int conv = s / (NCH * NG), chunk = (s / NG) % NCH, g = s % NG;
        constexpr int ntg = NTW - g * G < G ? NTW - g * G : G;
        constexpr int NP = NMT * G * 2;                 // 1 KB piece slots of a buffer
        const char* W = static_cast<const char*>(TWOTAP ? (conv ? p.W2p : p.W1p) : (conv ? p.W2 : p.W1));
#pragma unroll
        for (int i = 0; i < (NP + NW - 1) / NW; ++i) {
            const int pc = wave + NW * i;               // (uniform) piece = ((row tile * G + tap in group) * 2 + part)
            const int part = pc & 1, tgx = pc >> 1, mt = tgx / G, tg = tgx - mt * G;
            if (pc < NP && tg < ntg) {
                const char* src = W + ((((int64_t)(chunk * NMT + mt) * NTW + g * G + tg) * 2 + part) << 10) + lane * 16;
                __builtin_amdgcn_global_load_lds((rpx_gbl_t*)src, (rpx_lds_t*)(uintptr_t)__builtin_amdgcn_readfirstlane(lds0 + (s & 1) * K::WSLOT + pc * 1024), 16, 0, 0);
            }
        }
    };

    // ---- conv1 window: f32 rows -> registers -> lrelu, hi / lo -> LDS.  Thread: row (tid >> 2) of every 64-row block, 16-byte quad (tid & 3) of the
    // chunk's 64-byte row piece.  Interior tiles read one contiguous range: uniform base + lane offset + immediates.
    f32x4v rx[PAIR][NXC];
    const char* xwin = reinterpret_cast<const char*>(p.X) + (int64_t)wstart * (C * 4);   // (uniform; only dereferenced for interior tiles)
    const unsigned xlane = (unsigned)((tid >> 2) * (C * 4) + (tid & 3) * 16);
    auto load_pair = [&](int pr) {
        if (interior) {
            const char* b = xwin + pr * (PAIR * 64);
#pragma unroll
            for (int c = 0; c < PAIR; ++c)
#pragma unroll
                for (int i = 0; i < NXC; ++i) rx[c][i] = *reinterpret_cast<const f32x4v*>(b + (i * RB * C * 4 + c * 64) + (size_t)xlane);
        } else {
#pragma unroll
            for (int c = 0; c < PAIR; ++c)
#pragma unroll
                for (int i = 0; i < NXC; ++i) {
                    const int pos = min(max(wstart + i * RB + (tid >> 2), 0), NB - 1);
                    rx[c][i] = *reinterpret_cast<const f32x4v*>(p.X + (int64_t)pos * C + (pr * PAIR + c) * 16 + (tid & 3) * 4);
                }
        }
    };
    const unsigned x1w = lds0 + K::WREG + ((tid & 3) >> 1) * K::X1HALF + (tid >> 2) * 16 + (tid & 1) * 8;
    const float slope = p.slope;
    auto convert_one = [&](auto cc, auto edgec) {
        constexpr int c = decltype(cc)::value;
        constexpr bool EDGE = decltype(edgec)::value;
        rpx_for<0, NXC>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            f32x4v v = rx[c][i];
            if constexpr (EDGE) {
                const int pos = wstart + i * RB + (tid >> 2);
                if (pos < 0 || pos >= NB) v = f32x4v{0.f, 0.f, 0.f, 0.f};
            }
            bf16x4 h, l;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float x = fmaxf(v[e], v[e] * slope);   // leaky ReLU for 0 <= slope <= 1
                h[e] = (__bf16)x;
                l[e] = (__bf16)(x - (float)h[e]);
            }
            rpx_write_b64<i * RB * 16>(x1w, h);
            rpx_write_b64<K::X1PART + i * RB * 16>(x1w, l);
        });
    };
    auto convert = [&](auto cc) {
        if (interior) convert_one(cc, std::false_type{});
        else convert_one(cc, std::true_type{});
    };

    // ---- fragments.  C >= 32: v_mfma_f32_32x32x16_bf16, A = 32 output channels x 16 input channels of one tap, B = 16 channels x 32 positions; a wave
    // owns 32 rows x 64 positions.  C = 16 (TWOTAP): v_mfma_f32_16x16x32_bf16, the 32-deep K dimension carries TWO taps x 16 channels (k groups 0, 1 =
    // tap 2 tp, channels 0-7 / 8-15; groups 2, 3 = tap 2 tp + 1): A = [W_2tp | W_2tp+1] packed at load (pack_cl_pairs; the phantom tap of an odd kernel
    // size is zero weights), B lanes of k groups 2, 3 read the window one tap further; a wave owns 16 rows x 64 positions = four 16 x 16 accumulators
    // that SHARE the A fragment.  Round 2's variant of this halved the MFMAs on top of respair_cl's LDS traffic and gained nothing; here a tap pair costs
    // 2 + 8 fragment reads per 12 half-length MFMAs and no address arithmetic.  (Summation order: 32 products per MFMA instead of 2 x 16: not bit-identical
    // to the 32x32x16 kernels, f32-grade all the same.)
    constexpr int NB_ = TWOTAP ? 4 : 2;                // position tiles of a wave (16 or 32 positions each)
    constexpr int NRD = 2 + 2 * NB_;                   // fragment reads per weight step
    constexpr int NMF = 3 * NB_;                       // MFMAs per weight step
    struct Frags {
        bf16x8 ah, al, bh[NB_], bl[NB_];
    };
    const int lg = lane >> 4, l16 = lane & 15;         // (TWOTAP) k group / column of a 16x16x32 operand
    const unsigned abase = lds0 + lane * 16 + wm * (G * 2048);
    const unsigned b1base = TWOTAP ? lds0 + K::WREG + (lg & 1) * K::X1HALF + (wn * 64 + l16) * 16 + (lg >> 1) * d * 16
                                   : lds0 + K::WREG + lh * K::X1HALF + (wn * 64 + lcol) * 16;
    const unsigned b2base = TWOTAP ? lds0 + K::WREG + (lg & 1) * K::X2HALF + (wn * 64 + l16) * 16 + (lg >> 1) * 16
                                   : lds0 + K::WREG + lh * K::X2HALF + (wn * 64 + lcol) * 16;
    // (TWOTAP, odd kernel size) the lanes of the phantom tap read the row of the last real tap: finite data times zero weights
    const unsigned b1last = b1base - (lg >> 1) * d * 16, b2last = b2base - (lg >> 1) * 16;
    f32x16 acc[TWOTAP ? 1 : 2];
    f32x4v acc4[TWOTAP ? 4 : 1];
#if RPX_PROBE16
    f32x4v accq[2][4];     // (shape probe: the 32x32 tiles as four 16x16x32 accumulators each; results are WRONG)
#endif
    auto zero_acc = [&]() {
#if RPX_PROBE16
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) accq[j][q] = f32x4v{0.f, 0.f, 0.f, 0.f};
#endif
        if constexpr (TWOTAP) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc4[j] = f32x4v{0.f, 0.f, 0.f, 0.f};
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        }
    };
    // read r (0 .. NRD - 1) of the fragments of (conv, chunk, weight step ws = tap or tap pair); weight buffer s & 1, step tg of its group.
    // Order: al, bh0, ah, bh1, then the remaining B fragments in the order the MFMAs take them.
    auto read_one = [&](Frags& f, auto rc, auto convc, auto chunkc, auto wsc, auto sc, auto tgc) {
        constexpr int r = decltype(rc)::value;
        constexpr int conv = decltype(convc)::value, chunk = decltype(chunkc)::value, ws = decltype(wsc)::value;
        constexpr int aoff = (decltype(sc)::value & 1) * K::WSLOT + decltype(tgc)::value * 2048;
        if constexpr (r == 0) f.al = rpx_read_b128<aoff + 1024>(abase);
        else if constexpr (r == 2) f.ah = rpx_read_b128<aoff>(abase);
        else if constexpr (TWOTAP) {
            // B read order: bh0 (r 1), bh1 (3), bh2 (4), bh3 (5), bl0 .. bl3 (6 .. 9)
            constexpr int e = r == 1 ? 0 : r - 2;                    // 0 .. 7: bh0..3, bl0..3
            constexpr int j = e & 3, part = e >> 2;
            constexpr bool LASTP = (NTAPS & 1) && ws == NTW - 1;     // the pair with the phantom tap
            if constexpr (conv == 0) {
                const unsigned b = (LASTP ? b1last : b1base) + ws * 2 * d * 16;
                if constexpr (part == 0) f.bh[j] = rpx_read_b128<j * 256>(b);
                else f.bl[j] = rpx_read_b128<K::X1PART + j * 256>(b);
            } else {
                constexpr int o = ws * 2 * 16 + j * 256;
                if constexpr (part == 0) f.bh[j] = rpx_read_b128<o>(LASTP ? b2last : b2base);
                else f.bl[j] = rpx_read_b128<K::X2PART + o>(LASTP ? b2last : b2base);
            }
        } else if constexpr (conv == 0) {
            const unsigned b = b1base + ws * d * 16;
            if constexpr (r == 1) f.bh[0] = rpx_read_b128<0>(b);
            else if constexpr (r == 3) f.bh[1] = rpx_read_b128<512>(b);
            else if constexpr (r == 4) f.bl[0] = rpx_read_b128<K::X1PART>(b);
            else f.bl[1] = rpx_read_b128<K::X1PART + 512>(b);
        } else {
            constexpr int o = chunk * K::X2CH + ws * 16;
            if constexpr (r == 1) f.bh[0] = rpx_read_b128<o>(b2base);
            else if constexpr (r == 3) f.bh[1] = rpx_read_b128<o + 512>(b2base);
            else if constexpr (r == 4) f.bl[0] = rpx_read_b128<K::X2PART + o>(b2base);
            else f.bl[1] = rpx_read_b128<K::X2PART + o + 512>(b2base);
        }
    };
    // the wait is tied to the registers it covers: the MFMAs that consume them cannot be scheduled above it
    auto wait_frags = [&](Frags& f) {
        if constexpr (TWOTAP)
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(f.ah), "+v"(f.al), "+v"(f.bh[0]), "+v"(f.bh[1]), "+v"(f.bh[2]), "+v"(f.bh[3]), "+v"(f.bl[0]), "+v"(f.bl[1]), "+v"(f.bl[2]), "+v"(f.bl[3]));
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f.ah), "+v"(f.al), "+v"(f.bh[0]), "+v"(f.bh[1]), "+v"(f.bl[0]), "+v"(f.bl[1]));
    };
    auto mfma_one = [&](const Frags& f, auto nc) {   // term-major; per accumulator: lo*hi, hi*lo, hi*hi (conv_cl's order)
        constexpr int n = decltype(nc)::value;
        constexpr int t = n / NB_, j = n % NB_;
#if RPX_PROBE16
        auto& aq = accq;   // (named outside the discarded branch: the generic lambda captures it)
#endif
        if constexpr (TWOTAP) {
            if constexpr (t == 0) acc4[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.al, f.bh[j], acc4[j], 0, 0, 0);
            else if constexpr (t == 1) acc4[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.ah, f.bl[j], acc4[j], 0, 0, 0);
            else acc4[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.ah, f.bh[j], acc4[j], 0, 0, 0);
        } else {
#if RPX_PROBE16
            // the same FLOP from the same fragment registers as two v_mfma_f32_16x16x32_bf16 per 32x32x16 instruction (timing only)
            constexpr int sl = (t & 1) * 2;
            const bf16x8& a = t == 0 ? f.al : f.ah;
            const bf16x8& b = t == 1 ? f.bl[j] : f.bh[j];
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(aq[j & 1][sl]) : "v"(a), "v"(b));
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(aq[j & 1][sl + 1]) : "v"(a), "v"(b));
#else
            if constexpr (t == 0) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.al, f.bh[j], acc[j], 0, 0, 0);
            else if constexpr (t == 1) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah, f.bl[j], acc[j], 0, 0, 0);
            else acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah, f.bh[j], acc[j], 0, 0, 0);
#endif
        }
    };
    auto probe_fold = [&]() {   // (shape probe) the 16x16 accumulators -> the registers the epilogues read
#if RPX_PROBE16
        if constexpr (!TWOTAP) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[j][4 * q + e] = accq[j][q][e];
        }
#endif
    };
    // one weight group: its steps; the fragment reads of step t + 1 are dealt one per gap between the MFMAs of step t (a burst of reads in front of
    // the MFMAs fills the LDS command queue and leaves the matrix pipe idle while it drains)
    auto run_group = [&](auto sc) {
        constexpr int s = decltype(sc)::value;
        constexpr int conv = s / (NCH * NG), chunk = (s / NG) % NCH, g = s % NG;
        constexpr int ntg = NTW - g * G < G ? NTW - g * G : G;
        using CV = std::integral_constant<int, conv>;
        using CK = std::integral_constant<int, chunk>;
        Frags f[2];
        rpx_for<0, NRD>([&](auto rc) { read_one(f[0], rc, CV{}, CK{}, std::integral_constant<int, g * G>{}, sc, std::integral_constant<int, 0>{}); });
        rpx_for<0, ntg>([&](auto tc) {
            constexpr int tg = decltype(tc)::value;
            wait_frags(f[tg & 1]);
            __builtin_amdgcn_sched_barrier(0);
            rpx_for<0, NMF>([&](auto nc) {
                mfma_one(f[tg & 1], nc);
                if constexpr (tg + 1 < ntg && decltype(nc)::value < NRD)
                    read_one(f[(tg + 1) & 1], nc, CV{}, CK{}, std::integral_constant<int, g * G + tg + 1>{}, sc, std::integral_constant<int, tg + 1>{});
                __builtin_amdgcn_sched_barrier(0);
            });
        });
    };
    // the barrier in front of group s: its weights have landed (every wave's DMA pieces: vmcnt(0) then barrier), everybody is done with group s - 1
    // (its buffer takes group s + 1), and every LDS write issued so far is visible
    auto group_barrier = [&](auto sc) {
        constexpr int s = decltype(sc)::value;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (s + 1 < K::NSEQ) dma_group(std::integral_constant<int, s + 1>{});
    };

    // ================================================================================================================================
    dma_group(std::integral_constant<int, 0>{});
    load_pair(0);
    // The residual rows of the epilogue are requested NOW, together with the conv1 window that contains them: the second request for a line that is in
    // flight (or just landed) is served by L2.  Requested before conv2, as rounds 1-3 did, they were HBM reads again: with three 64 KB tiles per CU the
    // XCD's 4 MB L2 has turned over by then (PMC: 2.14 GB fetched per C = 32 launch for 0.94 GB of plane, profiles/r04f_pmc_hbm_traffic.csv).  Price: NIT x 4
    // registers live over the tile.
    const bool rres_early = !p.rres_late;
    constexpr int LPR = C == 16 ? 4 : 8;               // lanes per output row of the wave's transposed tile (32 channels = 128 bytes; C = 16: 64 bytes)
    constexpr int RPI = 64 / LPR, NIT = 64 / RPI;      // rows per iteration, iterations
    const int c4 = wm * 32 + (lane % LPR) * 4, rowi = lane / LPR;
    f32x4v rres[NIT];
    auto load_rres = [&]() {
        if (interior) {
            const char* rb = reinterpret_cast<const char*>(p.X) + (int64_t)n0 * (C * 4);   // (uniform)
            const unsigned rl = (unsigned)((wn * 64 + rowi) * (C * 4) + c4 * 4);
#pragma unroll
            for (int it = 0; it < NIT; ++it) rres[it] = *reinterpret_cast<const f32x4v*>(rb + it * RPI * C * 4 + (size_t)rl);
        } else {
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int64_t posr = min((int64_t)n0 + wn * 64 + it * RPI + rowi, (int64_t)NB - 1);
                rres[it] = *reinterpret_cast<const f32x4v*>(p.X + posr * C + c4);
            }
        }
    };
    if (rres_early) load_rres();
    {
        // biases and the keep flags of the intermediate's rows (position inside the batch and not masked), parked in LDS for both epilogues
        const float bval = tid < 128 ? (tid < 64 ? p.b1[min(tid, C - 1)] : p.b2[min(tid - 64, C - 1)]) : 0.f;
        constexpr int NMV = (K::ROWS2 + T - 1) / T;
        unsigned mval[NMV];
#pragma unroll
        for (int h = 0; h < NMV; ++h) {
            const int pos = t0 + tid + h * T;
            const int pc = min(max(pos, 0), NB - 1);
            const unsigned m = p.mask ? p.mask[pc >> p.mask_shift] : 1u;
            mval[h] = (pos >= 0 && pos < NB) ? m : 0u;
        }
        if (tid < 128) rpx_write_b32(lds0 + K::BIAS_OFF + tid * 4, bval);
#pragma unroll
        for (int h = 0; h < NMV; ++h)
            if (tid + h * T < K::ROWS2) rpx_write_b8(lds0 + K::MASK_OFF + tid + h * T, mval[h]);
    }
    convert(std::integral_constant<int, 0>{});
    RPX_STAMP(1);

    // ---- phase 1: t = lrelu(conv1(lrelu(y)) + b1) on positions [t0, t0 + NT) -> LDS ----------------------------------------------------
    zero_acc();
    rpx_for<0, NCH>([&](auto cc) {
        constexpr int chunk = decltype(cc)::value;
        if constexpr (chunk > 0) {
            // the window buffer is free once everybody has passed the barrier of this chunk's first group (all reads of chunk - 1 were waited for)
            group_barrier(std::integral_constant<int, chunk * NG>{});
            convert(std::integral_constant<int, chunk & (PAIR - 1)>{});
            if constexpr ((chunk & (PAIR - 1)) == PAIR - 1 && chunk + 1 < NCH) load_pair((chunk + 1) / PAIR);   // the next pair (its registers are free now)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            run_group(std::integral_constant<int, chunk * NG>{});
        } else {
            group_barrier(std::integral_constant<int, 0>{});
            run_group(std::integral_constant<int, 0>{});
        }
        if constexpr (chunk == 0) RPX_STAMP(7);
        rpx_for<1, NG>([&](auto gc) {
            constexpr int s = chunk * NG + decltype(gc)::value;
            group_barrier(std::integral_constant<int, s>{});
            run_group(std::integral_constant<int, s>{});
        });
        if constexpr (chunk == 0) RPX_STAMP(8);
    });
    RPX_STAMP(2);

    // ---- intermediate: + b1, lrelu, keep flag, hi / lo -> the window conv2 reads (aliases the conv1 window: behind a barrier) -------------
    constexpr int S2 = NCH * NG;   // first group of conv2
    group_barrier(std::integral_constant<int, S2>{});
    probe_fold();
    if constexpr (TWOTAP) {
        // accumulator tile jt: lane (column l16 = position, k group lg) holds channels 4 lg .. 4 lg + 3 of position wn * 64 + 16 jt + l16
        f32x4v bq = rpx_read_f128<0>(lds0 + K::BIAS_OFF + lg * 16);
        unsigned mk[4];
        rpx_for<0, 4>([&](auto jc) { mk[decltype(jc)::value] = rpx_read_u8<decltype(jc)::value * 16>(lds0 + K::MASK_OFF + wn * 64 + l16); });
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bq), "+v"(mk[0]), "+v"(mk[1]), "+v"(mk[2]), "+v"(mk[3]));
        const bool allkeep = __builtin_amdgcn_ballot_w64(mk[0] != 0 && mk[1] != 0 && mk[2] != 0 && mk[3] != 0) == ~0ull;
        const unsigned x2w = lds0 + K::WREG + (lg >> 1) * K::X2HALF + (wn * 64 + l16) * 16 + (lg & 1) * 8;
        auto mid = [&](auto keepc) {
            constexpr bool ALL = decltype(keepc)::value;
            rpx_for<0, 4>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                bf16x4 h, l;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float t = acc4[j][e] + bq[e];
                    t = fmaxf(t, t * slope);
                    if constexpr (!ALL) t = mk[j] != 0 ? t : 0.f;
                    h[e] = (__bf16)t;
                    l[e] = (__bf16)(t - (float)h[e]);
                }
                rpx_write_b64<j * 256>(x2w, h);
                rpx_write_b64<K::X2PART + j * 256>(x2w, l);
            });
        };
        if (allkeep) mid(std::true_type{});
        else mid(std::false_type{});
    } else {
        f32x4v bq[NQ];
        rpx_for<0, NQ>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            bq[q] = rpx_read_f128<q * 32>(lds0 + K::BIAS_OFF + (wm * 32 + 4 * lh) * 4);
        });
        unsigned mk0 = rpx_read_u8<0>(lds0 + K::MASK_OFF + wn * 64 + lcol), mk1 = rpx_read_u8<32>(lds0 + K::MASK_OFF + wn * 64 + lcol);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bq[0]), "+v"(bq[1]), "+v"(bq[2]), "+v"(bq[3]), "+v"(mk0), "+v"(mk1));
        const bool keep0 = mk0 != 0, keep1 = mk1 != 0;
        const bool allkeep = __builtin_amdgcn_ballot_w64(keep0 && keep1) == ~0ull;   // (uniform) the usual tile: no masked column, no batch end
        const unsigned x2w = lds0 + K::WREG + wm * 2 * K::X2CH + (wn * 64 + lcol) * 16 + lh * 8;
        auto mid = [&](auto keepc) {
            constexpr bool ALL = decltype(keepc)::value;
            rpx_for<0, 2>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                rpx_for<0, NQ>([&](auto qc) {
                    constexpr int q = decltype(qc)::value;
                    bf16x4 h, l;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float t = acc[j][4 * q + e] + bq[q][e];
                        t = fmaxf(t, t * slope);
                        if constexpr (!ALL) t = (j ? keep1 : keep0) ? t : 0.f;
                        h[e] = (__bf16)t;
                        l[e] = (__bf16)(t - (float)h[e]);
                    }
                    constexpr int o = (q >> 1) * K::X2CH + (q & 1) * K::X2HALF + j * 512;
                    rpx_write_b64<o>(x2w, h);
                    rpx_write_b64<K::X2PART + o>(x2w, l);
                });
            });
        };
        if (allkeep) mid(std::true_type{});
        else mid(std::false_type{});
    }
    RPX_STAMP(3);
    zero_acc();
    if (!rres_early) load_rres();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    RPX_STAMP(4);

    // ---- phase 2: conv2 over the LDS-resident intermediate -----------------------------------------------------------------------------------
    run_group(std::integral_constant<int, S2>{});
    rpx_for<S2 + 1, K::NSEQ>([&](auto sc) {
        group_barrier(sc);
        run_group(sc);
    });
    RPX_STAMP(5);
    probe_fold();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();   // the transpose tiles overlay the weight buffers and the window

    // ---- epilogue: + b2 + y, beta, accumulate, mask; full lines through a per-wave LDS transpose (no LDS-DMA is pending: plain LDS accesses) ---
    constexpr int TP = K::TPITCH;
    float* ttile = reinterpret_cast<float*>(smem) + wave * (64 * TP);
    if constexpr (TWOTAP) {
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4v*>(ttile + (j * 16 + l16) * TP + 4 * lg) = acc4[j];
    } else {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                f32x4v v = {acc[j][4 * q], acc[j][4 * q + 1], acc[j][4 * q + 2], acc[j][4 * q + 3]};
                *reinterpret_cast<f32x4v*>(ttile + (j * 32 + lcol) * TP + 8 * q + 4 * lh) = v;
            }
    }
    const float* bias_s = reinterpret_cast<const float*>(smem + K::BIAS_OFF);
    const unsigned char* mask_s = reinterpret_cast<const unsigned char*>(smem + K::MASK_OFF);
    const f32x4v b4 = *reinterpret_cast<const f32x4v*>(bias_s + 64 + c4);
    f32x4v rold[NIT];
    if (p.accumulate) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int64_t po = min((int64_t)n0 + wn * 64 + it * RPI + rowi, (int64_t)NB - 1);
            rold[it] = *reinterpret_cast<const f32x4v*>(p.Y + po * C + c4);
        }
    }
    const float beta = p.beta;
    // the stores of planes that do not fit the caches bypass them (p.nt_store, set by the launch: conv_clx.hip has the measurement); two copies of the loop
    // behind one uniform branch
    auto store_rows = [&](auto ntc) {
        constexpr bool NT = decltype(ntc)::value;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int row = it * RPI + rowi;
            const int o = wn * 64 + row;                    // output index inside the workgroup's range
            const int pos = n0 + o;                         // < 2^31 (checked by the caller)
            const f32x4v a = *reinterpret_cast<const f32x4v*>(ttile + row * TP + (lane % LPR) * 4);
            if (o >= nto || pos >= NB) continue;
            f32x4v v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (a[e] + b4[e] + rres[it][e]) * beta;
            if (p.accumulate) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += rold[it][e];
            }
            if (!mask_s[o + h2]) v = f32x4v{0.f, 0.f, 0.f, 0.f};   // position n0 + o = intermediate row o + h2
            if constexpr (NT) __builtin_nontemporal_store(v, reinterpret_cast<f32x4v*>(p.Y + (int64_t)pos * C + c4));
            else *reinterpret_cast<f32x4v*>(p.Y + (int64_t)pos * C + c4) = v;
        }
    };
    if (p.nt_store) store_rows(std::true_type{});
    else store_rows(std::false_type{});
    RPX_STAMP(6);
    RPX_STAMP(15);
    if constexpr (DIAG) {
        if (threadIdx.x == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) p.stamps[(size_t)blockIdx.x * 16 + i] = st_[i];
        }
    }
#undef RPX_STAMP
}

template <int C, int NTAPS, int DG, int GT, int WNP = 0>
static void launch_rpx(const ResPairParams& p, hipStream_t stream) {
    using K = RpxCfg<C, NTAPS, GT, WNP>;
    static_assert(K::LDS <= 160 * 1024, "LDS budget");
    auto kern = respair_clx_kernel<C, NTAPS, DG, GT, WNP>;
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

bool respair_clx_usable(const ResPairParams& p) {
    if (p.C == 16 && !(p.W1p && p.W2p)) return false;   // the 16-channel kernel wants the tap-pair packing of the weights (pack_cl_pairs)
    return p.split && !p.f16 && (p.C == 16 || p.C == 32 || p.C == 64) && (p.k == 3 || p.k == 7 || p.k == 11) && p.dil >= 1 && p.dil * (p.k - 1) <= 64 &&
           p.slope >= 0.f && p.slope <= 1.f && p.N >= 1 && (!p.mask || p.mask_shift >= 0);
}

template <int DG>
static void launch_rpx_any(const ResPairParams& p, hipStream_t stream) {
    // taps per weight group at C = 64: 2 = 52 KB of LDS, three workgroups per CU (11.75 vs 12.3 ms per step for the stage against groups of 4 / two per CU;
    // wider tiles on 6-wave workgroups measured 20-30 % slower: profiles/HISTORY.md)
#define RPX_CASE(CC, KK) \
    if (p.C == CC && p.k == KK) return launch_rpx<CC, KK, DG, 4>(p, stream);
    if (p.C == 64) {
        if (p.k == 3) return launch_rpx<64, 3, DG, 2>(p, stream);
        if (p.k == 7) return launch_rpx<64, 7, DG, 2>(p, stream);
        if (p.k == 11) return launch_rpx<64, 11, DG, 2>(p, stream);
    }
    RPX_CASE(16, 3) RPX_CASE(16, 7) RPX_CASE(16, 11) RPX_CASE(32, 3) RPX_CASE(32, 7) RPX_CASE(32, 11)
#undef RPX_CASE
    SBV2_REQUIRE(false, "respair_clx: shape not instantiated");
}

// p.mask_shift must be set (launch_respair_cl does it)
void launch_respair_clx(const ResPairParams& p0, hipStream_t stream) {
    SBV2_REQUIRE(respair_clx_usable(p0), "respair_clx: operands do not fit the kernel");
    ResPairParams p = p0;
    p.rres_late = 0;
    p.nt_store = (int64_t)p.N * p.C * 4 >= ((int64_t)128 << 20);
    launch_rpx_any<-1>(p, stream);
}
void launch_respair_clx_diag(const ResPairParams& p0, hipStream_t stream) {
    ResPairParams p = p0;
    p.mask_shift = 0;
    while (p.mask && (1 << p.mask_shift) < p.mask_div) ++p.mask_shift;
    SBV2_REQUIRE(respair_clx_usable(p) && p.stamps, "respair_clx diag: operands do not fit the kernel");
    launch_rpx_any<0>(p, stream);
}

}  // namespace sbv2
