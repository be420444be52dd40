// A whole HiFi-GAN ResBlock1 BRANCH (its three conv1 -> conv2 steps) in ONE launch, split-bf16 (round 6): the k = 3 branches of the decoder stages with
// C <= 128 channels and the k = 7 / 11 branches of the 16-channel stage.
//
//     y_0 = y;   y_q = keep * ( conv2_q( lrelu( conv1_q( lrelu(y_{q-1}), dilation d_q ) + b1_q ) * keep ) + b2_q + y_{q-1} ),  q = 1 .. 3
//     result = beta * y_3  [+ previous contents]
//
// (HifiGanResidualBlock.forward, transformers modeling_vits.py:455-463 = modules.ResBlock1 upstream; the graph scripts/convert/convert_model.py:97-110
// exports.)  respair_clx.hip runs ONE step per launch: 2 plane passes through HBM per step, 6 per branch.  At k = 3 those launches are HBM bound (4.4 - 4.9 TB/s at
// C = 16 / 32) or, at C = 64, pay their HBM time and their MFMA time one after the other (2.8 TB/s, MFMA busy 0.40: 0.39 ms of bytes + 0.27 ms of MFMA = the
// 0.68 ms they take).  This kernel keeps y_1 and y_2 on the chip: 2 plane passes per BRANCH.
//   * One window of R = 64 * WN rows per workgroup, indexed by absolute position for ALL six convolutions ("same" convolutions in place): row r of the window
//     is position w0 + r at every layer, a convolution reads rows r + (tap - h) * d, and what lies within a convolution's reach of the window's edge is simply
//     wrong from then on.  After the branch's six convolutions HALO = sum_q (d_q + 1) h rows per side are wrong (12 at k = 3, dilations 1, 3, 5) and are not
//     stored: R - 2 HALO outputs per workgroup, 1.10x recompute at R = 256 (1.23x at 128).  Rows outside the batch [0, N) and masked columns are forced to zero
//     at every layer by a select (the zero padding of every convolution): garbage in the margins / edge rows never reaches a row that is stored.
//   * Because a wave owns the same rows x channels at every layer, the f32 residual stream y_q never leaves its REGISTERS (accumulator layout); only its
//     bf16 hi / lo parts go through LDS, as the next convolution's operand.
//   * ONE operand window in LDS, reused in place: conv1 reads lrelu(y) parts, (barrier) the intermediate's parts overwrite them, conv2 reads those, (barrier)
//     the next step's lrelu(y') parts overwrite them.  Layout [part][chunk][channel half][MARG + R + MARG rows] of 16-byte cells as in respair_clx.hip: a B
//     fragment is ds_read_b128 of 32 consecutive cells at base + immediate.
//   * Weights of the six convolutions stream L2 -> LDS by LDS-DMA in groups of <= GT taps, double buffered across conv and step boundaries.
// Same operand split, same fragments and the same per-accumulator order (chunk, tap, lo*hi, hi*lo, hi*hi) as three respair_clx launches: BIT-IDENTICAL to them
// (tests/test_gpu_parity.py::test_resbranch_kernel_same_bits_as_three_respair_steps); C = 16 multiplies two taps per v_mfma_f32_16x16x32_bf16 exactly as
// respair_clx's C = 16 instance does.
#include <atomic>
#include <type_traits>

#include "common.h"

#pragma clang fp contract(off)

namespace sbv2 {

typedef __bf16 rb_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 rb_bf16x4 __attribute__((ext_vector_type(4)));
typedef float rb_f32x16 __attribute__((ext_vector_type(16)));
typedef float rb_f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void rb_lds_t;
typedef const __attribute__((address_space(1))) void rb_gbl_t;

template <int I, int N, class F>
__device__ __forceinline__ void rb_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        rb_for<I + 1, N>(f);
    }
}
// (LDS instructions take a 16-bit immediate offset: what lies beyond - the lo part of a 128-channel window - goes into the address, one add the compiler shares)
template <int OFF>
__device__ __forceinline__ rb_bf16x8 rb_read_b128(unsigned addr) {
    rb_bf16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr + (unsigned)(OFF & ~0xFFFF)), "i"(OFF & 0xFFFF));
    return v;
}
template <int OFF>
__device__ __forceinline__ rb_f32x4 rb_read_f128(unsigned addr) {
    rb_f32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr + (unsigned)(OFF & ~0xFFFF)), "i"(OFF & 0xFFFF));
    return v;
}
template <int OFF>
__device__ __forceinline__ unsigned rb_read_u8(unsigned addr) {
    unsigned v;
    asm volatile("ds_read_u8 %0, %1 offset:%2" : "=v"(v) : "v"(addr + (unsigned)(OFF & ~0xFFFF)), "i"(OFF & 0xFFFF));
    return v;
}
template <int OFF>
__device__ __forceinline__ void rb_write_b64(unsigned addr, rb_bf16x4 v) {
    asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(addr + (unsigned)(OFF & ~0xFFFF)), "v"(v), "i"(OFF & 0xFFFF) : "memory");
}
__device__ __forceinline__ void rb_write_b32(unsigned addr, float v) { asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ void rb_write_b8(unsigned addr, unsigned v) { asm volatile("ds_write_b8 %0, %1" ::"v"(addr), "v"(v) : "memory"); }

constexpr int rb_max(int a, int b) { return a > b ? a : b; }
// rows in front of / behind the window that a tap may reach: k = 3 with dilation <= 6 keeps the layout the sweeps below were measured on; the wider kernels
// (C = 16 only) cover dilation <= 5: 15 -> 16 rows at k = 7, 25 -> 26 at k = 11
constexpr int rb_margin(int k) { return k <= 3 ? kResBranchMargin : ((5 * (k - 1) / 2 + 1) & ~1); }

template <int C, int NTAPS, int WNP, int GT, int NBUFP>
struct RbCfg {
    static constexpr int NS = kResBranchSteps;         // steps of a branch
    static constexpr int NCH = C / 16;                 // 16-channel chunks (the K dimension of one MFMA)
    static constexpr int NMT = C >= 32 ? C / 32 : 1;   // 32-row tiles of the output channels (C = 16: one 16-row tile)
    static constexpr int WN = WNP;                     // 64-position groups
    static constexpr int NW = NMT * WN;                // waves
    static constexpr int T = 64 * NW;                  // threads
    static constexpr int R = 64 * WN;                  // rows of the window = positions every convolution is evaluated at
    static constexpr bool TWOTAP = C == 16;            // two taps per 32-deep MFMA (v_mfma_f32_16x16x32_bf16), as respair_clx's C = 16 instance
    static constexpr int NTW = TWOTAP ? (NTAPS + 1) / 2 : NTAPS;   // weight steps of a chunk: taps, or tap pairs
    static constexpr int G = NTW < GT ? NTW : GT;      // weight steps per group
    static constexpr int NG = (NTW + G - 1) / G;
    static constexpr int WSLOT = G * NMT * 2048;       // one weight buffer: [row tile][tap of the group][part][1 KB fragment block]
    static constexpr int NBUF = NBUFP;                 // ring of weight buffers: group gs lives in slot gs % NBUF and is requested NBUF - 1 groups ahead
    static constexpr int WREG = NBUF * WSLOT;
    static constexpr int NP = NMT * G * 2;             // 1 KB pieces of a full group, dealt round-robin over the waves
    static constexpr int MARG = rb_margin(NTAPS);      // rows in front of / behind the window that a tap may reach (contents: don't care)
    static constexpr int XROWS = R + 2 * MARG;
    // 16x16x32 (C = 16): a fragment read touches BOTH channel halves: they must be a multiple of 256 bytes apart (respair_clx.hip has the measurement)
    static constexpr int XHALF = TWOTAP ? (XROWS * 16 + 255) / 256 * 256 : XROWS * 16;
    static constexpr int XCH = 2 * XHALF;
    static constexpr int XPART = NCH * XCH;
    static constexpr int XREG = 2 * XPART;
    static constexpr int RB = T / 4;                   // window rows one load instruction of the workgroup covers (4 threads per 64-byte row piece)
    static constexpr int NXC = R / RB;                 // such blocks in the window
    static constexpr int TPITCH = TWOTAP ? 20 : 36;    // floats per row of the epilogue's transpose tiles
    static constexpr int TT = NW * 64 * TPITCH * 4;    // the epilogue's per-wave transpose tiles (overlay everything above)
    static constexpr int MAIN = rb_max(WREG + XREG, TT);
    static constexpr int BIAS_OFF = MAIN;              // 2 NS x C floats: b1, b2 of step 0, b1, b2 of step 1, ...
    static constexpr int MASK_OFF = MAIN + 2 * NS * C * 4;   // one byte per window row
    static constexpr int LDS = (MASK_OFF + R + 15) / 16 * 16;
    static constexpr int NSEQ = 2 * NCH * NG;          // weight groups of ONE step, in order: (conv, chunk, group)
    static constexpr int NTOT = NS * NSEQ;             // ... of the branch
};

// DG >= 0: diagnostic instantiation (phase stamps of thread 0 into p.stamps[16 per workgroup]; sbv2_debug_resbranch_clock)
template <int C, int NTAPS, int WNP, int GT, int NBUFP, int DG>
__global__ __launch_bounds__((RbCfg<C, NTAPS, WNP, GT, NBUFP>::T)) __attribute__((amdgpu_waves_per_eu(3))) void resbranch_clx_kernel(const ResBranchParams p) {
    using K = RbCfg<C, NTAPS, WNP, GT, NBUFP>;
    constexpr int T = K::T, NW = K::NW, WN = K::WN, RB = K::RB, R = K::R, NS = K::NS, MARG = K::MARG;
    constexpr bool DIAG = DG >= 0;
    constexpr int NCH = K::NCH, NMT = K::NMT, G = K::G, NG = K::NG, NXC = K::NXC, NTW = K::NTW, NBUF = K::NBUF, NSEQ = K::NSEQ, NTOT = K::NTOT;
    constexpr bool TWOTAP = K::TWOTAP;
    constexpr int NQ = 4;                              // 4-row groups of a 32 x 32 accumulator tile
    constexpr int h2 = (NTAPS - 1) / 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)smem);

    unsigned st_[16];
    auto stamp = [&](int i) __attribute__((always_inline)) {
        if constexpr (DIAG) st_[i] = (unsigned)(i >= 14 ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime());
    };
    if constexpr (DIAG) {
#pragma unroll
        for (int i = 0; i < 16; ++i) st_[i] = 0;
    }
    stamp(0);
    stamp(14);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = NMT >= 2 ? wave / WN : 0;           // this wave's 32-row tile of the output channels
    const int wn = NMT >= 2 ? wave - wm * WN : wave;   // ... and its 64 rows of the window
    const int lcol = lane & 31, lh = lane >> 5;
    const int NB = p.N, halo = p.halo, nto = R - 2 * halo;
    const int ntiles = (NB + nto - 1) / nto;
    // tiles are dealt to the XCDs in contiguous ranges (workgroup ids go round-robin over the 8 XCDs): neighbours share their halo rows in one L2
    const int per = (ntiles + 7) >> 3;
    const int tile = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (tile >= ntiles) return;
    const int n0 = tile * nto;                          // first output position
    const int w0 = n0 - halo;                           // position of window row 0
    const bool interior = w0 >= 0 && w0 + R <= NB;      // (uniform) every window row is a position of the batch

    // ---- weight groups by LDS-DMA.  Group gs = q * NSEQ + (conv * NCH + chunk) * NG + g of the branch lives in ring slot gs % NBUF and is requested NBUF - 1
    // groups ahead (behind the barrier of group gs - NBUF + 1, when everybody is done with the slot's previous group): an L2 -> LDS request takes ~1 us to
    // land under load, a group's MFMAs 0.1 - 0.3 us; with the double buffer of respair_clx.hip every group barrier of this kernel waited for its weights.
    auto dma_group = [&](auto gsc) __attribute__((always_inline)) {
        constexpr int gs = decltype(gsc)::value;
        constexpr int q = gs / NSEQ, s = gs % NSEQ;
        constexpr int conv = s / (NCH * NG), chunk = (s / NG) % NCH, g = s % NG;
        constexpr int ntg = NTW - g * G < G ? NTW - g * G : G;
        constexpr int NP = K::NP;
        const char* W = static_cast<const char*>(p.W[2 * q + conv]);
#pragma unroll
        for (int i = 0; i < (NP + NW - 1) / NW; ++i) {
            const int pc = wave + NW * i;               // (uniform) piece = ((row tile * G + tap in group) * 2 + part)
            const int part = pc & 1, tgx = pc >> 1, mt = tgx / G, tg = tgx - mt * G;
            if (pc < NP && tg < ntg) {
                const char* src = W + ((((int64_t)(chunk * NMT + mt) * NTW + g * G + tg) * 2 + part) << 10) + lane * 16;
                __builtin_amdgcn_global_load_lds((rb_gbl_t*)src, (rb_lds_t*)(uintptr_t)__builtin_amdgcn_readfirstlane(lds0 + (gs % NBUF) * K::WSLOT + pc * 1024), 16, 0, 0);
            }
        }
    };
    // pieces of group gs THIS wave requested (wave-uniform): the counted wait below must know how many younger requests may stay in flight
    auto pieces_of = [&](auto gsc) __attribute__((always_inline)) -> int {
        constexpr int gs = decltype(gsc)::value;
        constexpr int g = (gs % NSEQ) % NG;
        constexpr int ntg = NTW - g * G < G ? NTW - g * G : G;
        int n = 0;
#pragma unroll
        for (int i = 0; i < (K::NP + NW - 1) / NW; ++i) {
            const int pc = wave + NW * i;
            const int tgx = pc >> 1, mt = tgx / G, tg = tgx - mt * G;
            n += (pc < K::NP && tg < ntg) ? 1 : 0;
        }
        return n;
    };
    // ---- the window of y_0: f32 rows -> registers -> lrelu, hi / lo -> LDS, all chunks.  Thread: row (tid >> 2) of every RB-row block, 16-byte quad
    // (tid & 3) of a chunk's 64-byte row piece.
    rb_f32x4 rx[NCH][NXC];
    {
        const unsigned xlane = (unsigned)((tid >> 2) * (C * 4) + (tid & 3) * 16);
        if (interior) {
            const char* xwin = reinterpret_cast<const char*>(p.X) + (int64_t)w0 * (C * 4);   // (uniform)
#pragma unroll
            for (int c = 0; c < NCH; ++c)
#pragma unroll
                for (int i = 0; i < NXC; ++i) rx[c][i] = *reinterpret_cast<const rb_f32x4*>(xwin + (i * RB * C * 4 + c * 64) + (size_t)xlane);
        } else {
#pragma unroll
            for (int c = 0; c < NCH; ++c)
#pragma unroll
                for (int i = 0; i < NXC; ++i) {
                    const int pos = min(max(w0 + i * RB + (tid >> 2), 0), NB - 1);
                    rx[c][i] = *reinterpret_cast<const rb_f32x4*>(p.X + (int64_t)pos * C + c * 16 + (tid & 3) * 4);
                }
        }
    }
    rb_for<0, (NBUF - 1 < NTOT ? NBUF - 1 : NTOT)>([&](auto gsc) __attribute__((always_inline)) { dma_group(gsc); });

    // ---- the residual stream in ACCUMULATOR layout, from the same lines (L2 / L1 hits): 32x32 tile j, register 4 q + e of lane (lcol, lh) = channel
    // wm * 32 + 8 q + 4 lh + e of row wn * 64 + 32 j + lcol; C = 16: 16x16 tile j, register e of lane (l16, lg) = channel 4 lg + e of row wn * 64 + 16 j + l16
    const int lg = lane >> 4, l16 = lane & 15;         // (TWOTAP) k group / column of a 16x16x32 operand
    rb_f32x16 yres[TWOTAP ? 1 : 2];
    rb_f32x4 yres4[TWOTAP ? 4 : 1];
    if constexpr (TWOTAP) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int pos = min(max(w0 + wn * 64 + 16 * j + l16, 0), NB - 1);
            yres4[j] = *reinterpret_cast<const rb_f32x4*>(p.X + (int64_t)pos * C + 4 * lg);
        }
    } else {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int pos = min(max(w0 + wn * 64 + 32 * j + lcol, 0), NB - 1);
            const float* src = p.X + (int64_t)pos * C + wm * 32 + 4 * lh;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const rb_f32x4 v = *reinterpret_cast<const rb_f32x4*>(src + 8 * q);
#pragma unroll
                for (int e = 0; e < 4; ++e) yres[j][4 * q + e] = v[e];
            }
        }
    }
    {
        // biases of the six convolutions and the keep flags of the window's rows (position inside the batch and not masked), parked in LDS
        constexpr int NBV = (2 * NS * C + T - 1) / T;
#pragma unroll
        for (int h = 0; h < NBV; ++h) {
            const int i = tid + h * T;
            if (i < 2 * NS * C) rb_write_b32(lds0 + K::BIAS_OFF + i * 4, p.b[i / C][i % C]);
        }
        constexpr int NMV = (R + T - 1) / T;
#pragma unroll
        for (int h = 0; h < NMV; ++h) {
            const int r = tid + h * T;
            const int pos = w0 + r;
            const int pc = min(max(pos, 0), NB - 1);
            const unsigned m = p.mask ? p.mask[pc >> p.mask_shift] : 1u;
            if (r < R) rb_write_b8(lds0 + K::MASK_OFF + r, (pos >= 0 && pos < NB) ? m : 0u);
        }
    }
    const float slope = p.slope;
    {
        const unsigned x1w = lds0 + K::WREG + ((tid & 3) >> 1) * K::XHALF + (MARG + (tid >> 2)) * 16 + (tid & 1) * 8;
        auto convert_all = [&](auto edgec) __attribute__((always_inline)) {
            constexpr bool EDGE = decltype(edgec)::value;
            rb_for<0, NCH>([&](auto cc) __attribute__((always_inline)) {
                constexpr int c = decltype(cc)::value;
                rb_for<0, NXC>([&](auto ic) __attribute__((always_inline)) {
                    constexpr int i = decltype(ic)::value;
                    rb_f32x4 v = rx[c][i];
                    if constexpr (EDGE) {
                        const int pos = w0 + i * RB + (tid >> 2);
                        if (pos < 0 || pos >= NB) v = rb_f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                    rb_bf16x4 h, l;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float x = fmaxf(v[e], v[e] * slope);   // leaky ReLU for 0 <= slope <= 1
                        h[e] = (__bf16)x;
                        l[e] = (__bf16)(x - (float)h[e]);
                    }
                    rb_write_b64<c * K::XCH + i * RB * 16>(x1w, h);
                    rb_write_b64<K::XPART + c * K::XCH + i * RB * 16>(x1w, l);
                });
            });
        };
        if (interior) convert_all(std::false_type{});
        else convert_all(std::true_type{});
    }
    stamp(1);

    // ---- fragments (respair_clx.hip).  C >= 32: v_mfma_f32_32x32x16_bf16, a wave owns 32 channels x 64 rows.  C = 16: v_mfma_f32_16x16x32_bf16 whose 32-deep K
    // carries TWO taps x 16 channels (weights packed as tap pairs, pack_cl_pairs; the lanes of k groups 2, 3 read the window one tap further), 16 channels x 64
    // rows = four 16 x 16 accumulators that share the A fragment.
    constexpr int NB_ = TWOTAP ? 4 : 2;                // row tiles of a wave (16 or 32 rows each)
    constexpr int NRD = 2 + 2 * NB_;                   // fragment reads per weight step
    constexpr int NMF = 3 * NB_;                       // MFMAs per weight step
    struct Frags {
        rb_bf16x8 ah, al, bh[NB_], bl[NB_];
    };
    const unsigned abase = lds0 + lane * 16 + wm * (G * 2048);
    // row r of the window lives in cell MARG + r of its (part, chunk, half) plane
    const unsigned bbase = TWOTAP ? lds0 + K::WREG + (lg & 1) * K::XHALF + (MARG + wn * 64 + l16) * 16
                                  : lds0 + K::WREG + lh * K::XHALF + (MARG + wn * 64 + lcol) * 16;
    const unsigned c2base = bbase - h2 * 16 + (TWOTAP ? (lg >> 1) * 16 : 0);   // conv2 (dilation 1): tap ws reads row r + ws - h2
    const unsigned c2last = bbase - h2 * 16;           // (TWOTAP, odd kernel size) the lanes of the phantom tap read the row of the last real tap
    unsigned c1base = 0, c1last = 0;                   // conv1 of the current step (dilation d): set per step
    int dcur = 1;
    rb_f32x16 acc[TWOTAP ? 1 : 2];
    rb_f32x4 acc4[TWOTAP ? 4 : 1];
    auto zero_acc = [&]() __attribute__((always_inline)) {
        if constexpr (TWOTAP) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc4[j] = rb_f32x4{0.f, 0.f, 0.f, 0.f};
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        }
    };
    // read r (0 .. NRD - 1) of the fragments of (conv, chunk, weight step ws = tap or tap pair); weight buffer s & 1, step tg of its group.
    // Order: al, bh0, ah, bh1, then the remaining B fragments in the order the MFMAs take them.
    auto read_one = [&](Frags& f, auto rc, auto convc, auto chunkc, auto wsc, auto sc, auto tgc) __attribute__((always_inline)) {
        constexpr int r = decltype(rc)::value;
        constexpr int conv = decltype(convc)::value, chunk = decltype(chunkc)::value, ws = decltype(wsc)::value;
        constexpr int aoff = (decltype(sc)::value % NBUF) * K::WSLOT + decltype(tgc)::value * 2048;   // (sc: the group's index in the BRANCH)
        if constexpr (r == 0) f.al = rb_read_b128<aoff + 1024>(abase);
        else if constexpr (r == 2) f.ah = rb_read_b128<aoff>(abase);
        else if constexpr (TWOTAP) {
            // B read order: bh0 (r 1), bh1 (3), bh2 (4), bh3 (5), bl0 .. bl3 (6 .. 9)
            constexpr int e = r == 1 ? 0 : r - 2;                    // 0 .. 7: bh0..3, bl0..3
            constexpr int j = e & 3, part = e >> 2;
            constexpr bool LASTP = (NTAPS & 1) && ws == NTW - 1;     // the pair with the phantom tap
            if constexpr (conv == 0) {
                const unsigned b = (LASTP ? c1last : c1base) + ws * 2 * dcur * 16;
                if constexpr (part == 0) f.bh[j] = rb_read_b128<j * 256>(b);
                else f.bl[j] = rb_read_b128<K::XPART + j * 256>(b);
            } else {
                constexpr int o = ws * 2 * 16 + j * 256;
                if constexpr (part == 0) f.bh[j] = rb_read_b128<o>(LASTP ? c2last : c2base);
                else f.bl[j] = rb_read_b128<K::XPART + o>(LASTP ? c2last : c2base);
            }
        } else if constexpr (conv == 0) {
            const unsigned b = c1base + ws * dcur * 16;
            constexpr int o = chunk * K::XCH;
            if constexpr (r == 1) f.bh[0] = rb_read_b128<o>(b);
            else if constexpr (r == 3) f.bh[1] = rb_read_b128<o + 512>(b);
            else if constexpr (r == 4) f.bl[0] = rb_read_b128<K::XPART + o>(b);
            else f.bl[1] = rb_read_b128<K::XPART + o + 512>(b);
        } else {
            constexpr int o = chunk * K::XCH + ws * 16;
            if constexpr (r == 1) f.bh[0] = rb_read_b128<o>(c2base);
            else if constexpr (r == 3) f.bh[1] = rb_read_b128<o + 512>(c2base);
            else if constexpr (r == 4) f.bl[0] = rb_read_b128<K::XPART + o>(c2base);
            else f.bl[1] = rb_read_b128<K::XPART + o + 512>(c2base);
        }
    };
    // the wait is tied to the registers it covers: the MFMAs that consume them cannot be scheduled above it
    auto wait_frags = [&](Frags& f) __attribute__((always_inline)) {
        if constexpr (TWOTAP)
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(f.ah), "+v"(f.al), "+v"(f.bh[0]), "+v"(f.bh[1]), "+v"(f.bh[2]), "+v"(f.bh[3]), "+v"(f.bl[0]), "+v"(f.bl[1]), "+v"(f.bl[2]), "+v"(f.bl[3]));
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f.ah), "+v"(f.al), "+v"(f.bh[0]), "+v"(f.bh[1]), "+v"(f.bl[0]), "+v"(f.bl[1]));
    };
    auto mfma_one = [&](const Frags& f, auto nc) __attribute__((always_inline)) {   // term-major; per accumulator: lo*hi, hi*lo, hi*hi (conv_cl's order)
        constexpr int n = decltype(nc)::value;
        constexpr int t = n / NB_, j = n % NB_;
        if constexpr (TWOTAP) {
            if constexpr (t == 0) acc4[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.al, f.bh[j], acc4[j], 0, 0, 0);
            else if constexpr (t == 1) acc4[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.ah, f.bl[j], acc4[j], 0, 0, 0);
            else acc4[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.ah, f.bh[j], acc4[j], 0, 0, 0);
        } else {
            if constexpr (t == 0) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.al, f.bh[j], acc[j], 0, 0, 0);
            else if constexpr (t == 1) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah, f.bl[j], acc[j], 0, 0, 0);
            else acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah, f.bh[j], acc[j], 0, 0, 0);
        }
    };
    // one weight group (gs: its index in the branch): its steps; the fragment reads of step t + 1 are dealt one per gap between the MFMAs of step t
    auto run_group = [&](auto gsc) __attribute__((always_inline)) {
        constexpr int gs = decltype(gsc)::value, s = gs % NSEQ;
        constexpr int conv = s / (NCH * NG), chunk = (s / NG) % NCH, g = s % NG;
        constexpr int ntg = NTW - g * G < G ? NTW - g * G : G;
        using CV = std::integral_constant<int, conv>;
        using CK = std::integral_constant<int, chunk>;
        Frags f[2];
        rb_for<0, NRD>([&](auto rc) __attribute__((always_inline)) { read_one(f[0], rc, CV{}, CK{}, std::integral_constant<int, g * G>{}, gsc, std::integral_constant<int, 0>{}); });
        rb_for<0, ntg>([&](auto tc) __attribute__((always_inline)) {
            constexpr int tg = decltype(tc)::value;
            wait_frags(f[tg & 1]);
            __builtin_amdgcn_sched_barrier(0);
            rb_for<0, NMF>([&](auto nc) __attribute__((always_inline)) {
                mfma_one(f[tg & 1], nc);
                if constexpr (tg + 1 < ntg && decltype(nc)::value < NRD)
                    read_one(f[(tg + 1) & 1], nc, CV{}, CK{}, std::integral_constant<int, g * G + tg + 1>{}, gsc, std::integral_constant<int, tg + 1>{});
                __builtin_amdgcn_sched_barrier(0);
            });
        });
    };
    // The barrier in front of group gs: its weights have landed (every wave waits for its own pieces: vmcnt is in order, so "at most the pieces of the
    // younger groups outstanding", then the barrier), everybody is done with group gs - 1 (its slot takes group gs + NBUF - 1), and every LDS write issued
    // so far is visible.
    auto group_barrier = [&](auto gsc) __attribute__((always_inline)) {
        constexpr int gs = decltype(gsc)::value;
        constexpr int ahead = NTOT - 1 - gs < NBUF - 2 ? NTOT - 1 - gs : NBUF - 2;   // younger groups already requested
        int young = 0;
        rb_for<0, ahead>([&](auto ac) __attribute__((always_inline)) { young += pieces_of(std::integral_constant<int, gs + 1 + decltype(ac)::value>{}); });
        // (a few immediates: vmcnt takes no register)
        if (young <= 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        else if (young == 1) asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");
        else if (young == 2) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
        else if (young == 3) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
        else if (young == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        else if (young <= 6) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");   // (stricter than needed is always correct)
        else asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (gs + NBUF - 1 < NTOT) dma_group(std::integral_constant<int, gs + NBUF - 1>{});
    };

    // the keep flags of this lane's rows (constant over the branch), read once the bytes other threads parked are visible (behind the first barrier)
    unsigned mk[NB_];
#pragma unroll
    for (int j = 0; j < NB_; ++j) mk[j] = 0;
    auto load_flags = [&]() __attribute__((always_inline)) {
        if constexpr (TWOTAP) {
            rb_for<0, 4>([&](auto jc) __attribute__((always_inline)) { mk[decltype(jc)::value] = rb_read_u8<decltype(jc)::value * 16>(lds0 + K::MASK_OFF + wn * 64 + l16); });
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(mk[0]), "+v"(mk[1]), "+v"(mk[2]), "+v"(mk[3]));
        } else {
            mk[0] = rb_read_u8<0>(lds0 + K::MASK_OFF + wn * 64 + lcol);
            mk[1] = rb_read_u8<32>(lds0 + K::MASK_OFF + wn * 64 + lcol);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(mk[0]), "+v"(mk[1]));
        }
    };
    bool allkeep = false;

    // the window cell this lane's accumulator registers go to as operand parts (the mid / step epilogues)
    const unsigned xw = TWOTAP ? lds0 + K::WREG + (lg >> 1) * K::XHALF + (MARG + wn * 64 + l16) * 16 + (lg & 1) * 8
                               : lds0 + K::WREG + wm * 2 * K::XCH + (MARG + wn * 64 + lcol) * 16 + lh * 8;
    // acc (+ bias [+ residual]) -> masked value; RES: the residual stream is added and replaced (the step's result), else the intermediate.  The value's
    // lrelu is split into the window (PARTS) unless this is the branch's last result.
    auto emit = [&](auto resc, auto partsc, auto keepc, int bias_row) __attribute__((always_inline)) {
        constexpr bool RES = decltype(resc)::value, PARTS = decltype(partsc)::value, ALL = decltype(keepc)::value;
        if constexpr (TWOTAP) {
            rb_f32x4 bq = rb_read_f128<0>(lds0 + K::BIAS_OFF + bias_row * (C * 4) + lg * 16);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bq));
            rb_for<0, 4>([&](auto jc) __attribute__((always_inline)) {
                constexpr int j = decltype(jc)::value;
                rb_bf16x4 h, l;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float t = acc4[j][e] + bq[e];
                    if constexpr (RES) t = t + yres4[j][e];
                    else t = fmaxf(t, t * slope);
                    if constexpr (!ALL) t = mk[j] != 0 ? t : 0.f;
                    if constexpr (RES) {
                        yres4[j][e] = t;
                        t = fmaxf(t, t * slope);
                    }
                    h[e] = (__bf16)t;
                    l[e] = (__bf16)(t - (float)h[e]);
                }
                if constexpr (PARTS) {
                    rb_write_b64<j * 256>(xw, h);
                    rb_write_b64<K::XPART + j * 256>(xw, l);
                }
            });
        } else {
            rb_f32x4 bq[NQ];
            rb_for<0, NQ>([&](auto qc) __attribute__((always_inline)) {
                constexpr int q = decltype(qc)::value;
                bq[q] = rb_read_f128<q * 32>(lds0 + K::BIAS_OFF + bias_row * (C * 4) + (wm * 32 + 4 * lh) * 4);
            });
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bq[0]), "+v"(bq[1]), "+v"(bq[2]), "+v"(bq[3]));
            rb_for<0, 2>([&](auto jc) __attribute__((always_inline)) {
                constexpr int j = decltype(jc)::value;
                rb_for<0, NQ>([&](auto qc) __attribute__((always_inline)) {
                    constexpr int q = decltype(qc)::value;
                    rb_bf16x4 h, l;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float t = acc[j][4 * q + e] + bq[q][e];
                        if constexpr (RES) t = t + yres[j][4 * q + e];
                        else t = fmaxf(t, t * slope);
                        if constexpr (!ALL) t = mk[j] != 0 ? t : 0.f;
                        if constexpr (RES) {
                            yres[j][4 * q + e] = t;
                            t = fmaxf(t, t * slope);
                        }
                        h[e] = (__bf16)t;
                        l[e] = (__bf16)(t - (float)h[e]);
                    }
                    if constexpr (PARTS) {
                        constexpr int o = (q >> 1) * K::XCH + (q & 1) * K::XHALF + j * 512;
                        rb_write_b64<o>(xw, h);
                        rb_write_b64<K::XPART + o>(xw, l);
                    }
                });
            });
        }
    };
    auto emit_any = [&](auto resc, auto partsc, int bias_row) __attribute__((always_inline)) {
        if (allkeep) emit(resc, partsc, std::true_type{}, bias_row);
        else emit(resc, partsc, std::false_type{}, bias_row);
    };

    // ================================================================================================================================
    rb_for<0, NS>([&](auto qc) __attribute__((always_inline)) {
        constexpr int q = decltype(qc)::value;
        constexpr bool more = q + 1 < NS;
        constexpr int S0 = q * NSEQ;           // first group of the step (conv1), S2: first group of conv2
        constexpr int S2 = S0 + NCH * NG;
        dcur = p.dil[q];
        c1base = bbase - h2 * dcur * 16 + (TWOTAP ? (lg >> 1) * dcur * 16 : 0);
        c1last = bbase - h2 * dcur * 16;
        // ---- conv1 (dilation d) over the window's lrelu(y) parts ---------------------------------------------------------------------------
        zero_acc();
        rb_for<S0, S2>([&](auto gsc) __attribute__((always_inline)) {
            group_barrier(gsc);
            if constexpr (decltype(gsc)::value == 0) {   // the keep flags other threads parked are visible now
                load_flags();
                if constexpr (TWOTAP) allkeep = __builtin_amdgcn_ballot_w64(mk[0] != 0 && mk[1] != 0 && mk[2] != 0 && mk[3] != 0) == ~0ull;
                else allkeep = __builtin_amdgcn_ballot_w64(mk[0] != 0 && mk[1] != 0) == ~0ull;
            }
            run_group(gsc);
        });
        // ---- intermediate: + b1, lrelu, keep flag, hi / lo IN PLACE of the window conv1 read (behind the barrier: everybody is done reading it) ----
        group_barrier(std::integral_constant<int, S2>{});
        emit_any(std::false_type{}, std::true_type{}, 2 * q);
        zero_acc();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---- conv2 over the intermediate ----------------------------------------------------------------------------------------------------
        run_group(std::integral_constant<int, S2>{});
        rb_for<S2 + 1, S0 + NSEQ>([&](auto gsc) __attribute__((always_inline)) {
            group_barrier(gsc);
            run_group(gsc);
        });
        // ---- the step's result: + b2 + y, keep flag -> the residual registers; its lrelu parts IN PLACE as the next step's window (behind a barrier) ----
        if constexpr (more) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            emit_any(std::true_type{}, std::true_type{}, 2 * q + 1);
        } else {
            emit_any(std::true_type{}, std::false_type{}, 2 * q + 1);
        }
        stamp(2 + q);
    });
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();   // the transpose tiles overlay the weight buffers and the window

    // ---- epilogue: beta, accumulate; full lines through a per-wave LDS transpose (no LDS-DMA is pending: plain LDS accesses) ----------------------
    constexpr int TP = K::TPITCH;
    constexpr int LPR = C == 16 ? 4 : 8;               // lanes per output row of the wave's transposed tile (32 channels = 128 bytes; C = 16: 64 bytes)
    constexpr int RPI = 64 / LPR, NIT = 64 / RPI;      // rows per iteration, iterations
    const int c4 = wm * 32 + (lane % LPR) * 4, rowi = lane / LPR;
    float* ttile = reinterpret_cast<float*>(smem) + wave * (64 * TP);
    if constexpr (TWOTAP) {
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<rb_f32x4*>(ttile + (j * 16 + l16) * TP + 4 * lg) = yres4[j];
    } else {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                rb_f32x4 v = {yres[j][4 * q], yres[j][4 * q + 1], yres[j][4 * q + 2], yres[j][4 * q + 3]};
                *reinterpret_cast<rb_f32x4*>(ttile + (j * 32 + lcol) * TP + 8 * q + 4 * lh) = v;
            }
    }
    const unsigned char* mask_s = reinterpret_cast<const unsigned char*>(smem + K::MASK_OFF);
    rb_f32x4 rold[NIT];
    if (p.accumulate) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int64_t po = min(max((int64_t)w0 + wn * 64 + it * RPI + rowi, (int64_t)0), (int64_t)NB - 1);
            rold[it] = *reinterpret_cast<const rb_f32x4*>(p.Y + po * C + c4);
        }
    }
    const float beta = p.beta;
    auto store_rows = [&](auto ntc) __attribute__((always_inline)) {
        constexpr bool NT = decltype(ntc)::value;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int row = it * RPI + rowi;
            const int r = wn * 64 + row;                    // window row
            const int pos = w0 + r;                         // < 2^31 (checked by the caller)
            const rb_f32x4 a = *reinterpret_cast<const rb_f32x4*>(ttile + row * TP + (lane % LPR) * 4);
            if (r < halo || r >= R - halo || pos >= NB) continue;
            // (contraction is off in this file; respair_clx.hip / conv_cl.hip are compiled with hipcc's default, which fuses `x * beta + old` into one fma:
            // written out here, so that the branch's last step keeps their bits)
            rb_f32x4 v;
            if (p.accumulate) {
                // (C >= 128 replaces conv_cl / conv_clx launches, whose accumulate epilogue rounds the product before the sum)
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = C >= 128 ? a[e] * beta + rold[it][e] : __builtin_fmaf(a[e], beta, rold[it][e]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = a[e] * beta;
            }
            if (!mask_s[r]) v = rb_f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (NT) __builtin_nontemporal_store(v, reinterpret_cast<rb_f32x4*>(p.Y + (int64_t)pos * C + c4));
            else *reinterpret_cast<rb_f32x4*>(p.Y + (int64_t)pos * C + c4) = v;
        }
    };
    if (p.nt_store) store_rows(std::true_type{});
    else store_rows(std::false_type{});
    stamp(6);
    stamp(15);
    if constexpr (DIAG) {
        if (threadIdx.x == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) p.stamps[(size_t)blockIdx.x * 16 + i] = st_[i];
        }
    }
}

template <int C, int NTAPS, int WNP, int GT, int NBUFP, int DG>
static void launch_rb(const ResBranchParams& p, hipStream_t stream) {
    using K = RbCfg<C, NTAPS, WNP, GT, NBUFP>;
    static_assert(K::LDS <= 160 * 1024, "LDS budget");
    auto kern = resbranch_clx_kernel<C, NTAPS, WNP, GT, NBUFP, DG>;
    static std::atomic<uint64_t> lds_allowed{0};   // per (kernel instantiation, device)
    allow_full_lds(reinterpret_cast<const void*>(kern), lds_allowed);
    const int nto = K::R - 2 * p.halo;
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
        // (the fused steps' buckets; C = 128 replaces six conv_clx launches and is counted with them)
        conv_prof_add(C >= 128 ? 26 : (C == 64 ? 17 : 16), kResBranchSteps * 2.0 * 2.0 * p.C * (double)p.N * p.C * p.k, e0, e1);
    }
}

// 0: the unfused paths (three respair_clx launches; at C = 128 six conv_clx launches); 1 (default): the fused branch at C = 16 / 32 / 64 / 128 (k = 3) and
// at C = 16 (k = 7 / 11); 2: the k = 3 branches only
static std::atomic<int> g_rb{getenv("SBV2_RESBRANCH") ? atoi(getenv("SBV2_RESBRANCH")) : 1};   // sbv2_debug_set_resbranch
int set_resbranch(int on) { return g_rb.exchange(on); }
bool resbranch_enabled() { return g_rb.load(std::memory_order_relaxed) != 0; }
bool resbranch_wanted(int C, int k) {
    const int mode = g_rb.load(std::memory_order_relaxed);
    if (mode == 0 || !(C == 16 || C == 32 || C == 64 || C == 128)) return false;
    return k == 3 || mode == 1;
}

// rows of the window of the instance launch_rb_any picks (0: none)
static int rb_rows(int C, int k) {
    if (k == 3) return C == 128 ? 192 : (C == 64 ? 384 : ((C == 32 || C == 16) ? 256 : 0));
    if (C == 16 && (k == 7 || k == 11)) return 512;
    return 0;
}

bool resbranch_usable(const ResBranchParams& p) {
    const int R = rb_rows(p.C, p.k);
    if (R == 0 || p.N < 1 || !(p.slope >= 0.f && p.slope <= 1.f) || (p.mask && p.mask_shift < 0)) return false;
    int halo = 0;
    for (int q = 0; q < kResBranchSteps; ++q) {
        if (p.dil[q] < 1 || p.dil[q] * (p.k - 1) / 2 > rb_margin(p.k)) return false;
        if (!p.W[2 * q] || !p.W[2 * q + 1] || !p.b[2 * q] || !p.b[2 * q + 1]) return false;
        halo += (p.dil[q] + 1) * (p.k - 1) / 2;
    }
    return halo * 4 <= R;   // (leaves >= half of the window as output)
}

template <int DG>
static void launch_rb_any(const ResBranchParams& p, hipStream_t stream) {
    // Window sizes / weight rings, each the best of a same-box sweep (profiles/r06c, r06l, r06m, r06n *_probe*):
    //   C = 128: 192 rows on 12 waves, a chunk's three taps per weight group, 2 ring slots: 155 KB, ONE workgroup per CU (1.84 ms per half plane; 128 rows
    //            on 8 waves 2.14-2.28; the six conv_clx launches it replaces 2.2)
    //   C = 64:  384 rows on 12 waves, three taps per group, 3 slots: 137 KB, one per CU (1.01 ms; 512 rows on 16 waves 1.00-1.03; 256 rows on 8 waves 1.13;
    //            192 rows on 6 waves x TWO workgroups per CU 1.32; 128 rows on 4 waves x three per CU 1.25 = the three fused steps' 1.24)
    //   C = 32 / 16: 256 rows on 4 waves, three per CU (0.61 / 0.49 ms; 512 rows 0.74 / 0.49, 768 / 1024 rows 0.70 / 0.55)
    // The wide stages want ONE big workgroup (less recompute: 1.07-1.14x, a third of the weight bytes per output, 12 waves behind every barrier); the narrow
    // ones, bound by their own instruction issue, want several small ones.  (The same structure for ONE k = 7 / 11 step of the 64-channel stage - a 384-row
    // window, both convolutions over all of it - measured against respair_clx.hip: same bits, the stage's bucket 10.9 -> 12.0 ms per step, the step
    // +0.65 ms: profiles/r06p_one_step_big_workgroup_ab.txt.  A single MFMA-bound step has no plane pass to save and pays 1.03-1.19x recompute.  Not kept.)
    // The k = 7 / 11 branches of the 16-channel stage (halo 36 / 60 rows per side: 512-row windows on 8 waves, 59 / 74 KB, two workgroups per CU; C = 16 is
    // the one width whose fused STEP is HBM bound at these kernel sizes: 0.92 -> 0.76 and 1.11 -> 1.07 ms per half plane, same bits).  In the pipelined step
    // the window size matters beyond the isolated time: 1024-row windows on 16 waves (94 KB, one 1024-thread workgroup per CU) measure the same in isolation
    // (0.77 / 1.01 ms) and cost the step +0.7 ms, because the other context's kernels no longer fit beside them; 512-row windows: -0.25 ms.  C = 32 (k = 7 / 11
    // on 512 / 768 / 1024 rows: 1.25 -> 1.15 and 1.66 -> 1.62 ms isolated at 1024 rows, equal at 512) gains nothing in the step: not instantiated.
    // (profiles/r06w_resbranch_k_probe*.jsonl, r06x_bench_resbranch_k_ab*.txt)
    if (p.k == 3) {
        if (p.C == 128) return launch_rb<128, 3, 3, 3, 2, DG>(p, stream);
        if (p.C == 64) return launch_rb<64, 3, 6, 3, 3, DG>(p, stream);
        if (p.C == 32) return launch_rb<32, 3, 4, 4, 3, DG>(p, stream);
        if (p.C == 16) return launch_rb<16, 3, 4, 4, 3, DG>(p, stream);
    } else if (p.C == 16 && p.k == 7) {
        return launch_rb<16, 7, 8, 4, 3, DG>(p, stream);
    } else if (p.C == 16 && p.k == 11) {
        return launch_rb<16, 11, 8, 6, 3, DG>(p, stream);
    }
    SBV2_REQUIRE(false, "resbranch: shape not instantiated");
}

// p.mask_shift must be set by the caller (mask_div = 1 << mask_shift)
void launch_resbranch(const ResBranchParams& p0, hipStream_t stream) {
    SBV2_REQUIRE(resbranch_usable(p0), "resbranch: operands do not fit the kernel");
    ResBranchParams p = p0;
    p.halo = 0;
    for (int q = 0; q < kResBranchSteps; ++q) p.halo += (p.dil[q] + 1) * (p.k - 1) / 2;
    p.nt_store = (int64_t)p.N * p.C * 4 >= ((int64_t)128 << 20);
    if (p.stamps) launch_rb_any<0>(p, stream);
    else launch_rb_any<-1>(p, stream);
}

}  // namespace sbv2
