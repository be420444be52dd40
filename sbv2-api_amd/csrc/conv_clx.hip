// Channels-last split-bf16 convolution of the HiFi-GAN ResBlocks on PRE-SPLIT, PRE-ACTIVATED operands: the restructured conv_cl.
//
//   Y[pos][m] (+)= ( sum_tap sum_k W_tap[m][k] * Xs[pos + shift_tap][k] + b[m] + R[pos][m] ) * beta,   Xs = bf16 hi + bf16 lo of lrelu(x)
//   and / or   Ys = bf16 hi + lo of lrelu(that result)   (the operand of the NEXT convolution)
//
// Round 3 (profiles/r03_conv_cl_clock_probe.jsonl): conv_cl's chunk loop = MFMA issue + staging through registers, ADDED.  Here no wave stages anything:
//   * the activation operand exists in HBM as bf16 parts, written by the producing epilogue (this kernel's, or split_cl for a stage input), CHUNK-MAJOR:
//     for every 16-channel chunk and part a plane [front + N + back][16] bf16 (32-byte rows, zero halo rows), so a chunk's window [256 + span rows] is one
//     contiguous run per part -> LDS-DMA;
//   * the weight fragments of one (chunk, tap) = 4 KB for 64 rows are LDS-DMA'd into a ring of kClxWR slots, three steps ahead;
//   * a step = one tap of one chunk: lgkmcnt(0) [its fragments, requested a step ago] -> counted vmcnt + barrier -> MFMAs with the fragment reads of the next
//     step and the DMAs dealt BETWEEN them.  Waves 0, 1 issue the weight DMAs, waves 2, 3 the window DMAs (vmcnt is per wave and in order: a wave that
//     issued both kinds would have to wait for young window pieces to reach an old weight block).
//
// Round 5: v_mfma_f32_16x16x32_bf16 instead of 32x32x16.  These launches are POWER bound: the timeline of a workgroup's life (sbv2_debug_clx_timeline,
// profiles/r05*_clx_timeline*.jsonl) shows the chip trading clock for every cycle a denser schedule saves (epilogue 30 -> 20 us: launch -2 %, loop clock
// 1.70 -> 1.61 GHz, aggregate MFMA rate unchanged), and a probe build of this loop that issued the same FLOP from the same fragment registers as 16x16x32
// instructions ran 11-12.5 % faster at k = 7 / 11 (profiles/r05d_clx_shape_probe.jsonl; MI355X_MICROARCH.md, DVFS give-back item 7: 1.12-1.14x).
// Round 6: the 32-deep K dimension of that shape carries a PAIR of consecutive steps a, b (any two: the pairing runs over the global step sequence, across
// chunk boundaries), as respair_x16.hip:
//     A_part = [W_part(a) | W_part(b)] (k groups 0, 1 | 2, 3; packed at load: pack_clx16),   B_part = [X_part(a) ; X_part(b)] (k groups 0, 1 read step a's row
//     of its window, 2, 3 step b's),   and per accumulator   A_lo B_hi,  A_hi B_lo,  A_hi B_hi
// 16 fragment reads and no operand shuffling per 48 MFMAs.  (Round 5 put both cross terms of ONE step into an instruction, A = [W_hi | W_lo], B = [X_lo ; X_hi],
// and made a pair's hi x hi operands with 16 v_permlane32_swap + 4 more reads: same FLOP, 20 reads and 16 cross-lane VALU operations per pair; on one box the
// decoder's bucket 28.4 -> 26.8 ms, the flow's FFN pair 6.17 -> 5.65 ms, the step 68.99 -> 67.06 ms: profiles/r06za_clx_steppair_ab.txt.)  The LDS images are
// the same size (52 KB: three workgroups per CU), 3 K-products are issued per algorithmic product as before.  The summation order differs from conv_cl's (32
// products per instruction): results agree with conv_cl to f32 rounding (tests hold 1e-5 kernel against kernel), not bit for bit.
#include <atomic>
#include <type_traits>

#include "common.h"

// No floating-point contraction in this file: the epilogue's (sum) * beta + previous contents must round the same way whatever shape the surrounding
// control flow has (an fma there moved the waveform by 9e-7 between two epilogue variants that promise the same bits)
#pragma clang fp contract(off)

namespace sbv2 {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void clx_lds_t;
typedef const __attribute__((address_space(1))) void clx_gbl_t;

constexpr int kClxNT = 256;          // positions per workgroup
constexpr int kClxXR = 320;          // window rows per buffer (256 + span <= 64)

struct ClxKernelParams {
    ConvClxParams p;
    int xrows;     // window rows actually read (256 + tap span)
    int wshift0;   // first window row = n0 + wshift0
    int sh0, sh_step;
    int gy;        // row tiles (of 64 rows)
};

template <int I, int N, class F>
__device__ __forceinline__ void clx_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        clx_static_for<I + 1, N>(f);
    }
}
template <int N>
__device__ __forceinline__ void clx_wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void clx_wait_vm_dyn(int n) {   // wave-uniform n
    switch (n) {
        case 0: clx_wait_vm<0>(); break;
        case 1: clx_wait_vm<1>(); break;
        case 2: clx_wait_vm<2>(); break;
        case 3: clx_wait_vm<3>(); break;
        case 4: clx_wait_vm<4>(); break;
        case 5: clx_wait_vm<5>(); break;
        case 6: clx_wait_vm<6>(); break;
        case 7: clx_wait_vm<7>(); break;
        case 8: clx_wait_vm<8>(); break;
        case 9: clx_wait_vm<9>(); break;
        default: clx_wait_vm<10>(); break;
    }
}
template <int OFF>
__device__ __forceinline__ bf16x8 clx_read_b128o(unsigned addr) {
    bf16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF));
    return v;
}
__device__ __forceinline__ unsigned clx_opaque(unsigned x) {
    asm volatile("" : "+v"(x));
    return x;
}
__device__ __forceinline__ void clx_mfma16(f32x4v& c, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
// 4 waves, 64 rows x 256 positions per workgroup (wave: 64 rows x 64 positions = 4 x 4 accumulator tiles of 16 x 16); kClxWR weight ring slots (4 KB: one
// step), kClxXB window buffers of XR rows (both parts).  <= 53 KB of LDS and <= 168 registers: THREE workgroups per CU.
// EDGE: the launch has a partial last position tile (N % 256 != 0): its guarded epilogue is compiled in.  (Compiled into every instance, that rarely taken
// path set the register allocation of the whole kernel and the interior epilogue spilled in the middle of its load burst; the decoder's frame layout is
// rounded so that the wide stages' planes are whole tiles.)
// ... and the k-major epilogue (the flow's second FFN convolution) is an instance of its own (EPI 2; 1 = EDGE, 0 = whole tiles, 3 = whole tiles with non-temporal stores).
// PH: phased output rows (ConvClxParams::phase_rows: the polyphase transposed convolutions, round 6); whole-tile epilogues only.
template <int NTAPS, int kClxWR, int kClxXB, int XR, int EPI, bool PH = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(kClxWR * 4096 + kClxXB * 2 * XR * 32 <= 53 * 1024 ? 3 : 2))) void conv_clx_kernel(const ClxKernelParams kp) {
    constexpr bool EDGE = EPI == 1;
    static_assert(!PH || EPI == 0 || EPI == 3, "phased output: whole-tile channels-last epilogues only");
    constexpr bool NT = EPI == 3;   // EPI 0 with non-temporal stores (result planes larger than the caches)
    constexpr int NPW = 64;                    // positions per wave
    constexpr int NTW = 256;                   // positions per workgroup
    constexpr int NW = 4;                      // waves
    constexpr int NWW = 2;                     // weight-DMA waves (the others carry the window)
    constexpr int NXW = NW - NWW;
    constexpr int kClxPW = (2 * (XR / 32) + NXW - 1) / NXW;   // window DMA pieces per window wave and chunk (2 parts x XR / 32 pieces of 32 rows)
    constexpr int WSLOT = 4096;                // half a pair of steps: 2 row tiles of 16 x (hi block, lo block) of 1 KB; a pair = two consecutive slots
    constexpr int WBYTES = kClxWR * WSLOT;
    constexpr int XPART = XR * 32, XBUF = 2 * XPART;   // (XR = window rows per buffer: 320 holds every tap span <= 64; 288 those <= 32); hi plane, then lo plane
    const ConvClxParams& p = kp.p;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)smem);

    const unsigned long long st_entry = p.stamps ? __builtin_amdgcn_s_memrealtime() : 0ull;   // (diagnostics only)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wq = wave;
    // XCD-aware tile order as in conv_cl (1-D grid; the gy row tiles of one position tile share their window: ids 8 apart = same XCD; position tiles go
    // round-robin over the XCDs: contiguous ranges per XCD measured the same, profiles/r05a_*)
    const int bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;
    const int by = slot % kp.gy;
    const int bx = (slot / kp.gy) * 8 + xcd;
    if (bx * NTW >= p.N) return;
    // EPI 0 with a ragged N (launched only when nothing accumulates): the last tile is moved left to end at N; the positions it shares with its neighbour are
    // computed twice from the same operands in the same order, and stored twice with the same bits
    const int n0 = (EPI == 0 || EPI == 3) ? min(bx * NTW, p.N - NTW) : bx * NTW;
    const int m0 = by * 64;                    // first output row of the workgroup (every wave: all 64 rows, its own 64 positions)
    const int M = p.M, N = p.N;
    const int nchunks = p.K >> 4;              // (even: K is a multiple of 32)
    const int S = nchunks * NTAPS;             // steps (even)

    // ---- DMA sources.  All per-step state is incremental (running pointers and LDS offsets, wave-uniform where possible): the address
    // arithmetic of a step sits in front of its first MFMA, right behind the barrier, where nothing overlaps it.
    const bool wwave = wave < NWW;
    // Every DMA source is a wave-uniform pointer (scalar registers) + this lane's 16-byte piece of the 1 KB block: no per-lane 64-bit pointers.
    const unsigned lane16 = lane * 16;
    // weight wave w: 2 KB of each half pair (4 KB = two row tiles of 16 x [hi block, lo block]); block order [row tile of 64][pair][row tile of 16][part]
    // (pack_clx16): a row tile's pairs are one contiguous run
    const char* wptr = static_cast<const char*>(p.W) + ((int64_t)by * S) * WSLOT + (wave & (NWW - 1)) * 2048;   // next weight blocks to fetch (uniform)
    const unsigned wdst0 = lds0 + (wave & (NWW - 1)) * 2048;
    unsigned wdoff = 0;                                             // ring offset the next half pair goes to
    auto dma_w = [&]() {
        __builtin_amdgcn_global_load_lds((clx_gbl_t*)(wptr + lane16), (clx_lds_t*)(uintptr_t)(wdst0 + wdoff), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((clx_gbl_t*)(wptr + lane16), (clx_lds_t*)(uintptr_t)(wdst0 + wdoff), 16, 1024, 0);   // (the immediate offset applies to both addresses)
        wptr += WSLOT;
        wdoff = wdoff + WSLOT == WBYTES ? 0 : wdoff + WSLOT;
    };
    // window wave v = wave - NWW: pieces e = v, v + NXW, ... of the 2 * npc pieces of a chunk (piece = 32 rows of one part = 1 KB contiguous; this lane: row
    // 32 * piece + lane / 2, 16-byte half lane & 1): the LDS image is the plane's own layout (32-byte rows), which the 16x16x32 B-fragment reads take
    // without bank conflicts
    const int npc = (kp.xrows + 31) >> 5;      // pieces per part
    const int64_t xplane = (int64_t)(p.X.front + p.X.N + p.X.back) * 32;   // bytes of one (chunk, part) plane
    const int xv = wave - NWW;
    const char* xptr[kClxPW];                  // source of this wave's piece i in the next window to fetch (uniform)
    unsigned xdst[kClxPW];
    int nmine = 0;                             // pieces of this window wave
#pragma unroll
    for (int i = 0; i < kClxPW; ++i) {
        const int e = max(xv, 0) + i * NXW;
        const int ec = min(e, 2 * npc - 1);
        const int part = ec / npc, pc = ec - part * npc;
        xptr[i] = static_cast<const char*>(p.X.p) + (int64_t)part * xplane + ((int64_t)p.X.front + n0 + kp.wshift0 + pc * 32) * 32;
        xdst[i] = lds0 + WBYTES + part * XPART + pc * 1024;
        if (!wwave && e < 2 * npc) ++nmine;
    }
    unsigned xdoff = 0;                        // buffer offset the next window goes to
    auto dma_x = [&](auto ic) {                // piece i of the next window; the last piece of a window advances to the following chunk
        constexpr int i = decltype(ic)::value;
        if (i < nmine) {
            __builtin_amdgcn_global_load_lds((clx_gbl_t*)(xptr[i] + lane16), (clx_lds_t*)(uintptr_t)(xdst[i] + xdoff), 16, 0, 0);
            xptr[i] += 2 * xplane;
        }
    };
    auto next_window = [&]() { xdoff = xdoff + XBUF == kClxXB * XBUF ? 0 : xdoff + XBUF; };

    // ---- fragments of one PAIR of steps a, b: A_part[it] = rows 16 it .. + 15: lane (row l16, k group lg) holds W_part of step a (lg < 2) or b (lg >= 2),
    // channels 8 (lg & 1) .. + 7 of the step's chunk (pack_clx16: the pair's eight 1 KB blocks [row tile of 16][part]); B_part[jt] = positions 16 jt .. + 15 of
    // the wave's 64: lane (column l16, k group lg) reads 16 bytes of the row of step a (lg < 2) or b (lg >= 2) in the part's plane
    struct Frags {
        bf16x8 ah[4], al[4], bh[4], bl[4];
    };
    struct Pend {
        bf16x8 ah[4], bh[4];
    };
    const int l16 = lane & 15, lg = lane >> 4;
    const unsigned abase = lds0 + lane * 16;
    // this lane's B-fragment address in the hi plane of window buffer 0 at tap 0: 16-byte half lg & 1 of the row; a step's tap and buffer are added per pair
    // (k groups 0, 1: step a's, 2, 3: step b's), the lo plane is an immediate (XPART) further
    int sh0 = kp.sh0;
    if constexpr (PH) sh0 += p.phase_tap0[p.phase_group == 2 ? 2 * ((m0 >> 5) / (p.phase_rows >> 4)) : m0 / p.phase_rows] * kp.sh_step;   // (uniform) this row tile's phase
    const unsigned bhlane = lds0 + WBYTES + (wq * NPW + l16 + sh0) * 32 + ((lg & 1) << 4);
    const int shs32 = kp.sh_step * 32;
    f32x4v acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
    auto mfma_one = [&](const bf16x8 (&a)[4], const bf16x8 (&b)[4], auto nc) {   // MFMA n of a 16-instruction set: row tile n / 4, position tile n % 4
        constexpr int n = decltype(nc)::value;
        // (inline asm: the accumulator stays in ITS registers.  Given the builtin, hipcc wrote each result to another register quad and took the old one for
        // a fragment, then restored the mapping with ~200 v_mov per loop iteration.  An accumulate chain needs no wait states; the A / B operands were
        // written by LDS reads waited for before the set.)
        clx_mfma16(acc[n >> 2][n & 3], a[n >> 2], b[n & 3]);
    };
    // The operands of a set of MFMAs stay allocated until the set has been issued: the fragment reads dealt between the MFMAs return asynchronously, and the
    // compiler (which sees neither: inline asm) may otherwise hand a read the registers of an operand whose last MFMA in program order is still queued in
    // front of the matrix pipe (respair_x16.hip has the failure this produced)
    auto keep4 = [&](const bf16x8 (&v)[4]) { asm volatile("" ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3])); };

    // ---- The loop runs in PAIRS of consecutive steps a = 2 u, b = 2 u + 1 (step s = tap s % NTAPS of chunk s / NTAPS; NTAPS is odd and the chunk count even, so
    // the pairing runs across chunk boundaries and every two chunks hold NTAPS pairs), ONE barrier per pair, three sets of 16 MFMAs per pair whose 32-deep K
    // dimension carries the two steps (round 6; rounds 5's scheme - both cross terms of ONE step per instruction, the pair's hi x hi operands by
    // v_permlane32_swap + four more reads - cost 16 swaps and 4 reads per pair more):
    //   TOP    lgkmcnt(0) -> vmcnt -> barrier
    //   S0     A_hi B_hi of the PREVIOUS pair (its operands are registers): covers the reads of A_lo, B_hi of this pair, dealt in its first 8 gaps; the weight
    //          waves fetch the blocks of pair u + 1 (into the pair slot of u - 1, read by every wave before this barrier)
    //          lgkmcnt(0)
    //   S1     A_lo B_hi; the reads of A_hi, B_lo in its first 8 gaps; the window waves fetch their share of the next window
    //          lgkmcnt(0)
    //   S2     A_hi B_lo
    // What must have landed at TOP(u): the weight blocks of pair u (all a weight wave has in flight: vmcnt(0)), and the window of a chunk whose first step is
    // in pair u.  With chunks c0 (even), c0 + 1 in pairs 0 .. NTAPS - 1 of an iteration and J0 = (NTAPS - 1) / 2: window c0 + 1 (first read: step b of pair J0)
    // is fetched during pairs 0 .. J0 - 1 (its buffer's previous window was last read in the previous iteration) and waited for at TOP(J0); window c0 + 2 (first
    // read: the next iteration's first pair) during pairs J0 + 1 .. NTAPS - 1 (window c0 is last read by pair J0) and waited for at the next iteration's TOP(0).
    // At those two points a window wave has nothing else in flight: vmcnt(0).
    // (An EVEN kernel size - the phased transposed convolutions, whose phases share 2 or 4 input taps - pairs the taps inside a chunk, NTAPS / 2 pairs each: no
    // pair spans two chunks, window c0 + 1 is fetched during the pairs of chunk c0 and waited for at TOP(NTAPS / 2), window c0 + 2 during the pairs of c0 + 1.)
    constexpr bool EVEN = (NTAPS & 1) == 0;
    constexpr int J0 = EVEN ? NTAPS / 2 : (NTAPS - 1) / 2;   // the pair whose step b (even: both steps) opens chunk c0 + 1
    constexpr int PPP = (kClxPW + J0 - 1) / J0;        // window pieces per window wave and pair
    static_assert(PPP <= 16, "a pair's S1 gaps hold its window pieces");
    static_assert(kClxWR == 4 && kClxXB == 2, "the pair schedule assumes two pair slots of weights and two window buffers");
    if (wwave) {
        dma_w();
        dma_w();               // pair 0
    } else {
        clx_static_for<0, kClxPW>([&](auto ic) { dma_x(ic); });
        next_window();
    }
    unsigned wroff = 0;        // ring offset of the pair slot the fragment reads take
    Pend pend;                 // hi x hi operands of the previous pair (zeros in front of the first)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        pend.ah[i] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        pend.bh[i] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    }

    unsigned long long st_t0 = 0, st_r0 = 0;
    if (p.stamps) {
        st_t0 = __builtin_amdgcn_s_memtime();
        st_r0 = __builtin_amdgcn_s_memrealtime();
    }
    for (int chunk = 0; chunk < nchunks; chunk += 2) {
        const int s0 = chunk * NTAPS;
        clx_static_for<0, NTAPS>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            constexpr int ra = 2 * j, rb = 2 * j + 1;         // steps a, b relative to s0
            constexpr int tapa = ra % NTAPS, bufa = (ra / NTAPS) & 1, tapb = rb % NTAPS, bufb = (rb / NTAPS) & 1;
            // ---- TOP
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (wwave) clx_wait_vm<0>();
            else if constexpr (j == J0 || j == 0) clx_wait_vm<0>();
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            Frags f;
            // (fragment addresses are formed where they are used, from opaque copies of the lane bases: hoisted out of the loop they took up to 33 registers)
            const unsigned aaddr = clx_opaque(abase) + wroff;
            const unsigned b0 = clx_opaque(bhlane) + (unsigned)(lg < 2 ? tapa * shs32 + bufa * XBUF : tapb * shs32 + bufb * XBUF);
            wroff ^= 2 * WSLOT;
            // ---- S0: the previous pair's hi x hi; A_lo, B_hi of this pair; the weight blocks of the next pair
            {
                const bool wn = wwave && s0 + 2 * j + 2 < S;
                clx_static_for<0, 16>([&](auto nc) {
                    constexpr int n = decltype(nc)::value;
                    mfma_one(pend.ah, pend.bh, nc);
                    if constexpr (n < 4) f.al[n] = clx_read_b128o<n * 2048 + 1024>(aaddr);
                    else if constexpr (n < 8) f.bh[n - 4] = clx_read_b128o<(n - 4) * 512>(b0);
                    if constexpr (n == 2 || n == 6) {
                        if (wn) dma_w();
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
                keep4(pend.ah);
                keep4(pend.bh);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f.al[0]), "+v"(f.al[1]), "+v"(f.al[2]), "+v"(f.al[3]), "+v"(f.bh[0]), "+v"(f.bh[1]), "+v"(f.bh[2]), "+v"(f.bh[3]));
            __builtin_amdgcn_sched_barrier(0);
            // ---- S1: lo x hi; A_hi, B_lo; window pieces: pairs 0 .. J0 - 1 carry window chunk + 1, pairs J0 + 1 .. NTAPS - 1 window chunk + 2
            {
                constexpr bool first = j < J0;
                constexpr int jj = first ? j : j - J0 - (EVEN ? 0 : 1);
                const bool stx = !wwave && (first ? chunk + 1 < nchunks : chunk + 2 < nchunks);
                clx_static_for<0, 16>([&](auto nc) {
                    constexpr int n = decltype(nc)::value;
                    mfma_one(f.al, f.bh, nc);
                    if constexpr (n < 4) f.ah[n] = clx_read_b128o<n * 2048>(aaddr);
                    else if constexpr (n < 8) f.bl[n - 4] = clx_read_b128o<XPART + (n - 4) * 512>(b0);
                    if constexpr ((EVEN || j != J0) && n < PPP) {
                        constexpr int i = jj * PPP + n;
                        if constexpr (i < kClxPW) {
                            if (stx) dma_x(std::integral_constant<int, i>{});
                            if constexpr (i == kClxPW - 1) {
                                if (stx) next_window();
                            }
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
                keep4(f.al);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f.ah[0]), "+v"(f.ah[1]), "+v"(f.ah[2]), "+v"(f.ah[3]), "+v"(f.bl[0]), "+v"(f.bl[1]), "+v"(f.bl[2]), "+v"(f.bl[3]));
            __builtin_amdgcn_sched_barrier(0);
            // ---- S2: hi x lo
            clx_static_for<0, 16>([&](auto nc) {
                mfma_one(f.ah, f.bl, nc);
                __builtin_amdgcn_sched_barrier(0);
            });
            keep4(f.bl);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                pend.ah[i] = f.ah[i];
                pend.bh[i] = f.bh[i];
            }
        });
    }
    // the last pair's hi x hi
    clx_static_for<0, 16>([&](auto nc) { mfma_one(pend.ah, pend.bh, nc); });
    keep4(pend.ah);
    keep4(pend.bh);
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // (the accumulators are read by LDS writes next; the compiler does not see these MFMAs)
    if (p.stamps && tid == 0) {
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        unsigned long long* o = p.stamps + (size_t)blockIdx.x * kClxStampWords;
        o[0] = st_t0; o[1] = st_r0; o[2] = t1; o[3] = r1;
        o[4] = st_entry;
        // where this workgroup ran: HW_ID (wave / SIMD / CU / SH / SE) and XCC_ID
        o[7] = (unsigned long long)__builtin_amdgcn_s_getreg(4 | (31 << 11)) | ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (31 << 11)) << 32);
    }
    __syncthreads();   // the epilogue re-uses the rings as its transpose tiles
    unsigned long long st_e0 = 0, st_e1 = 0, st_e2 = 0;   // (diagnostics) behind the barrier / mask + bias + residual rows arrived / first half's stores issued
    if (p.stamps) st_e0 = __builtin_amdgcn_s_memrealtime();

    // accumulator tile [it][jt]: lane (column l16 = position 16 jt + l16 of the wave's 64, row group lg) holds rows 16 it + 4 lg .. + 3
    // ---- k-major result (the flow's second FFN convolution: Y[m][n] = (conv + b + R[m][n]) * mask): one accumulator register of a 16-lane group is 16
    // consecutive positions of one channel = a 64-byte run of the plane; bias, mask and residual of a row tile are requested before its first store
    if constexpr (EPI == 2) {
        // Round 6: through a wave-private LDS tile [32 rows][64 positions (+4)], so that a lane leaves with FOUR consecutive positions of one channel: 16-byte
        // residual loads and stores, 4 rows x 256 bytes per instruction (the direct form moved one 4-byte element per lane: 64 loads + 64 stores per wave
        // in runs of 64 bytes).  Same arithmetic per element in the same order: same bits.  Every residual row of a half is requested before its first store.
        float* tileT = reinterpret_cast<float*>(smem) + wave * (32 * 68);
        const int prow = lane >> 4, pq = (lane & 15) * 4;
        const int nq = n0 + wq * NPW + pq;                 // first of this lane's four positions
        bool keep4[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) keep4[e] = nq + e < N && (!p.mask || p.mask[min(nq + e, N - 1) >> p.mask_shift] != 0);
        const bool whole4 = nq + 4 <= N;
        const int nqc = whole4 ? nq : max(min(nq, N - 4), 0);   // (a clamped, valid address for the residual rows of a ragged tail: their values are not used)
        clx_static_for<0, 2>([&](auto hc) {
            constexpr int i2 = decltype(hc)::value;        // rows 32 i2 .. + 31 of the wave's 64
#pragma unroll
            for (int it2 = 0; it2 < 2; ++it2)
#pragma unroll
                for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) tileT[(it2 * 16 + 4 * lg + r) * 68 + jt * 16 + l16] = acc[2 * i2 + it2][jt][r];
            f32x4v rr[8];
            float brow[8];
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int mc = min(m0 + i2 * 32 + it * 4 + prow, M - 1);
                brow[it] = p.bias ? p.bias[mc] : 0.f;
                if (p.Rkm) {
                    if (whole4) rr[it] = *reinterpret_cast<const f32x4v*>(p.Rkm + (int64_t)mc * p.ldrkm + nqc);
                    else
#pragma unroll
                        for (int e = 0; e < 4; ++e) rr[it][e] = p.Rkm[(int64_t)mc * p.ldrkm + min(nq + e, N - 1)];
                } else {
                    rr[it] = f32x4v{0.f, 0.f, 0.f, 0.f};
                }
            }
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int m = m0 + i2 * 32 + it * 4 + prow;
                const f32x4v a = *reinterpret_cast<const f32x4v*>(tileT + (it * 4 + prow) * 68 + pq);
                if (m >= M) continue;
                f32x4v v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float x = a[e] + brow[it];
                    if (p.Rkm) x += rr[it][e];
                    x *= p.beta;
                    v[e] = keep4[e] ? x : 0.f;
                }
                if (whole4) {
                    *reinterpret_cast<f32x4v*>(p.Ykm + (int64_t)m * p.ldykm + nq) = v;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (nq + e < N) p.Ykm[(int64_t)m * p.ldykm + nq + e] = v[e];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        return;
    }
    if constexpr (EPI != 2) {
    // ---- channels-last epilogue: each wave transposes its two 32-row halves through a private LDS tile [64 positions][36] so that 8 consecutive lanes hold
    // one full 128-byte line of a row
    float* tile = reinterpret_cast<float*>(smem) + wave * (NPW * 36);
    const float beta = p.beta;
    const int64_t yplane = (int64_t)(p.Ys.front + p.Ys.N + p.Ys.back) * 32;
    const int c4 = (lane & 7) * 4;
    const int nfirst = n0 + wq * NPW + (lane >> 3);
    const float* trow = tile + (lane >> 3) * 36 + c4;
    const float sl = p.ys_slope;
    auto tile_write = [&](auto ic) {   // rows 32 i .. 32 i + 31 of the wave's 64 -> tile[position][row]
        constexpr int i = decltype(ic)::value;
#pragma unroll
        for (int it2 = 0; it2 < 2; ++it2)
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) *reinterpret_cast<f32x4v*>(tile + (jt * 16 + l16) * 36 + it2 * 16 + 4 * lg) = acc[2 * i + it2][jt];
    };
    // ---- interior tiles (every position of the tile exists; all but the batch's last tile): no per-lane conditions, and EVERY global read of both row
    // halves (mask bytes: one load per lane + a ballot, bias, residual rows) is requested before the first store.  Round 4's epilogue took 14 (conv1) to
    // 22-30 us (conv2) of a workgroup's 34-85 us (profiles/r05a_clx_timeline_before.jsonl): sixteen mask-byte loads each followed by s_waitcnt vmcnt(0),
    // the second half's loads queued behind the first one's stores (a load's data returns behind every older store's acknowledgement), and an
    // s_waitcnt vmcnt(0) at the join behind every conditional store.  A lane owns EIGHT channels of a position (4 lanes per 128-byte line, 16 positions per
    // iteration): its bf16 parts are one 16-byte store per part instead of two 8-byte ones (the store tail is bound by the number of store instructions:
    // with the step loop 15 % shorter the epilogue of the same bytes got 25 % LONGER, profiles/r05j_clx_ablate.txt).
    if (!EDGE || n0 + NTW <= N) {
        const int c8 = (lane & 3) * 8;
        const int nf16 = n0 + wq * NPW + (lane >> 2);
        const float* trow8 = tile + (lane >> 2) * 36 + c8;
        // lane (group g = lane >> 2, j = lane & 3) loads the flag of position nf16 + 16 j; bit 4 g + it of the ballot is this lane's flag of iteration it
        unsigned mv = 1u;
        if (p.mask) mv = p.mask[(nf16 + (lane & 3) * 16) >> p.mask_shift];
        // two sets of 4 x 2 quads: the residual rows of both halves; or, for an accumulating launch (a branch's last step: 4 of a step's 36 launches),
        // residual + previous contents of ONE half (the second half's are requested behind the first one's stores)
        f32x4v b8[2][2], ld[2][4][2];
        const bool acc_y = p.accumulate != 0;
        auto load_rows = [&](int set, const float* base, int ldb, int m) {
            const float* rp = base + (int64_t)nf16 * ldb + m;
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                ld[set][it][0] = *reinterpret_cast<const f32x4v*>(rp + (int64_t)it * 16 * ldb);
                ld[set][it][1] = *reinterpret_cast<const f32x4v*>(rp + (int64_t)it * 16 * ldb + 4);
            }
        };
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int h = 0; h < 2; ++h)
                b8[i][h] = p.bias ? *reinterpret_cast<const f32x4v*>(p.bias + m0 + i * 32 + c8 + 4 * h) : f32x4v{0.f, 0.f, 0.f, 0.f};
        if (p.R) load_rows(0, p.R, p.ldr, m0 + c8);
        unsigned mbits = 0xFu;
        bool allkeep = true;
        // NT (EPI 3): the stores of planes that do not fit the caches (a batch's 0.94 GB per stage; its consumer starts after the whole plane is written)
        // bypass them (same-box A/B profiles/r05o_nt_store_ab.txt: C = 256 k = 3 conv1 238 -> 217 us, C = 128 k = 3 conv2 760 -> 737, step -0.3 ms; a single
        // utterance's 29 MB planes are re-read from L2 / MALL and lose 0.1 ms per call with it: the launch decides by size).  An instance of its own: both
        // store flavours behind a uniform branch in one kernel spilled 132 bytes.
        clx_static_for<0, 2>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            __builtin_amdgcn_sched_barrier(0);
            tile_write(ic);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (i == 0) {
                // the second set of rows goes into the registers the first half's accumulators just left (requested with the first set, the 64 + 64 + 16
                // registers of both halves' accumulators, rows and bias spilled in the middle of the load burst)
                if (acc_y) load_rows(1, p.Y, p.ldy, m0 + c8);
                else if (p.R) load_rows(1, p.R, p.ldr, m0 + 32 + c8);
                asm volatile("" : "+v"(mv));   // (the compare stays behind the LDS writes: hoisted to the load, it waits for the load there)
                const unsigned long long bal = __builtin_amdgcn_ballot_w64(mv != 0);
                allkeep = bal == ~0ull;
                mbits = (unsigned)(bal >> ((lane >> 2) * 4)) & 0xFu;
                if (p.stamps) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    st_e1 = __builtin_amdgcn_s_memrealtime();
                }
            }
            const int m = m0 + i * 32 + c8;
            if constexpr (i == 1) {
                if (acc_y) {
                    if (p.R) load_rows(0, p.R, p.ldr, m);
                    load_rows(1, p.Y, p.ldy, m);
                }
            }
            // PH: this workgroup's 64 rows are channels mo .. mo + 63 of ONE phase (or, phase_group 2, 32 channels of two adjacent phases); position n of a
            // phase is output row n * out_stride + phase_off[phase]
            int ostride = 1, mch = m;
            int64_t orow = nf16;
            if constexpr (PH) {
                ostride = p.out_stride;
                if (p.phase_group == 2) {   // (uniform) 32-row blocks of two adjacent phases x 16 channels: lanes 0, 1 of a quad hold one phase, lanes 2, 3 the next
                    const int blk = (m0 >> 5) + i, nb = p.phase_rows >> 4;
                    const int pg = blk / nb;
                    mch = (blk - pg * nb) * 16 + (c8 & 15);
                    orow = (int64_t)nf16 * ostride + p.phase_off[2 * pg] + (c8 >> 4);
                } else {
                    const int ph = m0 / p.phase_rows;
                    mch = m - ph * p.phase_rows;
                    orow = (int64_t)nf16 * ostride + p.phase_off[ph];
                }
            }
            float* yp = p.Y ? p.Y + orow * p.ldy + mch : nullptr;
            const int64_t ystep = (int64_t)16 * ostride * p.ldy;
            const int64_t qstep = (int64_t)512 * ostride;
            // bf16 parts of lrelu(result), chunk-major: 8 channels = 16 bytes of a 32-byte row; the lo plane follows the hi plane
            char* qs = p.Ys.p ? static_cast<char*>(p.Ys.p) + ((int64_t)(mch >> 4) * 2) * yplane + ((int64_t)p.Ys.front + orow) * 32 + (mch & 15) * 2 : nullptr;
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                f32x4v v[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const f32x4v a = *reinterpret_cast<const f32x4v*>(trow8 + it * 16 * 36 + 4 * h);
                    v[h] = a + b8[i][h];
                    if (p.R) v[h] += acc_y ? ld[0][it][h] : ld[i][it][h];
                    v[h] *= beta;   // (x 1.0f is exact)
                    if (acc_y) v[h] += ld[1][it][h];
                    if (!allkeep && !((mbits >> it) & 1u)) v[h] = f32x4v{0.f, 0.f, 0.f, 0.f};
                }
                if (yp) {
                    if constexpr (NT) {
                        __builtin_nontemporal_store(v[0], reinterpret_cast<f32x4v*>(yp + it * ystep));
                        __builtin_nontemporal_store(v[1], reinterpret_cast<f32x4v*>(yp + it * ystep + 4));
                    } else {
                        *reinterpret_cast<f32x4v*>(yp + it * ystep) = v[0];
                        *reinterpret_cast<f32x4v*>(yp + it * ystep + 4) = v[1];
                    }
                }
                if (qs) {
                    bf16x8 h8, l8;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float ve = v[e >> 2][e & 3];
                        const float x = fmaxf(ve, ve * sl);   // leaky ReLU for 0 <= slope <= 1 (the same value as the select, one instruction less)
                        h8[e] = (__bf16)x;
                        l8[e] = (__bf16)(x - (float)h8[e]);
                    }
                    if constexpr (NT) {
                        __builtin_nontemporal_store(h8, reinterpret_cast<bf16x8*>(qs + it * qstep));
                        __builtin_nontemporal_store(l8, reinterpret_cast<bf16x8*>(qs + it * qstep + yplane));
                    } else {
                        *reinterpret_cast<bf16x8*>(qs + it * qstep) = h8;
                        *reinterpret_cast<bf16x8*>(qs + it * qstep + yplane) = l8;
                    }
                }
            }
            if constexpr (i == 0) {
                if (p.stamps) st_e2 = __builtin_amdgcn_s_memrealtime();
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        if (p.stamps && tid == 0) {   // (diagnostics) last store issued / every store of this wave acknowledged
            unsigned long long* o = p.stamps + (size_t)blockIdx.x * kClxStampWords;
            o[8] = st_e0; o[9] = st_e1; o[10] = st_e2;
            o[5] = __builtin_amdgcn_s_memrealtime();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            o[6] = __builtin_amdgcn_s_memrealtime();
        }
        return;
    }
    // ---- the batch's last position tile: the same arithmetic with clamped reads and guarded stores, one row at a time (a handful of workgroups per launch)
    if constexpr (EDGE) clx_static_for<0, 2>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        tile_write(ic);
        const int m = m0 + i * 32 + c4;
        const f32x4v b4 = p.bias ? *reinterpret_cast<const f32x4v*>(p.bias + m) : f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int it = 0; it < 8; ++it) {
            const int n = nfirst + it * 8;
            const int64_t pp = min(n, N - 1);
            const f32x4v a = *reinterpret_cast<const f32x4v*>(trow + it * 8 * 36);
            f32x4v v = a + b4;
            if (p.R) v += *reinterpret_cast<const f32x4v*>(p.R + pp * p.ldr + m);
            v *= beta;
            if (p.accumulate) v += *reinterpret_cast<const f32x4v*>(p.Y + pp * p.ldy + m);
            if (p.mask && !p.mask[pp >> p.mask_shift]) v = f32x4v{0.f, 0.f, 0.f, 0.f};
            if (n < N) {
                if (p.Y) *reinterpret_cast<f32x4v*>(p.Y + (int64_t)n * p.ldy + m) = v;
                if (p.Ys.p) {   // bf16 parts of lrelu(result), chunk-major (4 channels = 8 bytes of a 32-byte row; the lo plane follows the hi plane)
                    char* qs = static_cast<char*>(p.Ys.p) + ((int64_t)(m >> 4) * 2) * yplane + ((int64_t)p.Ys.front + n) * 32 + (m & 15) * 2;
                    bf16x4 h, l;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float x = fmaxf(v[e], v[e] * sl);
                        h[e] = (__bf16)x;
                        l[e] = (__bf16)(x - (float)h[e]);
                    }
                    *reinterpret_cast<bf16x4*>(qs) = h;
                    *reinterpret_cast<bf16x4*>(qs + yplane) = l;
                }
            }
        }
    });
    }   // EPI != 2
}

// ---- f32 channels-last plane -> chunk-major bf16 parts of lrelu(x) (a stage input; every other operand is written by an epilogue) ----------------
__global__ __launch_bounds__(256) void k_split_cl(const float* __restrict__ X, int ldx, int64_t N, int C, float slope, SplitClPlanes out) {
    const int64_t plane = (int64_t)(out.front + out.N + out.back) * 32;
    const int c4n = C >> 2;
    const int64_t total = N * c4n;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < total; q += (int64_t)gridDim.x * 256) {
        const int64_t n = q / c4n;
        const int c = (int)(q - n * c4n) * 4;
        const f32x4v v = *reinterpret_cast<const f32x4v*>(X + n * ldx + c);
        bf16x4 h, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float x = v[e] >= 0.f ? v[e] : v[e] * slope;
            h[e] = (__bf16)x;
            l[e] = (__bf16)(x - (float)h[e]);
        }
        char* dst = static_cast<char*>(out.p) + ((int64_t)(c >> 4) * 2) * plane + ((int64_t)out.front + n) * 32 + (c & 15) * 2;
        *reinterpret_cast<bf16x4*>(dst) = h;
        *reinterpret_cast<bf16x4*>(dst + plane) = l;
    }
}
// f32 k-major plane [C][ld] -> chunk-major bf16 parts of lrelu(x): a thread owns one position and 16 channels (the 16 loads of a wave are 256-byte
// runs of 16 rows; its two 32-byte rows per part are contiguous with its neighbours')
__global__ __launch_bounds__(256) void k_split_cl_km(const float* __restrict__ X, int ldx, int64_t N, int C, float slope, SplitClPlanes out) {
    const int64_t plane = (int64_t)(out.front + out.N + out.back) * 32;
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int chunk = blockIdx.y;
    if (n >= N) return;
    float v[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) v[c] = X[(int64_t)(chunk * 16 + c) * ldx + n];
    bf16x8 h[2], l[2];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const float x = v[c] >= 0.f ? v[c] : v[c] * slope;
        const __bf16 hh = (__bf16)x;
        h[c >> 3][c & 7] = hh;
        l[c >> 3][c & 7] = (__bf16)(x - (float)hh);
    }
    char* dst = static_cast<char*>(out.p) + ((int64_t)chunk * 2) * plane + ((int64_t)out.front + n) * 32;
    *reinterpret_cast<bf16x8*>(dst) = h[0];
    *reinterpret_cast<bf16x8*>(dst + 16) = h[1];
    *reinterpret_cast<bf16x8*>(dst + plane) = l[0];
    *reinterpret_cast<bf16x8*>(dst + plane + 16) = l[1];
}
// zero halo rows in front of and behind every (chunk, part) plane: the zero padding of the convolutions at the ends of the batch
__global__ __launch_bounds__(256) void k_clx_zero_halo(SplitClPlanes s) {
    const int64_t rows = (int64_t)s.front + s.N + s.back;
    const int nplanes = (s.C >> 4) * 2;
    const int per = (s.front + s.back) * 2;   // 16-byte pieces per plane
    const int64_t total = (int64_t)nplanes * per;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < total; q += (int64_t)gridDim.x * 256) {
        const int pl = (int)(q / per);
        int e = (int)(q - (int64_t)pl * per);
        int64_t row = e >> 1;
        if (row >= s.front) row += s.N;
        *reinterpret_cast<f32x4v*>(static_cast<char*>(s.p) + ((int64_t)pl * rows + row) * 32 + (e & 1) * 16) = f32x4v{0.f, 0.f, 0.f, 0.f};
    }
}

size_t split_cl_bytes(int C, int64_t N) { return (size_t)(C >> 4) * 2 * (size_t)(kClxFront + N + kClxBack) * 32; }

SplitClPlanes make_split_cl(void* mem, int C, int64_t N, hipStream_t stream) {
    SplitClPlanes s;
    s.p = mem;
    s.C = C;
    s.N = N;
    s.front = kClxFront;
    s.back = kClxBack;
    const int64_t total = (int64_t)(C >> 4) * 2 * (s.front + s.back) * 2;
    hipLaunchKernelGGL(k_clx_zero_halo, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 1024)), dim3(256), 0, stream, s);
    HIP_CHECK(hipGetLastError());
    return s;
}

void split_cl(const float* X, int ldx, int64_t N, int C, float slope, const SplitClPlanes& out, hipStream_t stream) {
    SBV2_REQUIRE((C & 15) == 0 && out.C == C && out.N == N && (ldx & 3) == 0, "split_cl: shape mismatch");
    const int64_t total = N * (C >> 2);
    hipLaunchKernelGGL(k_split_cl, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 4096)), dim3(256), 0, stream, X, ldx, N, C, slope, out);
    HIP_CHECK(hipGetLastError());
}

void split_cl_km(Plane x, float slope, const SplitClPlanes& out, hipStream_t stream) {
    SBV2_REQUIRE((x.C & 15) == 0 && out.C == x.C && out.N == x.L, "split_cl_km: shape mismatch");
    hipLaunchKernelGGL(k_split_cl_km, dim3((unsigned)((x.L + 255) / 256), x.C >> 4), dim3(256), 0, stream, x.p, x.ld, (int64_t)x.L, x.C, slope, out);
    HIP_CHECK(hipGetLastError());
}

bool conv_clx_usable(const ConvClxParams& p) {
    if (!(p.ntaps == 3 || p.ntaps == 5 || p.ntaps == 7 || p.ntaps == 11 || (p.phase_rows && (p.ntaps == 2 || p.ntaps == 4)))) return false;
    if (p.Ykm && (p.Y || p.Ys.p || p.accumulate || p.R)) return false;   // the k-major epilogue writes Ykm only
    if ((p.K & 31) || (p.M & 63) || p.K != p.X.C) return false;            // K: pairs of 16-channel chunks (the step pairs of the 16x16x32 products)
    int tapmax = p.ntaps - 1;
    if (p.phase_rows)
        for (int q = 0; q < kMaxPhases && q < p.M / std::max(p.phase_rows, 1); ++q) tapmax = std::max(tapmax, p.ntaps - 1 + p.phase_tap0[q]);
    const int span = tapmax * std::abs(p.shift_step);
    if (span > kClxXR - kClxNT || p.shift0 < -kClxFront || p.shift0 + span > 64) return false;
    if (p.mask && p.mask_shift < 0) return false;
    if (p.Y && (p.ldy & 3)) return false;
    if (p.R && (p.ldr & 3)) return false;
    if (p.phase_rows) {   // phased output (a polyphase transposed convolution)
        if ((p.phase_rows & 63) || p.M % p.phase_rows || p.M / p.phase_rows > kMaxPhases || p.out_stride < 1 || p.R || p.accumulate || p.Ykm || p.N < kClxNT ||
            !(p.ntaps >= 2 && p.ntaps <= 5)) return false;
        if (p.Y && p.ldy < p.phase_rows) return false;
        if (p.Ys.p && (p.Ys.C != p.phase_rows || p.Ys.N != (int64_t)p.N * p.out_stride)) return false;
        for (int q = 0; q < p.M / p.phase_rows; ++q)
            if (p.phase_off[q] < 0 || p.phase_off[q] >= p.out_stride) return false;
        if (p.phase_group == 2) {
            if ((p.M / p.phase_rows) & 1) return false;
            for (int q = 0; q < p.M / p.phase_rows; q += 2)
                if (p.phase_off[q + 1] != p.phase_off[q] + 1 || p.phase_tap0[q + 1] != p.phase_tap0[q]) return false;
        } else if (p.phase_group != 1) return false;
        for (int q = 0; q < p.M / p.phase_rows; ++q)
            if (p.phase_tap0[q] < 0 || p.phase_tap0[q] > 8) return false;
    } else if (p.Ys.p && (p.Ys.C != p.M || p.Ys.N != p.N)) return false;
    return p.N >= 1 && p.X.N == p.N;
}

static thread_local int64_t* g_clx_grid_only = nullptr;   // clx_grid_workgroups: report the grid instead of launching

template <int NTAPS, int WR, int XB, int XR, int EPI, bool PH = false>
static void launch_clx_e(ClxKernelParams kp, hipStream_t stream);

template <int NTAPS, int WR, int XB, int XR>
static void launch_clx(const ClxKernelParams& kp, hipStream_t stream) {
    if constexpr (NTAPS <= 5 && XR == 288) {
        if (kp.p.phase_rows) {   // (conv_clx_usable: whole tiles, nothing accumulates)
            if ((int64_t)kp.p.N * kp.p.M * 4 >= ((int64_t)128 << 20)) return launch_clx_e<NTAPS, WR, XB, XR, 3, true>(kp, stream);
            return launch_clx_e<NTAPS, WR, XB, XR, 0, true>(kp, stream);
        }
    }
    SBV2_REQUIRE(!kp.p.phase_rows, "conv_clx: phased output is instantiated for 2 .. 5 taps on 288-row windows");
    if constexpr ((NTAPS & 1) == 0) {
        SBV2_REQUIRE(false, "conv_clx: even kernel sizes are instantiated for phased output only");
    } else {
    if (kp.p.Ykm) launch_clx_e<NTAPS, WR, XB, XR, 2>(kp, stream);
    else if (kp.p.N % kClxNT == 0 || (!kp.p.accumulate && kp.p.N >= kClxNT)) {
        // result planes of >= 128 MB (a batch's decoder stages; a long utterance) leave through non-temporal stores (EPI 3); smaller ones (a single utterance: 29 MB;
        // the flow's 88 MB FFN intermediate, which its second conv reads straight back: 6.05 -> 6.2 ms with them) are re-read from the caches
        if ((int64_t)kp.p.N * kp.p.M * 4 >= ((int64_t)128 << 20)) launch_clx_e<NTAPS, WR, XB, XR, 3>(kp, stream);
        else launch_clx_e<NTAPS, WR, XB, XR, 0>(kp, stream);
    }
    else launch_clx_e<NTAPS, WR, XB, XR, 1>(kp, stream);
    }
}

template <int NTAPS, int WR, int XB, int XR, int EPI, bool PH>
static void launch_clx_e(ClxKernelParams kp, hipStream_t stream) {
    const ConvClxParams& p = kp.p;
    SBV2_REQUIRE(kp.xrows <= XR, "conv_clx: tap span exceeds the window buffer of this configuration");
    kp.gy = p.M / 64;
    const int ntx = round_up((p.N + kClxNT - 1) / kClxNT, 8);
    if (g_clx_grid_only) {
        *g_clx_grid_only = (int64_t)ntx * kp.gy;
        return;
    }
    const size_t lds = std::max<size_t>((size_t)WR * 4096 + (size_t)XB * 2 * XR * 32, (size_t)4 * 64 * 36 * sizeof(float));
    auto kern = conv_clx_kernel<NTAPS, WR, XB, XR, EPI, PH>;
    static std::atomic<uint64_t> lds_allowed{0};
    allow_full_lds(reinterpret_cast<const void*>(kern), lds_allowed);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const bool prof = conv_prof_active();
    if (prof) {
        HIP_CHECK(hipEventCreate(&e0));
        HIP_CHECK(hipEventCreate(&e1));
        HIP_CHECK(hipEventRecord(e0, stream));
    }
    hipLaunchKernelGGL(kern, dim3(ntx * kp.gy), dim3(256), lds, stream, kp);
    HIP_CHECK(hipGetLastError());
    if (prof) {
        HIP_CHECK(hipEventRecord(e1, stream));
        // 28: the flow's FFN convs; 8: a phased launch = a transposed convolution, counted with conv_cl<2,split-bf16> where the upsamplers always were
        conv_prof_add(PH ? 8 : (p.Ykm || p.ntaps == 5 ? 28 : 26), p.prof_flops > 0.0 ? p.prof_flops : 2.0 * p.M * (double)p.N * p.K * p.ntaps, e0, e1);
    }
}

int64_t clx_grid_workgroups(const ConvClxParams& p) {
    int64_t g = 0;
    g_clx_grid_only = &g;
    try {
        launch_conv_clx(p, nullptr);
    } catch (...) {
        g_clx_grid_only = nullptr;
        throw;
    }
    g_clx_grid_only = nullptr;
    return g;
}

void launch_conv_clx(const ConvClxParams& p, hipStream_t stream) {
    SBV2_REQUIRE(conv_clx_usable(p), "conv_clx: operands do not fit the pre-split channels-last kernel");
    ClxKernelParams kp;
    kp.p = p;
    const int step = p.shift_step;
    int tapmax = p.ntaps - 1;          // the largest tap index any row tile reads (phased output: a phase's own taps start at phase_tap0)
    if (p.phase_rows)
        for (int q = 0; q < p.M / p.phase_rows; ++q) tapmax = std::max(tapmax, p.ntaps - 1 + p.phase_tap0[q]);
    const int smin = step >= 0 ? p.shift0 : p.shift0 + tapmax * step;
    const int smax = step >= 0 ? p.shift0 + tapmax * step : p.shift0;
    kp.wshift0 = smin;
    kp.xrows = kClxNT + (smax - smin);
    kp.sh0 = p.shift0 - smin;
    kp.sh_step = step;
    // Every launch whose tap span fits 288-row window buffers (k = 3 / 5 / 7 at every dilation of the model, k = 11 at dilations 1 and 3 and in every conv2)
    // runs on a 4-slot weight ring + two such buffers = 52 KB, THREE workgroups per CU; k = 11 at dilation 5 (span 50 rows) takes three 320-row buffers
    // (76 KB), two per CU.  (Round 4 measured the other ring shapes, 128-row workgroups and 128-position tiles: profiles/HISTORY.md.)
    if (kp.xrows <= 288) {
        if (p.ntaps == 2) launch_clx<2, 4, 2, 288>(kp, stream);
        else if (p.ntaps == 4) launch_clx<4, 4, 2, 288>(kp, stream);
        else if (p.ntaps == 3) launch_clx<3, 4, 2, 288>(kp, stream);
        else if (p.ntaps == 5) launch_clx<5, 4, 2, 288>(kp, stream);
        else if (p.ntaps == 7) launch_clx<7, 4, 2, 288>(kp, stream);
        else launch_clx<11, 4, 2, 288>(kp, stream);
    } else {
        SBV2_REQUIRE(p.ntaps & 1, "conv_clx: even kernel sizes run on 288-row windows");
        if (p.ntaps == 3) launch_clx<3, 4, 2, 320>(kp, stream);
        else if (p.ntaps == 5) launch_clx<5, 4, 2, 320>(kp, stream);
        else if (p.ntaps == 7) launch_clx<7, 4, 2, 320>(kp, stream);
        else launch_clx<11, 4, 2, 320>(kp, stream);
    }
}

}  // namespace sbv2
