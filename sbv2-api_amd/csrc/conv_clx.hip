// Channels-last split-bf16 convolution of the HiFi-GAN ResBlocks on PRE-SPLIT, PRE-ACTIVATED operands: the restructured conv_cl.
//
//   Y[pos][m] (+)= ( sum_tap sum_k W_tap[m][k] * Xs[pos + shift_tap][k] + b[m] + R[pos][m] ) * beta,   Xs = bf16 hi + bf16 lo of lrelu(x)
//   and / or   Ys = bf16 hi + lo of lrelu(that result)   (the operand of the NEXT convolution)
//
// Why (profiles/r03_conv_cl_clock_probe.jsonl, C = 128, k = 7): conv_cl's chunk loop costs 72.8k cycles per workgroup = 23.7k of MFMA issue
// per wave + 44.7k of staging (global loads, f32 -> hi / lo conversion, LDS stores, two barriers per chunk): with one 8-wave workgroup per CU
// (VGPR bound) every wave is in the same phase, so the halves ADD; the MFMA stream alone reaches 448 TFLOP/s against 302 for the kernel.
// Here no wave ever stages through registers:
//   * the activation operand exists in HBM as bf16 parts already, written by the producing epilogue (this kernel's, or split_cl for a stage
//     input), CHUNK-MAJOR: for every 16-channel chunk and part a plane [front + N + back][16] bf16 (32-byte rows, zero halo rows), so a
//     chunk's window [256 + span rows] is one contiguous run per part -> LDS-DMA straight into conv_cl's window layout (the 16-byte half
//     swizzle is applied on the DMA's SOURCE address);
//   * the weight fragments of one (chunk, tap) = 8 KB for 128 rows are LDS-DMA'd into a ring of kWR slots, seven steps ahead;
//   * a step = one tap of one chunk: lgkmcnt(0) [its fragments, requested a step ago] -> counted vmcnt + barrier -> 12 MFMAs per wave with
//     the 8 fragment reads of the next step and the DMAs dealt BETWEEN them.  Waves 0 .. 3 issue the weight DMAs, waves 4 .. 7 the window
//     DMAs (vmcnt is per wave and in order: a wave that issued both kinds would have to wait for young window pieces to reach an old
//     weight block).
// Same fragments, same MFMA order (chunk, tap, lo*hi, hi*lo, hi*hi) as conv_cl: bit-identical results (tests/test_gpu_parity.py).
#include <atomic>
#include <type_traits>

#include "common.h"

// No floating-point contraction in this file: the epilogue's (sum) * beta + previous contents must round like conv_cl's (two roundings), whatever shape
// the surrounding control flow has (an fma there moved the waveform by 9e-7 against the conv_cl path)
#pragma clang fp contract(off)

namespace sbv2 {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void clx_lds_t;
typedef const __attribute__((address_space(1))) void clx_gbl_t;

constexpr int kClxNT = 256;          // positions per workgroup
constexpr int kClxXR = 320;          // window rows per buffer (256 + span <= 64)

struct ClxKernelParams {
    ConvClxParams p;
    int xrows;     // window rows actually read (256 + tap span)
    int wshift0;   // first window row = n0 + wshift0
    int sh0, sh_step;
    int gy;        // row tiles (of 64 * WM rows)
    int contig;    // position tiles dealt to the XCDs in contiguous ranges
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
__device__ __forceinline__ bf16x8 clx_read_b128(unsigned addr) {
    bf16x8 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}
template <int OFF>
__device__ __forceinline__ bf16x8 clx_read_b128o(unsigned addr) {
    bf16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF));
    return v;
}

// WM = 2: 8 waves, 128 rows x 256 positions (both 64-row wave groups read the one staged window), one workgroup per CU.
// WM = 1: 4 waves, 64 rows; <= 80 KB of LDS so that TWO workgroups share a CU: their barriers, prologues and epilogues interleave.
// kClxWR weight ring slots, kClxXB window buffers.
// TN: 32-position tiles per wave (2: 256 positions per workgroup; 1: 128, an experiment: 36 KB and <= 128 registers = FOUR workgroups per CU)
template <int NTAPS, int WM, int kClxWR, int kClxXB, int XR = kClxXR, bool FRONT = false, int TN = 2>
__global__ __launch_bounds__(256 * WM) __attribute__((amdgpu_waves_per_eu(WM == 1 && kClxWR * 4096 + kClxXB * 2 * XR * 32 <= 40 * 1024 ? 4 : (WM == 1 && kClxWR * 4096 + kClxXB * 2 * XR * 32 <= 53 * 1024 ? 3 : 2)))) void conv_clx_kernel(const ClxKernelParams kp) {
    constexpr int NPW = 32 * TN;               // positions per wave
    constexpr int NTW = 4 * NPW;               // positions per workgroup
    constexpr int NMF = 6 * TN;                // MFMAs per wave and step
    constexpr int NRD = 4 + 2 * TN;            // fragment reads per wave and step
    constexpr int G0 = NRD / 2;                // first MFMA gap without fragment reads
    constexpr int NW = 4 * WM;                 // waves
    constexpr int kClxPW = (2 * (XR / 32) + NW / 2 - 1) / (NW / 2);   // window DMA pieces per window wave and chunk (2 parts x XR / 32 pieces of 32 rows)
    // ... per tap (the last tap of a chunk carries none).  FRONT: as many as a tap's gaps hold (8) from the chunk's first tap on, so that the LAST piece of the next
    // window has the rest of the chunk to land (spread evenly its lead is one or two steps at every k)
    constexpr int PPT = FRONT ? (kClxPW < 8 ? kClxPW : 8) : (kClxPW + NTAPS - 2) / (NTAPS - 1);
    static_assert(PPT * (NTAPS - 1) >= kClxPW && PPT <= NMF - G0, "the chunk's taps and their MFMA gaps hold its window pieces");
    static_assert(PPT <= 8, "a tap's MFMA gaps hold its window pieces");
    constexpr int WSLOT = 2 * WM * 2 * 1024;   // one (chunk, tap): 2 WM row tiles x 2 parts x 1 KB
    constexpr int WBYTES = kClxWR * WSLOT;
    constexpr int XPART = XR * 32, XBUF = 2 * XPART;   // (XR = window rows per buffer: 320 holds every tap span <= 64; 288 those <= 32)
    constexpr int NWW = NW / 2;                // weight-DMA waves (the others carry the window)
    constexpr int WPW = (2 * WM * 2) / NWW;    // weight DMAs per weight wave and step (= 2)
    static_assert(WPW == 2, "two weight blocks per weight wave and step");
    const ConvClxParams& p = kp.p;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)smem);

    const unsigned long long st_entry = p.stamps ? __builtin_amdgcn_s_memrealtime() : 0ull;   // (diagnostics only)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wq = wave & 3, wm = wave >> 2;
    // XCD-aware tile order as in conv_cl (1-D grid; the gy row tiles of one position tile share their window: ids 8 apart = same XCD)
    const int bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;
    const int by = slot % kp.gy;
    // position tiles: round-robin over the XCDs (rounds 3: neighbours on different L2s re-fetch each other's halo rows from HBM), or (contig) a
    // contiguous range per XCD
    const int bx = kp.contig ? xcd * ((int)gridDim.x / (8 * kp.gy)) + slot / kp.gy : (slot / kp.gy) * 8 + xcd;
    const int n0 = bx * NTW;
    if (n0 >= p.N) return;
    const int m0 = (by * WM + wm) * 64;        // first output row of this WAVE's tile
    const int M = p.M, N = p.N;
    const int nchunks = p.K >> 4;
    const int S = nchunks * NTAPS;             // steps

    // ---- DMA sources.  All per-step state is incremental (running pointers and LDS offsets, wave-uniform where possible): the address
    // arithmetic of a step sits in front of its first MFMA, right behind the barrier, where nothing overlaps it.
    const bool wwave = wave < NWW;
    // weight wave w: row tile w of the workgroup's 2 WM, both parts (2 KB contiguous per step); fragment order [chunk][mtile][tap][part]
    const int mtw = min(by * 2 * WM + wave, kp.p.nmt - 1);
    const char* wptr = static_cast<const char*>(p.W) + ((int64_t)mtw * NTAPS * 2) * 1024 + lane * 16;   // next weight block to fetch
    const int64_t wjump = (int64_t)kp.p.nmt * NTAPS * 2 * 1024 - (int64_t)NTAPS * 2048;   // from a chunk's last tap to the next chunk's first
    int wtap = 0;                                                                            // tap of the block wptr points at
    const unsigned wdst0 = lds0 + (wave & (NWW - 1)) * 2048;
    unsigned wdoff = 0;                                                                      // ring offset the next weight block goes to
    auto dma_w = [&]() {
        __builtin_amdgcn_global_load_lds((clx_gbl_t*)wptr, (clx_lds_t*)(uintptr_t)(wdst0 + wdoff), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((clx_gbl_t*)(wptr + 1024), (clx_lds_t*)(uintptr_t)(wdst0 + wdoff + 1024), 16, 0, 0);
        wptr += 2048;
        if (++wtap == NTAPS) {
            wtap = 0;
            wptr += wjump;
        }
        wdoff = wdoff + WSLOT == WBYTES ? 0 : wdoff + WSLOT;
    };
    // window wave v = wave - NWW: pieces e = v, v + NXW, ... of the 2 * npc pieces of a chunk (piece = 32 rows of one part); this lane:
    // row = 32 * piece + lane / 2, 16-byte half (lane & 1) ^ ((row >> 3) & 1)  [conv_cl's LDS swizzle, applied on the source]
    constexpr int NXW = NW - NWW;
    const int npc = (kp.xrows + 31) >> 5;      // pieces per part
    const int64_t xplane = (int64_t)(p.X.front + p.X.N + p.X.back) * 32;   // bytes of one (chunk, part) plane
    const int xv = wave - NWW;
    const char* xptr[kClxPW];                  // source of this wave's piece i in the next window to fetch
    unsigned xdst[kClxPW];
    int nmine = 0;                             // pieces of this window wave
#pragma unroll
    for (int i = 0; i < kClxPW; ++i) {
        const int e = max(xv, 0) + i * NXW;
        const int ec = min(e, 2 * npc - 1);
        const int part = ec / npc, pc = ec - part * npc;
        const int row = pc * 32 + (lane >> 1);
        const int half = (lane & 1) ^ ((row >> 3) & 1);
        xptr[i] = static_cast<const char*>(p.X.p) + (int64_t)part * xplane + ((int64_t)p.X.front + n0 + kp.wshift0 + row) * 32 + half * 16;
        xdst[i] = __builtin_amdgcn_readfirstlane(lds0 + WBYTES + part * XPART + pc * 1024);
        if (!wwave && e < 2 * npc) ++nmine;
    }
    nmine = __builtin_amdgcn_readfirstlane(nmine);
    unsigned xdoff = 0;                        // buffer offset the next window goes to
    auto dma_x = [&](auto ic) {                // piece i of the next window; the last piece of a window advances to the following chunk
        constexpr int i = decltype(ic)::value;
        if (i < nmine) {
            __builtin_amdgcn_global_load_lds((clx_gbl_t*)xptr[i], (clx_lds_t*)(uintptr_t)(xdst[i] + xdoff), 16, 0, 0);
            xptr[i] += 2 * xplane;
        }
    };
    auto next_window = [&]() { xdoff = xdoff + XBUF == kClxXB * XBUF ? 0 : xdoff + XBUF; };

    // ---- fragments
    struct Frags {
        bf16x8 ah[2], al[2], bh[TN], bl[TN];
    };
    const int lcol = lane & 31, lh = lane >> 5;
    const unsigned abase = lds0 + (wm * 2) * 2048 + lane * 16;
    // window offsets of this lane's B fragments per tap (the XOR swizzle depends on the row).  The second position tile is 32 rows further: + 1024 bytes
    // with the SAME swizzle bit ((r + 32) >> 3 has the parity of r >> 3), an immediate in its reads instead of a second register per tap
    unsigned boff0[NTAPS];
#pragma unroll
    for (int t = 0; t < NTAPS; ++t) {
        const int r0 = wq * NPW + lcol + kp.sh0 + t * kp.sh_step;
        boff0[t] = lds0 + WBYTES + r0 * 32 + (((lh ^ (r0 >> 3)) & 1) << 4);
    }
    auto read_frag = [&](Frags& f, auto rc, unsigned aaddr, unsigned b0) {
        constexpr int r = decltype(rc)::value;   // 0..3: A (row tile, part); 4..7: B (position tile, part)
        if constexpr (r == 0) f.ah[0] = clx_read_b128o<0>(aaddr);
        else if constexpr (r == 1) f.al[0] = clx_read_b128o<1024>(aaddr);
        else if constexpr (r == 2) f.ah[1] = clx_read_b128o<2048>(aaddr);
        else if constexpr (r == 3) f.al[1] = clx_read_b128o<3072>(aaddr);
        else if constexpr (r == 4) f.bh[0] = clx_read_b128o<0>(b0);
        else if constexpr (r == 5) f.bl[0] = clx_read_b128o<XPART>(b0);
        else if constexpr (r == 6 && TN == 2) f.bh[TN - 1] = clx_read_b128o<1024>(b0);
        else if constexpr (TN == 2) f.bl[TN - 1] = clx_read_b128o<XPART + 1024>(b0);
    };
    unsigned wroff = 0;        // ring offset of the weight slot the NEXT fragment reads take (step s + 1 while step s runs)
    unsigned xroff = 0;        // buffer offset of the window those reads take

    f32x16 acc[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // MFMA n of a step: term-major (lo*hi for the four tiles, hi*lo, hi*hi): per accumulator the order of conv_cl
    auto mfma_one = [&](const Frags& f, auto nc) {
        constexpr int n = decltype(nc)::value;
        constexpr int t = n / (2 * TN), i = (n % (2 * TN)) / TN, j = n % TN;
        if constexpr (t == 0) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.al[i], f.bh[j], acc[i][j], 0, 0, 0);
        else if constexpr (t == 1) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[i], f.bl[j], acc[i][j], 0, 0, 0);
        else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[i], f.bh[j], acc[i][j], 0, 0, 0);
    };

    // ---- prologue: the whole weight ring, the windows of chunks 0 .. kClxXB - 2 (iteration `chunk` stages chunk + kClxXB - 1)
    if (wwave) {
        const int npre = min(kClxWR, S);
        for (int u = 0; u < npre; ++u) dma_w();
        if (npre == kClxWR) clx_wait_vm<2 * (kClxWR - 1)>();
        else clx_wait_vm<0>();
    } else {
        const int nwin = min(kClxXB - 1, nchunks);
        for (int c = 0; c < nwin; ++c) {
            clx_static_for<0, kClxPW>([&](auto ic) { dma_x(ic); });
            next_window();
        }
        clx_wait_vm_dyn((nwin - 1) * nmine);   // chunk 0's window has landed
    }
    __builtin_amdgcn_s_barrier();
    Frags fa, fb;
    clx_static_for<0, NRD>([&](auto rc) { read_frag(fa, rc, abase, boff0[0]); });
    wroff = WSLOT == WBYTES ? 0 : WSLOT;

    // One step: tap `tap` of chunk `chunk` (s = chunk * NTAPS + tap).  LAST = the step's successor opens a new chunk.
    auto step = [&](int s, int chunk, auto tapc, Frags& cur, Frags& nxt) {
        constexpr int tap = decltype(tapc)::value;
        constexpr bool LAST = tap == NTAPS - 1;
        constexpr int tapn = LAST ? 0 : tap + 1;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the fragments of step s (requested a step ago): its weight slot is free after the barrier
        if (wwave) {
            // the weight blocks of step s + 1 have landed: everything but the blocks of steps s + 2 .. s + kClxWR - 1 (all issued in the steady state)
            if (s + kClxWR - 1 < S) clx_wait_vm<2 * (kClxWR - 2)>();
            else clx_wait_vm<0>();
        } else if (LAST && chunk + 1 < nchunks) {
            // the window of chunk + 1 has landed: everything but the windows of chunks + 2 .. + kClxXB - 1 (all issued in the steady state)
            if (chunk + kClxXB - 1 < nchunks) clx_wait_vm_dyn((kClxXB - 2) * nmine);
            else clx_wait_vm<0>();
        }
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const bool rd = s + 1 < S;
        if (LAST) xroff = xroff + XBUF == kClxXB * XBUF ? 0 : xroff + XBUF;   // the next step reads the next chunk's window
        const unsigned aaddr = abase + wroff, b0 = boff0[tapn] + xroff;
        wroff = wroff + WSLOT == WBYTES ? 0 : wroff + WSLOT;
        const bool stw = wwave && s + kClxWR < S;
        const bool stx = !wwave && chunk + kClxXB - 1 < nchunks;
        clx_static_for<0, NMF>([&](auto nc) {
            constexpr int n = decltype(nc)::value;
            mfma_one(cur, nc);
            if constexpr (n < G0) {
                if (rd) {
                    read_frag(nxt, std::integral_constant<int, 2 * n>{}, aaddr, b0);
                    read_frag(nxt, std::integral_constant<int, 2 * n + 1>{}, aaddr, b0);
                }
            } else {
                if constexpr (n == (TN == 2 ? 5 : G0)) {
                    if (stw) dma_w();   // the blocks of step s + kClxWR, into the slot of step s (released by this step's barrier)
                }
                // window pieces of chunk + kClxXB - 1 (its buffer held chunk - 1): PPT per tap, one per gap, none with the chunk's last tap
                if constexpr (tap < NTAPS - 1 && n - G0 < PPT) {
                    constexpr int i = tap * PPT + (n - G0);
                    if constexpr (i < kClxPW) {
                        if (stx) dma_x(std::integral_constant<int, i>{});
                    }
                }
                if constexpr (LAST && n == G0) {
                    if (stx) next_window();
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    };

    unsigned long long st_t0 = 0, st_r0 = 0;
    if (p.stamps) {
        st_t0 = __builtin_amdgcn_s_memtime();
        st_r0 = __builtin_amdgcn_s_memrealtime();
    }
    if (p.variant & 4) __builtin_amdgcn_s_setprio(2);   // (experiment) the step loop's instructions outrank the other workgroups' prologues / epilogues
    for (int chunk = 0; chunk < nchunks; chunk += 2) {
        // two chunks per iteration: NTAPS is odd, the fragment register sets ping-pong per step
        const int s0 = chunk * NTAPS;
        clx_static_for<0, NTAPS>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            if constexpr ((t & 1) == 0) step(s0 + t, chunk, tc, fa, fb);
            else step(s0 + t, chunk, tc, fb, fa);
        });
        if (chunk + 1 < nchunks) {
            clx_static_for<0, NTAPS>([&](auto tc) {
                constexpr int t = decltype(tc)::value;
                if constexpr (((NTAPS + t) & 1) == 0) step(s0 + NTAPS + t, chunk + 1, tc, fa, fb);
                else step(s0 + NTAPS + t, chunk + 1, tc, fb, fa);
            });
        }
    }
    if (p.stamps && tid == 0) {
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        unsigned long long* o = p.stamps + (size_t)blockIdx.x * 8;
        o[0] = st_t0; o[1] = st_r0; o[2] = t1; o[3] = r1;
        o[4] = st_entry;
        // where this workgroup ran: HW_ID (wave / SIMD / CU / SH / SE) and XCC_ID
        o[7] = (unsigned long long)__builtin_amdgcn_s_getreg(4 | (31 << 11)) | ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (31 << 11)) << 32);
    }
    if (p.variant & 4) __builtin_amdgcn_s_setprio(0);
    __syncthreads();   // the epilogue re-uses the rings as its transpose tiles

    // ---- k-major result (the flow's second FFN convolution: Y[m][n] = (conv + b + R[m][n]) * mask): one accumulator register of a half-wave
    // is 32 consecutive positions of one channel = a 128-byte run of the plane; bias, mask and residual of a row tile are requested before its
    // first store (conv_cl's k-major epilogue)
    if (p.Ykm) {
        int nn[TN];
        bool nok[TN], keepn[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            nn[j] = n0 + wq * NPW + j * 32 + lcol;
            nok[j] = nn[j] < N;
            const int nc = min(nn[j], N - 1);
            keepn[j] = !p.mask || p.mask[nc >> p.mask_shift] != 0;
        }
        clx_static_for<0, 2>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            float brow[16], rr[16][TN];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int mc = min(m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, M - 1);
                brow[r] = p.bias ? p.bias[mc] : 0.f;
#pragma unroll
                for (int j = 0; j < TN; ++j) rr[r][j] = p.Rkm ? p.Rkm[(int64_t)mc * p.ldrkm + min(nn[j], N - 1)] : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m >= M) continue;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (!nok[j]) continue;
                    float v = acc[i][j][r] + brow[r];
                    if (p.Rkm) v += rr[r][j];
                    v *= p.beta;
                    p.Ykm[(int64_t)m * p.ldykm + nn[j]] = keepn[j] ? v : 0.f;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        return;
    }

    // ---- epilogue (conv_cl's channels-last epilogue): each wave transposes its 32 x 64 sub-tiles through a private LDS tile [64 positions][36]
    // so that 8 consecutive lanes hold one full 128-byte line of a row; everything read from global memory is requested before the first store.
    float* tile = reinterpret_cast<float*>(smem) + wave * (NPW * 36);
    const float beta = p.beta;
    const int64_t yplane = (int64_t)(p.Ys.front + p.Ys.N + p.Ys.back) * 32;
    const int c4 = (lane & 7) * 4;
    const int nfirst = n0 + wq * NPW + (lane >> 3);
    const float* trow = tile + (lane >> 3) * 36 + c4;
    const float sl = p.ys_slope;
    // ---- interior tiles (every position and row of the tile exists; all but the batch's last tile): no per-lane conditions, and EVERY global read of both
    // row tiles (mask bytes: one load per lane + a ballot, bias, residual rows) is requested before the first store.  Round 4's epilogue took 14 (conv1) to
    // 22-30 us (conv2) of a workgroup's 34-85 us (profiles/r05a_clx_timeline.jsonl): sixteen mask-byte loads each followed by s_waitcnt vmcnt(0), the
    // second row tile's loads queued behind the first one's stores (a load's data returns behind every older store's acknowledgement), and an
    // s_waitcnt vmcnt(0) at the join behind every conditional store.
    if (!(p.variant & 2) && n0 + NTW <= N) {
        // lane (group g = lane >> 3, j = lane & 7) loads the flag of position nfirst + 8 j; bit 8 g + it of the ballot is this lane's flag of iteration it
        unsigned mv = 1u;
        if (p.mask) mv = p.mask[(nfirst + (lane & 7) * 8) >> p.mask_shift];
        // two sets of 4 TN rows: the residual rows of both row tiles; or, for an accumulating launch (a branch's last step: 4 of a step's 36 launches),
        // residual + previous contents of ONE row tile (the second tile's are requested behind the first one's stores)
        f32x4v b4[2], ld[2][4 * TN];
        const bool acc_y = p.accumulate != 0;
        auto load_rows = [&](int set, const float* base, int ldb, int m) {
            const float* rp = base + (int64_t)nfirst * ldb + m;
#pragma unroll
            for (int it = 0; it < 4 * TN; ++it) ld[set][it] = *reinterpret_cast<const f32x4v*>(rp + (int64_t)it * 8 * ldb);
        };
#pragma unroll
        for (int i = 0; i < 2; ++i) b4[i] = p.bias ? *reinterpret_cast<const f32x4v*>(p.bias + m0 + i * 32 + c4) : f32x4v{0.f, 0.f, 0.f, 0.f};
        if (p.R) load_rows(0, p.R, p.ldr, m0 + c4);
        if (acc_y) load_rows(1, p.Y, p.ldy, m0 + c4);
        else if (p.R) load_rows(1, p.R, p.ldr, m0 + 32 + c4);
        unsigned mbits = 0xFFu;
        bool allkeep = true;
        clx_static_for<0, 2>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4v v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                    *reinterpret_cast<f32x4v*>(tile + (j * 32 + lcol) * 36 + 8 * q + 4 * lh) = v;
                }
            if constexpr (i == 0) {
                asm volatile("" : "+v"(mv));   // (the compare stays behind the LDS writes: hoisted to the load, it waits for the load there)
                const unsigned long long bal = __builtin_amdgcn_ballot_w64(mv != 0);
                allkeep = bal == ~0ull;
                mbits = (unsigned)(bal >> ((lane >> 3) * 8)) & 0xFFu;
            }
            const int m = m0 + i * 32 + c4;
            if constexpr (i == 1) {
                if (acc_y) {
                    if (p.R) load_rows(0, p.R, p.ldr, m);
                    load_rows(1, p.Y, p.ldy, m);
                }
            }
            float* yp = p.Y ? p.Y + (int64_t)nfirst * p.ldy + m : nullptr;
            const int64_t ystep = (int64_t)8 * p.ldy;
            char* qs = p.Ys.p ? static_cast<char*>(p.Ys.p) + ((int64_t)(m >> 4) * 2) * yplane + ((int64_t)p.Ys.front + nfirst) * 32 + (m & 15) * 2 : nullptr;
#pragma unroll
            for (int it = 0; it < 4 * TN; ++it) {
                const f32x4v a = *reinterpret_cast<const f32x4v*>(trow + it * 8 * 36);
                f32x4v v = a + b4[i];
                if (p.R) v += acc_y ? ld[0][it] : ld[i][it];
                v *= beta;   // (x 1.0f is exact: the generic path's test for beta != 1 changes no bit)
                if (acc_y) v += ld[1][it];
                if (!allkeep && !((mbits >> it) & 1u)) v = f32x4v{0.f, 0.f, 0.f, 0.f};
                if (yp) *reinterpret_cast<f32x4v*>(yp + it * ystep) = v;
                if (qs) {
                    bf16x4 h, l;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float x = v[e] >= 0.f ? v[e] : v[e] * sl;
                        h[e] = (__bf16)x;
                        l[e] = (__bf16)(x - (float)h[e]);
                    }
                    *reinterpret_cast<bf16x4*>(qs + it * 256) = h;
                    *reinterpret_cast<bf16x4*>(qs + it * 256 + yplane) = l;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        if (p.stamps && tid == 0) {
            unsigned long long* o = p.stamps + (size_t)blockIdx.x * 8;
            o[5] = __builtin_amdgcn_s_memrealtime();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            o[6] = __builtin_amdgcn_s_memrealtime();
        }
        return;
    }
    clx_static_for<0, 2>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4v v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                *reinterpret_cast<f32x4v*>(tile + (j * 32 + lcol) * 36 + 8 * q + 4 * lh) = v;
            }
        const int m = m0 + i * 32 + c4;
        const bool mok = m < M;
        const int mc = mok ? m : 0;
        const f32x4v b4 = p.bias ? *reinterpret_cast<const f32x4v*>(p.bias + mc) : f32x4v{0.f, 0.f, 0.f, 0.f};
        f32x4v rold[4 * TN], rres[4 * TN];
        unsigned char keep[4 * TN];
#pragma unroll
        for (int it = 0; it < 4 * TN; ++it) {
            const int64_t pp = min(nfirst + it * 8, N - 1);
            if (p.accumulate) rold[it] = *reinterpret_cast<const f32x4v*>(p.Y + pp * p.ldy + mc);
            if (p.R) rres[it] = *reinterpret_cast<const f32x4v*>(p.R + pp * p.ldr + mc);
            keep[it] = p.mask ? p.mask[pp >> p.mask_shift] : 1;
        }
        float* yp = p.Y ? p.Y + (int64_t)nfirst * p.ldy + m : nullptr;
        const int64_t ystep = (int64_t)8 * p.ldy;
        // bf16 parts of lrelu(result), chunk-major (4 channels = 8 bytes of a 32-byte row; the lo plane follows the hi plane)
        char* qs = p.Ys.p ? static_cast<char*>(p.Ys.p) + ((int64_t)(m >> 4) * 2) * yplane + ((int64_t)p.Ys.front + nfirst) * 32 + (m & 15) * 2 : nullptr;
#pragma unroll
        for (int it = 0; it < 4 * TN; ++it) {
            const int n = nfirst + it * 8;
            const f32x4v a = *reinterpret_cast<const f32x4v*>(trow + it * 8 * 36);
            if (n < N && mok) {
                f32x4v v = a + b4;
                if (p.R) v += rres[it];
                if (beta != 1.0f) v *= beta;
                if (p.accumulate) v += rold[it];
                if (!keep[it]) v = f32x4v{0.f, 0.f, 0.f, 0.f};
                if (yp) *reinterpret_cast<f32x4v*>(yp) = v;
                if (qs) {
                    bf16x4 h, l;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float x = v[e] >= 0.f ? v[e] : v[e] * sl;
                        h[e] = (__bf16)x;
                        l[e] = (__bf16)(x - (float)h[e]);
                    }
                    *reinterpret_cast<bf16x4*>(qs) = h;
                    *reinterpret_cast<bf16x4*>(qs + yplane) = l;
                }
            }
            if (yp) yp += ystep;
            if (qs) qs += 8 * 32;
        }
        __builtin_amdgcn_sched_barrier(0);   // (the second row tile's loads stay behind this one's stores: hoisted, the two tiles' registers spill)
    });
    if (p.stamps && tid == 0) {   // (diagnostics) last store issued / every store of this wave acknowledged
        unsigned long long* o = p.stamps + (size_t)blockIdx.x * 8;
        o[5] = __builtin_amdgcn_s_memrealtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        o[6] = __builtin_amdgcn_s_memrealtime();
    }
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
    if (!(p.ntaps == 3 || p.ntaps == 5 || p.ntaps == 7 || p.ntaps == 11)) return false;
    if (p.Ykm && (p.Y || p.Ys.p || p.accumulate || p.R)) return false;   // the k-major epilogue writes Ykm only
    if ((p.K & 15) || (p.M & 63) || p.K != p.X.C || p.nmt * 32 < p.M || (p.nmt & 1)) return false;
    const int span = (p.ntaps - 1) * std::abs(p.shift_step);
    if (span > kClxXR - kClxNT || p.shift0 < -kClxFront || p.shift0 + span > 64) return false;
    if (p.mask && p.mask_shift < 0) return false;
    if (p.Y && (p.ldy & 3)) return false;
    if (p.R && (p.ldr & 3)) return false;
    if (p.Ys.p && (p.Ys.C != p.M || p.Ys.N != p.N)) return false;
    return p.N >= 1 && p.X.N == p.N;
}

static thread_local int64_t* g_clx_grid_only = nullptr;   // clx_grid_workgroups: report the grid instead of launching

template <int NTAPS, int WM, int WR, int XB, int XR = kClxXR, bool FRONT = false, int TN = 2>
static void launch_clx(ClxKernelParams kp, hipStream_t stream) {
    constexpr int NTW = 128 * TN;
    kp.xrows += NTW - kClxNT;
    const ConvClxParams& p = kp.p;
    SBV2_REQUIRE(kp.xrows <= XR, "conv_clx: tap span exceeds the window buffer of this configuration");
    kp.gy = p.M / (64 * WM);
    static const int contig = getenv("SBV2_CLX_CONTIG") ? atoi(getenv("SBV2_CLX_CONTIG")) : 0;
    kp.contig = contig || (p.variant & 1);
    if (g_clx_grid_only) {
        *g_clx_grid_only = (int64_t)round_up((p.N + NTW - 1) / NTW, 8) * kp.gy;
        return;
    }
    const size_t lds = std::max<size_t>((size_t)WR * (2 * WM * 2 * 1024) + (size_t)XB * 2 * XR * 32, (size_t)4 * WM * 32 * TN * 36 * sizeof(float));
    auto kern = conv_clx_kernel<NTAPS, WM, WR, XB, XR, FRONT, TN>;
    static std::atomic<uint64_t> lds_allowed{0};
    allow_full_lds(reinterpret_cast<const void*>(kern), lds_allowed);
    const int ntx = round_up((p.N + NTW - 1) / NTW, 8);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const bool prof = conv_prof_active();
    if (prof) {
        HIP_CHECK(hipEventCreate(&e0));
        HIP_CHECK(hipEventCreate(&e1));
        HIP_CHECK(hipEventRecord(e0, stream));
    }
    hipLaunchKernelGGL(kern, dim3(ntx * kp.gy), dim3(256 * WM), lds, stream, kp);
    HIP_CHECK(hipGetLastError());
    if (prof) {
        HIP_CHECK(hipEventRecord(e1, stream));
        conv_prof_add(p.Ykm || p.ntaps == 5 ? 28 : 26, 2.0 * p.M * (double)p.N * p.K * p.ntaps, e0, e1);   // 28: the flow's FFN convs
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
    static const int env_variant = getenv("SBV2_CLX_VARIANT") ? atoi(getenv("SBV2_CLX_VARIANT")) : 0;   // (round-5 experiments; see ConvClxParams::variant)
    kp.p.variant |= env_variant;
    const int step = p.shift_step;
    const int smin = step >= 0 ? p.shift0 : p.shift0 + (p.ntaps - 1) * step;
    const int smax = step >= 0 ? p.shift0 + (p.ntaps - 1) * step : p.shift0;
    kp.wshift0 = smin;
    kp.xrows = kClxNT + (smax - smin);
    kp.sh0 = p.shift0 - smin;
    kp.sh_step = step;
    // 8 (default, round 4): every launch whose tap span fits 288-row window buffers (k = 3 / 5 / 7 at every dilation, k = 11 at dilations 1 and 3 and in
    // every conv2) runs on a 4-slot weight ring + two such buffers = 52 KB, THREE 64-row workgroups per CU: the prologue and epilogue of one overlap the
    // others' loops (conv_clx 33.1 -> 31.0 ms, the flow's FFN convs 6.57 -> 6.2 ms per step, same boxes); k = 11 at dilation 5 (span 50 rows: 320-row
    // buffers, 56 KB) stays at two per CU.  7 = the same without k = 11; 1 = two per CU for every k (round 3); 2 = 128-row workgroups (one per CU); 3 / 4 / 6
    // = other ring shapes (measured, slower: DESIGN 5.3)
    static const int cfg = getenv("SBV2_CLX_CFG") ? atoi(getenv("SBV2_CLX_CFG")) : 8;
    static const int front = getenv("SBV2_CLX_FRONT") ? atoi(getenv("SBV2_CLX_FRONT")) : 0;
    // SBV2_CLX_NT128: mask of kernel sizes (1: k = 3, 2: k = 5, 4: k = 7, 8: k = 11) that run on 128-position workgroups (36 KB, <= 110 registers: four per CU)
    static const int nt128 = getenv("SBV2_CLX_NT128") ? atoi(getenv("SBV2_CLX_NT128")) : 0;
    if (cfg == 2 && (p.M & 127) == 0 && p.ntaps != 5) {
        if (p.ntaps == 3) launch_clx<3, 2, 8, 3>(kp, stream);
        else if (p.ntaps == 7) launch_clx<7, 2, 8, 3>(kp, stream);
        else launch_clx<11, 2, 8, 3>(kp, stream);
    } else if (cfg == 8 && kp.xrows <= 288 && ((nt128 >> (p.ntaps == 3 ? 0 : p.ntaps == 5 ? 1 : p.ntaps == 7 ? 2 : 3)) & 1)) {   // 128-position workgroups, four per CU, for the kernel sizes of the mask
        if (p.ntaps == 3) launch_clx<3, 1, 4, 2, 160, false, 1>(kp, stream);
        else if (p.ntaps == 5) launch_clx<5, 1, 4, 2, 160, false, 1>(kp, stream);
        else if (p.ntaps == 7) launch_clx<7, 1, 4, 2, 160, false, 1>(kp, stream);
        else launch_clx<11, 1, 4, 2, 160, false, 1>(kp, stream);
    } else if (cfg == 8 && front && kp.xrows <= 288 && p.ntaps != 3) {   // (experiment: window pieces front-loaded)
        if (p.ntaps == 5) launch_clx<5, 1, 4, 2, 288, true>(kp, stream);
        else if (p.ntaps == 7) launch_clx<7, 1, 4, 2, 288, true>(kp, stream);
        else launch_clx<11, 1, 4, 2, 288, true>(kp, stream);
    } else if (cfg == 8 && kp.xrows <= 288 && p.ntaps == 11) {   // k = 11 at dilations 1 and 3 (and every conv2) fits the 288-row buffers too
        launch_clx<11, 1, 4, 2, 288>(kp, stream);
    } else if ((cfg == 5 || cfg == 6 || cfg == 7 || cfg == 8) && kp.xrows <= 288 && p.ntaps != 11 && (p.ntaps != 5 || cfg >= 7)) {
        // 4-slot weight ring + two 288-row window buffers (tap spans <= 32: k = 3 and k = 7 at every dilation of the model): 52 KB, THREE workgroups per CU
        if (p.ntaps == 3 && front) launch_clx<3, 1, 4, 2, 288, true>(kp, stream);
        else if (p.ntaps == 3) launch_clx<3, 1, 4, 2, 288>(kp, stream);
        else if (p.ntaps == 5) launch_clx<5, 1, 4, 2, 288>(kp, stream);
        else launch_clx<7, 1, 4, 2, 288>(kp, stream);
    } else if (cfg == 6 && p.ntaps == 11) {
        launch_clx<11, 1, 3, 2>(kp, stream);
    } else if (cfg == 4 && p.ntaps != 5) {   // 52 KB of LDS (3-slot weight ring, 2 window buffers) and <= 168 registers: THREE workgroups per CU
        if (p.ntaps == 3) launch_clx<3, 1, 3, 2>(kp, stream);
        else if (p.ntaps == 7) launch_clx<7, 1, 3, 2>(kp, stream);
        else launch_clx<11, 1, 3, 2>(kp, stream);
    } else if (cfg == 3 && p.ntaps != 5) {
        if (p.ntaps == 3) launch_clx<3, 1, 8, 2>(kp, stream);
        else if (p.ntaps == 7) launch_clx<7, 1, 8, 2>(kp, stream);
        else launch_clx<11, 1, 8, 2>(kp, stream);
    } else {
        if (p.ntaps == 3) launch_clx<3, 1, 4, 3>(kp, stream);
        else if (p.ntaps == 5) launch_clx<5, 1, 4, 3>(kp, stream);
        else if (p.ntaps == 7) launch_clx<7, 1, 4, 3>(kp, stream);
        else launch_clx<11, 1, 4, 3>(kp, stream);
    }
}

}  // namespace sbv2
