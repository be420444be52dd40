// Split-bf16 1x1 products on k-major planes, gfx950 bf16 matrix cores (v_mfma_f32_32x32x16_bf16), operands pre-split.
//
//   Y[m][n] = epi( sum_k W[m][k] * X[k][n] ),   W = sum_p Wp, X = sum_p Xp  (bf16 parts, most significant first)
//
//   PARTS = 2 ("bf16x3"): lo*hi + hi*lo + hi*hi, three MFMAs per product, dropped term 2^-16 relative.
//   PARTS = 3 ("bf16x6"): lo*hi + hi*lo + mid*mid + mid*hi + hi*mid + hi*hi, six MFMAs; the dropped terms (mid*lo, lo*mid, lo*lo) are
//                         2^-24 relative, i.e. the rounding of an f32 product: f32-grade results at 6/16 of the f32 MFMA's cost.
//   F16 ("f16x3", PARTS = 2): f16 hi + f16 lo' with lo' = (x - hi) 2^11 (common.h): hi*hi into the accumulator, lo'*hi + hi*lo' into a second
//                         accumulator added with 2^-11 by the epilogue; three v_mfma_f32_32x32x16_f16, dropped term 2^-22 relative.
//
// What the reference does here: nothing -- these are the MatMul / Gemm / 1x1 Conv nodes ONNX Runtime executes inside `session.run`
// (crates/sbv2_core/src/bert.rs:11, model.rs:91); the reference itself offers reduced precision for them (TensorRT fp16 for BERT,
// model.rs:11-17; TF32 on CUDA, model.rs:27-30).
//
// Why a third GEMM kernel: conv_cl.hip converts f32 -> bf16 parts while staging, which at ONE tap costs as much as the MFMAs it feeds
// (43-75 TFLOP/s measured), and gemm_conv.hip is pinned to the f32 pipe (157 TFLOP/s peak).  Here the parts exist in HBM already (written by
// the producer of the plane: LayerNorm, the previous product's epilogue, split_planes), so a chunk's tiles go L2 -> LDS by LDS-DMA
// (global_load_lds_dwordx4) into a ring, the loop holds no VALU work besides addresses, and
//   * A (weights) is packed at load time as MFMA fragments (1 KB lane-linear blocks -> conflict-free ds_read_b128),
//   * B (activations) stays a k-major image [16 k][NT n] and is read with ds_read_b64_tr_b16 (the transposing LDS read: a lane receives 4
//     consecutive k of its column); the 64-byte segments of a row are XOR-swizzled on the DMA's SOURCE address so that the four k rows a
//     half-wave touches land on four different bank quarters (conflict-free; cdna_hip_programming.md T10 / rule 21).
// One workgroup = 4 waves = (32 TM WM) x (32 TN WN) outputs; per 16-deep chunk: barrier, fragment reads of chunk c + 1 into the second
// register set, DMAs of chunk c + NSLOT - 1, MFMAs of chunk c.  The per-element summation order (chunk, then the term order above) does
// not depend on the tile shape, so a batch row equals the single-utterance call bit for bit whatever configuration either picks.
#include <atomic>
#include <type_traits>

#include "common.h"

namespace sbv2 {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void bfs_lds_t;
typedef const __attribute__((address_space(1))) void bfs_gbl_t;
typedef __attribute__((address_space(3))) s16x4 bfs_lds_s16x4;

struct BfsKernelParams {
    GemmBfsParams p;
    int mask_shift;
    int gm, gn, total;
    int ksplit;   // SK instances: workgroups per output tile (total = gm * gn * ksplit)
};

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

template <int N>
__device__ __forceinline__ void bfs_wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// LDS reads the compiler does not count (see the kernel): 16 bytes per lane / the transposing 4 x 16-bit read
template <int OFF>
__device__ __forceinline__ bf16x8 bfs_read_b128(unsigned addr) {
    bf16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF));
    return v;
}
template <int OFF>
__device__ __forceinline__ s16x4 bfs_read_tr(unsigned addr) {
    s16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF));
    return v;
}

// XOR applied to the 64-byte segment index of row k of the B image (row = RB bytes): rows k .. k + 3 of one segment column then fall on
// four different quarters of the 256-byte bank row
template <int RB>
__device__ __forceinline__ int bfs_swz(int k) {
    if (RB >= 256) return k & 3;
    if (RB == 128) return (k >> 1) & 1;
    return 0;
}

// SK (round 5, small grids): a single utterance's DeBERTa products are 32 - 128 tiles on 256 CUs and each workgroup streams its tile's whole K range through
// ONE CU's L2 -> LDS path: 10.8 us at K = 1024, 29.9 at K = 4096 (4.4 us + 0.1 us per 16-deep chunk = 2 MB at 70 GB/s; splitting the chunks over more waves of
// the SAME workgroup changed nothing, tools/ksplit_probe.py).  With SK, kp.ksplit workgroups share a tile: workgroup (tile, g) multiplies the g-th contiguous
// K / ksplit range, leaves its accumulators in p.sk.ws and counts itself in; the one that arrives last adds the ksplit partial sums IN GROUP ORDER (its own
// included, from memory: the result does not depend on who was last) and runs the epilogue.  Another summation order than the unsplit kernel: a batch row and
// the single-utterance call agree to f32 rounding instead of bit for bit (sbv2_debug_set_ksplit(0): the unsplit dispatch, which the bit-equality tests run on).
// KG > 1 (round 5, the single utterance's K = 1024 products): the workgroup is KG groups of WM x WN waves (here: four groups of ONE wave, a 32 x 32 output tile);
// group g owns the g-th contiguous K / KG range and its own ring, every group takes the same number of barriers, and behind the loop groups 1 .. KG - 1 hand
// their accumulators to group 0 through their (now idle) rings, which adds them in group order and runs the epilogue.  What this buys is not the shorter loop per
// wave (the same split on 64 x 64 tiles measured nothing) but HALF the bytes per CU: a 32 x 32 tile streams (32 + 32) rows of K where a 64 x 64 one streams
// (64 + 64), on four times as many CUs.  Summation order: K / KG partial sums (sbv2_debug_set_ksplit(0) selects the unsplit kernel).
template <int PARTS, int TM, int TN, int WM, int WN, int KSUB, int NSLOT, bool F16 = false, bool SK = false, int KG = 1>
__global__ __launch_bounds__(64 * WM * WN * KG) __attribute__((amdgpu_waves_per_eu(KG > 1 ? 1 : (F16 && TM * TN >= 4 ? 2 : (TM * TN == 2 && NSLOT <= 4 ? 3 : 1))))) void gemm_bfs_kernel(const BfsKernelParams kp) {
    static_assert(!F16 || PARTS == 2, "f16x3 has two planes");
    constexpr int NW = WM * WN;   // waves of a group
    static_assert(NW == 4 || NW == 1, "4 waves per group, or one");
    static_assert(!(SK && KG > 1), "one kind of K split at a time");
    constexpr int MT = 32 * TM * WM, NT = 32 * TN * WN, RB = NT * 2;
    constexpr int NMT = MT / 32;
    constexpr int A_BYTES = NMT * PARTS * 1024, B_PART = 16 * RB, B_BYTES = PARTS * B_PART;
    constexpr int CH = A_BYTES + B_BYTES;     // one 16-deep chunk
    constexpr int SLOT = KSUB * CH;           // a ring slot = KSUB chunks = the span between two barriers
    constexpr int GA = A_BYTES / 1024, GBP = B_PART / 1024, GB = GBP * PARTS;
    static_assert((GA + GB) % NW == 0 && GBP >= 1, "the DMAs of a chunk are dealt evenly over the group's waves");
    constexpr int PERW = (GA + GB) / NW;
    static_assert((NSLOT - 1) * KSUB * PERW <= 60, "vmcnt is a 6-bit counter");
    const GemmBfsParams& p = kp.p;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int grp = KG > 1 ? __builtin_amdgcn_readfirstlane((tid >> 6) / NW) : 0;   // K group (KG > 1)
    const int wave = KG > 1 ? __builtin_amdgcn_readfirstlane((tid >> 6) % NW) : tid >> 6;   // wave inside its group
    const int wm = wave / WN, wn = wave % WN;
    // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs; each XCD walks a contiguous range of tiles (m fastest), so the
    // tiles co-resident on one L2 share their weight rows / activation columns.  Speed only.
    const int per = gridDim.x >> 3;
    const int t_all = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (t_all >= kp.total) return;
    const int ntile = kp.gm * kp.gn;
    const int t = SK ? t_all % ntile : t_all;        // output tile
    const int kg = SK ? t_all / ntile : 0;           // ... and which part of K
    const int tmi = t % kp.gm, tni = t / kp.gm;
    const int m0 = tmi * MT, n0 = tni * NT;
    const int M = p.M, N = p.N;
    const int nchunks = SK ? (p.K >> 4) / kp.ksplit : (p.K >> 4) / KG;   // of this workgroup / group (the launch checks divisibility)

    // ---- DMA descriptors: DMA g of a chunk (g < GA: weight fragment blocks, else 1 KB pieces of the activation parts) belongs to wave g % 4
    const char* src[PERW];
    int64_t step[PERW];
    int dst[PERW];
#pragma unroll
    for (int q = 0; q < PERW; ++q) {
        const int gi = wave + NW * q;
        if (gi < GA) {
            const int mt = min(m0 / 32 + gi / PARTS, p.W.nmt - 1), part = gi % PARTS;   // row tiles outside the problem only feed outputs never stored
            src[q] = static_cast<const char*>(p.W.w) + ((int64_t)mt * PARTS + part) * 1024 + lane * 16;
            step[q] = (int64_t)p.W.nmt * PARTS * 1024;
            dst[q] = gi * 1024;
        } else {
            const int gb = gi - GA;
            const int part = gb / GBP, r = gb % GBP;
            const int pos = r * 1024 + lane * 16;
            const int k = pos / RB, bq = pos % RB;
            const int seg = (bq >> 6) ^ bfs_swz<RB>(k);
            int n = n0 + (((bq & 63) | (seg << 6)) >> 1);
            if (n >= N) n = 0;   // columns outside the problem only feed outputs never stored
            src[q] = static_cast<const char*>(p.X.p) + ((int64_t)part * p.X.pstride + (int64_t)k * p.X.ld + n) * 2;
            step[q] = (int64_t)16 * p.X.ld * 2;
            dst[q] = A_BYTES + part * B_PART + r * 1024;
        }
    }
    // LDS destinations are wave-uniform (M0): made provably so once, or hipcc re-derives them per DMA with v_readfirstlane
    int sdst[PERW];
#pragma unroll
    for (int q = 0; q < PERW; ++q) {
        sdst[q] = __builtin_amdgcn_readfirstlane(dst[q]);
        if (SK) src[q] += (int64_t)kg * nchunks * step[q];
        if (KG > 1) src[q] += (int64_t)grp * nchunks * step[q];
    }
    const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)smem) + (KG > 1 ? grp * (NSLOT * SLOT) : 0);
    auto dma = [&](int q, int off) {   // DMA q of the next chunk to stage (chunk image at LDS offset off); advances its source pointer
        __builtin_amdgcn_global_load_lds((bfs_gbl_t*)src[q], (bfs_lds_t*)(uintptr_t)(lds0 + off + sdst[q]), 16, 0, 0);
        src[q] += step[q];
    };

    // ---- fragment addresses
    const int g16 = lane >> 4, q4 = (lane >> 2) & 3, p4 = lane & 3;
    const int kq = 8 * (g16 >> 1) + q4;
    int boff[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int bcol = ((wn * TN + j) * 32 + 16 * (g16 & 1) + 4 * p4) * 2;
        const int seg = (bcol >> 6) ^ bfs_swz<RB>(kq);
        boff[j] = A_BYTES + kq * RB + ((bcol & 63) | (seg << 6));
    }
    const int aoff = (wm * TM) * PARTS * 1024 + lane * 16;

    // Fragment reads are inline asm: hipcc waits vmcnt(0) in front of the ds_read_tr builtin whenever an LDS-DMA is pending (it treats the
    // intrinsic as a possible write to the DMA's LDS range), which drains the ring every chunk; and for its own ds_reads it waits lgkmcnt(0)
    // before the MFMAs of chunk c although they only need the reads issued one chunk earlier.  With asm reads the kernel counts for itself:
    // one lgkmcnt(0) per chunk, placed BEFORE the next chunk's reads are issued (cdna_hip_programming.md 5.7 form iii).
    struct Frags {
        bf16x8 a[TM][PARTS];
        s16x4 blo[TN][PARTS], bhi[TN][PARTS];
    };
    constexpr int NRA = TM * PARTS, NR = NRA + TN * PARTS * 2;   // LDS reads of one chunk's fragments
    // read r of a chunk: r < NRA the weight fragments (row tile, part), then per column tile (part, k half)
    auto read_one = [&](Frags& f, auto rc, unsigned aaddr, const unsigned (&baddr)[TN]) {
        constexpr int r = decltype(rc)::value;
        if constexpr (r < NRA) {
            f.a[r / PARTS][r % PARTS] = bfs_read_b128<r * 1024>(aaddr);
        } else {
            constexpr int e = r - NRA, j = e / (2 * PARTS), pp = (e % (2 * PARTS)) / 2, half = e & 1;
            if constexpr (half == 0) f.blo[j][pp] = bfs_read_tr<pp * B_PART>(baddr[j]);
            else f.bhi[j][pp] = bfs_read_tr<pp * B_PART + 4 * RB>(baddr[j]);
        }
    };
    auto frag_b = [](const Frags& f, int j, int pp) {
        const s16x4 lo = f.blo[j][pp], hi = f.bhi[j][pp];
        s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, v);
    };

    f32x16 acc[TM][TN];
    f32x16 accx[F16 ? TM : 1][F16 ? TN : 1];   // f16x3: the cross terms (scaled by 2^11)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[i][j][r] = 0.f;
                if (F16) accx[i][j][r] = 0.f;
            }
    // MFMA n of a chunk: term-major (every accumulator takes term t before any takes t + 1: TM * TN independent MFMAs between two that
    // share an accumulator); the terms in ascending magnitude
    constexpr int NT_ = PARTS == 2 ? 3 : 6, NM = NT_ * TM * TN;
    auto mfma_one = [&](const Frags& f, auto nc) {
        constexpr int n = decltype(nc)::value;
        constexpr int t = n / (TM * TN), i = (n % (TM * TN)) / TN, j = n % TN;
        constexpr int pa = PARTS == 2 ? (t == 0 ? 1 : 0) : (t == 0 ? 2 : (t == 2 || t == 3 ? 1 : 0));
        constexpr int pb = PARTS == 2 ? (t == 1 ? 1 : 0) : (t == 1 ? 2 : (t == 2 || t == 4 ? 1 : 0));
        if constexpr (F16) {
            const f16x8 a = __builtin_bit_cast(f16x8, f.a[i][pa]), b = __builtin_bit_cast(f16x8, frag_b(f, j, pb));
            if constexpr (t == 2) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i][j], 0, 0, 0);
            else accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, accx[i][j], 0, 0, 0);
        } else {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i][pa], frag_b(f, j, pb), acc[i][j], 0, 0, 0);
        }
    };

    // ---- ring of NSLOT slots of KSUB chunks.  Chunk c lives at LDS offset (c / KSUB % NSLOT) * SLOT + (c % KSUB) * CH.
    // Prologue: chunks 0 .. (NSLOT - 1) KSUB are staged.  Iteration c: [the fragments of chunk c, requested an iteration ago, have landed: lgkmcnt(0)];
    // when chunk c + 1 opens a new slot: {this wave's DMAs of that slot have landed (counted vmcnt), barrier: everybody's have, and
    // everybody is done reading the slot of chunk c}; then the MFMAs of chunk c with the fragment reads of chunk c + 1 dealt two per gap
    // over the first gaps and the DMAs of chunk c + 1 + (NSLOT - 1) KSUB (it goes to the slot just released) one per gap after them.
    // An MFMA holds the matrix pipe for 32 cycles during which the wave can issue a few other instructions; issued as one block in front
    // of the MFMAs, the 12 reads and 4 DMAs (~100 cycles each to issue) left the pipe idle for as long as the MFMAs keep it busy (28 % of
    // peak with one wave per SIMD); reads placed late in the chunk exposed their latency at the next lgkmcnt(0).  sched_barrier pins the
    // interleave (the reads are asm and stay where they are while the MFMAs would move).
    constexpr int AHEAD = (NSLOT - 1) * KSUB;   // chunks staged beyond c + 1 in the steady state
    const int npre = min(AHEAD + 1, nchunks);   // chunks 0 .. AHEAD: iteration c then stages chunk c + 1 + AHEAD
    for (int c = 0; c < npre; ++c)
#pragma unroll
        for (int q = 0; q < PERW; ++q) dma(q, (c / KSUB) * SLOT + (c % KSUB) * CH);
    if (npre == AHEAD + 1) bfs_wait_vm<(AHEAD + 1 - KSUB) * PERW>();   // slot 0 has landed
    else bfs_wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    Frags fa, fb;
    {
        unsigned baddr[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) baddr[j] = lds0 + boff[j];
        static_for<0, NR>([&](auto rc) { read_one(fa, rc, lds0 + aoff, baddr); });
    }
    int roff = KSUB > 1 ? CH : SLOT;   // LDS offset of chunk c + 1 (nchunks >= 2 whenever it is used)
    int rsub = KSUB > 1 ? 1 : 0;
    // LDS offset of chunk g = c + 1 + AHEAD (the next one to stage): it goes to the slot of chunks c + 1 - KSUB .. c, released at the barrier
    int dsub = (AHEAD + 1) % KSUB;
    int doff = (((AHEAD + 1) / KSUB) % NSLOT) * SLOT + dsub * CH;
    constexpr int DG = NM >= PERW + (NR + 1) / 2 ? PERW : 1;                     // gaps that carry DMAs (after the reads)
    constexpr int RPG_MIN = (NR + (NM - DG) - 1) / (NM - DG);
    constexpr int RPG = RPG_MIN > 2 ? RPG_MIN : 2;                                // fragment reads per MFMA gap
    constexpr int RG = (NR + RPG - 1) / RPG;                                      // gaps that carry reads
    constexpr int DPG = (PERW + DG - 1) / DG;                                     // DMAs per gap
    static_assert(RG + DG <= NM && DG * DPG >= PERW, "the chunk's MFMA gaps hold its reads and DMAs");
    auto step_chunk = [&](int c, Frags& cur, Frags& nxt, auto mainc, auto barc) {
        constexpr bool MAIN = decltype(mainc)::value;   // steady state: chunk c + 1 + AHEAD exists, no conditions in the body
        constexpr bool BAR = decltype(barc)::value;     // chunk c + 1 opens a new slot
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the fragments of chunk c (requested one chunk ago)
        if (BAR) {
            if (MAIN) bfs_wait_vm<(AHEAD - KSUB) * PERW>();
            else bfs_wait_vm<0>();
            __builtin_amdgcn_s_barrier();
        }
        __builtin_amdgcn_sched_barrier(0);
        const bool rd = MAIN || c + 1 < nchunks, st = MAIN || c + 1 + AHEAD < nchunks;
        unsigned baddr[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) baddr[j] = lds0 + roff + boff[j];
        const unsigned aaddr = lds0 + roff + aoff;
        static_for<0, NM>([&](auto nc) {
            constexpr int n = decltype(nc)::value;
            mfma_one(cur, nc);
            if constexpr (n < RG) {
                if (rd) static_for<n * RPG, (n + 1) * RPG < NR ? (n + 1) * RPG : NR>([&](auto rc) { read_one(nxt, rc, aaddr, baddr); });
            } else if constexpr (n < RG + DG) {
                if (st) static_for<(n - RG) * DPG, (n - RG + 1) * DPG < PERW ? (n - RG + 1) * DPG : PERW>([&](auto qc) { dma(decltype(qc)::value, doff); });
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        // advance the two running LDS offsets by one chunk
        if (KSUB == 1) {
            roff = roff + SLOT == NSLOT * SLOT ? 0 : roff + SLOT;
            doff = doff + SLOT == NSLOT * SLOT ? 0 : doff + SLOT;
        } else {
            if (++rsub == KSUB) {
                rsub = 0;
                roff += CH;
                if (roff == NSLOT * SLOT) roff = 0;
            } else {
                roff += CH;
            }
            if (++dsub == KSUB) {
                dsub = 0;
                doff += CH;
                if (doff == NSLOT * SLOT) doff = 0;
            } else {
                doff += CH;
            }
        }
    };
    // chunk c + 1 opens a new slot iff (c + 1) % KSUB == 0; the loop is unrolled by two (fragment register sets ping-pong), KSUB <= 2
    static_assert(KSUB == 1 || KSUB == 2, "one or two chunks per slot");
    using BarEven = std::integral_constant<bool, KSUB == 1>;   // c even
    using BarOdd = std::true_type;                             // c odd
    int c = 0;
    const int nmain = nchunks - 1 - AHEAD;   // c < nmain: chunk c + 1 + AHEAD exists
    for (; c + 2 <= nmain; c += 2) {
        step_chunk(c, fa, fb, std::true_type{}, BarEven{});
        step_chunk(c + 1, fb, fa, std::true_type{}, BarOdd{});
    }
    for (; c + 2 <= nchunks; c += 2) {
        step_chunk(c, fa, fb, std::false_type{}, BarEven{});
        step_chunk(c + 1, fb, fa, std::false_type{}, BarOdd{});
    }
    if (c < nchunks) step_chunk(c, fa, fb, std::false_type{}, BarEven{});
    __syncthreads();   // the epilogue re-uses the ring as its transpose tiles
    if constexpr (KG > 1) {
        // groups 1 .. KG - 1: accumulators -> their own ring, lane-linear 16-byte cells [wave][quad][lane]; group 0 adds them in group order
        constexpr int NQ = TM * TN * (F16 ? 8 : 4);
        static_assert(NW * NQ * 1024 <= NSLOT * SLOT, "a group's ring holds its waves' accumulators");
        f32x4v* part = reinterpret_cast<f32x4v*>(smem + (size_t)grp * (NSLOT * SLOT)) + wave * (NQ * 64) + lane;
        if (grp > 0) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        part[((i * TN + j) * (F16 ? 8 : 4) + q) * 64] = f32x4v{acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                        if (F16) part[((i * TN + j) * 8 + 4 + q) * 64] = f32x4v{accx[i][j][4 * q], accx[i][j][4 * q + 1], accx[i][j][4 * q + 2], accx[i][j][4 * q + 3]};
                    }
        }
        __syncthreads();
        if (grp > 0) return;
#pragma unroll 1
        for (int g = 1; g < KG; ++g) {
            const f32x4v* src_g = part + (size_t)g * (NSLOT * SLOT / 16);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4v a = src_g[((i * TN + j) * (F16 ? 8 : 4) + q) * 64];
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[i][j][4 * q + e] += a[e];
                        if (F16) {
                            const f32x4v x = src_g[((i * TN + j) * 8 + 4 + q) * 64];
#pragma unroll
                            for (int e = 0; e < 4; ++e) accx[i][j][4 * q + e] += x[e];
                        }
                    }
        }
        // (group 0's waves write their transpose tiles into group 0's ring only: the other groups' partial sums are not overwritten while a sibling reads)
    }
    if constexpr (SK) {
        // accumulators -> p.sk.ws[tile][group][wave][quad][lane] (16-byte cells: every store / load is 1 KB per wave, coalesced).  The partial sums cross
        // XCDs (one L2 each).  Agent-scope fences around the count cost 25 us per launch (a release writes the XCD's whole L2 back, an acquire invalidates
        // it: 37.9 us against the unsplit kernel's 11.1); instead every access to the scratch is itself agent-coherent (sc1: stores write through, loads
        // are not served from this XCD's L2), like the counter's atomic.
        constexpr int NQ = TM * TN * (F16 ? 8 : 4);
        const int ks = kp.ksplit;
        f32x4v* wsp = reinterpret_cast<f32x4v*>(p.sk.ws) + ((size_t)t * ks * NW + wave) * (NQ * 64) + lane;
        auto st = [](f32x4v* dst, const f32x4v& v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(v) : "memory"); };
        {
            f32x4v* mine = wsp + (size_t)kg * NW * (NQ * 64);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        st(mine + ((i * TN + j) * (F16 ? 8 : 4) + q) * 64, f32x4v{acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]});
                        if (F16) st(mine + ((i * TN + j) * 8 + 4 + q) * 64, f32x4v{accx[i][j][4 * q], accx[i][j][4 * q + 1], accx[i][j][4 * q + 2], accx[i][j][4 * q + 3]});
                    }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this thread's partial sums are in memory ...
        __syncthreads();                                    // ... every thread's are ...
        int* arrived = reinterpret_cast<int*>(smem);
        if (tid == 0) *arrived = (int)__hip_atomic_fetch_add(p.sk.counters + t, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ... before the count
        __syncthreads();
        const bool last = *arrived == ks - 1;
        if (!last) return;
        if (tid == 0) __hip_atomic_store(p.sk.counters + t, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // zero again for the next launch
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    acc[i][j][r] = 0.f;
                    if (F16) accx[i][j][r] = 0.f;
                }
        if constexpr (TM * TN == 1) {
        // four groups' partial sums are requested before the first of them is added (two memory round trips for ks = 8 instead of eight), added in group order
        constexpr int kBatch = 4;
#pragma unroll 1
        for (int g0 = 0; g0 < ks; g0 += kBatch) {
            f32x4v v[kBatch][NQ];
#pragma unroll
            for (int gi = 0; gi < kBatch; ++gi)
                if (g0 + gi < ks) {
                    const f32x4v* src_g = wsp + (size_t)(g0 + gi) * NW * (NQ * 64);
#pragma unroll
                    for (int q = 0; q < NQ; ++q) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v[gi][q]) : "v"(src_g + q * 64) : "memory");
                }
#pragma unroll
            for (int gi = 0; gi < kBatch; ++gi)
                if (g0 + gi < ks) {
                    // (waits for everything requested: the first wait of a batch is the only one that waits)
                    if constexpr (NQ == 8)
                        asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[gi][0]), "+v"(v[gi][1]), "+v"(v[gi][2]), "+v"(v[gi][3]), "+v"(v[gi][4]), "+v"(v[gi][5]), "+v"(v[gi][6]), "+v"(v[gi][7]));
                    else
                        asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[gi][0]), "+v"(v[gi][1]), "+v"(v[gi][2]), "+v"(v[gi][3]));
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            acc[0][0][4 * q + e] += v[gi][q][e];
                            if (F16) accx[0][0][4 * q + e] += v[gi][4 + q][e];
                        }
                }
        }
        } else {
            static_assert(TM * TN == 1, "the K split serves the small-grid configuration (one 32 x 32 tile per wave)");
        }
        __syncthreads();   // (wave 0's transpose tile starts at smem: `arrived` has been read by everybody)
    }

    // ---- epilogue: each wave passes its 32 x 32 tiles through a private LDS tile and leaves with 16-byte stores (8 lanes = 128 bytes of a row)
    float* tile = reinterpret_cast<float*>(smem) + wave * (32 * 36);
    const int lcol = lane & 31, lh = lane >> 5;
    const int lrow = lane >> 3, c4 = (lane & 7) * 4;
    // Interior tiles whose output formats are the kernel's own (round 5): no per-lane conditions and no loops around the stores, the global reads of sub-tile
    // s + 1 requested before the first store of sub-tile s.  In the generic path below hipcc cannot count the stores of split_store4's loop over the parts,
    // so every use of a loaded value behind them waits s_waitcnt vmcnt(0): three exposed store acknowledgements per 32 x 32 sub-tile, on top of the loads
    // queued behind the previous sub-tile's stores.  Same arithmetic in the same order: same bits.
    const bool ys_own = !p.Ys.parts || (p.Ys.parts == PARTS && (p.Ys.f16 != 0) == F16);
    // (y_rows: the default 0x7fffffff = every row; a bound inside the matrix must fall on a 32-row sub-tile for the uniform toY test below)
    if (m0 + MT <= M && n0 + NT <= N && ys_own && (p.y_rows >= M || (p.y_rows & 31) == 0) && (p.ys_row0 & 31) == 0 && (!p.mask || kp.mask_shift == 0)) {
        constexpr int NS = TM * TN;
        unsigned nclamp = 0;   // f16 pair: values this lane's split clamped to +-65504 (counted into p.Ys.sat once, behind the last store)
        f32x4v rr[2][4];
        float brow[2][4];
        unsigned mk[2];
        auto loads = [&](int s, int buf) {
            const int i = s / TN, j = s % TN;
            const int n = n0 + (wn * TN + j) * 32 + c4, mb = m0 + (wm * TM + i) * 32 + lrow;
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                brow[buf][ps] = p.bias ? p.bias[mb + ps * 8] : 0.f;
                if (p.R) rr[buf][ps] = *reinterpret_cast<const f32x4v*>(p.R + (int64_t)(mb + ps * 8) * p.ldr + n);
            }
            mk[buf] = p.mask ? *reinterpret_cast<const unsigned*>(p.mask + n) : 0x01010101u;   // four keep flags (n is a multiple of 4)
        };
        loads(0, 0);
        static_for<0, NS>([&](auto sc) {
            constexpr int s = decltype(sc)::value, i = s / TN, j = s % TN, buf = s & 1;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                tile[((r & 3) + 8 * (r >> 2) + 4 * lh) * 36 + lcol] = F16 ? acc[i][j][r] + accx[i][j][r] * (1.0f / kF16LoScale) : acc[i][j][r];
            if constexpr (s + 1 < NS) loads(s + 1, buf ^ 1);
            const int n = n0 + (wn * TN + j) * 32 + c4, mb = m0 + (wm * TM + i) * 32 + lrow;
            const int mt0 = m0 + (wm * TM + i) * 32;                       // (uniform) first row of the sub-tile
            const bool toY = p.Y && mt0 < p.y_rows, toYs = p.Ys.parts && mt0 >= p.ys_row0;
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                const f32x4v av = *reinterpret_cast<const f32x4v*>(tile + (ps * 8 + lrow) * 36 + c4);
                const int m = mb + ps * 8;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float x = av[e] + brow[buf][ps];
                    if (p.act == ACT_RELU) x = fmaxf(x, 0.f);
                    else if (p.act == ACT_GELU) x = 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
                    else if (p.act == ACT_TANH) x = tanhf(x);
                    x *= p.alpha;
                    if (p.R) x += rr[buf][ps][e];
                    x *= p.beta;
                    v[e] = ((mk[buf] >> (8 * e)) & 0xFFu) ? x : 0.f;
                }
                if (toY) *reinterpret_cast<f32x4v*>(p.Y + (int64_t)m * p.ldy + n) = f32x4v{v[0], v[1], v[2], v[3]};
                if (toYs) {
                    const int64_t off = (int64_t)m * p.Ys.ld + n;
                    if constexpr (F16) {
                        typedef _Float16 f16x4v __attribute__((ext_vector_type(4)));
                        f16x4v h, l;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {   // split_store4's f16 pair: finite values beyond f16's range saturate, NaN / infinity pass through
                            const float c = (v[e] != v[e] || fabsf(v[e]) == __builtin_inff()) ? v[e] : fminf(fmaxf(v[e], -65504.f), 65504.f);
                            nclamp += (c != v[e] && v[e] == v[e]) ? 1u : 0u;
                            h[e] = (_Float16)c;
                            l[e] = (_Float16)((c - (float)h[e]) * kF16LoScale);
                        }
                        *reinterpret_cast<f16x4v*>(static_cast<_Float16*>(p.Ys.p) + off) = h;
                        *reinterpret_cast<f16x4v*>(static_cast<_Float16*>(p.Ys.p) + p.Ys.pstride + off) = l;
                    } else {
                        typedef __bf16 b16x4v __attribute__((ext_vector_type(4)));
                        float res[4] = {v[0], v[1], v[2], v[3]};
#pragma unroll
                        for (int pp = 0; pp < PARTS; ++pp) {
                            b16x4v h;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                h[e] = (__bf16)res[e];
                                res[e] -= (float)h[e];
                            }
                            *reinterpret_cast<b16x4v*>(static_cast<__bf16*>(p.Ys.p) + (int64_t)pp * p.Ys.pstride + off) = h;
                        }
                    }
                }
            }
        });
        if constexpr (F16) {
            if (p.Ys.parts && p.Ys.sat && nclamp) atomicAdd(p.Ys.sat, (unsigned long long)nclamp);
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                tile[((r & 3) + 8 * (r >> 2) + 4 * lh) * 36 + lcol] = F16 ? acc[i][j][r] + accx[i][j][r] * (1.0f / kF16LoScale) : acc[i][j][r];
            const int n = n0 + (wn * TN + j) * 32 + c4;
            const int mb = m0 + (wm * TM + i) * 32 + lrow;
            f32x4v av[4], rr[4];
            float brow[4];
            // everything read from global memory before the first store (loads and stores share the in-order vmcnt queue)
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                av[ps] = *reinterpret_cast<const f32x4v*>(tile + (ps * 8 + lrow) * 36 + c4);
                const int m = min(mb + ps * 8, M - 1);
                brow[ps] = p.bias ? p.bias[m] : 0.f;
                if (p.R) rr[ps] = *reinterpret_cast<const f32x4v*>(p.R + (int64_t)m * p.ldr + min(n, N - 4));
            }
            bool keep[4] = {true, true, true, true};
            if (p.mask) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int oc = min(n + e, N - 1);
                    keep[e] = p.mask[kp.mask_shift >= 0 ? (oc >> kp.mask_shift) : (oc / p.mask_div)] != 0;
                }
            }
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                const int m = mb + ps * 8;
                if (m >= M || n >= N) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float x = av[ps][e] + brow[ps];
                    if (p.act == ACT_RELU) x = fmaxf(x, 0.f);
                    else if (p.act == ACT_GELU) x = 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
                    else if (p.act == ACT_TANH) x = tanhf(x);
                    x *= p.alpha;
                    if (p.R) x += rr[ps][e];
                    x *= p.beta;
                    v[e] = keep[e] ? x : 0.f;
                }
                if (p.Y && m < p.y_rows) *reinterpret_cast<f32x4v*>(p.Y + (int64_t)m * p.ldy + n) = f32x4v{v[0], v[1], v[2], v[3]};
                if (p.Ys.parts && m >= p.ys_row0) split_store4(p.Ys, (int64_t)m * p.Ys.ld + n, v);
            }
        }
}

// ---- f32 plane -> bf16 parts (for producers that do not emit the parts themselves) ---------------------------------------------------
__global__ __launch_bounds__(256) void k_split_planes(Plane in, SplitPlanes out) {
    const int64_t nq = (int64_t)in.C * (in.ld >> 2);
    const int lq = in.ld >> 2;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < nq; q += (int64_t)gridDim.x * 256) {
        const int c = (int)(q / lq), j = (int)(q - (int64_t)c * lq) * 4;
        if (j >= out.ld) continue;
        const f32x4v v = *reinterpret_cast<const f32x4v*>(in.p + (int64_t)c * in.ld + j);
        const float res[4] = {v[0], v[1], v[2], v[3]};
        split_store4(out, (int64_t)c * out.ld + j, res);
    }
}

void split_planes(Plane in, SplitPlanes out, hipStream_t stream) {
    SBV2_REQUIRE(in.C == out.C && out.parts >= 1 && out.parts <= 3 && (!out.f16 || out.parts == 2) && (in.ld & 3) == 0 && (out.ld & 3) == 0 && out.ld <= in.ld + 63,
                 "split_planes: shape mismatch");
    const int64_t nq = (int64_t)in.C * (in.ld >> 2);
    const int grid = (int)std::min<int64_t>((nq + 255) / 256, 2048);
    hipLaunchKernelGGL(k_split_planes, dim3(grid), dim3(256), 0, stream, in, out);
    HIP_CHECK(hipGetLastError());
}

// ---- host side -------------------------------------------------------------------------------------------------------------------------
bool gemm_bfs_usable(const GemmBfsParams& p) {
    return p.W.w && (p.W.parts == 2 || p.W.parts == 3) && p.X.p && p.X.parts == p.W.parts && p.X.f16 == p.W.f16 && (!p.W.f16 || p.W.parts == 2) &&
           (p.K & 15) == 0 && p.K >= 16 && (p.N & 3) == 0 &&
           (p.X.ld & 7) == 0 && (!p.Y || (p.ldy & 3) == 0) && (!p.R || (p.ldr & 3) == 0) && (!p.Ys.parts || (p.Ys.ld & 3) == 0) && p.N >= 4;
}

template <int PARTS, int TM, int TN, int WM, int WN, int KSUB, int NSLOT, bool F16 = false, bool SK = false, int KG = 1>
static void launch_bfs_cfg(BfsKernelParams kp, hipStream_t stream, int ksplit = 1) {
    constexpr int MT = 32 * TM * WM, NT = 32 * TN * WN;
    constexpr int SLOT = KSUB * ((MT / 32) * PARTS * 1024 + PARTS * 16 * NT * 2);
    const GemmBfsParams& p = kp.p;
    kp.gm = (p.M + MT - 1) / MT;
    kp.gn = (p.N + NT - 1) / NT;
    kp.ksplit = SK ? ksplit : 1;
    kp.total = kp.gm * kp.gn * kp.ksplit;
    if (SK) {
        constexpr size_t kPerWg = (size_t)WM * WN * TM * TN * (F16 ? 8 : 4) * 1024;   // bytes of partial sums per workgroup
        SBV2_REQUIRE((p.K >> 4) % ksplit == 0 && p.sk.ws && p.sk.counters && kp.gm * kp.gn <= p.sk.ncounters && (size_t)kp.total * kPerWg <= p.sk.ws_bytes,
                     "gemm_bfs: K split without room for it");
    }
    const size_t lds = std::max<size_t>((size_t)KG * NSLOT * SLOT, (size_t)WM * WN * 32 * 36 * sizeof(float));
    SBV2_REQUIRE(KG == 1 || (p.K >> 4) % KG == 0, "gemm_bfs: K does not split into equal groups");
    auto kern = gemm_bfs_kernel<PARTS, TM, TN, WM, WN, KSUB, NSLOT, F16, SK, KG>;
    static std::atomic<uint64_t> lds_allowed{0};
    allow_full_lds(reinterpret_cast<const void*>(kern), lds_allowed);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const bool prof = conv_prof_active();
    if (prof) {
        HIP_CHECK(hipEventCreate(&e0));
        HIP_CHECK(hipEventCreate(&e1));
        HIP_CHECK(hipEventRecord(e0, stream));
    }
    hipLaunchKernelGGL(kern, dim3(round_up(kp.total, 8)), dim3(64 * WM * WN * KG), lds, stream, kp);
    HIP_CHECK(hipGetLastError());
    if (prof) {
        HIP_CHECK(hipEventRecord(e1, stream));
        conv_prof_add(F16 ? 27 : (PARTS == 2 ? 24 : 25), 2.0 * p.M * (double)p.N * p.K, e0, e1);
    }
}

void launch_gemm_bfs(const GemmBfsParams& p, hipStream_t stream) {
    SBV2_REQUIRE(gemm_bfs_usable(p), "gemm_bfs: operands do not fit the split-bf16 kernel");
    SBV2_REQUIRE(p.X.C >= p.K && p.W.K == p.K && p.W.M == p.M, "gemm_bfs: shape mismatch");
    if (p.M <= 0 || p.N <= 0) return;
    BfsKernelParams kp;
    kp.p = p;
    kp.mask_shift = -1;
    if (p.mask && p.mask_div > 0 && (p.mask_div & (p.mask_div - 1)) == 0) {
        int s = 0;
        while ((1 << s) < p.mask_div) ++s;
        kp.mask_shift = s;
    }
    auto blocks = [&](int mt, int nt) { return (int64_t)((p.M + mt - 1) / mt) * ((p.N + nt - 1) / nt); };
    const bool big = blocks(128, 128) >= 128;
    // Ring depth = bytes in flight per CU: a chunk's DMAs take ~1 us to land under load, a chunk's MFMAs 0.2 - 0.4 us.  Grids that leave one
    // workgroup per CU take the whole LDS (144 KB); larger grids run two workgroups per CU with 80 / 72 KB each.
    const int ncu = device_cu_count();   // (256 on MI355X)
    const bool lone = blocks(128, 128) <= ncu;
    const bool k32 = (p.K & 31) == 0;
    if (p.W.f16) {
        // f16x3 (the default format): small grids (a single utterance) on 64 x 64 tiles with an 8-slot ring; everything else on 64 x 128 tiles with a 4-slot
        // ring = 48 KB and 99 registers: THREE workgroups per CU instead of two 128 x 128 ones (80 KB, 222 registers): gemm_bfs 8.07 -> 7.13 ms per step in
        // round 4.  Same per-element summation order on every tile.  (Rounds 3-4 measured the other tile / ring shapes behind knobs: 128 x 128 everywhere,
        // 64 x 128 on 3 / 6 slots, 128 x 64, two chunks per barrier, 64 x 64 for the 1024-row products, 4 / 12 / 16 slots for small grids: all slower
        // or equal; profiles/HISTORY.md.)
        if (big) {
            // (Round 6 measured two more shapes here, profiles/r06g_bfs_batch_ksplit_probe.txt: the cross-workgroup K split, two workgroups per 64 x 128 tile of
            // DeBERTa's K = 4096 product, 272 -> 544 workgroups on 768 slots: 90.2 -> 86.9-88.0 us, and 26.8 -> 35.9 at K = 1024; and 128 x 128 tiles on 8 waves
            // of 32 x 64, two workgroups per CU: 26.9 / 63.2 / 86.5 / 92.3 -> 28.7 / 58.0 / 84.3 / 92.1 us on the four DeBERTa shapes.  Neither the slot count nor
            // 1.5x fewer staged bytes per FLOP moves these launches: ~0.35 us per 16-deep chunk and workgroup at every occupancy.  Not kept.)
            launch_bfs_cfg<2, 1, 2, 2, 2, 1, 4, true>(kp, stream);
        } else {
            // Small grids with a long K loop and scratch from the caller (DeBERTa's FFN down projection in a single-utterance call, K = 4096: 30.1 -> 12-18 us):
            // up to 8 workgroups per tile, as many as keep the grid within one workgroup per CU and every group at >= 32 chunks: handing over and adding the
            // partial sums costs 3 - 6 us, what 30 - 60 chunks take (K = 1024 split 2 - 8 ways measured 13.3 - 17.4 us against 11.1 - 12.3 unsplit)
            const int64_t tiles = blocks(64, 64);
            const int nch = p.K >> 4;
            int ks = 1;
            if (ksplit_enabled() && p.sk.ws && p.sk.counters && tiles <= p.sk.ncounters && nch >= 128)
                while (ks < 8 && tiles * (ks * 2) <= ncu && nch % (ks * 2) == 0 && nch / (ks * 2) >= 32 && (size_t)tiles * (ks * 2) * 32768 <= p.sk.ws_bytes) ks *= 2;
            // ... and the shorter K loops on 32 x 32 tiles whose four waves each take a quarter of K (the kernel's KG): half the bytes per CU on four times the
            // CUs, as long as that is still one workgroup per CU (DeBERTa's 1024 x 1024 product at 68 columns: 96 workgroups, 11.0 -> 8.1 us; the 3072- and
            // 4096-row products would be 288 / 384 workgroups and measured 16.3 us against 11.6 - 12.1)
            const bool kg4 = ksplit_enabled() && ks == 1 && nch % 4 == 0 && nch >= 32 && blocks(32, 32) <= ncu;
            if (ks > 1) launch_bfs_cfg<2, 1, 1, 2, 2, 1, 8, true, true>(kp, stream, ks);
            else if (kg4) launch_bfs_cfg<2, 1, 1, 1, 1, 1, 4, true, false, 4>(kp, stream);
            else launch_bfs_cfg<2, 1, 1, 2, 2, 1, 8, true>(kp, stream);
        }
    } else if (p.W.parts == 2) {   // bf16x3 (opt-in)
        if (!big) launch_bfs_cfg<2, 1, 1, 2, 2, 1, 8>(kp, stream);
        else if (k32 && lone) launch_bfs_cfg<2, 2, 2, 2, 2, 2, 4>(kp, stream);
        else launch_bfs_cfg<2, 2, 2, 2, 2, 1, 5>(kp, stream);
    } else {                       // bf16x6 (the f16x3 fallback with bf16's exponent range)
        if (!big) launch_bfs_cfg<3, 1, 1, 2, 2, 1, 6>(kp, stream);
        else if (lone) launch_bfs_cfg<3, 2, 2, 2, 2, 1, 6>(kp, stream);
        else launch_bfs_cfg<3, 2, 2, 2, 2, 1, 3>(kp, stream);
    }
}

}  // namespace sbv2
