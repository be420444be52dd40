// Grouped implicit-GEMM 1-D convolution on the gfx950 f32 matrix cores.
//
//   C[m][n] (+)= epilogue( sum_tap sum_k A_tap[k][m] * pre(B[k][n + shift_tap]) )
//
// Both operands are k-major ("planes": [K][ld], the m / n axis contiguous), so the same kernel serves
//   * Conv1d / dilated Conv1d (A = weights packed [tap][Cin][Cout], B = activation plane, shifts = taps),
//   * ConvTranspose1d as polyphase convolutions (row m = phase * Cout + co, strided output columns),
//   * Linear / 1x1 conv (one tap), with either operand in the "weight" role (bias per row or per column),
//   * the attention products Q^T K, P V, position terms (A and B both activation planes, grouped by
//     (utterance, head) through GemmGroup descriptors).
//
// What the reference does here: nothing — these are the Conv/MatMul/Gemm nodes ONNX Runtime executes
// inside `session.run` (crates/sbv2_core/src/model.rs:91, bert.rs:11).
//
// Machine mapping (MI355X_MICROARCH.md / cdna_hip_programming.md §3 "FP32-input MFMA"):
//   v_mfma_f32_32x32x2_f32 (or 16x16x4 for 16-row problems): exact f32, 64 FLOP/clk/SIMD.
//   A lane layout  A[i = lane & 31][k = lane >> 5], B[k = lane >> 5][j = lane & 31]  -> with k-major LDS
//   tiles every operand fetch is a conflict-free ds_read_b32 of 32 consecutive dwords per half-wave.
//   One workgroup = 4 waves computes an MT x NT tile; per input-channel chunk (KC rows) the B window
//   [KC][NT + tap span] is staged ONCE in LDS (leaky-ReLU fused into the staging) and re-used by every tap;
//   the per-tap weight tile [KC][MT] is double-buffered; global loads for step s+1 are issued before the
//   MFMA block of step s and written to LDS after it (one barrier per step).
#include <atomic>

#include "common.h"

namespace sbv2 {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kThreads = 256;
constexpr int kMaxSpan = 64;  // max (max shift - floor4(min shift)), host-checked

struct KernelParams {
    ConvParams p;
    int xw;       // LDS pitch of the B window (floats)
    int wshift0;  // floor4(min shift)
    int mask_shift;
};

template <int MF>
struct Mfma;
template <>
struct Mfma<32> {
    using acc_t = f32x16;
    static constexpr int KS = 2, NACC = 16;
    static __device__ __forceinline__ acc_t run(float a, float b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ int row(int lane, int r) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }
};
template <>
struct Mfma<16> {
    using acc_t = f32x4;
    static constexpr int KS = 4, NACC = 4;
    static __device__ __forceinline__ acc_t run(float a, float b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ int row(int lane, int r) { return (lane >> 4) * 4 + r; }
};

__device__ __forceinline__ float4 load4_guard(const float* src, int j, int lo, int hi) {
    // src points at column j; columns outside [lo, hi) read as zero.  src is 16-byte aligned when j % 4 == 0.
    if (j >= lo && j + 3 < hi) return *reinterpret_cast<const float4*>(src);
    float4 v;
    v.x = (j >= lo && j < hi) ? src[0] : 0.f;
    v.y = (j + 1 >= lo && j + 1 < hi) ? src[1] : 0.f;
    v.z = (j + 2 >= lo && j + 2 < hi) ? src[2] : 0.f;
    v.w = (j + 3 >= lo && j + 3 < hi) ? src[3] : 0.f;
    return v;
}

// RING = true (1x1 products with K % 16 == 0, no shift, no groups): both operand tiles of a chunk are row-contiguous f32 images, so they go
// L2 -> LDS by LDS-DMA (global_load_lds_dwordx4) into a kRingSlots-deep ring, kRingSlots - 1 chunks ahead, with no staging register and
// no staging VALU.  An ablation of the register-staged loop (4096 x 2112 x 1024: 247 us) showed the staging path alone (loads, zero-fill
// selects, LDS stores) at 111 us and the MFMAs alone at 114 us (= the 155 TFLOP/s the f32 pipe sustains on random data), adding up instead
// of overlapping, and 64 x 128 tiles pulling 7.5 TB/s from L2 at that rate: the ring removes the staging instructions, and having no
// staging registers makes 128 x 128 tiles (half the L2 bytes per FLOP) affordable.  Same fragments, same MFMA order, same epilogue: same bits.
constexpr int kRingSlots = 4;
template <int N>
__device__ __forceinline__ void ring_wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
typedef __attribute__((address_space(3))) void ring_lds_t;
typedef const __attribute__((address_space(1))) void ring_gbl_t;

template <int MF, int TM, int TN, int WM, int WN, int KC, bool RING = false>
__global__ __launch_bounds__(kThreads) void conv_gemm_kernel(const KernelParams kp) {
    using MM = Mfma<MF>;
    constexpr int MT = MF * TM * WM;
    constexpr int NT = MF * TN * WN;
    constexpr int F4W = KC * MT / 4;
    constexpr int NW = (F4W + kThreads - 1) / kThreads;
    constexpr int NX = (KC * (NT + kMaxSpan + (MF == 16 ? 32 : 0)) / 4 + kThreads - 1) / kThreads;  // MF16 pads the pitch to 16 mod 32
    static_assert(WM * WN == 4, "4 waves per workgroup");

    const ConvParams& p = kp.p;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int XW = RING ? MF * TN * WN : kp.xw;
    float* ws = smem;                 // [2][KC][MT]          (RING: slot s = [KC][MT] then [KC][NT] at smem + s * KC * (MT + NT))
    float* xs = smem + 2 * KC * MT;   // [2][KC][XW]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm0 = (wave / WN) * (MF * TM);
    const int wn0 = (wave % WN) * (MF * TN);

    int M = p.M, N = p.N, K = p.K, nb = p.nb;
    const float* Ag = p.A;
    const float* Bg = p.B;
    float* Cg = p.C;
    const float* Rg = p.R;
    if (p.groups) {
        const GemmGroup g = p.groups[blockIdx.z];
        M = g.M; N = g.N; K = g.K; nb = g.nb;
        Ag += g.a_off; Bg += g.b_off; Cg += g.c_off;
        if (Rg) Rg += g.r_off;
    }
    const int m0 = blockIdx.y * MT;
    const int n0 = blockIdx.x * NT;
    if (m0 >= M || n0 >= N) return;

    const int ntaps = p.ntaps;
    const int nchunks = (K + KC - 1) / KC;
    const int wstart = n0 + kp.wshift0;
    const int xw4 = XW >> 2;
    const int f4x = KC * xw4;
    const float slope = p.pre_slope;

    typename MM::acc_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < MM::NACC; ++r) acc[i][j][r] = 0.f;

    typedef float f32x4r __attribute__((ext_vector_type(4)));
    f32x4r rw[NW], rx[NX];     // staging registers of the next step / chunk
    f32x4r rw2[NW], rx2[NX];   // second set: 1x1 products prefetch two chunks ahead (see the GEMM loop below)

    // Staging loads are BRANCH-FREE (clamped always-valid addresses; out-of-range elements are zeroed when the registers
    // are written to LDS) so that they stay in flight across the MFMA block: any control flow around a load makes hipcc
    // wait vmcnt(0) at the join.  A float4 at column j < n is always inside the row (pitches are multiples of 4).
    auto load_w = [&](auto& regs, int tap, int k0) {
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int idx = min(tid + i * kThreads, F4W - 1);
            const int k = min(k0 + idx / (MT / 4), K - 1);
            const int m = m0 + (idx % (MT / 4)) * 4;
            const int mm = m < M ? m : 0;
            regs[i] = *reinterpret_cast<const f32x4r*>(Ag + (int64_t)tap * p.a_tap_stride + (int64_t)k * p.lda + mm);
        }
    };
    auto store_w = [&](auto& regs, int buf, int k0) {
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int idx = tid + i * kThreads;
            if (idx < F4W) {
                const int k = k0 + idx / (MT / 4);
                const int m = m0 + (idx % (MT / 4)) * 4;
                f32x4r v = regs[i];
                const bool kin = k < K;
                v[0] = (kin && m < M) ? v[0] : 0.f;
                v[1] = (kin && m + 1 < M) ? v[1] : 0.f;
                v[2] = (kin && m + 2 < M) ? v[2] : 0.f;
                v[3] = (kin && m + 3 < M) ? v[3] : 0.f;
                *reinterpret_cast<f32x4r*>(ws + buf * (KC * MT) + idx * 4) = v;
            }
        }
    };
    auto load_x = [&](auto& regs, int k0) {
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int idx = min(tid + i * kThreads, f4x - 1);
            const int kr = idx / xw4;
            const int j = wstart + (idx - kr * xw4) * 4;
            const int k = min(k0 + kr, K - 1);
            const int jj = (j >= 0 && j < nb) ? j : 0;
            regs[i] = *reinterpret_cast<const f32x4r*>(Bg + (int64_t)k * p.ldb + jj);
        }
    };
    // zero fill + the fused input activation happen here, AFTER the MFMA block
    auto store_x = [&](auto& regs, int buf, int k0) {
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int idx = tid + i * kThreads;
            if (idx < f4x) {
                const int kr = idx / xw4;
                const int j = wstart + (idx - kr * xw4) * 4;
                const bool kin = (k0 + kr < K) && j >= 0;
                f32x4r v = regs[i];
                v[0] = (kin && j < nb) ? v[0] : 0.f;
                v[1] = (kin && j + 1 < nb) ? v[1] : 0.f;
                v[2] = (kin && j + 2 < nb) ? v[2] : 0.f;
                v[3] = (kin && j + 3 < nb) ? v[3] : 0.f;
                if (slope != 1.0f) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] >= 0.f ? v[e] : v[e] * slope;
                }
                *reinterpret_cast<f32x4r*>(xs + buf * (KC * XW) + idx * 4) = v;
            }
        }
    };

    const int lrow = lane / MF;  // k row inside one MFMA k-step
    const int lcol = lane % MF;
    auto compute = [&](int wbuf, int xbuf, int shift) {
        const float* wsb = ws + wbuf * (KC * MT) + wm0 + lcol;
        const float* xsb = xs + xbuf * (KC * XW) + wn0 + lcol + (shift - kp.wshift0);
#pragma unroll
        for (int kk = 0; kk < KC / MM::KS; ++kk) {
            const int kr = kk * MM::KS + lrow;
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = wsb[kr * MT + i * MF];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = xsb[kr * XW + j * MF];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = MM::run(a[i], b[j], acc[i][j]);
        }
    };

    if constexpr (RING) {
        constexpr int SLOT = KC * (MT + NT);                 // floats per ring slot
        constexpr int LA = MT / 4, LB = NT / 4;              // lanes per row of the A / B tile
        constexpr int GA = KC * LA / 64, GB = KC * LB / 64;  // DMA instructions per chunk
        static_assert((GA + GB) % 4 == 0 && GA >= 1 && GB >= 1, "ring: the DMAs of a chunk are dealt evenly over the four waves");
        constexpr int PERW = (GA + GB) / 4;
        // DMA g of a chunk (g < GA: rows of the weight tile, else rows of the activation tile) belongs to wave g % 4
        const float* src[PERW];
        int64_t step[PERW];
        int dst[PERW];
#pragma unroll
        for (int q = 0; q < PERW; ++q) {
            const int gi = wave + 4 * q;
            if (gi < GA) {
                const int row = gi * (64 / LA) + lane / LA, col = m0 + (lane % LA) * 4;
                src[q] = Ag + (int64_t)row * p.lda + (col < M ? col : 0);   // rows / columns outside the problem only feed outputs never stored
                step[q] = (int64_t)KC * p.lda;
                dst[q] = gi * 256;
            } else {
                const int gb = gi - GA;
                const int row = gb * (64 / LB) + lane / LB, col = n0 + (lane % LB) * 4;
                src[q] = Bg + (int64_t)row * p.ldb + (col < N ? col : 0);
                step[q] = (int64_t)KC * p.ldb;
                dst[q] = KC * MT + gb * 256;
            }
        }
        auto stage = [&](int c) {
            float* slot = smem + (c & (kRingSlots - 1)) * SLOT;
#pragma unroll
            for (int q = 0; q < PERW; ++q)
                __builtin_amdgcn_global_load_lds((ring_gbl_t*)(src[q] + c * step[q]), (ring_lds_t*)(slot + dst[q]), 16, 0, 0);
        };
        auto compute_ring = [&](int c) {
            const float* wsb = smem + (c & (kRingSlots - 1)) * SLOT + wm0 + (lane % MF);
            const float* xsb = wsb - wm0 + KC * MT + wn0;
            const int lrow_ = lane / MF;
            // every fragment of the chunk is requested before the first MFMA (left to hipcc each k-step was "ds_read, s_waitcnt lgkmcnt(0), two
            // MFMAs": one exposed LDS round trip per 128 cycles of matrix work); the register-staged loop above is better off without this
            // (its staging registers compete: measured 253 -> 289 us at 4096 x 2112 x 1024)
            constexpr int NK = KC / MM::KS;
            float a[NK][TM], b[NK][TN];
#pragma unroll
            for (int kk = 0; kk < NK; ++kk) {
                const int kr = kk * MM::KS + lrow_;
#pragma unroll
                for (int i = 0; i < TM; ++i) a[kk][i] = wsb[kr * MT + i * MF];
#pragma unroll
                for (int j = 0; j < TN; ++j) b[kk][j] = xsb[kr * NT + j * MF];
            }
            __builtin_amdgcn_sched_barrier(0);
            if (slope != 1.0f) {
#pragma unroll
                for (int kk = 0; kk < NK; ++kk)
#pragma unroll
                    for (int j = 0; j < TN; ++j) b[kk][j] = b[kk][j] >= 0.f ? b[kk][j] : b[kk][j] * slope;
            }
#pragma unroll
            for (int kk = 0; kk < NK; ++kk)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = MM::run(a[kk][i], b[kk][j], acc[i][j]);
        };
        for (int c = 0; c < min(kRingSlots - 1, nchunks); ++c) stage(c);
        for (int c = 0; c < nchunks; ++c) {
            // this wave's DMAs of chunk c have landed (loads retire in order; at the tail fewer chunks are behind it: wait for all)
            if (c + kRingSlots - 2 < nchunks) ring_wait_vm<(kRingSlots - 2) * PERW>();
            else ring_wait_vm<0>();
            __builtin_amdgcn_s_barrier();   // ... and everybody else's; every wave is also done with chunk c - 1, whose slot is refilled now
            if (c + kRingSlots - 1 < nchunks) stage(c + kRingSlots - 1);
            compute_ring(c);
        }
        __syncthreads();   // the epilogue re-uses the ring as its transpose tiles
    } else {
    load_x(rx, 0);
    load_w(rw, 0, 0);
    store_x(rx, 0, 0);
    store_w(rw, 0, 0);

    if (ntaps == 1 && nchunks >= 3) {
        // 1x1 products (Linear layers, attention): every step is a new chunk, so a one-step prefetch only hides one MFMA block
        // (~1024 cycles) of the ~2 us global latency.  Two register sets keep the loads of chunks c+1 and c+2 in flight.
        const int sh0 = p.shift[0];
        load_x(rx, KC);
        load_w(rw, 0, KC);
        __syncthreads();
        for (int c = 0; c < nchunks; c += 2) {
            load_x(rx2, min(c + 2, nchunks - 1) * KC);
            load_w(rw2, 0, min(c + 2, nchunks - 1) * KC);
            compute(0, 0, sh0);
            if (c + 1 < nchunks) {
                store_x(rx, 1, (c + 1) * KC);
                store_w(rw, 1, (c + 1) * KC);
            }
            __syncthreads();
            if (c + 1 >= nchunks) break;
            load_x(rx, min(c + 3, nchunks - 1) * KC);
            load_w(rw, 0, min(c + 3, nchunks - 1) * KC);
            compute(1, 1, sh0);
            if (c + 2 < nchunks) {
                store_x(rx2, 0, (c + 2) * KC);
                store_w(rw2, 0, (c + 2) * KC);
            }
            __syncthreads();
        }
    } else {
        __syncthreads();
        const int nsteps = nchunks * ntaps;
        int tap = 0, chunk = 0;
        for (int s = 0; s < nsteps; ++s) {
            int ntap = tap + 1, nchunk = chunk;
            if (ntap == ntaps) { ntap = 0; nchunk = chunk + 1; }
            const bool has_next = s + 1 < nsteps;
            const bool new_chunk = has_next && ntap == 0;
            if (has_next) load_w(rw, ntap, nchunk * KC);
            if (new_chunk) load_x(rx, nchunk * KC);
            compute(s & 1, chunk & 1, p.shift[tap]);
            if (has_next) store_w(rw, (s + 1) & 1, nchunk * KC);
            if (new_chunk) store_x(rx, nchunk & 1, nchunk * KC);
            __syncthreads();
            tap = ntap;
            chunk = nchunk;
        }
    }
    }   // !RING

    // ---- epilogue --------------------------------------------------------------------------------
    const bool phased = p.phase_rows < (1 << 30);
    if constexpr (MF == 32 && TM * TN <= 4) {   // (the 8-tile experiment configuration keeps the scalar form: its unrolled body spills)
        // Vector form (everything but the polyphase launches): in the accumulators a lane owns ONE column and 16 rows, i.e. a direct
        // store is sixteen 4-byte stores per 32 x 32 tile (and as many 4-byte residual loads): ablating that epilogue was worth 4 ms of
        // the 23 ms these kernels take per step.  Each wave passes its tiles through a private 32 x 36 LDS tile instead and leaves with
        // four 16-byte stores per lane: 8 lanes cover 128 bytes of a row, a wave-instruction 8 rows.  Same arithmetic per element.
        if (!phased && (p.ldc & 3) == 0 && (!Rg || (p.ldr & 3) == 0)) {
            float* tile = smem + wave * (32 * 36);
            const int lrow = lane >> 3, c4 = (lane & 7) * 4;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
#pragma unroll
                    for (int r = 0; r < MM::NACC; ++r) tile[MM::row(lane, r) * 36 + lcol] = acc[i][j][r];
                    // Interior sub-tiles (round 5): every global read of the four row groups (bias, residual, previous contents, the four keep flags as
                    // one word) is requested before the first store, and the stores carry no per-lane conditions.  The loop below reads inside its
                    // iterations: each of them opens with load round trips queued behind the previous iteration's store (loads and stores share the
                    // in-order vmcnt queue).  Same arithmetic in the same order per element.
                    if (m0 + wm0 + i * MF + 32 <= M && n0 + wn0 + j * MF + 32 <= N && (!p.mask || kp.mask_shift == 0)) {
                        const int n = n0 + wn0 + j * MF + c4, mb = m0 + wm0 + i * MF + lrow;
                        float brow[4];
                        float4 rr[4], old[4];
                        float bcv[4] = {0.f, 0.f, 0.f, 0.f};
                        if (p.bias_mode == BIAS_COL) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) bcv[e] = p.bias[n + e];
                        }
                        const unsigned mk = p.mask ? *reinterpret_cast<const unsigned*>(p.mask + n) : 0x01010101u;
#pragma unroll
                        for (int ps = 0; ps < 4; ++ps) {
                            const int m = mb + ps * 8;
                            brow[ps] = (p.bias_mode == BIAS_ROW) ? p.bias[m] : 0.f;
                            rr[ps] = Rg ? *reinterpret_cast<const float4*>(Rg + (int64_t)m * p.ldr + n) : make_float4(0.f, 0.f, 0.f, 0.f);
                            old[ps] = p.accumulate ? *reinterpret_cast<const float4*>(Cg + (int64_t)m * p.ldc + n) : make_float4(0.f, 0.f, 0.f, 0.f);
                        }
#pragma unroll
                        for (int ps = 0; ps < 4; ++ps) {
                            const float4 a = *reinterpret_cast<const float4*>(tile + (ps * 8 + lrow) * 36 + c4);
                            const float av[4] = {a.x, a.y, a.z, a.w}, rv[4] = {rr[ps].x, rr[ps].y, rr[ps].z, rr[ps].w},
                                        ov[4] = {old[ps].x, old[ps].y, old[ps].z, old[ps].w};
                            float v[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                float x = av[e] + brow[ps];
                                if (p.bias_mode == BIAS_COL) x += bcv[e];
                                if (p.act == ACT_RELU) x = fmaxf(x, 0.f);
                                else if (p.act == ACT_GELU) x = 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
                                else if (p.act == ACT_TANH) x = tanhf(x);
                                x *= p.alpha;
                                if (Rg) x += rv[e];
                                x *= p.beta;
                                if (p.accumulate) x += ov[e];
                                v[e] = ((mk >> (8 * e)) & 0xFFu) ? x : 0.f;
                            }
                            *reinterpret_cast<float4*>(Cg + (int64_t)(mb + ps * 8) * p.ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
                        }
                        continue;
                    }
                    for (int ps = 0; ps < 4; ++ps) {
                        const int row = ps * 8 + lrow;
                        const float4 a = *reinterpret_cast<const float4*>(tile + row * 36 + c4);
                        const int m = m0 + wm0 + i * MF + row;
                        const int n = n0 + wn0 + j * MF + c4;
                        if (m < M && n < N) {
                        const float brow = (p.bias_mode == BIAS_ROW) ? p.bias[m] : 0.f;
                        float v[4] = {a.x + brow, a.y + brow, a.z + brow, a.w + brow};
                        const bool full = n + 3 < N;
                        float* dst = Cg + (int64_t)m * p.ldc + n;
                        float rr[4] = {0.f, 0.f, 0.f, 0.f}, old[4] = {0.f, 0.f, 0.f, 0.f};
                        if (full) {
                            if (Rg) {
                                const float4 t = *reinterpret_cast<const float4*>(Rg + (int64_t)m * p.ldr + n);
                                rr[0] = t.x; rr[1] = t.y; rr[2] = t.z; rr[3] = t.w;
                            }
                            if (p.accumulate) {
                                const float4 t = *reinterpret_cast<const float4*>(dst);
                                old[0] = t.x; old[1] = t.y; old[2] = t.z; old[3] = t.w;
                            }
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (n + e < N) {
                                    if (Rg) rr[e] = Rg[(int64_t)m * p.ldr + n + e];
                                    if (p.accumulate) old[e] = dst[e];
                                }
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float x = v[e];
                            if (p.bias_mode == BIAS_COL) x += p.bias[min(n + e, N - 1)];
                            if (p.act == ACT_RELU) x = fmaxf(x, 0.f);
                            else if (p.act == ACT_GELU) x = 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
                            else if (p.act == ACT_TANH) x = tanhf(x);
                            x *= p.alpha;
                            if (Rg) x += rr[e];
                            x *= p.beta;
                            if (p.accumulate) x += old[e];
                            if (p.mask) {
                                const int oc = min(n + e, N - 1);
                                const int mi = kp.mask_shift >= 0 ? (oc >> kp.mask_shift) : (oc / p.mask_div);
                                if (!p.mask[mi]) x = 0.f;
                            }
                            v[e] = x;
                        }
                        if (full) {
                            *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (n + e < N) dst[e] = v[e];
                        }
                        }
                    }
                }
            return;
        }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < MM::NACC; ++r) {
            const int m = m0 + wm0 + i * MF + MM::row(lane, r);
            if (m >= M) continue;
            int orow = m, po = 0, ostride = 1;
            if (phased) {
                const int ph = m / p.phase_rows;
                orow = m - ph * p.phase_rows;
                ostride = p.out_stride;
#pragma unroll
                for (int q = 0; q < kMaxPhases; ++q) po = (ph == q) ? p.phase_off[q] : po;
            }
            const float brow = (p.bias_mode == BIAS_ROW) ? p.bias[orow] : 0.f;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn0 + j * MF + lcol;
                if (n >= N) continue;
                float v = acc[i][j][r] + brow;
                if (p.bias_mode == BIAS_COL) v += p.bias[n];
                if (p.act == ACT_RELU) v = fmaxf(v, 0.f);
                else if (p.act == ACT_GELU) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
                else if (p.act == ACT_TANH) v = tanhf(v);
                v *= p.alpha;
                const int ocol = n * ostride + po;
                if (Rg) v += Rg[(int64_t)orow * p.ldr + ocol];
                v *= p.beta;
                float* dst = Cg + (int64_t)orow * p.ldc + ocol;
                if (p.accumulate) v += *dst;
                if (p.mask) {
                    const int mi = kp.mask_shift >= 0 ? (ocol >> kp.mask_shift) : (ocol / p.mask_div);
                    if (!p.mask[mi]) v = 0.f;
                }
                *dst = v;
            }
        }
    }
}

// ---- optional per-launch timing with HIP events on the launch stream (bench.py's roofline leg) -----------------
namespace {
struct ProfRec {
    hipEvent_t e0, e1;
    int cfg;
    double flops;
};
bool g_prof_on = false;
std::vector<ProfRec> g_prof;
const char* kCfgNames[] = {"conv_gemm<16,1,4,1,4,16>", "conv_gemm<32,1,2,1,4,16>", "conv_gemm<32,1,1,1,4,16>", "conv_gemm<32,2,4,2,2,16>",
                           "conv_gemm<32,2,2,2,2,16>", "conv_gemm<32,2,2,1,4,16>", "conv_gemm<32,1,2,2,2,16>", "conv_gemm<32,1,1,2,2,16>",
                           "conv_cl<2,split-bf16>",    "conv_cl<1,split-bf16>",    "conv_cl<2,bf16>",          "conv_cl<1,bf16>",
                           "conv_cl_km<2,split-bf16>", "conv_cl_km<1,split-bf16>", "conv_cl_km<2,bf16>",       "conv_cl_km<1,bf16>",
                           "respair_cl<C<=32>",        "respair_cl<C=64>",         "resblock_cl",
                           "conv_cl<2,f16>",           "conv_cl<1,f16>",           "conv_cl_km<2,f16>",        "conv_cl_km<1,f16>",
                           "gemm_skinny<16x16x4>",     "gemm_bfs<bf16x3>",         "gemm_bfs<bf16x6>",
                           "conv_clx<split-bf16>",     "gemm_bfs<f16x3>",          "conv_clx_ffn<split-bf16>"};
constexpr int kNumCfg = 29;
constexpr int cfg_id(int MF, int TM, int TN, int WM) {
    return MF == 16 ? 0 : (WM == 1 ? (TM == 1 ? (TN == 2 ? 1 : 2) : 5) : (TM == 2 ? (TN == 4 ? 3 : 4) : (TN == 2 ? 6 : 7)));
}
}  // namespace

bool conv_prof_active() { return g_prof_on; }
void conv_prof_add(int cfg, double flops, hipEvent_t e0, hipEvent_t e1) { g_prof.push_back(ProfRec{e0, e1, cfg, flops}); }

void conv_prof_begin() {
    for (auto& r : g_prof) {
        (void)hipEventDestroy(r.e0);
        (void)hipEventDestroy(r.e1);
    }
    g_prof.clear();
    g_prof_on = true;
}
// returns one line per tile configuration: name, launches, total ms, total algorithmic FLOP
std::string conv_prof_end() {
    g_prof_on = false;
    double ms[kNumCfg] = {0}, fl[kNumCfg] = {0};
    long cnt[kNumCfg] = {0};
    for (auto& r : g_prof) {
        (void)hipEventSynchronize(r.e1);
        float t = 0.f;
        (void)hipEventElapsedTime(&t, r.e0, r.e1);
        ms[r.cfg] += t;
        fl[r.cfg] += r.flops;
        cnt[r.cfg]++;
        (void)hipEventDestroy(r.e0);
        (void)hipEventDestroy(r.e1);
    }
    g_prof.clear();
    std::string out = "[";
    bool first = true;
    for (int c = 0; c < kNumCfg; ++c) {
        if (!cnt[c]) continue;
        char buf[256];
        snprintf(buf, sizeof(buf), "%s{\"kernel\": \"%s\", \"launches\": %ld, \"ms\": %.6f, \"flop\": %.6e}", first ? "" : ", ", kCfgNames[c],
                 cnt[c], ms[c], fl[c]);
        out += buf;
        first = false;
    }
    return out + "]";
}

static std::atomic<int> g_skinny_max{128};   // sbv2_debug_set_skinny_max; 0 = off
int set_skinny_max(int v) { return g_skinny_max.exchange(v); }
int small_grid_max() { return g_skinny_max.load(std::memory_order_relaxed); }

template <int MF, int TM, int TN, int WM, int WN, int KC, bool RING = false>
static void launch_cfg(const KernelParams& kp0, int Mx, int Nx, hipStream_t stream) {
    constexpr int MT = MF * TM * WM;
    constexpr int NT = MF * TN * WN;
    KernelParams kp = kp0;
    const ConvParams& p = kp.p;
    int smin = p.shift[0], smax = p.shift[0];
    for (int t = 1; t < p.ntaps; ++t) {
        smin = std::min(smin, p.shift[t]);
        smax = std::max(smax, p.shift[t]);
    }
    const int w0 = (smin >= 0) ? (smin / 4) * 4 : -(((-smin) + 3) / 4) * 4;
    int span = round_up(smax - w0, 4);
    SBV2_REQUIRE(span <= kMaxSpan, "conv tap span too large for the staged window");
    int xw = NT + span;
    if (MF == 16) {  // rows k and k+1 of a half-wave must land on different bank halves: pitch = 16 (mod 32)
        while ((xw & 31) != 16) xw += 4;
    }
    kp.xw = xw;
    kp.wshift0 = w0;
    const size_t lds = RING ? sizeof(float) * kRingSlots * KC * (MT + NT)
                            : std::max(sizeof(float) * (2 * KC * MT + 2 * KC * xw), sizeof(float) * 4 * 32 * 36);   // staging | epilogue tiles
    dim3 grid((Nx + NT - 1) / NT, (Mx + MT - 1) / MT, p.groups ? p.ngroups : 1);
    auto kern = conv_gemm_kernel<MF, TM, TN, WM, WN, KC, RING>;
    static std::atomic<uint64_t> lds_allowed{0};   // per (kernel instantiation, device)
    allow_full_lds(reinterpret_cast<const void*>(kern), lds_allowed);
    ProfRec rec;
    if (g_prof_on) {
        rec.cfg = cfg_id(MF, TM, TN, WM);
        rec.flops = p.groups ? p.flops_hint : 2.0 * p.M * (double)p.N * p.K * p.ntaps;
        HIP_CHECK(hipEventCreate(&rec.e0));
        HIP_CHECK(hipEventCreate(&rec.e1));
        HIP_CHECK(hipEventRecord(rec.e0, stream));
    }
    hipLaunchKernelGGL(kern, grid, dim3(kThreads), lds, stream, kp);
    HIP_CHECK(hipGetLastError());
    if (g_prof_on) {
        HIP_CHECK(hipEventRecord(rec.e1, stream));
        g_prof.push_back(rec);
    }
}

void launch_conv(const ConvParams& p, hipStream_t stream) {
    SBV2_REQUIRE(p.ntaps >= 1 && p.ntaps <= kMaxTaps, "bad tap count");
    SBV2_REQUIRE((p.lda & 3) == 0 && (p.ldb & 3) == 0, "operand pitch must be a multiple of 4 floats");
    KernelParams kp;
    kp.p = p;
    kp.mask_shift = -1;
    if (p.mask && p.mask_div > 0 && (p.mask_div & (p.mask_div - 1)) == 0) {
        int s = 0;
        while ((1 << s) < p.mask_div) ++s;
        kp.mask_shift = s;
    }
    const int Mx = p.groups ? p.maxM : p.M;
    const int Nx = p.groups ? p.maxN : p.N;
    if (Mx <= 0 || Nx <= 0) return;
    const int64_t z = p.groups ? p.ngroups : 1;
    auto blocks = [&](int mt, int nt) { return z * ((Mx + mt - 1) / mt) * (int64_t)((Nx + nt - 1) / nt); };
    // tile choice: largest tile that still gives >= 2 workgroups per CU; 16-row MFMA for 16-row problems
    if (Mx <= 16) return launch_cfg<16, 1, 4, 1, 4, 16>(kp, Mx, Nx, stream);
    if (Mx <= 32) {
        if (blocks(32, 256) >= 512) return launch_cfg<32, 1, 2, 1, 4, 16>(kp, Mx, Nx, stream);
        return launch_cfg<32, 1, 1, 1, 4, 16>(kp, Mx, Nx, stream);
    }
    // measured on MI355X: 64-row tiles (81 VGPRs, 48 KB LDS -> 3 workgroups per CU) beat 128x256 (1-2 per CU) by 20-30 %
    if (p.ntaps == 1) {
        // 1x1 products (measured on the DeBERTa / flow shapes with the two-deep prefetch): 64x128 beats 64x256, and 64x64 wins when
        // the grid would otherwise be under two workgroups per CU
        // small grids first (single-utterance calls): one-wave 16 x 16 tiles fed through an LDS-DMA ring (gemm_skinny.hip: same bits, see there;
        // long products pay the tiled kernel's per-chunk round trip 64 times or more, so their threshold is higher)
        const int skinny_max = g_skinny_max.load(std::memory_order_relaxed);
        if (blocks(64, 64) < (p.K >= 512 ? skinny_max + skinny_max / 2 : skinny_max) && launch_gemm_skinny(p, kp.mask_shift, stream)) return;
        const bool ring = !p.groups && p.shift[0] == 0 && (p.K & 15) == 0 && p.K >= 48 && (p.lda & 3) == 0 && (p.ldb & 3) == 0 &&
                          p.N <= p.nb && p.phase_rows >= (1 << 30);
        // 128 x 128 tiles (half the L2 bytes per FLOP; 64 KB of ring: two workgroups per CU) when they fill the chip in ONE round: 3072 x 2112 is
        // 408 workgroups on 512 slots (160 vs 177 us), 4096 x 2112 would be 544 = two rounds (251 vs 222 us)
        if (ring && blocks(128, 128) >= 384 && blocks(128, 128) <= 512) return launch_cfg<32, 2, 2, 2, 2, 16, true>(kp, Mx, Nx, stream);
        // ... and 64 x 64 tiles otherwise: 32 KB of ring = five workgroups per CU keep the MFMA pipe fed across the chunk barriers and quantise
        // better than 64 x 128 (4096 x 2112 x 1024: 211 vs 222 us; 768 x 28704 x 192: 113 vs 118)
        if (ring) return launch_cfg<32, 1, 1, 2, 2, 16, true>(kp, Mx, Nx, stream);
        if (blocks(64, 128) >= 512) return launch_cfg<32, 1, 2, 2, 2, 16>(kp, Mx, Nx, stream);
        // single-utterance calls (DeBERTa at 64 tokens: 16-64 workgroups of 64 rows on 256 CUs, 47 us per launch whatever the size):
        // 32-row tiles double the workgroup count; the per-element summation order does not depend on the tile, so batch rows stay
        // bit-identical to single calls
        // ... and 64-channel chunks: a workgroup of a small grid is alone on its CU, its K loop is a chain of global round trips
        // (two chunks in flight: ~0.7 us per 16-channel chunk whatever the MFMA work), so it asks for four times as much per trip
        // ... superseded where it applies by one-wave 16 x 16 tiles fed through an LDS-DMA ring (gemm_skinny.hip: same bits, see there)
        // (long products pay the tiled kernel's per-chunk round trip 64 times or more: the FFN-up Linear of DeBERTa at 66 tokens, 128 workgroups,
        // is 43 us tiled, 28 us as 1280 single waves)
        if (blocks(64, 64) < 128 && Nx <= 128) return launch_cfg<32, 1, 1, 1, 4, 64>(kp, Mx, Nx, stream);
        if (ring) return launch_cfg<32, 1, 1, 2, 2, 16, true>(kp, Mx, Nx, stream);
        return launch_cfg<32, 1, 1, 2, 2, 16>(kp, Mx, Nx, stream);
    }
    if (blocks(64, 64) < g_skinny_max.load(std::memory_order_relaxed) && launch_gemm_skinny_taps(p, kp.mask_shift, stream)) return;
    if (blocks(64, 256) >= 512) return launch_cfg<32, 2, 2, 1, 4, 16>(kp, Mx, Nx, stream);
    if (blocks(64, 128) >= 256) return launch_cfg<32, 1, 2, 2, 2, 16>(kp, Mx, Nx, stream);
    return launch_cfg<32, 1, 1, 2, 2, 16>(kp, Mx, Nx, stream);
}

}  // namespace sbv2
