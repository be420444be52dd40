// Fused window-relative attention of the VITS encoders (attentions.MultiHeadAttention.attention, upstream style_bert_vits2;
// same block as transformers modeling_vits.py:844-997) on k-major planes, exact f32 MFMA, online softmax:
//
//   s[i][j]  = (q_i . k_j + q_i . emb_rel_k[j - i + w]  (|j - i| <= w)) / sqrt(dk)
//   p        = softmax_j(s)
//   ctx[:, i] = sum_j p[i][j] v_j + sum_{|r| <= w} p[i][i + r] emb_rel_v[r + w]
//
// The unfused path (grouped GEMM S^T = K^T Q -> k_vits_softmax -> grouped GEMM V P^T -> k_vits_relv_add) writes the T x T score block
// of every (utterance, head) to HBM and sweeps it five times; at 897 frames x 2 heads x 32 utterances x 24 flow layers that is
// ~20 GB per step, and the block grows quadratically for long-form input (784 MB per head at 14 001 frames).  Here a wave owns
// 32 query columns, keeps q in registers (one value per lane and k-step: exactly the B operand of v_mfma_f32_32x32x2_f32), and walks
// the keys in tiles of 32 staged through LDS for the four waves of the workgroup:
//
//   S^T tile = K_tile^T Q       A = K[d][j] (row j = lane & 31, k = d),  B = q            -> 16 accumulators: rows j, column i = lane
//   online softmax per column   (16 registers + one cross-half exchange), band scores (|j - i| <= w) kept in LDS for the emb_rel_v term
//   ctx     += V_tile P         the accumulator registers ARE the B operand: MFMA k-step s, lane half h consumes key row
//                               (s & 3) + 8 (s >> 2) + 4 h, which is exactly the row accumulator register s holds in half h; the A
//                               operand reads V[d][that row] from the LDS tile (pitch 33: conflict free)
//
// Summation order is fixed (no atomics, no scheduling dependence): a column's result does not depend on which other utterances share
// the batch, as the packed-batch contract requires.
#include <atomic>
#include <cfloat>
#include <type_traits>
#include <vector>
#include <cstdio>

#include "common.h"
#include "ops.h"

// No floating-point contraction in this file: the three split-bf16 kernels below (converting, pre-split, pre-split + pipelined) promise the SAME bits
// (a batch row runs on one, the single-utterance call on another), and whether `a * b + c` becomes one fma is otherwise the optimiser's choice per
// kernel (it differed: x * qs2 - max fused in one kernel and not in the other, 3e-7 on the waveform).  Every fma that is wanted is written as fmaf.
#pragma clang fp contract(off)

namespace sbv2 {

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {
constexpr int kFaThreads = 256;
constexpr int kFaPitch = 33;
constexpr int kFaMaxWin = 4;
constexpr int kFaBand = 2 * kFaMaxWin + 1;
constexpr float kFaNegBig = -1e30f;

// The two halves of a wave exchange a value (lane l <-> lane l ^ 32) with ONE v_permlane32_swap instead of a round trip through the LDS crossbar
// (ds_bpermute: ~100 cycles of latency, twice per key step on the softmax's critical path).  lo = the value of lane l & 31, hi = that of lane (l & 31) + 32.
__device__ __forceinline__ void fa_halves(float x, float& lo, float& hi) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    lo = __uint_as_float(r[0]);
    hi = __uint_as_float(r[1]);
}

template <int DT>
__global__ __launch_bounds__(kFaThreads) __attribute__((amdgpu_waves_per_eu(2))) void k_vits_flash(const AttnGroup* groups, const float* Q, const float* K, const float* V, int ld,
                                                            float* ctx, int ldc, int dk, const float* erk, const float* erv, int w,
                                                            float qscale) {
    constexpr int DR = DT * 32;   // head dimension padded to whole 32-row MFMA tiles
    constexpr int NS = DR / 2;    // k-steps of the 32x32x2 MFMA over the head dimension
    constexpr int NP = DR / 8;    // rows of a tile each thread stages (256 threads = 8 rows x 32 columns per pass)
    __shared__ float Kt[DR * kFaPitch], Vt[DR * kFaPitch];
    __shared__ float erk_s[kFaBand * DR], erv_s[kFaBand * DR];
    __shared__ float rk_s[4][kFaBand][32], band_s[4][kFaBand][32];

    const AttnGroup g = groups[blockIdx.y];
    const int T = g.T;
    const int q0 = blockIdx.x * 128;
    if (q0 >= T) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31, kh = lane >> 5;
    const int i0 = q0 + wave * 32;
    const bool active = i0 < T;           // wave-uniform: inactive waves only help staging
    const int i = i0 + col;
    const int ic = min(i, T - 1);
    const int64_t off = (int64_t)g.head * dk * ld + g.col0;
    const float* Qg = Q + off;
    const float* Kg = K + off;
    const float* Vg = V + off;
    const int nb = 2 * w + 1;

    for (int idx = tid; idx < kFaBand * DR; idx += kFaThreads) {
        const int r = idx / DR, d = idx - r * DR;
        const bool in = r < nb && d < dk;
        erk_s[idx] = in ? erk[r * dk + d] : 0.f;
        erv_s[idx] = in ? erv[r * dk + d] : 0.f;
    }

    // staging registers: thread t owns column t & 31 and rows (t >> 5) + 8 p of both tiles
    const int sc = tid & 31, sr = tid >> 5;
    float kreg[NP], vreg[NP];
    auto load_tile = [&](int j0) {
        const int jc = min(j0 + sc, T - 1);
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int d = min(sr + 8 * p, dk - 1);
            kreg[p] = Kg[(int64_t)d * ld + jc];
            vreg[p] = Vg[(int64_t)d * ld + jc];
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int d = sr + 8 * p;
            const bool in = d < dk;
            Kt[d * kFaPitch + sc] = in ? kreg[p] : 0.f;
            Vt[d * kFaPitch + sc] = in ? vreg[p] : 0.f;
        }
    };
    load_tile(0);

    __syncthreads();   // erk_s / erv_s

    // relative-key logits of this wave's columns: rk[r][i] = qscale * q_i . emb_rel_k[r].  A rolled loop that re-reads q from the
    // cache (once per workgroup): fully unrolled on the register copy of q it cost 250 VGPRs + AGPR spills and halved the occupancy.
    {
        float part[kFaBand];
#pragma unroll
        for (int r = 0; r < kFaBand; ++r) part[r] = 0.f;
        // (blocks of 16 k-steps: the 16 loads of a block are in flight together; one load per step with its use right behind it exposed the
        // global latency 24 times per workgroup, 12-20 us of an 84 us single-utterance launch.  Same order of additions.)
#pragma unroll 1
        for (int sb = 0; sb < NS; sb += 16) {
            float qd[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int d = 2 * (sb + u) + kh;
                const float x = Qg[(int64_t)min(d, dk - 1) * ld + ic];   // (unconditional: no branch around the load)
                qd[u] = d < dk ? x : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int d = 2 * (sb + u) + kh;
#pragma unroll
                for (int r = 0; r < kFaBand; ++r) part[r] = fmaf(qd[u], erk_s[r * DR + d], part[r]);
            }
        }
#pragma unroll
        for (int r = 0; r < kFaBand; ++r) {
            float lo, hi;                                   // fixed order: (even d) + (odd d) in both halves
            fa_halves(part[r], lo, hi);
            rk_s[wave][r][col] = (lo + hi) * qscale;
            band_s[wave][r][col] = kFaNegBig;
        }
    }

    // q: one value per lane and k-step (B operand), constant over the key loop
    float qv[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int d = 2 * s + kh;
        const float x = Qg[(int64_t)min(d, dk - 1) * ld + ic];
        qv[s] = d < dk ? x : 0.f;
    }

    f32x16 cacc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) cacc[dt][r] = 0.f;
    float m = kFaNegBig, l = 0.f;

    const int ntiles = (T + 31) >> 5;
    for (int jt = 0; jt < ntiles; ++jt) {
        const int j0 = jt * 32;
        store_tile();
        __syncthreads();
        if (jt + 1 < ntiles) load_tile(j0 + 32);   // lands behind this tile's MFMAs
        if (active) {
            f32x16 sacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
            for (int s = 0; s < NS; ++s) sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(Kt[(2 * s + kh) * kFaPitch + col], qv[s], sacc, 0, 0, 0);
            const bool diag = j0 <= i0 + 31 + w && j0 + 31 >= i0 - w;   // wave-uniform: the tile touches the +-w band
            float mt = kFaNegBig;
            if (diag) {   // (one wave-uniform branch per key step instead of one per element)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int j = j0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                    float sv = sacc[r] * qscale;
                    const int rr = j - i + w;
                    if (rr >= 0 && rr < nb && j < T) {
                        sv += rk_s[wave][rr][col];
                        band_s[wave][rr][col] = sv;
                    }
                    sv = j < T ? sv : kFaNegBig;
                    sacc[r] = sv;
                    mt = fmaxf(mt, sv);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int j = j0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                    const float sv = j < T ? sacc[r] * qscale : kFaNegBig;
                    sacc[r] = sv;
                    mt = fmaxf(mt, sv);
                }
            }
            {
                float lo, hi;
                fa_halves(mt, lo, hi);
                mt = fmaxf(mt, kh ? lo : hi);
            }
            const float mn = fmaxf(m, mt);
            const float alpha = expf(m - mn);
            float ps = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = expf(sacc[r] - mn);
                sacc[r] = e;
                ps += e;
            }
            {
                float lo, hi;
                fa_halves(ps, lo, hi);
                l = fmaf(l, alpha, lo + hi);
            }
            m = mn;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) cacc[dt][r] *= alpha;
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int jj = (s & 3) + 8 * (s >> 2) + 4 * kh;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
                    cacc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(Vt[(dt * 32 + col) * kFaPitch + jj], sacc[s], cacc[dt], 0, 0, 0);
            }
        }
        __syncthreads();   // every wave is done with the tiles before they are overwritten
    }
    if (!active) return;

    // relative-value term from the band probabilities, normalisation, store
    const float inv = 1.0f / l;
    float pb[kFaBand];
#pragma unroll
    for (int r = 0; r < kFaBand; ++r) pb[r] = expf(band_s[wave][r][col] - m) * inv;   // untouched entries: exp(-1e30 - m) = 0
    float* Cg = ctx + (int64_t)g.head * dk * ldc + g.col0;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int d = dt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            float v = cacc[dt][r] * inv;
#pragma unroll
            for (int b = 0; b < kFaBand; ++b) v = fmaf(pb[b], erv_s[b * DR + d], v);
            if (i < T && d < dk) Cg[(int64_t)d * ldc + i] = v;
        }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Split-bf16 variant for the flow (everything after the integer durations): the two big products run on v_mfma_f32_32x32x16_bf16 with
// hi/lo operands (3 MFMAs per product, ~1e-5 relative like the decoder convs), 18 + 18 MFMAs of 32 cycles per 32-key tile instead of
// 96 f32 MFMAs of 64.  Softmax, the relative terms and all accumulation stay f32.  Layouts:
//   K tile  -> LDS [32 keys][dk] bf16 hi | lo (row pitch 2 dk + 16 bytes: 16 consecutive keys hit 16 distinct 16-byte slots);
//              A fragment of key row j, k-step s = 16 bytes at d = 16 s + 8 h
//   q       -> registers: B fragments (8 consecutive d per lane), hi and lo, 6 k-steps
//   V tile  -> LDS [dk channels][32 keys] bf16 hi | lo, keys stored in the order the accumulator registers hold them
//              (bits 2 and 3 of the key index swapped), so that the A fragment of PV's k-step s' is one 16-byte read and the B fragment is
//              simply the eight score registers 8 s' .. 8 s' + 7 converted to bf16 hi / lo
// ---------------------------------------------------------------------------------------------------------------------------------
typedef __bf16 fa_bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void fa_split8(const float (&v)[8], fa_bf16x8& hi, fa_bf16x8& lo) {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        hi[t] = (__bf16)v[t];
        lo[t] = (__bf16)(v[t] - (float)hi[t]);
    }
}

template <int DT>
__global__ __launch_bounds__(kFaThreads) __attribute__((amdgpu_waves_per_eu(2))) void k_vits_flash_x3(
    const AttnGroup* groups, const float* Q, const float* K, const float* V, int ld, float* ctx, int ldc, int dk, const float* erk,
    const float* erv, int w, float qscale) {
    constexpr int DR = DT * 32;
    // (DR / 8 elements of each matrix per thread and tile)
    constexpr int KS = DR / 16;            // bf16 k-steps over the (padded) head dimension
    constexpr int PK = 2 * DR + 16;        // bytes per key row of the K tile
    constexpr int PV = 2 * 32 + 16;        // bytes per channel row of the V tile
    __shared__ __attribute__((aligned(16))) char kt_hi[32 * PK], kt_lo[32 * PK];
    __shared__ __attribute__((aligned(16))) char vt_hi[DR * PV], vt_lo[DR * PV];
    __shared__ float erk_s[kFaBand * DR], erv_s[kFaBand * DR];
    __shared__ float rk_s[4][kFaBand][32], band_s[4][kFaBand][32];

    const AttnGroup g = groups[blockIdx.y];
    const int T = g.T;
    const int q0 = blockIdx.x * 128;
    if (q0 >= T) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31, kh = lane >> 5;
    const int i0 = q0 + wave * 32;
    const bool active = i0 < T;
    const int i = i0 + col;
    const int ic = min(i, T - 1);
    const int64_t off = (int64_t)g.head * dk * ld + g.col0;
    const float* Qg = Q + off;
    const float* Kg = K + off;
    const float* Vg = V + off;
    const int nb = 2 * w + 1;
    // logits are kept in the base-2 domain (scores * log2 e): every exponential of the online softmax is then ONE v_exp_f32 instead of the
    // twelve-instruction expf (argument reduction + ldexp + range checks), 17 of them per lane and key tile
    const float qs2 = qscale * 1.4426950408889634f;

    for (int idx = tid; idx < kFaBand * DR; idx += kFaThreads) {
        const int r = idx / DR, d = idx - r * DR;
        const bool in = r < nb && d < dk;
        erk_s[idx] = in ? erk[r * dk + d] : 0.f;
        erv_s[idx] = in ? erv[r * dk + d] : 0.f;
    }
    // Staging (round 4): a thread owns FOUR consecutive elements along the axis its LDS image is contiguous in, so that every LDS store is 8 bytes
    // (round 3: one element per store: 48 two-byte ds_write per thread and tile, LDS bank-conflict ratio 0.36, the staging phase as long as the tile's
    // MFMAs).  K image [key][d]: key = tid & 31, d = 4 g .. 4 g + 3 with g = (tid >> 5) + 8 p: four scalar loads (coalesced over the keys), one store
    // per part.  V image [d][key, bits 2 and 3 swapped]: d = (tid >> 3) + 32 p, keys 4 q .. 4 q + 3 with q = tid & 7 (a group of four consecutive keys stays
    // consecutive under the swap): ONE 16-byte load, one store per part.  Same values in the same places: bit-identical results.
    typedef __bf16 fa_bf16x4 __attribute__((ext_vector_type(4)));
    typedef float fa_f32x4 __attribute__((ext_vector_type(4)));
    const int sc = tid & 31, sg = tid >> 5;            // K: key, first d group
    const int vq = tid & 7, vd = tid >> 3;             // V: key quad, first d
    const int vqp = (vq & 4) | ((vq & 1) << 1) | ((vq >> 1) & 1);   // the quad's place in the V image (bits 2 and 3 of the key index swapped)
    float kreg[DT][4];
    fa_f32x4 vreg[DT];
    auto load_tile = [&](int j0) {
        const int jc = min(j0 + sc, T - 1);
#pragma unroll
        for (int p = 0; p < DT; ++p)
#pragma unroll
            for (int e = 0; e < 4; ++e) kreg[p][e] = Kg[(int64_t)min(4 * (sg + 8 * p) + e, dk - 1) * ld + jc];
        if (j0 + 32 <= T) {   // (uniform) whole tile inside the utterance: 16-byte loads (columns are 16-byte aligned: starts and pitches are multiples of 4)
#pragma unroll
            for (int p = 0; p < DT; ++p) vreg[p] = *reinterpret_cast<const fa_f32x4*>(Vg + (int64_t)min(vd + 32 * p, dk - 1) * ld + j0 + 4 * vq);
        } else {
#pragma unroll
            for (int p = 0; p < DT; ++p)
#pragma unroll
                for (int e = 0; e < 4; ++e) vreg[p][e] = Vg[(int64_t)min(vd + 32 * p, dk - 1) * ld + min(j0 + 4 * vq + e, T - 1)];
        }
    };
    auto store_tile = [&](int j0) {
        const bool jin = j0 + sc < T;     // keys beyond the utterance: zero operands (their scores are masked anyway, V must not be NaN)
#pragma unroll
        for (int p = 0; p < DT; ++p) {
            const int g = sg + 8 * p;
            fa_bf16x4 h, l;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float kv = (4 * g + e < dk && jin) ? kreg[p][e] : 0.f;
                h[e] = (__bf16)kv;
                l[e] = (__bf16)(kv - (float)h[e]);
            }
            *reinterpret_cast<fa_bf16x4*>(kt_hi + sc * PK + g * 8) = h;
            *reinterpret_cast<fa_bf16x4*>(kt_lo + sc * PK + g * 8) = l;
        }
#pragma unroll
        for (int p = 0; p < DT; ++p) {
            const int d = vd + 32 * p;
            fa_bf16x4 h, l;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float vv = (d < dk && j0 + 4 * vq + e < T) ? vreg[p][e] : 0.f;
                h[e] = (__bf16)vv;
                l[e] = (__bf16)(vv - (float)h[e]);
            }
            *reinterpret_cast<fa_bf16x4*>(vt_hi + d * PV + vqp * 8) = h;
            *reinterpret_cast<fa_bf16x4*>(vt_lo + d * PV + vqp * 8) = l;
        }
    };
    load_tile(0);
    __syncthreads();   // erk_s / erv_s

    {   // relative-key logits (f32, as in k_vits_flash)
        float part[kFaBand];
#pragma unroll
        for (int r = 0; r < kFaBand; ++r) part[r] = 0.f;
#pragma unroll 1
        for (int sb = 0; sb < DR / 2; sb += 16) {   // blocks of 16 loads in flight together (see k_vits_flash)
            float qd[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int d = 2 * (sb + u) + kh;
                const float x = Qg[(int64_t)min(d, dk - 1) * ld + ic];   // (unconditional: no branch around the load)
                qd[u] = d < dk ? x : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int d = 2 * (sb + u) + kh;
#pragma unroll
                for (int r = 0; r < kFaBand; ++r) part[r] = fmaf(qd[u], erk_s[r * DR + d], part[r]);
            }
        }
#pragma unroll
        for (int r = 0; r < kFaBand; ++r) {
            float lo, hi;
            fa_halves(part[r], lo, hi);
            rk_s[wave][r][col] = (lo + hi) * qs2;
            band_s[wave][r][col] = kFaNegBig;
        }
    }
    // q fragments: lane (column i, half h) holds d = 16 s + 8 h + t
    fa_bf16x8 qh[KS], ql[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        float v[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int d = 16 * s + 8 * kh + t;
            const float x = Qg[(int64_t)min(d, dk - 1) * ld + ic];
            v[t] = d < dk ? x : 0.f;
        }
        fa_split8(v, qh[s], ql[s]);
    }

    f32x16 cacc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) cacc[dt][r] = 0.f;
    float m = kFaNegBig, l = 0.f;

    const int ntiles = (T + 31) >> 5;
    for (int jt = 0; jt < ntiles; ++jt) {
        const int j0 = jt * 32;
        store_tile(j0);
        __syncthreads();
        if (jt + 1 < ntiles) load_tile(j0 + 32);
        if (active) {
            f32x16 sacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const fa_bf16x8 ah = *reinterpret_cast<const fa_bf16x8*>(kt_hi + col * PK + (16 * s + 8 * kh) * 2);
                const fa_bf16x8 al = *reinterpret_cast<const fa_bf16x8*>(kt_lo + col * PK + (16 * s + 8 * kh) * 2);
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, qh[s], sacc, 0, 0, 0);
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, ql[s], sacc, 0, 0, 0);
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, qh[s], sacc, 0, 0, 0);
            }
            const bool diag = j0 <= i0 + 31 + w && j0 + 31 >= i0 - w;
            const bool tail = j0 + 32 > T;   // (uniform) only the utterance's last key step has keys to mask
            float mt = kFaNegBig;
            // three copies of the loop behind ONE wave-uniform branch: tested per element, `diag` and `tail` were 32 taken branches per key step
            // (a quarter of the step's issue time); interior steps, all but two or three per wave, are 16 multiplies and 16 maxima
            if (diag) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int j = j0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                    float sv = sacc[r] * qs2;
                    const int rr = j - i + w;
                    if (rr >= 0 && rr < nb && j < T) {
                        sv += rk_s[wave][rr][col];
                        band_s[wave][rr][col] = sv;
                    }
                    if (tail) sv = j < T ? sv : kFaNegBig;
                    sacc[r] = sv;
                    mt = fmaxf(mt, sv);
                }
            } else if (tail) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int j = j0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                    const float sv = j < T ? sacc[r] * qs2 : kFaNegBig;
                    sacc[r] = sv;
                    mt = fmaxf(mt, sv);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    sacc[r] *= qs2;
                    mt = fmaxf(mt, sacc[r]);
                }
            }
            {
                float lo, hi;
                fa_halves(mt, lo, hi);
                mt = fmaxf(mt, kh ? lo : hi);
            }
            const float mn = fmaxf(m, mt);
            const float alpha = __builtin_amdgcn_exp2f(m - mn);
            float ps = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = __builtin_amdgcn_exp2f(sacc[r] - mn);
                sacc[r] = e;
                ps += e;
            }
            {
                float lo, hi;
                fa_halves(ps, lo, hi);
                l = fmaf(l, alpha, lo + hi);
            }
            m = mn;
            // (the running maximum of most columns stops moving after the first key steps: multiplying by exactly 1 changes nothing, so the 48
            // multiplies are skipped whenever no lane of the wave has a new maximum: same bits)
            if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) cacc[dt][r] *= alpha;
            }
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) {
                float pv8[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) pv8[t] = sacc[8 * sp + t];
                fa_bf16x8 ph, pl;
                fa_split8(pv8, ph, pl);
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const fa_bf16x8 vh = *reinterpret_cast<const fa_bf16x8*>(vt_hi + (dt * 32 + col) * PV + (16 * sp + 8 * kh) * 2);
                    const fa_bf16x8 vl = *reinterpret_cast<const fa_bf16x8*>(vt_lo + (dt * 32 + col) * PV + (16 * sp + 8 * kh) * 2);
                    cacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl, ph, cacc[dt], 0, 0, 0);
                    cacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, pl, cacc[dt], 0, 0, 0);
                    cacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, ph, cacc[dt], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
    if (!active) return;

    const float inv = 1.0f / l;
    float pb[kFaBand];
#pragma unroll
    for (int r = 0; r < kFaBand; ++r) pb[r] = __builtin_amdgcn_exp2f(band_s[wave][r][col] - m) * inv;
    float* Cg = ctx + (int64_t)g.head * dk * ldc + g.col0;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int d = dt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            float v = cacc[dt][r] * inv;
#pragma unroll
            for (int b = 0; b < kFaBand; ++b) v = fmaf(pb[b], erv_s[b * DR + d], v);
            if (i < T && d < dk) Cg[(int64_t)d * ldc + i] = v;
        }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The same split-bf16 attention on PRE-SPLIT keys and values (round 3).  k_vits_flash_x3 converts every K / V tile f32 -> bf16 hi / lo while
// staging it, once per 128-query workgroup: T / 128 times per layer (110 times at 14 001 frames), 24 conversions + 48 two-byte LDS stores per
// thread and 32-key tile.  Here the parts exist in HBM (written next to the f32 plane by the q | k | v product's epilogue: the same
// hi = bf16(x), lo = bf16(x - hi)), a staged tile is 64 keys (two 32-key softmax steps per barrier pair: the arithmetic and its order are
// those of k_vits_flash_x3, bit for bit), moved with 8-byte loads and 8-byte LDS stores, and both tiles keep the k-major shape of the planes:
//   K image [part][d][64 keys] (row pitch 160 B): the A fragment of S^T = K^T Q (key row = lane & 31, 8 consecutive d) is two transposing
//           reads (ds_read_b64_tr_b16), conflict free at this pitch (rows 8 banks apart, 4 x 8-byte columns per row);
//   V image [part][d][64 keys] (row pitch 136 B): the A fragment of ctx += V P (channel row = lane & 31) takes the keys in the order the score
//           registers hold them, 16 sp + 4 h + {0..3, 8..11}: two 8-byte reads (34-dword pitch: 16 rows on 16 distinct bank pairs).
// ---------------------------------------------------------------------------------------------------------------------------------
typedef short fa_s16x4 __attribute__((ext_vector_type(4)));
typedef short fa_s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) fa_s16x4 fa_lds_s16x4;
constexpr int kFpKeys = 64;
constexpr int kFpKS = 160;   // bytes per d row of the K image
constexpr int kFpVS = 136;   // ... of the V image

template <int DT>
__global__ __launch_bounds__(kFaThreads) __attribute__((amdgpu_waves_per_eu(2))) void k_vits_flash_x3p(
    const AttnGroup* groups, const float* Q, int ld, const __bf16* Kp, const __bf16* Vp, int64_t pstride, int ldp, float* ctx, int ldc, int dk,
    const float* erk, const float* erv, int w, float qscale) {
    constexpr int DR = DT * 32;
    constexpr int KS = DR / 16;            // bf16 k-steps over the (padded) head dimension
    constexpr int NL = DR / 16;            // 8-byte pieces per thread, part and matrix of one 64-key tile (16 rows per pass)
    extern __shared__ __attribute__((aligned(16))) char fp_smem[];
    char* kt = fp_smem;                                   // [2][DR][kFpKS]
    char* vt = kt + 2 * DR * kFpKS;                       // [2][DR][kFpVS]
    float* erk_s = reinterpret_cast<float*>(vt + 2 * DR * kFpVS);   // [kFaBand][DR]
    float* erv_s = erk_s + kFaBand * DR;
    float (*rk_s)[kFaBand][32] = reinterpret_cast<float (*)[kFaBand][32]>(erv_s + kFaBand * DR);
    float (*band_s)[kFaBand][32] = rk_s + 4;

    const AttnGroup g = groups[blockIdx.y];
    const int T = g.T;
    const int q0 = blockIdx.x * 128;
    if (q0 >= T) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31, kh = lane >> 5;
    const int i0 = q0 + wave * 32;
    const bool active = i0 < T;
    const int i = i0 + col;
    const int ic = min(i, T - 1);
    const float* Qg = Q + (int64_t)g.head * dk * ld + g.col0;
    const int64_t poff = (int64_t)g.head * dk * ldp + g.col0;
    const int nb = 2 * w + 1;
    const float qs2 = qscale * 1.4426950408889634f;   // base-2 logits, as in k_vits_flash_x3

    for (int idx = tid; idx < kFaBand * DR; idx += kFaThreads) {
        const int r = idx / DR, d = idx - r * DR;
        const bool in = r < nb && d < dk;
        erk_s[idx] = in ? erk[r * dk + d] : 0.f;
        erv_s[idx] = in ? erv[r * dk + d] : 0.f;
    }
    // staging: thread (tx, ty) moves keys 4 tx .. 4 tx + 3 of rows ty + 16 p
    const int tx = tid & 15, ty = tid >> 4;
    uint2 kreg[2][NL], vreg[2][NL];
    auto load_tile = [&](int j0) {
        const int jq = j0 + 4 * tx;
        const int jc = min(jq, (T - 1) & ~3);          // an aligned quad that starts inside the utterance (its tail may lie behind T: in the row)
#pragma unroll
        for (int p = 0; p < NL; ++p) {
            const int d = min(ty + 16 * p, dk - 1);
            const int64_t o = poff + (int64_t)d * ldp + jc;
#pragma unroll
            for (int part = 0; part < 2; ++part) {
                kreg[part][p] = *reinterpret_cast<const uint2*>(Kp + part * pstride + o);
                vreg[part][p] = *reinterpret_cast<const uint2*>(Vp + part * pstride + o);
            }
        }
    };
    auto store_tile = [&](int j0) {
        if (j0 + kFpKeys <= T && dk == DR) {   // (uniform) an interior tile of a full-width head: nothing to mask, 24 stores and no VALU work
#pragma unroll
            for (int p = 0; p < NL; ++p) {
                const int d = ty + 16 * p;
#pragma unroll
                for (int part = 0; part < 2; ++part) {
                    *reinterpret_cast<uint2*>(kt + (part * DR + d) * kFpKS + 8 * tx) = kreg[part][p];
                    *reinterpret_cast<uint2*>(vt + (part * DR + d) * kFpVS + 8 * tx) = vreg[part][p];
                }
            }
            return;
        }
        // keys beyond the utterance and rows beyond dk: zero operands (their scores are masked anyway, V must not carry garbage)
        const int nv = min(max(T - (j0 + 4 * tx), 0), 4);
        const unsigned m0 = nv >= 2 ? 0xffffffffu : (nv == 1 ? 0x0000ffffu : 0u);
        const unsigned m1 = nv >= 4 ? 0xffffffffu : (nv == 3 ? 0x0000ffffu : 0u);
#pragma unroll
        for (int p = 0; p < NL; ++p) {
            const int d = ty + 16 * p;
            const bool din = d < dk;
#pragma unroll
            for (int part = 0; part < 2; ++part) {
                uint2 kv = kreg[part][p], vv = vreg[part][p];
                kv.x = din ? kv.x & m0 : 0u; kv.y = din ? kv.y & m1 : 0u;
                vv.x = din ? vv.x & m0 : 0u; vv.y = din ? vv.y & m1 : 0u;
                *reinterpret_cast<uint2*>(kt + (part * DR + d) * kFpKS + 8 * tx) = kv;
                *reinterpret_cast<uint2*>(vt + (part * DR + d) * kFpVS + 8 * tx) = vv;
            }
        }
    };
    load_tile(0);
    __syncthreads();   // erk_s / erv_s

    {   // relative-key logits (f32, as in k_vits_flash)
        float part[kFaBand];
#pragma unroll
        for (int r = 0; r < kFaBand; ++r) part[r] = 0.f;
#pragma unroll 1
        for (int sb = 0; sb < DR / 2; sb += 16) {   // blocks of 16 loads in flight together (see k_vits_flash)
            float qd[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int d = 2 * (sb + u) + kh;
                const float x = Qg[(int64_t)min(d, dk - 1) * ld + ic];   // (unconditional: no branch around the load)
                qd[u] = d < dk ? x : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int d = 2 * (sb + u) + kh;
#pragma unroll
                for (int r = 0; r < kFaBand; ++r) part[r] = fmaf(qd[u], erk_s[r * DR + d], part[r]);
            }
        }
#pragma unroll
        for (int r = 0; r < kFaBand; ++r) {
            float lo, hi;
            fa_halves(part[r], lo, hi);
            rk_s[wave][r][col] = (lo + hi) * qs2;
            band_s[wave][r][col] = kFaNegBig;
        }
    }
    fa_bf16x8 qh[KS], ql[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        float v[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int d = 16 * s + 8 * kh + t;
            const float x = Qg[(int64_t)min(d, dk - 1) * ld + ic];
            v[t] = d < dk ? x : 0.f;
        }
        fa_split8(v, qh[s], ql[s]);
    }

    f32x16 cacc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) cacc[dt][r] = 0.f;
    float m = kFaNegBig, l = 0.f;

    // fragment addresses inside a tile image (LDS byte offsets)
    const int g16 = lane >> 4, q4 = (lane >> 2) & 3, p4 = lane & 3;
    const unsigned kt0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)kt);
    const unsigned kfrag = kt0 + (8 * (g16 >> 1) + q4) * kFpKS + (16 * (g16 & 1) + 4 * p4) * 2;   // + 16 s rows + 64 h2 bytes (+ 4 rows: the upper half)
    const char* vfrag = vt + col * kFpVS + 8 * kh;                                              // + dt 32 rows + (32 sp + 64 h2) bytes (+ 16: keys + 8)
    auto k_frag = [&](int part, int s, int h2) {
        const unsigned a = kfrag + (part * DR + 16 * s) * kFpKS + 64 * h2;
        const fa_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((fa_lds_s16x4*)(uintptr_t)a);
        const fa_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((fa_lds_s16x4*)(uintptr_t)(a + 4 * kFpKS));
        const fa_s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(fa_bf16x8, v);
    };
    auto v_frag = [&](int part, int dt, int sp, int h2) {
        const char* a = vfrag + (part * DR + dt * 32) * kFpVS + 32 * sp + 64 * h2;
        const uint2 x = *reinterpret_cast<const uint2*>(a), y = *reinterpret_cast<const uint2*>(a + 16);
        const uint4 v = {x.x, x.y, y.x, y.y};
        return __builtin_bit_cast(fa_bf16x8, v);
    };

    const int ntiles = (T + kFpKeys - 1) / kFpKeys;
    for (int jt = 0; jt < ntiles; ++jt) {
        const int jbase = jt * kFpKeys;
        store_tile(jbase);
        __syncthreads();
        if (jt + 1 < ntiles) load_tile(jbase + kFpKeys);
        if (active) {
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const int j0 = jbase + 32 * h2;
                if (j0 >= T) break;   // (wave-uniform: the 32-key step k_vits_flash_x3 would not have run either)
                f32x16 sacc;
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const fa_bf16x8 ah = k_frag(0, s, h2), al = k_frag(1, s, h2);
                    sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, qh[s], sacc, 0, 0, 0);
                    sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, ql[s], sacc, 0, 0, 0);
                    sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, qh[s], sacc, 0, 0, 0);
                }
                const bool diag = j0 <= i0 + 31 + w && j0 + 31 >= i0 - w;
                const bool tail = j0 + 32 > T;   // (uniform) only the utterance's last key step has keys to mask
                float mt = kFaNegBig;
                // three copies of the loop behind ONE wave-uniform branch: tested per element, `diag` and `tail` were 32 taken branches per key step
                // (a quarter of the step's issue time); interior steps, all but two or three per wave, are 16 multiplies and 16 maxima
                if (diag) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int j = j0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                        float sv = sacc[r] * qs2;
                        const int rr = j - i + w;
                        if (rr >= 0 && rr < nb && j < T) {
                            sv += rk_s[wave][rr][col];
                            band_s[wave][rr][col] = sv;
                        }
                        if (tail) sv = j < T ? sv : kFaNegBig;
                        sacc[r] = sv;
                        mt = fmaxf(mt, sv);
                    }
                } else if (tail) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int j = j0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                        const float sv = j < T ? sacc[r] * qs2 : kFaNegBig;
                        sacc[r] = sv;
                        mt = fmaxf(mt, sv);
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        sacc[r] *= qs2;
                        mt = fmaxf(mt, sacc[r]);
                    }
                }
                {
                float lo, hi;
                fa_halves(mt, lo, hi);
                mt = fmaxf(mt, kh ? lo : hi);
            }
                const float mn = fmaxf(m, mt);
                const float alpha = __builtin_amdgcn_exp2f(m - mn);
                float ps = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float e = __builtin_amdgcn_exp2f(sacc[r] - mn);
                    sacc[r] = e;
                    ps += e;
                }
                {
                    float lo, hi;
                    fa_halves(ps, lo, hi);
                    l = fmaf(l, alpha, lo + hi);
                }
                m = mn;
                // (the running maximum of most columns stops moving after the first key steps: multiplying by exactly 1 changes nothing, so the 48
                // multiplies are skipped whenever no lane of the wave has a new maximum: same bits)
                if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) cacc[dt][r] *= alpha;
                }
#pragma unroll
                for (int sp = 0; sp < 2; ++sp) {
                    float pv8[8];
#pragma unroll
                    for (int t = 0; t < 8; ++t) pv8[t] = sacc[8 * sp + t];
                    fa_bf16x8 ph, pl;
                    fa_split8(pv8, ph, pl);
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) {
                        const fa_bf16x8 vh = v_frag(0, dt, sp, h2), vl = v_frag(1, dt, sp, h2);
                        cacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl, ph, cacc[dt], 0, 0, 0);
                        cacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, pl, cacc[dt], 0, 0, 0);
                        cacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, ph, cacc[dt], 0, 0, 0);
                    }
                }
            }
        }
        __syncthreads();
    }
    if (!active) return;

    const float inv = 1.0f / l;
    float pb[kFaBand];
#pragma unroll
    for (int r = 0; r < kFaBand; ++r) pb[r] = __builtin_amdgcn_exp2f(band_s[wave][r][col] - m) * inv;
    float* Cg = ctx + (int64_t)g.head * dk * ldc + g.col0;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int d = dt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            float v = cacc[dt][r] * inv;
#pragma unroll
            for (int b = 0; b < kFaBand; ++b) v = fmaf(pb[b], erv_s[b * DR + d], v);
            if (i < T && d < dk) Cg[(int64_t)d * ldc + i] = v;
        }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The pre-split attention, software-pipelined (round 4): k_vits_flash_x3q.  Measured on k_vits_flash_x3 / x3p (ISA + kernel trace): a 32-key step
// costs a lone wave ~6 700 cycles for 1 152 cycles of MFMA: every phase waits for the one before it (fragment reads -> 18 chained MFMAs -> 190
// VALU instructions of softmax -> fragment reads -> 18 MFMAs), the staging registers (48) leave the compiler no room to read fragments ahead, and two
// waves per SIMD only interleave 1.3x.  Here
//   * K / V tiles (64 keys, both parts) go L2 -> LDS by LDS-DMA (global_load_lds_dwordx4), double-buffered, no staging registers and no staging
//     instructions; a tile image is [part][d][64 keys] with 128-byte rows, its 16-byte segments XOR-swizzled on the DMA's SOURCE address
//     (K: segment ^ 4 ((d >> 1) & 1): ds_read_b64_tr_b16 banks over 32-LANE groups (MI355X_MICROARCH.md's LDS table), whose lanes touch rows d .. d + 3
//     x four 16-byte segments: rows d, d + 2 share a bank half and take segments 0-3 / 4-7; rounds 4-5 swizzled by 2 ((d >> 1) & 3), right for 16-lane
//     groups, and measured a conflict ratio of 0.37.  V: segment ^ ((d >> 1) & 7): the 16 rows a 16-lane group of 8-byte reads touches fall on 16 distinct
//     bank pairs; over the 32 lanes of one k-half only 16 of a row pitch's 32 bank pairs are reachable: 2-way, which no segment swizzle removes);
//   * the wave's program is pipelined by one key step: QK(t + 1)'s 18 MFMAs are issued with the softmax of step t dealt into their gaps (scale / max,
//     exponentials, the split of P), then PV(t)'s 18 MFMAs, term-major over the three accumulators; fragment reads run two MFMA groups ahead;
//   * one barrier per key step (K tile u + 1 is requested at the barrier of step 2u - 1 and needed at that of step 2u + 1; V tile u + 1 at 2u / 2u + 2).
// The arithmetic and its order per accumulator are those of k_vits_flash_x3: same bits (tests/test_gpu_parity.py).
// The utterance's last tile, when T is no multiple of 64, is completed by a fix-up pass (keys >= T zero: V must not carry garbage; the segment that
// straddles T copied element-wise: the DMA of such a segment reads a clamped in-row address instead).
// ---------------------------------------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void fq_lds_t;
typedef const __attribute__((address_space(1))) void fq_gbl_t;
template <int OFF>
__device__ __forceinline__ fa_s16x4 fq_read_tr(unsigned addr) {
    fa_s16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF));
    return v;
}
template <int OFF>
__device__ __forceinline__ uint2 fq_read_b64(unsigned addr) {
    uint2 v;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF));
    return v;
}
template <int I, int N, class F>
__device__ __forceinline__ void fq_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        fq_static_for<I + 1, N>(f);
    }
}
template <int N>
__device__ __forceinline__ void fq_wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// DG: builder's timeline variant (a diagnostic instantiation, not launched by the product path): s_memtime at every barrier / phase boundary of every wave of workgroup (0, 0), kept in LDS
template <int DT, int NW, bool DG = false>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_vits_flash_x3q(const AttnGroup* groups, const float* Q, int ld, const __bf16* Kp, const __bf16* Vp,
                                                              int64_t pstride, int ldp, float* ctx, int ldc, int dk, const float* erk, const float* erv,
                                                              int w, float qscale, unsigned long long* stamps = nullptr) {
    constexpr int DR = DT * 32;
    constexpr int KS = DR / 16;            // bf16 k-steps over the head dimension
    constexpr int NB8 = DR / 8;            // 8-row DMA blocks per part (1 KB each)
    constexpr int PART = DR * 128;         // bytes of one part of a tile image
    constexpr int IMG = 2 * PART;          // one tile image
    constexpr int NI = 2 * NB8 / NW;       // DMAs per wave, matrix and tile
    static_assert((2 * NB8) % NW == 0 && NI >= 1, "a tile's DMA blocks are dealt evenly over the waves");
    constexpr int NTH = 64 * NW;
    extern __shared__ __attribute__((aligned(16))) char fq_smem[];
    // LDS: K images [2] | V images [2] | erk | erv | rk [NW] | band [NW]
    float* erk_s = reinterpret_cast<float*>(fq_smem + 4 * IMG);
    float* erv_s = erk_s + kFaBand * DR;
    float (*rk_s)[kFaBand][32] = reinterpret_cast<float (*)[kFaBand][32]>(erv_s + kFaBand * DR);
    float (*band_s)[kFaBand][32] = rk_s + NW;
    const unsigned lds_k0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)fq_smem);
    const unsigned lds_v0 = lds_k0 + 2 * IMG;
    const unsigned lds_dummy = lds_k0 + 4 * IMG + sizeof(float) * (2 * kFaBand * DR + 2 * NW * kFaBand * 32);   // NW x 64 floats
    // the lo parts of q: in LDS as B fragments [NW][KS][64 lanes][16 bytes] for the 4-wave workgroup (one wave per SIMD, the registers go to deeper
    // fragment prefetch); in registers for the 8-wave one (two waves per SIMD hide each other's fragment latency; its LDS holds 18 KB of band state)
    constexpr bool QL_LDS = NW == 4;
    constexpr int QL_BYTES = QL_LDS ? NW * KS * 1024 : 0;
    const unsigned lds_ql = lds_dummy + NW * 64 * 4;
    constexpr int kStampMax = 160;   // per wave
    const unsigned lds_st = lds_ql + QL_BYTES;
    int nst = 0;
    auto stamp = [&]() {
        if constexpr (DG) {
            if (nst < kStampMax) {
                const unsigned long long tm = __builtin_amdgcn_s_memtime();
                const unsigned a = lds_st + ((threadIdx.x >> 6) * kStampMax + nst) * 8;
                asm volatile("ds_write_b64 %0, %1" ::"v"(a), "v"(tm) : "memory");
                ++nst;
            }
        }
    };

    const AttnGroup g = groups[blockIdx.y];
    const int T = g.T;
    const int q0 = blockIdx.x * (32 * NW);
    if (q0 >= T) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31, kh = lane >> 5;
    const int i0 = q0 + wave * 32;
    const bool active = i0 < T;
    const int i = i0 + col;
    const int ic = min(i, T - 1);
    const float* Qg = Q + (int64_t)g.head * dk * ld + g.col0;
    const int64_t poff = (int64_t)g.head * dk * ldp + g.col0;
    const int nb = 2 * w + 1;
    const float qs2 = qscale * 1.4426950408889634f;   // base-2 logits, as in k_vits_flash_x3
    const int ntiles = (T + 63) >> 6, nsteps = (T + 31) >> 5;

    // ---- DMA: instruction i of this wave moves block e = wave + NW i (part e / NB8, rows 8 (e % NB8) .. + 7); lane -> row lane / 8, LDS segment lane % 8
    const int drow = lane >> 3, dseg = lane & 7;
    const int kgseg = dseg ^ (4 * ((drow >> 1) & 1)), vgseg = dseg ^ ((drow >> 1) & 3);   // the GLOBAL segment this lane's LDS segment holds (V: even blocks)
    const int nblk = dk >> 3;              // blocks that exist (dk is a multiple of 8); blocks beyond repeat the last one (their q / outputs are unused)
    // (V's swizzle takes bit 3 of the row as well: odd 8-row blocks hold global segment ^ 4.)  A DMA's address = a wave-uniform base (matrix, part,
    // block, tile: scalar registers, recomputed per request) + a 32-bit per-lane byte offset (row in the block, segment): no per-request 64-bit lane pointers
    // (hoisted out of the loop they were spilled, and every reload waited vmcnt(0): for the DMAs just issued, 2 400 cycles per K request)
    const unsigned k_loff = ((unsigned)drow * (unsigned)ldp + 8u * kgseg) * 2u;
    const unsigned v_loff0 = ((unsigned)drow * (unsigned)ldp + 8u * vgseg) * 2u, v_loff1 = ((unsigned)drow * (unsigned)ldp + 8u * (vgseg ^ 4)) * 2u;
    auto dma_tile = [&](const __bf16* base, unsigned loff_even, unsigned loff_odd, int gseg, int odd_xor, int u, unsigned img) {
        const bool whole = u * 64 + 64 <= T;   // (uniform) every segment of the tile lies inside the utterance
#pragma unroll
        for (int q = 0; q < NI; ++q) {
            const int e = wave + NW * q;
            const int part = e / NB8, blk = e - part * NB8;
            const char* ub = reinterpret_cast<const char*>(base + poff + (int64_t)part * pstride + (int64_t)min(blk, nblk - 1) * 8 * ldp + u * 64);   // wave-uniform
            const unsigned dst = img + part * PART + blk * 1024;
            if (whole) {
                __builtin_amdgcn_global_load_lds((fq_gbl_t*)(ub + ((blk & 1) ? loff_odd : loff_even)), (fq_lds_t*)(uintptr_t)dst, 16, 0, 0);
            } else {
                // a segment that is not whole inside the utterance reads a clamped in-row address instead (the fix-up pass rewrites it)
                const int js = 8 * (gseg ^ ((blk & 1) ? odd_xor : 0));
                const int jq = u * 64 + js;
                const int colq = jq + 8 <= T ? jq : max(min(jq, T - 8), 0);
                __builtin_amdgcn_global_load_lds((fq_gbl_t*)(ub + ((int64_t)drow * ldp + (colq - u * 64)) * 2), (fq_lds_t*)(uintptr_t)dst, 16, 0, 0);
            }
        }
    };
    auto dma_k = [&](int u) { dma_tile(Kp, k_loff, k_loff, kgseg, 0, u, lds_k0 + (u & 1) * IMG); };
    auto dma_v = [&](int u) { dma_tile(Vp, v_loff0, v_loff1, vgseg, 4, u, lds_v0 + (u & 1) * IMG); };
    // the utterance's last tile, T % 64 != 0: segments k0 / 8 .. 7 of every row rewritten (valid keys copied, the rest zero).  One (matrix, part, row) per
    // thread and pass: the <= 7 element loads of the segment that straddles T are in flight together, then 16-byte stores
    auto fixup_tail = [&]() {
        const int u = ntiles - 1, tl = T - u * 64, sg = tl >> 3, nv = tl & 7;
        for (int it = tid; it < 4 * DR; it += NTH) {
            const int d = it % DR, mp = it / DR, part = mp & 1, mat = mp >> 1;
            const unsigned short* src = reinterpret_cast<const unsigned short*>(mat ? Vp : Kp) + (int64_t)part * pstride + poff + (int64_t)min(d, dk - 1) * ldp + u * 64 + 8 * sg;
            unsigned short v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (e < nv && d < dk) ? src[e] : (unsigned short)0;
            const int swz = mat ? ((d >> 1) & 7) : 4 * ((d >> 1) & 1);
            char* row = fq_smem + (mat ? 2 * IMG : 0) + (u & 1) * IMG + part * PART + d * 128;
            const uint4 o = {(unsigned)v[0] | ((unsigned)v[1] << 16), (unsigned)v[2] | ((unsigned)v[3] << 16), (unsigned)v[4] | ((unsigned)v[5] << 16),
                             (unsigned)v[6] | ((unsigned)v[7] << 16)};
            *reinterpret_cast<uint4*>(row + ((sg ^ swz) << 4)) = o;
            for (int seg = sg + 1; seg < 8; ++seg) *reinterpret_cast<uint4*>(row + ((seg ^ swz) << 4)) = uint4{0u, 0u, 0u, 0u};
        }
    };
    const bool has_tail = (T & 63) != 0;
    const int t_fix = 2 * (ntiles - 1) - 1;   // the barrier at which the tail tile is completed (-1 = the one in front of the loop)

    dma_k(0);
    dma_v(0);
    for (int idx = tid; idx < kFaBand * DR; idx += NTH) {
        const int r = idx / DR, d = idx - r * DR;
        const bool in = r < nb && d < dk;
        erk_s[idx] = in ? erk[r * dk + d] : 0.f;
        erv_s[idx] = in ? erv[r * dk + d] : 0.f;
    }
    __syncthreads();   // erk_s / erv_s
    {   // relative-key logits (f32, as in k_vits_flash)
        float part[kFaBand];
#pragma unroll
        for (int r = 0; r < kFaBand; ++r) part[r] = 0.f;
#pragma unroll 1
        for (int sb = 0; sb < DR / 2; sb += 16) {
            float qd[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int d = 2 * (sb + u) + kh;
                const float x = Qg[(int64_t)min(d, dk - 1) * ld + ic];
                qd[u] = d < dk ? x : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int d = 2 * (sb + u) + kh;
#pragma unroll
                for (int r = 0; r < kFaBand; ++r) part[r] = fmaf(qd[u], erk_s[r * DR + d], part[r]);
            }
        }
#pragma unroll
        for (int r = 0; r < kFaBand; ++r) {
            float lo, hi;
            fa_halves(part[r], lo, hi);
            rk_s[wave][r][col] = (lo + hi) * qs2;
            band_s[wave][r][col] = kFaNegBig;
        }
    }
    // q: the hi parts stay in registers; the lo parts live in LDS (24 registers the pipelined loop needs), read with the K fragments of their k-step
    fa_bf16x8 qh[KS], qlr[QL_LDS ? 1 : KS];
    const unsigned ql_a = lds_ql + (wave * KS * 64 + lane) * 16;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        float v[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int d = 16 * s + 8 * kh + t;
            const float x = Qg[(int64_t)min(d, dk - 1) * ld + ic];
            v[t] = d < dk ? x : 0.f;
        }
        fa_bf16x8 qlo;
        fa_split8(v, qh[s], qlo);
        if constexpr (QL_LDS) asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(ql_a), "v"(qlo), "n"(s * 1024) : "memory");
        else qlr[s] = qlo;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    // ---- fragment addresses.  A byte offset inside a tile image = row * 128 + (segment << 4) + sub; the swizzles are XORs on the segment bits, so
    // the address of (key half h2, sp, keys + 8) is ONE per-lane base XOR a constant: no per-(h2, sp) address registers
    const int g16 = lane >> 4, q4 = (lane >> 2) & 3, p4 = lane & 3;
    const int krow = 8 * (g16 >> 1) + q4, ksegl = 2 * (g16 & 1) + (p4 >> 1);
    const unsigned kb_lo = krow * 128 + ((ksegl ^ (4 * ((krow >> 1) & 1))) << 4) + (p4 & 1) * 8;                  // rows d .. d + 3 (h2 = 0; h2 = 1: ^ 64)
    const unsigned kb_hi = (krow + 4) * 128 + ((ksegl ^ (4 * (((krow + 4) >> 1) & 1))) << 4) + (p4 & 1) * 8;      // rows d + 4 .. d + 7
    const unsigned vb = col * 128 + (((col >> 1) & 7) << 4) + 8 * kh;                                            // segment 0 (segment g: ^ (g << 4))
    // the wave's row of rk_s / band_s (+ rr * 128), and a dummy slot the band stores of elements outside the band go to
    const unsigned rk_a = (unsigned)(uintptr_t)((__attribute__((address_space(3))) float*)&rk_s[wave][0][col]);
    constexpr unsigned band_off = NW * kFaBand * 32 * 4;   // band_s[wave][rr][col] = rk_s[wave][rr][col] + band_off

    struct KF {
        fa_s16x4 hl, hh, ll, lh;   // hi part: rows d .. d + 3 | d + 4 .. d + 7; lo part
        fa_bf16x8 ql;              // the lo part of q of the same k-step
    };
    struct VF {
        uint2 h0, h1, l0, l1;      // hi part: keys .. + 3 | + 8 .. + 11; lo part
    };
    auto read_k = [&](KF& f, auto sc, unsigned alo, unsigned ahi) {
        constexpr int s = decltype(sc)::value;
        f.hl = fq_read_tr<s * 2048>(alo);
        f.hh = fq_read_tr<s * 2048>(ahi);
        f.ll = fq_read_tr<PART + s * 2048>(alo);
        f.lh = fq_read_tr<PART + s * 2048>(ahi);
        if constexpr (QL_LDS) {
            const unsigned qa = ql_a;   // (a local: an asm operand cannot name a captured variable)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f.ql) : "v"(qa), "n"(s * 1024));
        }
    };
    auto read_v = [&](VF& f, auto dtc, unsigned a0, unsigned a1) {
        constexpr int dt = decltype(dtc)::value;
        f.h0 = fq_read_b64<dt * 4096>(a0);
        f.h1 = fq_read_b64<dt * 4096>(a1);
        f.l0 = fq_read_b64<PART + dt * 4096>(a0);
        f.l1 = fq_read_b64<PART + dt * 4096>(a1);
    };
    auto k_hi = [](const KF& f) {
        const fa_s16x8 v = {f.hl[0], f.hl[1], f.hl[2], f.hl[3], f.hh[0], f.hh[1], f.hh[2], f.hh[3]};
        return __builtin_bit_cast(fa_bf16x8, v);
    };
    auto k_lo = [](const KF& f) {
        const fa_s16x8 v = {f.ll[0], f.ll[1], f.ll[2], f.ll[3], f.lh[0], f.lh[1], f.lh[2], f.lh[3]};
        return __builtin_bit_cast(fa_bf16x8, v);
    };
    auto v_hi = [](const VF& f) {
        const uint4 v = {f.h0.x, f.h0.y, f.h1.x, f.h1.y};
        return __builtin_bit_cast(fa_bf16x8, v);
    };
    auto v_lo = [](const VF& f) {
        const uint4 v = {f.l0.x, f.l0.y, f.l1.x, f.l1.y};
        return __builtin_bit_cast(fa_bf16x8, v);
    };
    // the reads are asm the compiler does not count: waits are explicit and tied to the registers they release
    auto wait_k = [](KF& f, auto nc) {
        if constexpr (QL_LDS) asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(f.hl), "+v"(f.hh), "+v"(f.ll), "+v"(f.lh), "+v"(f.ql) : "n"(decltype(nc)::value));
        else asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(f.hl), "+v"(f.hh), "+v"(f.ll), "+v"(f.lh) : "n"(decltype(nc)::value));
    };
    auto wait_v = [](VF& f, auto nc) {
        asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(f.h0), "+v"(f.h1), "+v"(f.l0), "+v"(f.l1) : "n"(decltype(nc)::value));
    };

    f32x16 cacc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) cacc[dt][r] = 0.f;
    float m = kFaNegBig, l = 0.f;

    // The barrier in front of step t (t = -1: in front of the loop).  Outstanding DMA groups of this wave, oldest first, when it arrives: the tile
    // this step needs (K tile (t + 1) / 2 for odd t, V tile t / 2 for even t: requested two barriers ago) and the one requested at the last barrier.
    auto step_barrier = [&](int t) {
        const int uy = t < 0 ? 1 : (t & 1 ? (t - 1) / 2 + 1 : t / 2 + 1);   // tile of the younger group (K tile for even t, V tile for odd t)
        if (t == t_fix && has_tail) {
            fq_wait_vm<0>();
            __builtin_amdgcn_s_barrier();
            fixup_tail();
            __syncthreads();
        } else {
            if (t >= 0 && uy < ntiles) fq_wait_vm<NI>();
            else fq_wait_vm<0>();
            __builtin_amdgcn_s_barrier();
        }
        // requests: odd t (and -1): K tile (t + 3) / 2; even t: V tile t / 2 + 1.  (Dealt into the step's first MFMA gaps instead they cost more than they hid:
        // 58 -> 76 us per launch for a single utterance: address arithmetic and a uniform branch per request inside the in-order MFMA / softmax stream.)
        if (t & 1) {
            const int u = (t + 3) / 2;
            if (u < ntiles) dma_k(u);
        } else {
            const int u = t / 2 + 1;
            if (u < ntiles) dma_v(u);
        }
    };

    // QK(step): 18 MFMAs into acc with `gap(n)` run behind MFMA n
    KF kf[2];   // (two sets: the reads of k-step s + 1 are issued behind the wait for s; three MFMA gaps full of VALU work cover their latency)
    auto qk = [&](f32x16& acc, int step, auto&& gap) {
        const unsigned kbase = lds_k0 + ((step >> 1) & 1) * IMG, hx = (step & 1) << 6;
        const unsigned alo = kbase + (kb_lo ^ hx), ahi = kbase + (kb_hi ^ hx);
        read_k(kf[0], std::integral_constant<int, 0>{}, alo, ahi);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        fq_static_for<0, 3 * KS>([&](auto nc) {
            constexpr int n = decltype(nc)::value;
            constexpr int s = n / 3, term = n % 3;
            KF& f = kf[s % 2];
            if constexpr (term == 0) {
                wait_k(f, std::integral_constant<int, 0>{});
                if constexpr (s + 1 < KS) read_k(kf[(s + 1) % 2], std::integral_constant<int, s + 1>{}, alo, ahi);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k_lo(f), qh[s], acc, 0, 0, 0);
            } else if constexpr (term == 1) {
                if constexpr (QL_LDS) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k_hi(f), f.ql, acc, 0, 0, 0);
                else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k_hi(f), qlr[s], acc, 0, 0, 0);
            } else {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k_hi(f), qh[s], acc, 0, 0, 0);
            }
            gap(nc);
            __builtin_amdgcn_sched_barrier(0);
        });
    };

    // One pipelined step.  `cur` holds the base-2 logits of step t (already scaled; mt = their maximum over this lane's 16 when the step is interior).
    // Phase A: QK(t + 1) -> `nxt` (when NEXT) with the softmax of step t and the split of P's first key half in its gaps.  Phase B: PV(t), term-paired,
    // with the split of P's second key half and the scaling / maximum of `nxt` in its gaps.
    VF vf[4];   // set g % 4 of the 2 DT (key half, row tile) groups of a step
    auto iter = [&](int t, f32x16& cur, f32x16& nxt, float& mt, auto nextc) {
        constexpr bool NEXT = decltype(nextc)::value;
        const int j0 = t * 32;
        float mn = 0.f, alpha = 0.f, ps = 0.f, mtn = kFaNegBig;
        fa_bf16x8 ph[2], pl[2];
        const unsigned va0 = lds_v0 + ((t >> 1) & 1) * IMG + (vb ^ ((t & 1) << 6));   // (sp = 0, keys + 0); sp: ^ 32; keys + 8: ^ 16
        const bool diag = j0 <= i0 + 31 + w && j0 + 31 >= i0 - w, tail = j0 + 32 > T;   // (wave-uniform)
        if (diag) {
            // the step touches the +-w band of this wave's queries: relative-key logits added, band scores kept for the relative-value term.  Branch
            // free (elements outside the band add nothing and store to a dummy slot), LDS through asm (a compiler-visible LDS access waits for every
            // pending LDS-DMA): 4 000-6 000 cycles per such step as compiled from the plain loop
            const int rr0 = j0 - i + w + 4 * kh;
            mt = kFaNegBig;
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {   // (two halves: eight values in flight)
                float rk[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int r = 8 * hf + q, ro = (r & 3) + 8 * (r >> 2);
                    const int rr = rr0 + ro;
                    const bool in = (unsigned)rr < (unsigned)nb && j0 + 4 * kh + ro < T;
                    const unsigned a = rk_a + (in ? rr : 0) * 128;
                    asm volatile("ds_read_b32 %0, %1" : "=v"(rk[q]) : "v"(a));
                }
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(rk[0]), "+v"(rk[1]), "+v"(rk[2]), "+v"(rk[3]), "+v"(rk[4]), "+v"(rk[5]), "+v"(rk[6]), "+v"(rk[7]));
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int r = 8 * hf + q, ro = (r & 3) + 8 * (r >> 2);
                    const int rr = rr0 + ro;
                    const bool jin = j0 + 4 * kh + ro < T;
                    const bool in = (unsigned)rr < (unsigned)nb && jin;
                    float sv = cur[r];
                    const float sb = sv + rk[q];
                    sv = in ? sb : sv;
                    const unsigned a = in ? rk_a + band_off + rr * 128 : lds_dummy + (wave * 64 + lane) * 4;
                    asm volatile("ds_write_b32 %0, %1" ::"v"(a), "v"(sv) : "memory");
                    if (tail) sv = jin ? sv : kFaNegBig;
                    cur[r] = sv;
                    mt = fmaxf(mt, sv);
                }
            }
        } else if (tail) {
            mt = kFaNegBig;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = j0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                const float sv = j < T ? cur[r] : kFaNegBig;
                cur[r] = sv;
                mt = fmaxf(mt, sv);
            }
        }
        auto read_vg = [&](auto gc) {   // the V fragments of group gc of PV(t)
            constexpr int gq = decltype(gc)::value;
            if constexpr (gq < 2 * DT) {
                constexpr int sp = gq / DT, dt = gq % DT;
                read_v(vf[gq % 4], std::integral_constant<int, dt>{}, va0 ^ (32 * sp), va0 ^ (32 * sp + 16));
            }
        };
        auto split2 = [&](auto spc, auto ec) {   // elements e, e + 1 of key half sp of P -> bf16 hi / lo
            constexpr int sp = decltype(spc)::value, e0 = decltype(ec)::value;
#pragma unroll
            for (int e = e0; e < e0 + 2; ++e) {
                const float x = cur[8 * sp + e];
                const __bf16 hh = (__bf16)x;
                ph[sp][e] = hh;
                pl[sp][e] = (__bf16)(x - (float)hh);
            }
            if constexpr (e0 == 6) asm volatile("" : "+v"(ph[sp]), "+v"(pl[sp]));   // (pinned here: sunk to its use it would run un-overlapped)
        };
        // phase A: the softmax of step t in 18 slices
        auto slice = [&](auto nc) {
            constexpr int n = decltype(nc)::value;
            if constexpr (n == 0) {
                float lo, hi;
                fa_halves(mt, lo, hi);
                mt = fmaxf(mt, kh ? lo : hi);
                mn = fmaxf(m, mt);
                alpha = __builtin_amdgcn_exp2f(m - mn);
            } else if constexpr (n < 9) {
#pragma unroll
                for (int r = 2 * (n - 1); r < 2 * (n - 1) + 2; ++r) {
                    const float e = __builtin_amdgcn_exp2f(cur[r] - mn);
                    cur[r] = e;
                    ps += e;
                }
            } else if constexpr (n == 9) {
                float lo, hi;
                fa_halves(ps, lo, hi);
                l = fmaf(l, alpha, lo + hi);
                m = mn;
            } else if constexpr (n < 14) {
                split2(std::integral_constant<int, 0>{}, std::integral_constant<int, 2 * (n - 10)>{});
            } else if constexpr (n >= 15) {
                read_vg(std::integral_constant<int, n - 15>{});   // groups 0 .. 2 (behind QK's last fragment wait)
            }
        };
        if constexpr (NEXT) {
            qk(nxt, t + 1, slice);
            if constexpr (3 * KS < 18) fq_static_for<3 * KS, 18>([&](auto nc) { slice(nc); });
        } else {
            fq_static_for<0, 18>([&](auto nc) { slice(nc); });
        }
        read_vg(std::integral_constant<int, 3>{});
        stamp();
        if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) cacc[dt][r] *= alpha;
        }
        stamp();
        // phase B: PV(t).  Groups g = (key half sp, row tile dt) = 0 .. 2 DT - 1, three MFMAs each (lo * hi, hi * lo, hi * hi on accumulator dt), in PAIRS with
        // their MFMAs interleaved (two independent accumulators back to back).  Four fragment sets: groups 0 .. 3 are on their way when the phase starts,
        // groups 4 and 5 are requested behind the last MFMA of groups 0 and 1.
        constexpr int NG = 2 * DT;
        auto pv_mfma = [&](auto gc, auto termc) {
            constexpr int gq = decltype(gc)::value, term = decltype(termc)::value;
            constexpr int sp = gq / DT, dt = gq % DT;
            const VF& f = vf[gq % 4];
            if constexpr (term == 0) cacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v_lo(f), ph[sp], cacc[dt], 0, 0, 0);
            else if constexpr (term == 1) cacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v_hi(f), pl[sp], cacc[dt], 0, 0, 0);
            else cacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v_hi(f), ph[sp], cacc[dt], 0, 0, 0);
        };
        auto wait_vg = [&](auto gc) {   // groups requested behind group g when its wait is reached: up to group min(NG - 1, g < 2 ? 3 : 5)
            constexpr int gq = decltype(gc)::value;
            constexpr int last = (gq < 2 ? 3 : 5) < NG - 1 ? (gq < 2 ? 3 : 5) : NG - 1;
            wait_v(vf[gq % 4], std::integral_constant<int, 4 * (last - gq)>{});
        };
        // gap n of phase B (behind its MFMA n): the second key half of P, then the next step's logits scaled and their maximum
        auto bgap = [&](auto nc) {
            constexpr int n = decltype(nc)::value;
            if constexpr (n < 4) {
                split2(std::integral_constant<int, 1>{}, std::integral_constant<int, 2 * n>{});
            } else if constexpr (NEXT && n >= 6 && n < 14) {
#pragma unroll
                for (int r = 2 * (n - 6); r < 2 * (n - 6) + 2; ++r) {
                    nxt[r] *= qs2;
                    mtn = fmaxf(mtn, nxt[r]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        if constexpr (DT == 1) {   // one accumulator: its two groups in order (the second key half of P split in front of them)
            fq_static_for<0, 4>([&](auto nc) { split2(std::integral_constant<int, 1>{}, std::integral_constant<int, 2 * decltype(nc)::value>{}); });
            wait_vg(std::integral_constant<int, 0>{});
            fq_static_for<0, 3>([&](auto tc) { pv_mfma(std::integral_constant<int, 0>{}, tc); });
            wait_vg(std::integral_constant<int, 1>{});
            fq_static_for<0, 3>([&](auto tc) { pv_mfma(std::integral_constant<int, 1>{}, tc); });
            fq_static_for<6, 14>([&](auto nc) { bgap(nc); });
        } else {
            fq_static_for<0, NG / 2>([&](auto pc) {
                constexpr int p = decltype(pc)::value, g0 = 2 * p, g1 = g0 + 1;
                wait_vg(std::integral_constant<int, g0>{});
                pv_mfma(std::integral_constant<int, g0>{}, std::integral_constant<int, 0>{});
                bgap(std::integral_constant<int, 6 * p>{});
                wait_vg(std::integral_constant<int, g1>{});
                pv_mfma(std::integral_constant<int, g1>{}, std::integral_constant<int, 0>{});
                bgap(std::integral_constant<int, 6 * p + 1>{});
                pv_mfma(std::integral_constant<int, g0>{}, std::integral_constant<int, 1>{});
                bgap(std::integral_constant<int, 6 * p + 2>{});
                pv_mfma(std::integral_constant<int, g1>{}, std::integral_constant<int, 1>{});
                bgap(std::integral_constant<int, 6 * p + 3>{});
                pv_mfma(std::integral_constant<int, g0>{}, std::integral_constant<int, 2>{});
                if constexpr (p == 0) read_vg(std::integral_constant<int, 4>{});      // into group 0's set
                bgap(std::integral_constant<int, 6 * p + 4>{});
                pv_mfma(std::integral_constant<int, g1>{}, std::integral_constant<int, 2>{});
                if constexpr (p == 0) read_vg(std::integral_constant<int, 5>{});      // into group 1's set
                bgap(std::integral_constant<int, 6 * p + 5>{});
            });
            if constexpr (3 * NG < 14) fq_static_for<3 * NG, 14>([&](auto nc) { bgap(nc); });
        }
        mt = mtn;   // (of the next step, when it is interior; a step on the band or at the utterance's end computes its own)
    };

    // steps 0 .. nsteps - 2 with QK of their successor in flight, two per loop iteration (the two accumulators swap roles); the last one on its own
    f32x16 sa, sb;
    float mt = kFaNegBig;
    stamp();
    step_barrier(-1);
    stamp();
    if (active) {
        qk(sa, 0, [](auto) {});
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            sa[r] *= qs2;
            mt = fmaxf(mt, sa[r]);
        }
    }
    const int n1 = nsteps - 1;
    int t = 0;
    for (; t + 2 <= n1; t += 2) {
        stamp();
        step_barrier(t);
        stamp();
        if (active) iter(t, sa, sb, mt, std::true_type{});
        stamp();
        step_barrier(t + 1);
        stamp();
        if (active) iter(t + 1, sb, sa, mt, std::true_type{});
    }
    if (t < n1) {
        stamp();
        step_barrier(t);
        stamp();
        if (active) iter(t, sa, sb, mt, std::true_type{});
        stamp();
        step_barrier(t + 1);
        stamp();
        if (active) iter(t + 1, sb, sa, mt, std::false_type{});
    } else {
        stamp();
        step_barrier(t);
        stamp();
        if (active) iter(t, sa, sb, mt, std::false_type{});
    }
    stamp();
    if constexpr (DG) {
        __syncthreads();
        if (blockIdx.x == 0 && blockIdx.y == 0 && stamps)
            for (int q = threadIdx.x; q < NW * kStampMax; q += NTH) stamps[q] = q % kStampMax < nst ? reinterpret_cast<unsigned long long*>(fq_smem + 4 * IMG + sizeof(float) * (2 * kFaBand * DR + 2 * NW * kFaBand * 32) + NW * 64 * 4 + QL_BYTES)[q] : 0ull;
    }
    if (!active) return;

    const float inv = 1.0f / l;
    float pb[kFaBand];
#pragma unroll
    for (int r = 0; r < kFaBand; ++r) pb[r] = __builtin_amdgcn_exp2f(band_s[wave][r][col] - m) * inv;
    float* Cg = ctx + (int64_t)g.head * dk * ldc + g.col0;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int d = dt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            float v = cacc[dt][r] * inv;
#pragma unroll
            for (int b = 0; b < kFaBand; ++b) v = fmaf(pb[b], erv_s[b * DR + d], v);
            if (i < T && d < dk) Cg[(int64_t)d * ldc + i] = v;
        }
}

template <int DT, int NW>
void launch_flash_x3q(const AttnGroup* groups, int ngroups, int maxT, const float* Q, int ld, const SplitPlanes& kv, int k_row0, int v_row0, float* ctx,
                      int ldc, int dk, const float* erk, const float* erv, int window, float qscale, hipStream_t s) {
    constexpr int DR = DT * 32;
    constexpr size_t lds = 4 * 2 * DR * 128 + sizeof(float) * (2 * kFaBand * DR + 2 * NW * kFaBand * 32) + NW * 64 * 4 + (NW == 4 ? NW * (DR / 16) * 1024 : 0);
    const __bf16* base = static_cast<const __bf16*>(kv.p);
    const dim3 grid((maxT + 32 * NW - 1) / (32 * NW), ngroups);
    // (the timeline instantiation <DT, NW, true> of round 4 - s_memtime stamps at every phase boundary, profiles/r04i_flash_attention_timeline.txt - is not
    // launched by the library any more)
    auto kern = k_vits_flash_x3q<DT, NW>;
    static std::atomic<uint64_t> lds_allowed{0};   // per (kernel instantiation, device)
    allow_full_lds(reinterpret_cast<const void*>(kern), lds_allowed);
    hipLaunchKernelGGL(kern, grid, dim3(64 * NW), lds, s, groups, Q, ld, base + (int64_t)k_row0 * kv.ld, base + (int64_t)v_row0 * kv.ld, kv.pstride, kv.ld,
                       ctx, ldc, dk, erk, erv, window, qscale, (unsigned long long*)nullptr);
}

template <int DT>
void launch_flash_x3p(dim3 grid, const AttnGroup* groups, const float* Q, int ld, const SplitPlanes& kv, int k_row0, int v_row0, float* ctx, int ldc,
                      int dk, const float* erk, const float* erv, int window, float qscale, hipStream_t s) {
    constexpr int DR = DT * 32;
    constexpr size_t lds = 2 * DR * (kFpKS + kFpVS) + sizeof(float) * (2 * kFaBand * DR + 2 * 4 * kFaBand * 32);
    auto kern = k_vits_flash_x3p<DT>;
    static std::atomic<uint64_t> lds_allowed{0};   // per (kernel instantiation, device)
    allow_full_lds(reinterpret_cast<const void*>(kern), lds_allowed);
    const __bf16* base = static_cast<const __bf16*>(kv.p);
    hipLaunchKernelGGL(kern, grid, dim3(kFaThreads), lds, s, groups, Q, ld, base + (int64_t)k_row0 * kv.ld, base + (int64_t)v_row0 * kv.ld, kv.pstride,
                       kv.ld, ctx, ldc, dk, erk, erv, window, qscale);
}
}  // namespace

// the software-pipelined kernel (k_vits_flash_x3q) wherever its DMA blocks fit (head dimensions that are multiples of 8)
bool flash_pipelined_usable(int dk) { return (dk & 7) == 0 && dk <= 96; }

// keys / values from bf16 hi / lo planes (rows k_row0 .. + heads dk and v_row0 .. of kv: two bf16 parts, the columns of Q's plane)
void vits_flash_attention_parts(const AttnGroup* groups, int ngroups, int maxT, const float* Q, int ld, const SplitPlanes& kv, int k_row0,
                                int v_row0, float* ctx, int ldc, int dk, const float* erk, const float* erv, int window, float qscale,
                                hipStream_t s, int pipelined) {
    SBV2_REQUIRE(window <= kFaMaxWin, "relative attention window larger than the compiled maximum");
    SBV2_REQUIRE(dk >= 2 && dk <= 96 && (dk & 1) == 0, "flash attention: head dimension must be even and <= 96");
    SBV2_REQUIRE(kv.parts == 2 && !kv.f16 && (kv.ld & 3) == 0, "flash attention: keys / values must be two bf16 parts");
    if (ngroups <= 0 || maxT <= 0) return;
    if (pipelined && flash_pipelined_usable(dk)) {
        // 128-query workgroups (4 waves, one per SIMD) while they leave at most one workgroup per CU; beyond, 256-query workgroups of 8 waves (two per SIMD)
        const bool wide = pipelined == 2 || (int64_t)((maxT + 127) / 128) * ngroups > device_cu_count();   // (2: the test forces the 8-wave shape on small batches)
        if (wide) {
            if (dk <= 32) launch_flash_x3q<1, 8>(groups, ngroups, maxT, Q, ld, kv, k_row0, v_row0, ctx, ldc, dk, erk, erv, window, qscale, s);
            else if (dk <= 64) launch_flash_x3q<2, 8>(groups, ngroups, maxT, Q, ld, kv, k_row0, v_row0, ctx, ldc, dk, erk, erv, window, qscale, s);
            else launch_flash_x3q<3, 8>(groups, ngroups, maxT, Q, ld, kv, k_row0, v_row0, ctx, ldc, dk, erk, erv, window, qscale, s);
        } else if (dk <= 32) launch_flash_x3q<1, 4>(groups, ngroups, maxT, Q, ld, kv, k_row0, v_row0, ctx, ldc, dk, erk, erv, window, qscale, s);
        else if (dk <= 64) launch_flash_x3q<2, 4>(groups, ngroups, maxT, Q, ld, kv, k_row0, v_row0, ctx, ldc, dk, erk, erv, window, qscale, s);
        else launch_flash_x3q<3, 4>(groups, ngroups, maxT, Q, ld, kv, k_row0, v_row0, ctx, ldc, dk, erk, erv, window, qscale, s);
        HIP_CHECK(hipGetLastError());
        return;
    }
    const dim3 grid((maxT + 127) / 128, ngroups);
    if (dk <= 32) launch_flash_x3p<1>(grid, groups, Q, ld, kv, k_row0, v_row0, ctx, ldc, dk, erk, erv, window, qscale, s);
    else if (dk <= 64) launch_flash_x3p<2>(grid, groups, Q, ld, kv, k_row0, v_row0, ctx, ldc, dk, erk, erv, window, qscale, s);
    else launch_flash_x3p<3>(grid, groups, Q, ld, kv, k_row0, v_row0, ctx, ldc, dk, erk, erv, window, qscale, s);
    HIP_CHECK(hipGetLastError());
}

void vits_flash_attention(const AttnGroup* groups, int ngroups, int maxT, const float* Q, const float* K, const float* V, int ld, float* ctx,
                          int ldc, int dk, const float* erk, const float* erv, int window, float qscale, bool split_bf16, hipStream_t s) {
    SBV2_REQUIRE(window <= kFaMaxWin, "relative attention window larger than the compiled maximum");
    SBV2_REQUIRE(dk >= 2 && dk <= 96 && (dk & 1) == 0, "flash attention: head dimension must be even and <= 96");
    if (ngroups <= 0 || maxT <= 0) return;
    const dim3 grid((maxT + 127) / 128, ngroups), block(kFaThreads);
    if (split_bf16) {   // the flow: split-bf16 matrix cores (exact f32 is reserved for what decides the integer durations)
        if (dk <= 32) hipLaunchKernelGGL(k_vits_flash_x3<1>, grid, block, 0, s, groups, Q, K, V, ld, ctx, ldc, dk, erk, erv, window, qscale);
        else if (dk <= 64) hipLaunchKernelGGL(k_vits_flash_x3<2>, grid, block, 0, s, groups, Q, K, V, ld, ctx, ldc, dk, erk, erv, window, qscale);
        else hipLaunchKernelGGL(k_vits_flash_x3<3>, grid, block, 0, s, groups, Q, K, V, ld, ctx, ldc, dk, erk, erv, window, qscale);
    } else if (dk <= 32) hipLaunchKernelGGL(k_vits_flash<1>, grid, block, 0, s, groups, Q, K, V, ld, ctx, ldc, dk, erk, erv, window, qscale);
    else if (dk <= 64) hipLaunchKernelGGL(k_vits_flash<2>, grid, block, 0, s, groups, Q, K, V, ld, ctx, ldc, dk, erk, erv, window, qscale);
    else hipLaunchKernelGGL(k_vits_flash<3>, grid, block, 0, s, groups, Q, K, V, ld, ctx, ldc, dk, erk, erv, window, qscale);
    HIP_CHECK(hipGetLastError());
}

}  // namespace sbv2
