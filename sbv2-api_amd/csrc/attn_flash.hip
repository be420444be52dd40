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

#include "common.h"
#include "ops.h"

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
                for (int r = 0; r < kFaBand; ++r) part[r] += qd[u] * erk_s[r * DR + d];
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
                l = l * alpha + (lo + hi);
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
            for (int b = 0; b < kFaBand; ++b) v += pb[b] * erv_s[b * DR + d];
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
                for (int r = 0; r < kFaBand; ++r) part[r] += qd[u] * erk_s[r * DR + d];
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
                l = l * alpha + (lo + hi);
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
            for (int b = 0; b < kFaBand; ++b) v += pb[b] * erv_s[b * DR + d];
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
                for (int r = 0; r < kFaBand; ++r) part[r] += qd[u] * erk_s[r * DR + d];
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
                    l = l * alpha + (lo + hi);
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
            for (int b = 0; b < kFaBand; ++b) v += pb[b] * erv_s[b * DR + d];
            if (i < T && d < dk) Cg[(int64_t)d * ldc + i] = v;
        }
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

// keys / values from bf16 hi / lo planes (rows k_row0 .. + heads dk and v_row0 .. of kv: two bf16 parts, the columns of Q's plane)
void vits_flash_attention_parts(const AttnGroup* groups, int ngroups, int maxT, const float* Q, int ld, const SplitPlanes& kv, int k_row0,
                                int v_row0, float* ctx, int ldc, int dk, const float* erk, const float* erv, int window, float qscale,
                                hipStream_t s) {
    SBV2_REQUIRE(window <= kFaMaxWin, "relative attention window larger than the compiled maximum");
    SBV2_REQUIRE(dk >= 2 && dk <= 96 && (dk & 1) == 0, "flash attention: head dimension must be even and <= 96");
    SBV2_REQUIRE(kv.parts == 2 && !kv.f16 && (kv.ld & 3) == 0, "flash attention: keys / values must be two bf16 parts");
    if (ngroups <= 0 || maxT <= 0) return;
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
