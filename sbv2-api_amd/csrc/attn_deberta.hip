// Fused DeBERTa-v2 disentangled self-attention for short sequences (T <= 64 tokens per utterance, head dim <= 64), exact f32 MFMA.
//
// Reference arithmetic (transformers modeling_deberta_v2.py:232-253, 276-346, share_att_key, pos_att_type = c2p | p2c):
//
//   s[i][j] = ( q_i . k_j  +  q_i . posk[ clamp(bucket(i - j) + span) ]  +  k_j . posq[ clamp(-bucket(j - i) + span) ] ) / sqrt(3 d)
//   p = softmax_j(s)  (masked pairs: -FLT_MAX before the softmax),   ctx_i = sum_j p[i][j] v_j
//
// The unfused path runs this as four grouped GEMM launches (S^T = K^T Q, c2p = posK^T Q, p2c = K^T posQ, ctx = V P^T) and a softmax /
// gather launch per layer: 110 launches of 64 x 64 (x 127) problems per forward, ~6 ms of an 18 ms DeBERTa at batch 32 and most of its
// launch count at batch 1.  Here one workgroup owns one (utterance, head): the two relative-position products go to LDS once
// (c2p^T [window][query], p2c [key][window]; the window is the range of bucket indices a sequence of this length can reach, <= 128),
// each wave computes a 32 x 32 tile of S^T = K^T Q, gathers the two bias terms by bucket index, the column softmax is combined
// across the two key tiles through LDS, P goes to LDS (over the c2p buffer) and each wave computes one 32 x 32 tile of ctx = V P^T.
// Operands are read straight from the k-major planes (they are L2 resident; one value per lane and k-step is exactly the MFMA's
// operand shape).  Fixed summation order: results do not depend on the batch composition.
#include <cfloat>

#include "common.h"
#include "ops.h"

namespace sbv2 {

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {
constexpr int kDaT = 64;         // max tokens
constexpr int kDaW = 128;        // max window of relative-position buckets
constexpr int kDaPc = kDaT + 1;  // pitch of c2p_s / p_s rows  (floats)
constexpr int kDaPp = kDaW + 1;  // pitch of p2c_s rows

__device__ __forceinline__ int acc_row(int r, int kh) { return (r & 3) + 8 * (r >> 2) + 4 * kh; }

__global__ __launch_bounds__(256) void k_deberta_attn(const AttnGroup* groups, const float* Q, const float* K, int ld, const float* V,
                                                      const float* posk, const float* posq, int ldp, int win_lo, int wlen, const int* tab,
                                                      int tab_center, int span, float inv_scale, const unsigned char* tok_mask, int dh,
                                                      float* ctx, int ldc) {
    extern __shared__ __attribute__((aligned(16))) float da_smem[];   // 67.8 KB: above the 64 KB static limit
    float* c2p_s = da_smem;                          // [w][i]; reused for P[j][i] after the scores are formed
    float* p2c_s = c2p_s + kDaW * kDaPc;             // [j][w]
    float (*redm)[kDaT] = reinterpret_cast<float (*)[kDaT]>(p2c_s + kDaT * kDaPp);
    float (*reds)[kDaT] = redm + 2;
    int* tab_s = reinterpret_cast<int*>(reds + 2);   // [2 * kDaT]
    const AttnGroup g = groups[blockIdx.x];
    const int T = g.T;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31, kh = lane >> 5;
    const int64_t hoff = (int64_t)g.head * dh;
    const float* Qg = Q + hoff * ld + g.col0;
    const float* Kg = K + hoff * ld + g.col0;
    const float* Pk = posk + hoff * ldp + win_lo;
    const float* Pq = posq + hoff * ldp + win_lo;
    const float* Vg = V + hoff * ld + g.col0;   // k-major like q and k: rows = channels
    const int ns = dh >> 1;   // k-steps over the head dimension

    if (tid < 2 * T - 1) tab_s[tid] = tab[tab_center - (T - 1) + tid];   // tab_s[(i - j) + T - 1] = bucket(i - j)
    if (tid < 2 * kDaT) (&redm[0][0])[tid] = -FLT_MAX;

    auto tile_product = [&](const float* Abase, int lda_, int arow, int amax, const float* Bbase, int ldb_, int bcol, int bmax) {
        // 32 x 32 tile of sum_d A[d][arow + row] * B[d][bcol + col]; rows / columns beyond amax / bmax read as zero
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const int ar = arow + col, bc = bcol + col;
        const bool aok = ar < amax, bok = bc < bmax;
        const float* ap = Abase + min(ar, amax - 1);
        const float* bp = Bbase + min(bc, bmax - 1);
        float av[32], bv[32];
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            const int dd = min(2 * s + kh, dh - 1);
            av[s] = ap[(int64_t)dd * lda_];
            bv[s] = bp[(int64_t)dd * ldb_];
        }
#pragma unroll
        for (int s = 0; s < 32; ++s)
            if (s < ns) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aok ? av[s] : 0.f, bok ? bv[s] : 0.f, acc, 0, 0, 0);
        return acc;
    };

    // ---- c2p^T[w][i] = sum_d posk[d][w] q[d][i]   (8 tiles: wave -> window tile `wave`, both query tiles) ----------------------
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const f32x16 a = tile_product(Pk, ldp, wave * 32, wlen, Qg, ld, it * 32, T);
#pragma unroll
        for (int r = 0; r < 16; ++r) c2p_s[(wave * 32 + acc_row(r, kh)) * kDaPc + it * 32 + col] = a[r];
    }
    // ---- p2c[j][w] = sum_d k[d][j] posq[d][w]     (8 tiles: wave -> key tile wave & 1, window tiles 2 (wave >> 1) + {0, 1}) --------
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int jt = wave & 1, wt = 2 * (wave >> 1) + t;
        const f32x16 a = tile_product(Kg, ld, jt * 32, T, Pq, ldp, wt * 32, wlen);
#pragma unroll
        for (int r = 0; r < 16; ++r) p2c_s[(jt * 32 + acc_row(r, kh)) * kDaPp + wt * 32 + col] = a[r];
    }
    __syncthreads();

    // ---- scores: wave -> (key tile jt, query tile it) ----------------------------------------------------------------------------
    const int jt = wave >> 1, it = wave & 1;
    const int i = it * 32 + col;
    const bool iok = i < T;
    const int ic = min(i, T - 1);
    f32x16 sacc = tile_product(Kg, ld, jt * 32, T, Qg, ld, it * 32, T);
    const bool mi = tok_mask[g.col0 + ic] != 0;
    const int hi = 2 * span - 1;
    float mloc = -FLT_MAX;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int j = jt * 32 + acc_row(r, kh);
        const int jc = min(j, T - 1);
        const int d1 = min(max(tab_s[(ic - jc) + T - 1] + span, 0), hi) - win_lo;
        const int d2 = min(max(-tab_s[(jc - ic) + T - 1] + span, 0), hi) - win_lo;
        float v = sacc[r] * inv_scale + c2p_s[d1 * kDaPc + ic] * inv_scale + p2c_s[jc * kDaPp + d2] * inv_scale;
        if (!(mi && tok_mask[g.col0 + jc])) v = -FLT_MAX;
        sacc[r] = v;
        if (j < T) mloc = fmaxf(mloc, v);
    }
    mloc = fmaxf(mloc, __shfl_xor(mloc, 32));
    if (kh == 0 && jt * 32 < T) redm[jt][i] = mloc;
    __syncthreads();   // (also: every wave is done reading c2p_s, which P overwrites below)
    const float mx = fmaxf(redm[0][i], redm[1][i]);
    float sloc = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int j = jt * 32 + acc_row(r, kh);
        const float e = j < T ? expf(sacc[r] - mx) : 0.f;
        sacc[r] = e;
        sloc += e;
    }
    {
        const float other = __shfl_xor(sloc, 32);
        sloc = kh ? other + sloc : sloc + other;   // (rows of half 0) + (rows of half 1) in both halves
    }
    if (kh == 0) reds[jt][i] = sloc;
    __syncthreads();
    const float sum = reds[0][i] + reds[1][i];
    float* p_s = c2p_s;   // P[j][i]
#pragma unroll
    for (int r = 0; r < 16; ++r) p_s[(jt * 32 + acc_row(r, kh)) * kDaPc + i] = sacc[r] / sum;
    __syncthreads();

    // ---- ctx[dd][i] = sum_j v[dd][j] p[j][i]: wave -> (channel tile ddt, query tile it) ---------------------------------------------
    // V is a k-major plane (rows = channels): the A operand wants, per lane, channel row `col` at key j.  Read coalesced (lanes along j)
    // into the p2c buffer, which is free now, and picked up transposed from LDS.
    float* v_s = p2c_s;   // [dd][j], pitch kDaPc
    for (int idx = tid; idx < 64 * 64; idx += 256) {
        const int dd = idx >> 6, j = idx & 63;
        v_s[dd * kDaPc + j] = (dd < dh && j < T) ? Vg[(int64_t)dd * ld + j] : 0.f;
    }
    __syncthreads();
    const int ddt = wave >> 1;
    if (ddt * 32 < dh) {
        f32x16 cacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) cacc[r] = 0.f;
        const float* vrow = v_s + (ddt * 32 + col) * kDaPc;   // A row: channel
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            const int j = 2 * s + kh;
            if (2 * s < T) cacc = __builtin_amdgcn_mfma_f32_32x32x2f32(vrow[j], p_s[j * kDaPc + i], cacc, 0, 0, 0);
        }
        float* Cg = ctx + hoff * ldc + g.col0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int d2 = ddt * 32 + acc_row(r, kh);
            if (d2 < dh && iok) Cg[(int64_t)d2 * ldc + i] = cacc[r];
        }
    }
}

// ---- the same attention for 65 .. 128 tokens (the reference's TensorRT profile allows 100: crates/sbv2_core/src/model.rs:15) -----------
// 4 x 4 score tiles on 8 waves (two key tiles of one query tile per wave).  The relative-position products of a whole (utterance, head)
// no longer fit LDS (c2p^T alone is 256 x 128 floats), but a 32 x 32 score tile only needs the 63 bucket indices its (i - j) range
// reaches (the bucket function is monotone with slope <= 1), so every wave computes, per score tile, a 64 x 32 piece of c2p^T and a 32 x 64
// piece of p2c into its private 8.4 KB scratch, gathers the bias terms from there and moves on.  P ([128][129] floats) and the scratch
// fill 135 KB of LDS; V is staged over the scratch once the scores are done.  Arithmetic per element as in k_deberta_attn.
constexpr int kDbT = 128;
constexpr int kDbPp = kDbT + 1;          // pitch of P / V rows
constexpr int kDbScr = 64 * 33;          // floats of per-wave scratch (c2p piece [64][33]; p2c piece [32][65] = 2080 fits)

__global__ __launch_bounds__(512) void k_deberta_attn128(const AttnGroup* groups, const float* Q, const float* K, int ld, const float* V,
                                                         const float* posk, const float* posq, int ldp, int win_lo, int wlen, const int* tab,
                                                         int tab_center, int span, float inv_scale, const unsigned char* tok_mask, int dh,
                                                         float* ctx, int ldc) {
    extern __shared__ __attribute__((aligned(16))) float db_smem[];
    float* p_s = db_smem;                                   // P[j][i]
    float* scr = p_s + kDbT * kDbPp;                        // [8 waves][kDbScr]; V[dd][j] after the scores
    float (*redm)[kDbT] = reinterpret_cast<float (*)[kDbT]>(scr + 8 * kDbScr);
    float (*reds)[kDbT] = redm + 4;
    int* tab_s = reinterpret_cast<int*>(reds + 4);          // [2 * kDbT]
    const AttnGroup g = groups[blockIdx.x];
    const int T = g.T;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31, kh = lane >> 5;
    const int64_t hoff = (int64_t)g.head * dh;
    const float* Qg = Q + hoff * ld + g.col0;
    const float* Kg = K + hoff * ld + g.col0;
    const float* Pk = posk + hoff * ldp + win_lo;
    const float* Pq = posq + hoff * ldp + win_lo;
    const float* Vg = V + hoff * ld + g.col0;
    const int ns = dh >> 1;
    const int hi = 2 * span - 1;

    if (tid < 2 * T - 1) tab_s[tid] = tab[tab_center - (T - 1) + tid];   // tab_s[(i - j) + T - 1] = bucket(i - j)
    for (int e = tid; e < 4 * kDbT; e += 512) (&redm[0][0])[e] = -FLT_MAX;
    __syncthreads();

    auto tile_product = [&](const float* Abase, int lda_, int arow, int amax, const float* Bbase, int ldb_, int bcol, int bmax) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const int ar = arow + col, bc = bcol + col;
        const bool aok = ar >= 0 && ar < amax, bok = bc >= 0 && bc < bmax;
        const float* ap = Abase + min(max(ar, 0), amax - 1);
        const float* bp = Bbase + min(max(bc, 0), bmax - 1);
        // operands eight k-steps at a time (the 64-token kernel preloads all 32: with two score tiles per wave that spills)
#pragma unroll 1
        for (int s0 = 0; s0 < ns; s0 += 8) {
            float av[8], bv[8];
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const int dd = min(2 * (s0 + s) + kh, dh - 1);
                av[s] = ap[(int64_t)dd * lda_];
                bv[s] = bp[(int64_t)dd * ldb_];
            }
#pragma unroll
            for (int s = 0; s < 8; ++s)
                if (s0 + s < ns) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aok ? av[s] : 0.f, bok ? bv[s] : 0.f, acc, 0, 0, 0);
        }
        return acc;
    };
    auto widx = [&](int rel, bool neg) {   // window index of bucket(rel) (c2p) or of -bucket(-rel) (p2c)
        const int b = neg ? -tab_s[-rel + T - 1] : tab_s[rel + T - 1];
        return min(max(b + span, 0), hi) - win_lo;
    };

    // ---- scores: wave -> query tile it, key tiles jt0 and jt0 + 1 --------------------------------------------------------------------
    const int it = wave & 3, jt0 = 2 * (wave >> 2);
    const int i = it * 32 + col;
    const bool iok = i < T;
    const int ic = min(i, T - 1);
    const bool mi = tok_mask[g.col0 + ic] != 0;
    float* my = scr + wave * kDbScr;
    f32x16 sacc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int jt = jt0 + t;
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[t][r] = -FLT_MAX;
        if (it * 32 >= T || jt * 32 >= T) continue;   // (wave-uniform)
        f32x16 a = tile_product(Kg, ld, jt * 32, T, Qg, ld, it * 32, T);
        // relative positions of this tile pair: i - j in [rlo, rhi]; the bucket index is monotone in it
        const int rlo = max(it * 32 - min(jt * 32 + 31, T - 1), -(T - 1)), rhi = min(min(it * 32 + 31, T - 1) - jt * 32, T - 1);
        const int wb1 = widx(rlo, false), wb2 = widx(rlo, true);
        (void)rhi;
        // c2p^T piece: [w - wb1][i] = sum_d posk[d][w] q[d][i]
#pragma unroll
        for (int wt = 0; wt < 2; ++wt) {
            const f32x16 c = tile_product(Pk, ldp, wb1 + wt * 32, wlen, Qg, ld, it * 32, T);
#pragma unroll
            for (int r = 0; r < 16; ++r) my[(wt * 32 + acc_row(r, kh)) * 33 + col] = c[r];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int jc = min(jt * 32 + acc_row(r, kh), T - 1);
            const int d1 = min(max(widx(ic - jc, false) - wb1, 0), 63);
            a[r] = a[r] * inv_scale + my[d1 * 33 + col] * inv_scale;
        }
        // p2c piece: [j][w - wb2] = sum_d k[d][j] posq[d][w]
#pragma unroll
        for (int wt = 0; wt < 2; ++wt) {
            const f32x16 c = tile_product(Kg, ld, jt * 32, T, Pq, ldp, wb2 + wt * 32, wlen);
#pragma unroll
            for (int r = 0; r < 16; ++r) my[acc_row(r, kh) * 65 + wt * 32 + col] = c[r];
        }
        float mloc = -FLT_MAX;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int jr = acc_row(r, kh), j = jt * 32 + jr;
            const int jc = min(j, T - 1);
            // the p2c piece is indexed [key row of this tile][window]: this lane needs row jr at the window index of ITS query column,
            // which another lane computed: read it from the scratch (same wave: LDS operations execute in order)
            const int d2 = min(max(widx(ic - jc, true) - wb2, 0), 63);
            float v = a[r] + my[jr * 65 + d2] * inv_scale;
            if (!(mi && tok_mask[g.col0 + jc])) v = -FLT_MAX;
            if (j >= T) v = -FLT_MAX;
            sacc[t][r] = v;
            if (j < T) mloc = fmaxf(mloc, v);
        }
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32));
        if (kh == 0) redm[jt][i] = mloc;
    }
    __syncthreads();
    const float mx = fmaxf(fmaxf(redm[0][i], redm[1][i]), fmaxf(redm[2][i], redm[3][i]));
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int jt = jt0 + t;
        float sloc = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = jt * 32 + acc_row(r, kh);
            const float e = (j < T && it * 32 < T) ? expf(sacc[t][r] - mx) : 0.f;
            sacc[t][r] = e;
            sloc += e;
        }
        const float other = __shfl_xor(sloc, 32);
        sloc = kh ? other + sloc : sloc + other;
        if (kh == 0) reds[jt][i] = sloc;
    }
    __syncthreads();
    const float sum = (reds[0][i] + reds[1][i]) + (reds[2][i] + reds[3][i]);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) p_s[((jt0 + t) * 32 + acc_row(r, kh)) * kDbPp + i] = sum > 0.f ? sacc[t][r] / sum : 0.f;
    // ---- V -> LDS over the scratch (coalesced along j), then ctx[dd][i] = sum_j v[dd][j] p[j][i]: wave -> (channel tile, query tile) -----
    float* v_s = scr;   // [dd][j], pitch kDbPp
    __syncthreads();    // every wave is done with its scratch; P is complete
    for (int idx = tid; idx < 64 * kDbT; idx += 512) {
        const int dd = idx >> 7, j = idx & 127;
        v_s[dd * kDbPp + j] = (dd < dh && j < T) ? Vg[(int64_t)dd * ld + j] : 0.f;
    }
    __syncthreads();
    const int ddt = wave >> 2;
    if (ddt * 32 < dh && it * 32 < T) {
        f32x16 cacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) cacc[r] = 0.f;
        const float* vrow = v_s + (ddt * 32 + col) * kDbPp;
#pragma unroll 8
        for (int s = 0; s < 64; ++s) {
            const int j = 2 * s + kh;
            if (2 * s < T) cacc = __builtin_amdgcn_mfma_f32_32x32x2f32(vrow[j], p_s[j * kDbPp + i], cacc, 0, 0, 0);
        }
        float* Cg = ctx + hoff * ldc + g.col0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int d2 = ddt * 32 + acc_row(r, kh);
            if (d2 < dh && iok) Cg[(int64_t)d2 * ldc + i] = cacc[r];
        }
    }
}

// ---- any length beyond 128 tokens (long-form text, BASELINE configs[4]): key-tile loop with an online softmax ----------------------------
// One workgroup per (utterance, head, 32-query tile); its four waves take the key tiles jt = wave, wave + 4, ... and keep a private
// running (max, sum, ctx) per query column, merged in wave order through LDS at the end (fixed order: results depend on T only).  Per
// 32 x 32 score tile the arithmetic is that of k_deberta_attn128 (K^T Q, a 64-index piece of c2p^T and of p2c in the wave's 8.4 KB
// scratch, gathered by bucket index); the query operand (the B side of K^T Q and of every c2p piece) stays in registers over the whole
// loop, the key operand over the tile.  ctx = V P^T needs no transpose of P: k-step s of the MFMA takes P rows acc_row(s, kh), which is
// exactly score register s of the lane, and the V tile (staged coalesced into the scratch once the gathers are done) is read at the same rows.
constexpr int kDlScr = 64 * 33;   // floats of per-wave scratch

__global__ __launch_bounds__(256) void k_deberta_attn_long(const AttnGroup* groups, const float* Q, const float* K, int ld, const float* V,
                                                          const float* posk, const float* posq, int ldp, int win_lo, int wlen, const int* tab,
                                                          int tab_center, int span, float inv_scale, const unsigned char* tok_mask, int dh,
                                                          float* ctx, int ldc) {
    extern __shared__ __attribute__((aligned(16))) float dl_smem[];
    float* scr = dl_smem;                                             // [4 waves][kDlScr]
    float (*cm)[32] = reinterpret_cast<float (*)[32]>(scr + 4 * kDlScr);   // running max / sum of each wave, for the merge
    float (*cl)[32] = cm + 4;
    int* tab_s = reinterpret_cast<int*>(cl + 4);                      // [2 T - 1]
    const AttnGroup g = groups[blockIdx.y];
    const int T = g.T;
    const int it = blockIdx.x;
    if (it * 32 >= T) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31, kh = lane >> 5;
    const int64_t hoff = (int64_t)g.head * dh;
    const float* Qg = Q + hoff * ld + g.col0;
    const float* Kg = K + hoff * ld + g.col0;
    const float* Pk = posk + hoff * ldp + win_lo;
    const float* Pq = posq + hoff * ldp + win_lo;
    const float* Vg = V + hoff * ld + g.col0;
    const int ns = dh >> 1;
    const int hi = 2 * span - 1;
    for (int e = tid; e < 2 * T - 1; e += 256) tab_s[e] = tab[tab_center - (T - 1) + e];   // tab_s[(i - j) + T - 1] = bucket(i - j)
    __syncthreads();

    // operand columns: lane (col, kh) holds x[d = 2 s + kh][c0 + col] for the 32 k-steps; columns outside [0, cmax) read as zero
    auto load_col = [&](const float* base, int ldx, int c0, int cmax, float (&out)[32]) {
        const int c = c0 + col;
        const bool ok = c >= 0 && c < cmax;
        const float* ptr = base + min(max(c, 0), cmax - 1);
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            const float v = ptr[(int64_t)min(2 * s + kh, dh - 1) * ldx];
            out[s] = ok ? v : 0.f;
        }
    };
    auto prod = [&](const float (&a)[32], const float (&b)[32]) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int s = 0; s < 32; ++s)
            if (s < ns) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc, 0, 0, 0);
        return acc;
    };
    auto widx = [&](int rel, bool neg) {   // window index of bucket(rel) (c2p) or of -bucket(-rel) (p2c)
        const int b = neg ? -tab_s[-rel + T - 1] : tab_s[rel + T - 1];
        return min(max(b + span, 0), hi) - win_lo;
    };

    const int i = it * 32 + col;
    const bool iok = i < T;
    const int ic = min(i, T - 1);
    const bool mi = tok_mask[g.col0 + ic] != 0;
    float* my = scr + wave * kDlScr;
    float bq[32];
    load_col(Qg, ld, it * 32, T, bq);
    f32x16 cacc[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) cacc[dt][r] = 0.f;
    float m = -FLT_MAX, l = 0.f;
    const int ntiles = (T + 31) >> 5;
    for (int jt = wave; jt < ntiles; jt += 4) {
        const int j0 = jt * 32;
        float ak[32], op[32];
        load_col(Kg, ld, j0, T, ak);
        f32x16 a = prod(ak, bq);
        // the V tile of these keys: requested now (lanes along the keys), parked in the scratch after the gathers
        float vreg[32];
#pragma unroll
        for (int q = 0; q < 32; ++q) {
            const int dd = 2 * q + kh;   // (idx = lane + 64 q: dd = idx >> 5, key = idx & 31 = col)
            vreg[q] = (dd < dh && j0 + col < T) ? Vg[(int64_t)min(dd, dh - 1) * ld + min(j0 + col, T - 1)] : 0.f;
        }
        const int rlo = max(it * 32 - min(j0 + 31, T - 1), -(T - 1));   // i - j over this tile pair starts here; the bucket index is monotone in it
        const int rhi = min(it * 32 + 31, T - 1) - j0;
        const int wb1 = widx(rlo, false), wb2 = widx(rlo, true);
        // beyond the exact range of the bucket function (|i - j| > buckets / 2) a tile reaches fewer than 32 bucket indices: one 32-index
        // piece instead of two (wave-uniform; the values gathered are the same products)
        const bool one1 = widx(rhi, false) - wb1 < 32, one2 = widx(rhi, true) - wb2 < 32;
        // c2p^T piece: [w - wb1][i] = sum_d posk[d][w] q[d][i]
#pragma unroll
        for (int wt = 0; wt < 2; ++wt) {
            if (wt == 1 && one1) continue;
            load_col(Pk, ldp, wb1 + wt * 32, wlen, op);
            const f32x16 c = prod(op, bq);
#pragma unroll
            for (int r = 0; r < 16; ++r) my[(wt * 32 + acc_row(r, kh)) * 33 + col] = c[r];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int jc = min(j0 + acc_row(r, kh), T - 1);
            const int d1 = min(max(widx(ic - jc, false) - wb1, 0), 63);
            a[r] = a[r] * inv_scale + my[d1 * 33 + col] * inv_scale;
        }
        // p2c piece: [j][w - wb2] = sum_d k[d][j] posq[d][w]
#pragma unroll
        for (int wt = 0; wt < 2; ++wt) {
            if (wt == 1 && one2) continue;
            load_col(Pq, ldp, wb2 + wt * 32, wlen, op);
            const f32x16 c = prod(ak, op);
#pragma unroll
            for (int r = 0; r < 16; ++r) my[acc_row(r, kh) * 65 + wt * 32 + col] = c[r];
        }
        float mt = -FLT_MAX;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int jr = acc_row(r, kh), j = j0 + jr;
            const int jc = min(j, T - 1);
            const int d2 = min(max(widx(ic - jc, true) - wb2, 0), 63);
            float v = a[r] + my[jr * 65 + d2] * inv_scale;
            if (!(mi && tok_mask[g.col0 + jc])) v = -FLT_MAX;
            a[r] = v;
            if (j < T) mt = fmaxf(mt, v);
        }
        mt = fmaxf(mt, __shfl_xor(mt, 32));
        const float mn = fmaxf(m, mt);
        const float alpha = expf(m - mn);
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float e = j0 + acc_row(r, kh) < T ? expf(a[r] - mn) : 0.f;
            a[r] = e;
            ps += e;
        }
        {
            const float other = __shfl_xor(ps, 32);
            ps = kh ? other + ps : ps + other;   // (rows of half 0) + (rows of half 1) in both halves
        }
        l = l * alpha + ps;
        m = mn;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) cacc[dt][r] *= alpha;
        // V tile -> scratch [dd][key] (the gathers above are done: LDS operations of one wave execute in order)
#pragma unroll
        for (int q = 0; q < 32; ++q) my[(2 * q + kh) * 33 + col] = vreg[q];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            if (dt * 32 >= dh) continue;
            const float* vrow = my + (dt * 32 + col) * 33;
#pragma unroll
            for (int s = 0; s < 16; ++s) cacc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vrow[acc_row(s, kh)], a[s], cacc[dt], 0, 0, 0);
        }
    }
    // ---- merge the four waves' (max, sum, ctx) in wave order ---------------------------------------------------------------------------------
    if (kh == 0) {
        cm[wave][col] = m;
        cl[wave][col] = l;
    }
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) my[(dt * 16 + r) * 64 + lane] = cacc[dt][r];
    __syncthreads();
    const float mx = fmaxf(fmaxf(cm[0][col], cm[1][col]), fmaxf(cm[2][col], cm[3][col]));
    float f[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) f[w] = expf(cm[w][col] - mx);
    const float sum = (cl[0][col] * f[0] + cl[1][col] * f[1]) + (cl[2][col] * f[2] + cl[3][col] * f[3]);
    const float inv = sum > 0.f ? 1.0f / sum : 0.f;
    float* Cg = ctx + hoff * ldc + g.col0;
    const int dt = wave >> 1;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int r = 8 * (wave & 1) + q;
        const int e = (dt * 16 + r) * 64 + lane;
        const float v = (scr[e] * f[0] + scr[kDlScr + e] * f[1]) + (scr[2 * kDlScr + e] * f[2] + scr[3 * kDlScr + e] * f[3]);
        const int d2 = dt * 32 + acc_row(r, kh);
        if (d2 < dh && iok) Cg[(int64_t)d2 * ldc + i] = v * inv;
    }
}
}  // namespace

bool deberta_attention_long_fits(int T, int dh) { return T > kDbT && T <= 8192 && dh <= 64 && (dh & 1) == 0; }

void deberta_attention_long(const AttnGroup* groups, int ngroups, int maxT, const float* Q, const float* K, int ld, const float* V,
                            const float* posk, const float* posq, int ldp, int win_lo, int wlen, const int* tab, int tab_center, int span,
                            float inv_scale, const unsigned char* tok_mask, int dh, float* ctx, int ldc, hipStream_t s) {
    if (ngroups <= 0) return;
    const size_t lds = sizeof(float) * (4 * kDlScr + 8 * 32) + sizeof(int) * (size_t)(2 * maxT);
    static std::atomic<uint64_t> lds_allowed{0};   // per (kernel instantiation, device)
    allow_full_lds(reinterpret_cast<const void*>(k_deberta_attn_long), lds_allowed);
    hipLaunchKernelGGL(k_deberta_attn_long, dim3((maxT + 31) / 32, ngroups), dim3(256), lds, s, groups, Q, K, ld, V, posk, posq, ldp, win_lo, wlen,
                       tab, tab_center, span, inv_scale, tok_mask, dh, ctx, ldc);
    HIP_CHECK(hipGetLastError());
}

// 65 .. 128 tokens: the tiled variant (any window length: pieces are cut per score tile)
bool deberta_attention128_fits(int T, int dh) { return T > kDaT && T <= kDbT && dh <= 64 && (dh & 1) == 0; }

void deberta_attention128(const AttnGroup* groups, int ngroups, const float* Q, const float* K, int ld, const float* V, const float* posk,
                          const float* posq, int ldp, int win_lo, int wlen, const int* tab, int tab_center, int span, float inv_scale,
                          const unsigned char* tok_mask, int dh, float* ctx, int ldc, hipStream_t s) {
    if (ngroups <= 0) return;
    constexpr size_t lds = sizeof(float) * (kDbT * kDbPp + 8 * kDbScr + 8 * kDbT + 2 * kDbT);
    static std::atomic<uint64_t> lds_allowed{0};   // per (kernel instantiation, device)
    allow_full_lds(reinterpret_cast<const void*>(k_deberta_attn128), lds_allowed);
    hipLaunchKernelGGL(k_deberta_attn128, dim3(ngroups), dim3(512), lds, s, groups, Q, K, ld, V, posk, posq, ldp, win_lo, wlen, tab,
                       tab_center, span, inv_scale, tok_mask, dh, ctx, ldc);
    HIP_CHECK(hipGetLastError());
}

bool deberta_attention_fits(int maxT, int wlen, int dh) { return maxT >= 1 && maxT <= kDaT && wlen <= kDaW && dh <= 64 && (dh & 1) == 0; }

void deberta_attention(const AttnGroup* groups, int ngroups, const float* Q, const float* K, int ld, const float* V, const float* posk,
                       const float* posq, int ldp, int win_lo, int wlen, const int* tab, int tab_center, int span, float inv_scale,
                       const unsigned char* tok_mask, int dh, float* ctx, int ldc, hipStream_t s) {
    if (ngroups <= 0) return;
    constexpr size_t lds = sizeof(float) * (kDaW * kDaPc + kDaT * kDaPp + 4 * kDaT + 2 * kDaT);
    static std::atomic<uint64_t> lds_allowed{0};   // per (kernel instantiation, device)
    allow_full_lds(reinterpret_cast<const void*>(k_deberta_attn), lds_allowed);
    hipLaunchKernelGGL(k_deberta_attn, dim3(ngroups), dim3(256), lds, s, groups, Q, K, ld, V, posk, posq, ldp, win_lo, wlen, tab,
                       tab_center, span, inv_scale, tok_mask, dh, ctx, ldc);
    HIP_CHECK(hipGetLastError());
}

}  // namespace sbv2
