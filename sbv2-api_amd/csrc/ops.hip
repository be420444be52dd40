// Non-GEMM kernels of the hot path: normalisation, softmax with relative-position terms, embeddings,
// the stochastic-duration flow (depthwise conv, rational-quadratic spline), duration -> frame expansion,
// and the final conv_post + tanh.  Activations are channel-major planes [C][ld] with the time axis
// contiguous, so "one thread per column, loop over channels" is the coalesced pattern used throughout.
//
// These are the ONNX nodes (LayerNormalization / Softmax / Gather / Erf / Where / CumSum / ...) that the
// reference leaves to ONNX Runtime inside session.run (crates/sbv2_core/src/model.rs:91, bert.rs:11); the
// arithmetic follows oracle/sbv2_oracle.py function by function.
#include "ops.h"

#include <cfloat>

namespace sbv2 {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float gelu_exact(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }

// ------------------------------------------------------------------------------------------------
// DeBERTa embeddings: gather + LayerNorm (modeling_deberta_v2.py:518-559), token-major row -> plane column
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_deberta_embed_ln(const int* ids, const float* emb, int H, const float* gamma,
                                                           const float* beta, float eps, Plane out) {
    __shared__ float red[4];
    const int n = blockIdx.x;
    const int id = ids[n];
    const int tid = threadIdx.x;
    if (id < 0) {
        for (int c = tid; c < H; c += 256) out.p[(size_t)c * out.ld + n] = 0.f;
        return;
    }
    const float* row = emb + (size_t)id * H;
    float s = 0.f;
    for (int c = tid; c < H; c += 256) s += row[c];
    s = wave_sum(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    const float mean = (red[0] + red[1] + red[2] + red[3]) / H;
    __syncthreads();
    float q = 0.f;
    for (int c = tid; c < H; c += 256) {
        const float d = row[c] - mean;
        q += d * d;
    }
    q = wave_sum(q);
    if ((tid & 63) == 0) red[tid >> 6] = q;
    __syncthreads();
    const float rstd = 1.0f / sqrtf((red[0] + red[1] + red[2] + red[3]) / H + eps);
    for (int c = tid; c < H; c += 256) out.p[(size_t)c * out.ld + n] = (row[c] - mean) * rstd * gamma[c] + beta[c];
}
void deberta_embed_ln(const int* ids, const float* emb, int H, const float* gamma, const float* beta, float eps, Plane out,
                      hipStream_t s) {
    hipLaunchKernelGGL(k_deberta_embed_ln, dim3(out.L), dim3(256), 0, s, ids, emb, H, gamma, beta, eps, out);
}

// ------------------------------------------------------------------------------------------------
// Channel LayerNorm over a plane: COLS columns x 256 / COLS channel groups per workgroup
// ------------------------------------------------------------------------------------------------
// CPT > 0: every thread keeps its <= CPT channel values in registers, so the plane is read once (the value is needed three times: mean,
// centred sum of squares, output); CPT == 0: generic fallback that re-reads it.  Same operations in the same order either way.
#ifndef LN_EARLY_MAX
#define LN_EARLY_MAX 32   // (-DLN_EARLY_MAX=0: the loads behind the reductions as in rounds 1-4; tools/ln_ab.sh is the same-box A/B)
#endif
template <bool DW, int CPT, int COLS>
__global__ __launch_bounds__(256) void k_layernorm_ch(Plane in, Plane out, const float* gamma, const float* beta, float eps,
                                                       int act, const float* res, int ldr, const unsigned char* mask,
                                                       const float* dw_w, const float* dw_b, int dil, SplitPlanes sp) {
    constexpr int G = 256 / COLS;   // channel groups: thread (tx, ty) owns column tx and channels ty, ty + G, ...
    __shared__ float red[G][COLS + 1];
    const int tx = threadIdx.x % COLS, ty = threadIdx.x / COLS;
    const int n = blockIdx.x * COLS + tx;
    const bool ok = n < in.L;
    const int nc = ok ? n : in.L - 1;
    const int C = in.C;
    auto val = [&](int c) -> float {
        if (!DW) return in.p[(size_t)c * in.ld + nc];
        const float* r = in.p + (size_t)c * in.ld;
        float v = dw_b[c];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int q = nc + (j - 1) * dil;
            if (q >= 0 && q < in.L) v += dw_w[c * 3 + j] * r[q];
        }
        return v;
    };
    constexpr int NV = CPT > 0 ? CPT : 1;
    float xv[NV];
    if (CPT > 0) {
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int c = ty + G * k;
            xv[k] = c < C ? val(c) : 0.f;
        }
    }
    // gamma / beta / residual of every owned channel are requested with the values (the compiler keeps loads behind a barrier where they are written: behind
    // the two reductions they were a second dependent memory round trip of the launch, 1 - 2 us of the 8 - 11 us a single-utterance LayerNorm takes) and
    // before the first store: a load issued after a store is not usable until that store is acknowledged (in-order vmcnt)
    // (same-box A/B, profiles/r05n_ln_ab.txt: 8.04 -> 7.84 us per launch at 24 channels per thread, 12.48 -> 12.13 at 32 in a single-utterance call: within noise of
    // nothing; kept because it is the order the comment above asks for)
    constexpr bool PRE = CPT > 0 && CPT <= 32, EARLY = CPT > 0 && CPT <= LN_EARLY_MAX;
    float gv[PRE ? NV : 1], bv[PRE ? NV : 1], rv[PRE ? NV : 1];
    auto load_params = [&]() {
#pragma unroll
        for (int k = 0; k < (PRE ? NV : 1); ++k) {
            const int c = min(ty + G * k, C - 1);
            gv[k] = gamma[c];
            bv[k] = beta[c];
            rv[k] = res ? res[(size_t)c * ldr + nc] : 0.f;
        }
    };
    if (EARLY) load_params();
    float s = 0.f;
    if (CPT > 0) {
#pragma unroll
        for (int k = 0; k < NV; ++k)
            if (ty + G * k < C) s += xv[k];
    } else {
        for (int c = ty; c < C; c += G) s += val(c);
    }
    red[ty][tx] = s;
    __syncthreads();
    float mean = 0.f;
#pragma unroll
    for (int g = 0; g < G; ++g) mean += red[g][tx];
    mean /= C;
    __syncthreads();
    float q = 0.f;
    if (CPT > 0) {
#pragma unroll
        for (int k = 0; k < NV; ++k)
            if (ty + G * k < C) {
                const float d = xv[k] - mean;
                q += d * d;
            }
    } else {
        for (int c = ty; c < C; c += G) {
            const float d = val(c) - mean;
            q += d * d;
        }
    }
    red[ty][tx] = q;
    __syncthreads();
    float var = 0.f;
#pragma unroll
    for (int g = 0; g < G; ++g) var += red[g][tx];
    const float rstd = 1.0f / sqrtf(var / C + eps);
    if (!ok) return;
    const bool keep = !mask || mask[n];
    // the result, and (sp.parts != 0) its bf16 parts for the split-bf16 products that read this plane (gemm_bfs.hip)
    auto put = [&](int c, float v) {
        out.p[(size_t)c * out.ld + n] = v;
        if (sp.parts) split_store1(sp, (int64_t)c * sp.ld + n, v);
    };
    auto emit = [&](int c, float x) {
        float v = (x - mean) * rstd * gamma[c] + beta[c];
        if (act == ACT_GELU) v = gelu_exact(v);
        if (res) v += res[(size_t)c * ldr + n];
        put(c, keep ? v : 0.f);
    };
    if (PRE) {
        if (!EARLY) load_params();   // (still before the first store)
#pragma unroll
        for (int k = 0; k < NV; ++k)
            if (ty + G * k < C) {
                float v = (xv[k] - mean) * rstd * gv[k] + bv[k];
                if (act == ACT_GELU) v = gelu_exact(v);
                if (res) v += rv[k];
                put(ty + G * k, keep ? v : 0.f);
            }
    } else if (CPT > 0) {
#pragma unroll
        for (int k = 0; k < NV; ++k)
            if (ty + G * k < C) emit(ty + G * k, xv[k]);
    } else {
        for (int c = ty; c < C; c += G) emit(c, val(c));
    }
}
// The same LayerNorm with FOUR consecutive columns per thread (round 6): 16-byte loads / stores of the plane and the residual, 8-byte stores of the operand
// parts (split_store4) instead of 4- and 2-byte ones: a quarter of the memory instructions (the scalar form above issues 5 x CPT of them per thread, 64 of
// them 2-byte stores, and moved DeBERTa's 35 MB per launch at 1.8 TB/s).  QC column quads x 256 / QC channel groups per workgroup; the column sums are reduced
// in two levels through LDS in a fixed order (another grouping than the scalar kernel's: f32 rounding apart from it).
template <int CPT, int QC>
__global__ __launch_bounds__(256) void k_layernorm_q4(Plane in, Plane out, const float* gamma, const float* beta, float eps, int act, const float* res,
                                                       int ldr, const unsigned char* mask, SplitPlanes sp) {
    typedef float ln_f4 __attribute__((ext_vector_type(4)));
    constexpr int G = 256 / QC, G2 = G / 8;
    static_assert(G % 8 == 0, "two-level reduction: groups of 8");
    __shared__ ln_f4 red[G][QC];
    __shared__ ln_f4 red2[G2][QC];
    const int tx = threadIdx.x % QC, ty = threadIdx.x / QC;
    const int n = (blockIdx.x * QC + tx) * 4;
    const bool ok = n < in.L;
    const int nc = ok ? n : 0;
    const int C = in.C;
    ln_f4 xv[CPT], rv[CPT];
    float gv[CPT], bv[CPT];
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        const int c = min(ty + G * k, C - 1);
        xv[k] = *reinterpret_cast<const ln_f4*>(in.p + (size_t)c * in.ld + nc);
        gv[k] = gamma[c];
        bv[k] = beta[c];
        rv[k] = res ? *reinterpret_cast<const ln_f4*>(res + (size_t)c * ldr + nc) : ln_f4{0.f, 0.f, 0.f, 0.f};
    }
    auto reduce = [&](ln_f4 part) -> ln_f4 {   // sum over the G channel groups of this thread's column quad, the same value in every thread of the quad
        red[ty][tx] = part;
        __syncthreads();
        if (ty < G2) {
            ln_f4 t = red[ty * 8][tx];
#pragma unroll
            for (int j = 1; j < 8; ++j) t += red[ty * 8 + j][tx];
            red2[ty][tx] = t;
        }
        __syncthreads();
        ln_f4 t = red2[0][tx];
#pragma unroll
        for (int j = 1; j < G2; ++j) t += red2[j][tx];
        return t;
    };
    ln_f4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < CPT; ++k)
        if (ty + G * k < C) s += xv[k];
    const ln_f4 mean = reduce(s) / (float)C;
    ln_f4 q = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < CPT; ++k)
        if (ty + G * k < C) {
            const ln_f4 d = xv[k] - mean;
            q += d * d;
        }
    const ln_f4 var = reduce(q);   // (its first barrier also orders the reads of red2 above before they are overwritten)
    ln_f4 rstd;
#pragma unroll
    for (int e = 0; e < 4; ++e) rstd[e] = 1.0f / sqrtf(var[e] / C + eps);
    if (!ok) return;
    bool keep[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) keep[e] = n + e < in.L && (!mask || mask[min(n + e, in.L - 1)]);
    const bool whole = n + 4 <= in.L;
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        const int c = ty + G * k;
        if (c >= C) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float y = (xv[k][e] - mean[e]) * rstd[e] * gv[k] + bv[k];
            if (act == ACT_GELU) y = gelu_exact(y);
            if (res) y += rv[k][e];
            v[e] = keep[e] ? y : 0.f;
        }
        if (whole) {
            *reinterpret_cast<ln_f4*>(out.p + (size_t)c * out.ld + n) = ln_f4{v[0], v[1], v[2], v[3]};
            if (sp.parts) split_store4(sp, (int64_t)c * sp.ld + n, v);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (n + e < in.L) {
                    out.p[(size_t)c * out.ld + n + e] = v[e];
                    if (sp.parts) split_store1(sp, (int64_t)c * sp.ld + n + e, v[e]);
                }
        }
    }
}

template <bool DW>
static void launch_layernorm(Plane in, Plane out, const float* gamma, const float* beta, float eps, int act, const float* res, int ldr,
                             const unsigned char* mask, const float* w, const float* b, int dil, hipStream_t s, SplitPlanes sp = SplitPlanes{}) {
    SBV2_REQUIRE(!sp.parts || (sp.C == out.C && sp.ld >= out.L), "layernorm: split output shape");
    const dim3 block(256);
    // A single utterance (66 tokens x 1024 channels, 897 frames or 257 symbols x 192): few columns per workgroup and few channels per thread, i.e. more
    // workgroups and a shorter chain of dependent loads per thread (12.4 -> ~7 us and 7.8 -> ~5 us per launch, 129 launches per call).  Another grouping of the
    // per-column sums than the batch's launch shape: f32-rounding apart from it, so it belongs to the small-grid dispatch that sbv2_debug_set_ksplit(0) turns off.
    if (ksplit_enabled() && in.C >= 512 && in.C <= 8 * 128 && in.L <= 256) {
        hipLaunchKernelGGL((k_layernorm_ch<DW, 8, 2>), dim3((in.L + 1) / 2), block, 0, s, in, out, gamma, beta, eps, act, res, ldr, mask, w, b, dil, sp);
        return;
    }
    if (ksplit_enabled() && in.C < 512 && in.C <= 6 * 32 && in.L <= 1024) {
        hipLaunchKernelGGL((k_layernorm_ch<DW, 6, 8>), dim3((in.L + 7) / 8), block, 0, s, in, out, gamma, beta, eps, act, res, ldr, mask, w, b, dil, sp);
        return;
    }
    // Round 6: four columns per thread (k_layernorm_q4) for the flow's / text side's <= 192 channels where the planes allow 16-byte accesses (every plane of
    // the library: pitches are multiples of 64 floats): 8 column quads x 32 channel groups per workgroup, 16.2 -> 13.1 us per launch at 192 x 28 832, 84
    // launches per step.  (DeBERTa's 1024 channels as 2 quads x 128 groups measured 22.1 us against the scalar kernel's 20.5: its reduction over 128 groups
    // costs what the wider accesses save; it stays on the scalar kernel.)  The choice depends on the channel count only, so a batch row and its single call
    // keep taking the same kernel on the size-independent dispatch.
    if constexpr (!DW) {
        if (in.C <= 6 * 32 && (in.ld & 3) == 0 && (out.ld & 3) == 0 && (!res || (ldr & 3) == 0) && (!sp.parts || (sp.ld & 3) == 0)) {
            hipLaunchKernelGGL((k_layernorm_q4<6, 8>), dim3((in.L + 31) / 32), block, 0, s, in, out, gamma, beta, eps, act, res, ldr, mask, sp);
            return;
        }
    }
    // few, wide columns (DeBERTa: 1024 channels x ~2k tokens): 8 columns x 32 channel groups per workgroup, or the grid is 64 workgroups
    if (!DW && in.C >= 512 && in.L <= 8192 && in.C <= 32 * 32) {
        hipLaunchKernelGGL((k_layernorm_ch<DW, 32, 8>), dim3((in.L + 7) / 8), block, 0, s, in, out, gamma, beta, eps, act, res, ldr, mask, w, b, dil, sp);
        return;
    }
    const dim3 grid((in.L + 31) / 32);
    const int cpt = (in.C + 7) / 8;
    if (cpt <= 8) hipLaunchKernelGGL((k_layernorm_ch<DW, 8, 32>), grid, block, 0, s, in, out, gamma, beta, eps, act, res, ldr, mask, w, b, dil, sp);
    else if (cpt <= 24) hipLaunchKernelGGL((k_layernorm_ch<DW, 24, 32>), grid, block, 0, s, in, out, gamma, beta, eps, act, res, ldr, mask, w, b, dil, sp);
    else if (cpt <= 32) hipLaunchKernelGGL((k_layernorm_ch<DW, 32, 32>), grid, block, 0, s, in, out, gamma, beta, eps, act, res, ldr, mask, w, b, dil, sp);
    else if (cpt <= 128) hipLaunchKernelGGL((k_layernorm_ch<DW, 128, 32>), grid, block, 0, s, in, out, gamma, beta, eps, act, res, ldr, mask, w, b, dil, sp);
    else hipLaunchKernelGGL((k_layernorm_ch<DW, 0, 32>), grid, block, 0, s, in, out, gamma, beta, eps, act, res, ldr, mask, w, b, dil, sp);
}
void layernorm_ch(Plane in, Plane out, const float* gamma, const float* beta, float eps, int act, const float* res, int ldr,
                  const unsigned char* mask, hipStream_t s, const SplitPlanes* split) {
    launch_layernorm<false>(in, out, gamma, beta, eps, act, res, ldr, mask, nullptr, nullptr, 1, s, split ? *split : SplitPlanes{});
}
void dds_dw_ln_gelu(Plane in, Plane out, const float* w, const float* b, int dil, const float* gamma, const float* beta,
                    const unsigned char* mask, hipStream_t s) {
    SBV2_REQUIRE(in.p != out.p, "depthwise conv cannot run in place");
    launch_layernorm<true>(in, out, gamma, beta, 1e-5f, (int)ACT_GELU, nullptr, 0, mask, w, b, dil, s);
}

// ------------------------------------------------------------------------------------------------
// DeBERTa disentangled-attention softmax (modeling_deberta_v2.py:233-253, 276-346).
// S holds the TRANSPOSED content scores S[j][i] = k_j . q_i / scale; one thread owns query column i.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_deberta_softmax(const AttnGroup* groups, float* S, const float* c2pT, const float* p2c,
                                                         const int* tab, int tab_center, int span, int win_lo, int win_ld,
                                                         float inv_scale, const unsigned char* tok_mask) {
    const AttnGroup g = groups[blockIdx.y];
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= g.T) return;
    float* Sg = S + g.s_off;
    const float* cg = c2pT + g.aux_off;
    const float* pg = p2c + g.aux2_off;
    const bool mi = tok_mask[g.col0 + i] != 0;
    const int hi = 2 * span - 1;
    float mx = -FLT_MAX;
    for (int j = 0; j < g.T; ++j) {
        const int d1 = min(max(tab[tab_center + (i - j)] + span, 0), hi) - win_lo;
        const int d2 = min(max(-tab[tab_center + (j - i)] + span, 0), hi) - win_lo;
        float v = Sg[(size_t)j * g.lds + i] + cg[(size_t)d1 * g.lds + i] * inv_scale + pg[(size_t)j * win_ld + d2] * inv_scale;
        if (!(mi && tok_mask[g.col0 + j])) v = -FLT_MAX;
        Sg[(size_t)j * g.lds + i] = v;
        mx = fmaxf(mx, v);
    }
    float sum = 0.f;
    for (int j = 0; j < g.T; ++j) {
        const float e = expf(Sg[(size_t)j * g.lds + i] - mx);
        Sg[(size_t)j * g.lds + i] = e;
        sum += e;
    }
    for (int j = 0; j < g.T; ++j) Sg[(size_t)j * g.lds + i] /= sum;
}
// Short sequences (the usual case: a sentence is <= 100 characters, model.rs:15): the column of scores lives in registers, so the
// score block is read once and written once instead of three times each, and the two relative-position gathers are done once.
// Same operations in the same order as k_deberta_softmax.
template <int TMAX>
__global__ __launch_bounds__(64) void k_deberta_softmax_reg(const AttnGroup* groups, float* S, const float* c2pT, const float* p2c,
                                                             const int* tab, int tab_center, int span, int win_lo, int win_ld,
                                                             float inv_scale, const unsigned char* tok_mask) {
    const AttnGroup g = groups[blockIdx.y];
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= g.T) return;
    float* Sg = S + g.s_off;
    const float* cg = c2pT + g.aux_off;
    const float* pg = p2c + g.aux2_off;
    const bool mi = tok_mask[g.col0 + i] != 0;
    const int hi = 2 * span - 1;
    float sv[TMAX];
    float mx = -FLT_MAX;
#pragma unroll
    for (int j = 0; j < TMAX; ++j) {
        sv[j] = -FLT_MAX;
        if (j < g.T) {
            const int d1 = min(max(tab[tab_center + (i - j)] + span, 0), hi) - win_lo;
            const int d2 = min(max(-tab[tab_center + (j - i)] + span, 0), hi) - win_lo;
            float v = Sg[(size_t)j * g.lds + i] + cg[(size_t)d1 * g.lds + i] * inv_scale + pg[(size_t)j * win_ld + d2] * inv_scale;
            if (!(mi && tok_mask[g.col0 + j])) v = -FLT_MAX;
            sv[j] = v;
            mx = fmaxf(mx, v);
        }
    }
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < TMAX; ++j)
        if (j < g.T) {
            sv[j] = expf(sv[j] - mx);
            sum += sv[j];
        }
#pragma unroll
    for (int j = 0; j < TMAX; ++j)
        if (j < g.T) Sg[(size_t)j * g.lds + i] = sv[j] / sum;
}
void deberta_softmax(const AttnGroup* groups, int ngroups, int maxT, float* S, const float* c2pT, const float* p2c, const int* tab,
                     int tab_center, int span, int win_lo, int win_ld, float inv_scale, const unsigned char* tok_mask,
                     hipStream_t s) {
    const dim3 grid((maxT + 63) / 64, ngroups);
    if (maxT <= 64)
        hipLaunchKernelGGL(k_deberta_softmax_reg<64>, grid, dim3(64), 0, s, groups, S, c2pT, p2c, tab, tab_center, span, win_lo, win_ld,
                           inv_scale, tok_mask);
    else if (maxT <= 128)
        hipLaunchKernelGGL(k_deberta_softmax_reg<128>, grid, dim3(64), 0, s, groups, S, c2pT, p2c, tab, tab_center, span, win_lo, win_ld,
                           inv_scale, tok_mask);
    else
        hipLaunchKernelGGL(k_deberta_softmax, grid, dim3(64), 0, s, groups, S, c2pT, p2c, tab, tab_center, span, win_lo, win_ld, inv_scale,
                           tok_mask);
}

// ------------------------------------------------------------------------------------------------
// VITS window-relative attention softmax (attentions.MultiHeadAttention.attention).
// S[j][i] = k_j . q_i / sqrt(dk) (transposed); adds q_i . emb_rel_k[j-i+w] / sqrt(dk) for |j-i| <= w, softmax over j,
// and saves the +-w band of the probabilities for the relative-value term.
// ------------------------------------------------------------------------------------------------
constexpr int kMaxWin = 4;
// 64 query columns x 4 row slices per workgroup: slice `ty` owns key rows j = ty, ty+4, ...; maxima and sums are combined in LDS.
__global__ __launch_bounds__(256) void k_vits_softmax(const AttnGroup* groups, float* S, const float* Q, int ldq, int dk,
                                                       const float* erk, int w, float qscale, float* pwin) {
    __shared__ float rks[2 * kMaxWin + 1][64];
    __shared__ float rkp[4][2 * kMaxWin + 1][64];
    __shared__ float red[4][64];
    const AttnGroup g = groups[blockIdx.y];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + tx;
    if (blockIdx.x * 64 >= g.T) return;  // whole workgroup out of range (uniform)
    const bool ok = i < g.T;
    const int ic = ok ? i : g.T - 1;
    float* Sg = S + g.s_off + ic;
    const size_t lds = g.lds;
    const int T = g.T;
    // relative-key logits q_i . emb_rel_k[r] (each slice sums a quarter of the head dimension)
    {
        float rk[2 * kMaxWin + 1];
#pragma unroll
        for (int r = 0; r < 2 * kMaxWin + 1; ++r) rk[r] = 0.f;
        const float* q = Q + g.qk_off + ic;
        for (int d = ty; d < dk; d += 4) {
            const float qv = q[(size_t)d * ldq];
#pragma unroll
            for (int r = 0; r < 2 * kMaxWin + 1; ++r)
                if (r < 2 * w + 1) rk[r] += qv * erk[r * dk + d];
        }
#pragma unroll
        for (int r = 0; r < 2 * kMaxWin + 1; ++r) rkp[ty][r][tx] = rk[r];
        __syncthreads();
        if (ty == 0) {
#pragma unroll
            for (int r = 0; r < 2 * kMaxWin + 1; ++r)  // fixed summation order: results must not depend on scheduling
                rks[r][tx] = (((rkp[0][r][tx] + rkp[1][r][tx]) + rkp[2][r][tx]) + rkp[3][r][tx]) * qscale;
        }
        __syncthreads();
    }
    auto score = [&](int j) -> float {
        float v = Sg[(size_t)j * lds];
        const int r = j - ic + w;
        if (r >= 0 && r <= 2 * w) v += rks[r][tx];
        return v;
    };
    // Two sweeps over the column instead of four (max | exp + store | scale | band): the first keeps a running maximum and the sum
    // rescaled to it (one read), the second stores exp(s - max) / sum (one read, one write).  The kernel is HBM bound (T^2 floats per
    // head and utterance), so this is 3 passes of traffic instead of 5.  Slices are combined in a fixed order: deterministic.
    float mx = -FLT_MAX, sum = 0.f;
    int j = ty;
    for (; j + 12 < T; j += 16) {
        const float a = score(j), b = score(j + 4), c = score(j + 8), d = score(j + 12);
        const float m4 = fmaxf(fmaxf(a, b), fmaxf(c, d));
        if (m4 > mx) {
            sum *= expf(mx - m4);   // exp(-inf) = 0 on the first block
            mx = m4;
        }
        sum += (expf(a - mx) + expf(b - mx)) + (expf(c - mx) + expf(d - mx));
    }
    for (; j < T; j += 4) {
        const float a = score(j);
        if (a > mx) {
            sum *= expf(mx - a);
            mx = a;
        }
        sum += expf(a - mx);
    }
    red[ty][tx] = mx;
    __syncthreads();
    const float mall = fmaxf(fmaxf(red[0][tx], red[1][tx]), fmaxf(red[2][tx], red[3][tx]));
    __syncthreads();
    red[ty][tx] = sum * expf(mx - mall);   // a slice without rows (T < 4) contributes 0 * exp(-FLT_MAX - mall) = 0
    __syncthreads();
    const float inv = 1.0f / ((red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]));
    if (!ok) return;
    j = ty;
    for (; j + 12 < T; j += 16) {
        const float a = expf(score(j) - mall) * inv, b = expf(score(j + 4) - mall) * inv, c = expf(score(j + 8) - mall) * inv,
                    d = expf(score(j + 12) - mall) * inv;
        Sg[(size_t)j * lds] = a;
        Sg[(size_t)(j + 4) * lds] = b;
        Sg[(size_t)(j + 8) * lds] = c;
        Sg[(size_t)(j + 12) * lds] = d;
    }
    for (; j < T; j += 4) Sg[(size_t)j * lds] = expf(score(j) - mall) * inv;
    __syncthreads();
    if (ty == 0) {
        float* pw = pwin + g.aux_off;
#pragma unroll
        for (int r = 0; r < 2 * kMaxWin + 1; ++r) {
            const int jj = i + r - w;
            if (r < 2 * w + 1) pw[(size_t)r * lds + i] = (jj >= 0 && jj < T) ? Sg[(size_t)jj * lds] : 0.f;
        }
    }
}
void vits_softmax(const AttnGroup* groups, int ngroups, int maxT, float* S, const float* Q, int ldq, int dk, const float* erk,
                  int window, float qscale, float* pwin, hipStream_t s) {
    SBV2_REQUIRE(window <= kMaxWin, "relative attention window larger than the compiled maximum");
    hipLaunchKernelGGL(k_vits_softmax, dim3((maxT + 63) / 64, ngroups), dim3(256), 0, s, groups, S, Q, ldq, dk, erk, window, qscale,
                       pwin);
}

__global__ __launch_bounds__(256) void k_vits_relv_add(const AttnGroup* groups, float* ctx, int ldc, int dk, const float* erv, int w,
                                                        const float* pwin) {
    const AttnGroup g = groups[blockIdx.y];
    const int i = blockIdx.x * 64 + (threadIdx.x & 63);
    const int ty = threadIdx.x >> 6;   // 4 channel slices per 64 columns
    if (i >= g.T) return;
    const float* pw = pwin + g.aux_off;
    float p[2 * kMaxWin + 1];
#pragma unroll
    for (int r = 0; r < 2 * kMaxWin + 1; ++r) p[r] = (r < 2 * w + 1) ? pw[(size_t)r * g.lds + i] : 0.f;
    float* c = ctx + g.qk_off + i;
    for (int d = ty; d < dk; d += 4) {
        float a = 0.f;
#pragma unroll
        for (int r = 0; r < 2 * kMaxWin + 1; ++r)
            if (r < 2 * w + 1) a += p[r] * erv[r * dk + d];
        c[(size_t)d * ldc] += a;
    }
}
void vits_relv_add(const AttnGroup* groups, int ngroups, int maxT, float* ctx, int ldc, int dk, const float* erv, int window,
                   const float* pwin, hipStream_t s) {
    hipLaunchKernelGGL(k_vits_relv_add, dim3((maxT + 63) / 64, ngroups), dim3(256), 0, s, groups, ctx, ldc, dk, erv, window, pwin);
}

// ------------------------------------------------------------------------------------------------
// small per-utterance vector ops
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_add_segvec(Plane x, const float* vec, int vec_ld, const int* seg_of, int div,
                                                     const unsigned char* mask) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    const int c = blockIdx.y;
    if (n >= x.L) return;
    const int q = n / div;
    const int sg = seg_of[q];
    float* p = x.p + (size_t)c * x.ld + n;
    *p = (sg >= 0 && (!mask || mask[q])) ? *p + vec[(size_t)sg * vec_ld + c] : 0.f;
}
void add_segvec(Plane x, const float* vec, int vec_ld, const int* seg_of, int div, const unsigned char* mask, hipStream_t s) {
    hipLaunchKernelGGL(k_add_segvec, dim3((x.L + 255) / 256, x.C), dim3(256), 0, s, x, vec, vec_ld, seg_of, div, mask);
}

__global__ __launch_bounds__(256) void k_linear_vec(const float* W, const float* bias, int M, int K, const float* v, int ldv,
                                                     float* out, int ldo) {
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int b = blockIdx.y;
    if (m >= M) return;
    const int lane = threadIdx.x & 63;
    float a = 0.f;
    for (int k = lane; k < K; k += 64) a += W[(size_t)m * K + k] * v[(size_t)b * ldv + k];
    a = wave_sum(a);
    if (lane == 0) out[(size_t)b * ldo + m] = a + (bias ? bias[m] : 0.f);
}
void linear_vec(const float* W, const float* bias, int M, int K, const float* v, int ldv, float* out, int ldo, int B,
                hipStream_t s) {
    hipLaunchKernelGGL(k_linear_vec, dim3((M + 3) / 4, B), dim3(256), 0, s, W, bias, M, K, v, ldv, out, ldo);
}

__global__ void k_gather_rows(const float* table, int K, const int* idx, float* out) {
    const int b = blockIdx.y;
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k < K) out[(size_t)b * K + k] = table[(size_t)idx[b] * K + k];
}
void gather_rows(const float* table, int K, const int* idx, float* out, int B, hipStream_t s) {
    hipLaunchKernelGGL(k_gather_rows, dim3((K + 255) / 256, B), dim3(256), 0, s, table, K, idx, out);
}

// TextEncoder front (models_jp_extra.TextEncoder.forward): (emb + tone + lang + bert_proj + style_proj) * sqrt(H)
__global__ __launch_bounds__(256) void k_text_embed(const int* phones, const int* tones, const int* langs, const int* seg_of,
                                                     const float* emb, const float* tone_emb, const float* lang_emb,
                                                     const float* bertproj, int ldb, const float* styleproj, int lds_, float scale,
                                                     Plane out) {
    const int n = blockIdx.x * 64 + (threadIdx.x & 63);
    if (n >= out.L) return;
    const int sg = seg_of[n];
    const int H = out.C;
    const int ph = sg >= 0 ? phones[n] : 0, tn = sg >= 0 ? tones[n] : 0, lg = sg >= 0 ? langs[n] : 0;
    for (int c = threadIdx.x >> 6; c < H; c += 4) {
        float v = 0.f;
        if (sg >= 0)
            v = (emb[(size_t)ph * H + c] + tone_emb[(size_t)tn * H + c] + lang_emb[(size_t)lg * H + c] + bertproj[(size_t)c * ldb + n] +
                 styleproj[(size_t)sg * lds_ + c]) * scale;
        out.p[(size_t)c * out.ld + n] = v;
    }
}
void text_embed(const int* phones, const int* tones, const int* langs, const int* seg_of, const float* emb, const float* tone_emb,
                const float* lang_emb, const float* bertproj, int ldb, const float* styleproj, int lds_, float scale, Plane out,
                hipStream_t s) {
    hipLaunchKernelGGL(k_text_embed, dim3((out.L + 63) / 64), dim3(256), 0, s, phones, tones, langs, seg_of, emb, tone_emb, lang_emb,
                       bertproj, ldb, styleproj, lds_, scale, out);
}

// ------------------------------------------------------------------------------------------------
// Stochastic duration predictor pieces (modules.ConvFlow / transforms.py; oracle: conv_flow_reverse, rq_spline_inverse)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_convflow_pre(const float* z0, const float* w, const float* b, Plane cond, Plane out,
                                                       const unsigned char* mask) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    const int c = blockIdx.y;
    if (n >= out.L) return;
    out.p[(size_t)c * out.ld + n] = mask[n] ? w[c] * z0[n] + b[c] + cond.p[(size_t)c * cond.ld + n] : 0.f;
}
void convflow_pre(const float* z0, const float* w, const float* b, Plane cond, Plane out, const unsigned char* mask, hipStream_t s) {
    hipLaunchKernelGGL(k_convflow_pre, dim3((out.L + 255) / 256, out.C), dim3(256), 0, s, z0, w, b, cond, out, mask);
}

template <int NB>
__global__ __launch_bounds__(64) void k_spline_inverse(Plane params, float* z0, float* z1, float tail, float inv_sqrt_f,
                                                        const unsigned char* mask, int L) {
    const int n = blockIdx.x * 64 + threadIdx.x;
    if (n >= L) return;
    if (!mask[n]) {
        z0[n] = 0.f;
        z1[n] = 0.f;
        return;
    }
    const float x = z1[n];
    if (!(x >= -tail && x <= tail)) return;  // linear tails: identity
    const float min_w = 1e-3f, min_h = 1e-3f, min_d = 1e-3f;
    const float* P = params.p + n;
    const size_t ld = params.ld;
    float cw[NB + 1], ch[NB + 1], dv[NB + 1];
    auto cum = [&](int base, float mn, float* c) {
        float u[NB];
        float mx = -FLT_MAX;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            u[b] = P[(size_t)(base + b) * ld] * inv_sqrt_f;
            mx = fmaxf(mx, u[b]);
        }
        float sum = 0.f;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            u[b] = expf(u[b] - mx);
            sum += u[b];
        }
        float run = 0.f;
        c[0] = -tail;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            run += mn + (1.0f - mn * NB) * (u[b] / sum);
            c[b + 1] = 2.0f * tail * run + (-tail);
        }
        c[NB] = tail;
    };
    cum(0, min_w, cw);
    cum(NB, min_h, ch);
    const float cst = logf(expf(1.0f - min_d) - 1.0f);
    auto softplus = [](float v) { return v > 20.f ? v : log1pf(expf(v)); };
    dv[0] = min_d + softplus(cst);
    dv[NB] = dv[0];
#pragma unroll
    for (int b = 1; b < NB; ++b) dv[b] = min_d + softplus(P[(size_t)(2 * NB + b - 1) * ld]);
    // bin search on the heights (inverse): count of locations <= x, last location nudged by 1e-6
    int bin = -1;
#pragma unroll
    for (int b = 0; b <= NB; ++b) bin += (x >= (b == NB ? ch[b] + 1e-6f : ch[b])) ? 1 : 0;
    bin = min(max(bin, 0), NB - 1);
    float in_cw = 0.f, in_w = 0.f, in_ch = 0.f, in_h = 0.f, in_d = 0.f, in_d1 = 0.f;
#pragma unroll
    for (int b = 0; b < NB; ++b)
        if (b == bin) {
            in_cw = cw[b];
            in_w = cw[b + 1] - cw[b];
            in_ch = ch[b];
            in_h = ch[b + 1] - ch[b];
            in_d = dv[b];
            in_d1 = dv[b + 1];
        }
    const float delta = in_h / in_w;
    const float i1 = in_d + in_d1 - 2.0f * delta;
    const float i2 = x - in_ch;
    const float i3 = i2 * i1;
    const float a = in_h * (delta - in_d) + i3;
    const float b2 = in_h * in_d - i3;
    const float c = -delta * i2;
    const float disc = fmaxf(b2 * b2 - 4.0f * a * c, 0.f);
    const float root = (2.0f * c) / (-b2 - sqrtf(disc));
    z1[n] = root * in_w + in_cw;
}
void spline_inverse(Plane params, float* z0, float* z1, int nbins, float tail, float inv_sqrt_f, const unsigned char* mask, int L,
                    hipStream_t s) {
    SBV2_REQUIRE(nbins == 10, "only the 10-bin spline of the JP-Extra duration flow is compiled");
    hipLaunchKernelGGL(k_spline_inverse<10>, dim3((L + 63) / 64), dim3(64), 0, s, params, z0, z1, tail, inv_sqrt_f, mask, L);
}

__global__ void k_swap_rows(float* a, float* b, int L) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n < L) {
        const float t = a[n];
        a[n] = b[n];
        b[n] = t;
    }
}
void swap_rows(float* a, float* b, int L, hipStream_t s) { hipLaunchKernelGGL(k_swap_rows, dim3((L + 255) / 256), dim3(256), 0, s, a, b, L); }

__global__ void k_flip_channels(Plane in, Plane out) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    const int c = blockIdx.y;
    if (n < in.L) out.p[(size_t)c * out.ld + n] = in.p[(size_t)(in.C - 1 - c) * in.ld + n];
}
void flip_channels(Plane in, Plane out, hipStream_t s) {
    hipLaunchKernelGGL(k_flip_channels, dim3((in.L + 255) / 256, in.C), dim3(256), 0, s, in, out);
}

__global__ void k_affine_reverse(float* z0, float* z1, const float* m, const float* logs, const float* scale, const unsigned char* mask, int L) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= L) return;
    const bool k = mask[n];
    // exp(-logs) is computed here unless the weight file carries it already folded (scale)
    z0[n] = k ? (z0[n] - m[0]) * (logs ? expf(-logs[0]) : scale[0]) : 0.f;
    z1[n] = k ? (z1[n] - m[1]) * (logs ? expf(-logs[1]) : scale[1]) : 0.f;
}
void affine_reverse(float* z0, float* z1, const float* m, const float* logs, const float* scale, const unsigned char* mask, int L, hipStream_t s) {
    hipLaunchKernelGGL(k_affine_reverse, dim3((L + 255) / 256), dim3(256), 0, s, z0, z1, m, logs, scale, mask, L);
}

// w_ceil = ceil(exp(logw) * mask * length_scale)   (SynthesizerTrn.infer)
__global__ void k_durations(const float* sdp, const float* dp, float ratio, float length_scale, const unsigned char* mask, int L,
                            float* logw, int* dur) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= L) return;
    const float lw = sdp[n] * ratio + dp[n] * (1.0f - ratio);
    logw[n] = mask[n] ? lw : 0.f;
    dur[n] = mask[n] ? (int)ceilf(expf(lw) * length_scale) : 0;
}
void durations(const float* sdp, const float* dp, float ratio, float length_scale, const unsigned char* mask, int L, float* logw,
               int* dur, hipStream_t s) {
    hipLaunchKernelGGL(k_durations, dim3((L + 255) / 256), dim3(256), 0, s, sdp, dp, ratio, length_scale, mask, L, logw, dur);
}

// ------------------------------------------------------------------------------------------------
// Counter-based normal noise, bit-compatible with sbv2-api_amd/synth.py hash_normal(noise_key(seed, utt, stream), n)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double hash_u01(uint64_t key, uint64_t e) {
    uint64_t z = (e + key) * 0x9E3779B97F4A7C15ull;
    z ^= z >> 30;
    z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27;
    z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (double)(float)(z >> 40) * (1.0 / 16777216.0);
}
__device__ __forceinline__ float hash_normal(uint64_t key, uint64_t e) {
    const double u1 = hash_u01(key, e), u2 = hash_u01(key ^ 0x5851F42D4C957F2Dull, e);
    return (float)(sqrt(-2.0 * log(1.0 - u1)) * cos(6.283185307179586476925286766559 * u2));
}
__device__ __forceinline__ uint64_t noise_key(uint64_t seed, int utt, int stream) {
    return seed + (uint64_t)(2 * utt + stream) * 0x9E3779B97F4A7C15ull;
}

// seg_utt[sg] = index of segment sg's utterance in the CALLER's batch (the noise streams are keyed by it, so a shard of a batch that was
// dealt to another GPU draws exactly the noise the whole batch would have drawn on one GPU)
__global__ void k_noise_fill(float* out, int ld, int rows, const int* seg_of, const int* seg_start, const int* seg_len, const int* seg_utt, int L,
                             uint64_t seed, int stream_id, float scale) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    const int r = blockIdx.y;
    if (n >= L) return;
    const int sg = seg_of[n];
    float v = 0.f;
    if (sg >= 0 && scale != 0.f) {
        const uint64_t e = (uint64_t)r * seg_len[sg] + (n - seg_start[sg]);
        v = hash_normal(noise_key(seed, seg_utt[sg], stream_id), e) * scale;
    }
    out[(size_t)r * ld + n] = v;
}
void noise_fill(float* out, int ld, int rows, const int* seg_of, const int* seg_start, const int* seg_len, const int* seg_utt, int L,
                uint64_t seed, int stream_id, float scale, hipStream_t s) {
    hipLaunchKernelGGL(k_noise_fill, dim3((L + 255) / 256, rows), dim3(256), 0, s, out, ld, rows, seg_of, seg_start, seg_len, seg_utt, L, seed,
                       stream_id, scale);
}

// generate_path + the two matmuls + prior sampling: z_p[c][y] = m_p[c][tok(y)] + randn * exp(logs_p[c][tok(y)]) * noise_scale
__global__ void k_expand_frames(Plane m_p, Plane logs_p, const int* tok_of_frame, const int* seg_of, const int* seg_start,
                                const int* seg_len, const int* seg_utt, uint64_t seed, float noise_scale, Plane out) {
    const int y = blockIdx.x * 256 + threadIdx.x;
    const int c = blockIdx.y;
    if (y >= out.L) return;
    const int tok = tok_of_frame[y];
    float v = 0.f;
    if (tok >= 0) {
        v = m_p.p[(size_t)c * m_p.ld + tok];
        if (noise_scale != 0.f) {
            const int sg = seg_of[y];
            const uint64_t e = (uint64_t)c * seg_len[sg] + (y - seg_start[sg]);
            v += hash_normal(noise_key(seed, seg_utt[sg], 1), e) * noise_scale * expf(logs_p.p[(size_t)c * logs_p.ld + tok]);
        }
    }
    out.p[(size_t)c * out.ld + y] = v;
}
void expand_frames(Plane m_p, Plane logs_p, const int* tok_of_frame, const int* seg_of, const int* seg_start, const int* seg_len,
                   const int* seg_utt, uint64_t seed, float noise_scale, Plane out, hipStream_t s) {
    hipLaunchKernelGGL(k_expand_frames, dim3((out.L + 255) / 256, out.C), dim3(256), 0, s, m_p, logs_p, tok_of_frame, seg_of,
                       seg_start, seg_len, seg_utt, seed, noise_scale, out);
}

// ------------------------------------------------------------------------------------------------
// Generator tail: leaky_relu(0.01) -> conv_post (C -> 1, k taps, no bias) -> tanh, written de-gapped per utterance
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_conv_post_tanh(Plane x, const float* w, int k, float slope, const int* seg_start,
                                                         const int* seg_len, const int64_t* pcm_off, int up, float* pcm) {
    const int sg = blockIdx.y;
    const int64_t len = (int64_t)seg_len[sg] * up;
    const int64_t sidx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (sidx >= len) return;
    const int64_t col = (int64_t)seg_start[sg] * up + sidx;
    const int half = k / 2;
    float a = 0.f;
    for (int c = 0; c < x.C; ++c) {
        const float* r = x.p + (size_t)c * x.ld;
        for (int j = 0; j < k; ++j) {
            const int64_t q = col + j - half;
            if (q >= 0 && q < x.L) {
                float v = r[q];
                v = v >= 0.f ? v : v * slope;
                a += w[c * k + j] * v;
            }
        }
    }
    pcm[pcm_off[sg] + sidx] = tanhf(a);
}
void conv_post_tanh(Plane x, const float* w, int k, float slope, const int* seg_start, const int* seg_len, const int64_t* pcm_off,
                    int nseg, int up, int64_t max_samples, float* pcm, hipStream_t s) {
    // grid.x covers the longest utterance; blocks past an utterance's end exit
    hipLaunchKernelGGL(k_conv_post_tanh, dim3((unsigned)((max_samples + 255) / 256), nseg), dim3(256), 0, s, x, w, k, slope, seg_start,
                       seg_len, pcm_off, up, pcm);
}

__global__ void k_transpose_out(Plane in, int col0, int T, float* out) {
    __shared__ float tile[32][33];
    const int t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, t = t0 + tx;
        tile[r][tx] = (c < in.C && t < T) ? in.p[(size_t)c * in.ld + col0 + t] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int t = t0 + r, c = c0 + tx;
        if (t < T && c < in.C) out[(size_t)t * in.C + c] = tile[tx][r];
    }
}
void transpose_out(Plane in, int col0, int T, float* out, hipStream_t s) {
    hipLaunchKernelGGL(k_transpose_out, dim3((T + 31) / 32, (in.C + 31) / 32), dim3(256), 0, s, in, col0, T, out);
}

__global__ void k_gather_cols(Plane in, const int* map, Plane out) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    const int c = blockIdx.y;
    if (n >= out.L) return;
    const int q = map[n];
    out.p[(size_t)c * out.ld + n] = q >= 0 ? in.p[(size_t)c * in.ld + q] : 0.f;
}
void gather_cols(Plane in, const int* map, Plane out, hipStream_t s) {
    hipLaunchKernelGGL(k_gather_cols, dim3((out.L + 255) / 256, out.C), dim3(256), 0, s, in, map, out);
}

__global__ void k_add_segvec_cl(float* x, int L, int C, const float* vec, int vec_ld, const int* seg_of, const unsigned char* mask) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;   // float4 index
    const int c4 = C >> 2;
    if (e >= (int64_t)L * c4) return;
    const int n = (int)(e / c4), c = (int)(e % c4) * 4;
    const int sg = seg_of[n];
    float4* p = reinterpret_cast<float4*>(x + (int64_t)n * C + c);
    float4 v = *p;
    if (sg >= 0 && mask[n]) {
        const float4 g = *reinterpret_cast<const float4*>(vec + (int64_t)sg * vec_ld + c);
        v.x += g.x; v.y += g.y; v.z += g.z; v.w += g.w;
    } else {
        v = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    *p = v;
}
void add_segvec_cl(float* x, int L, int C, const float* vec, int vec_ld, const int* seg_of, const unsigned char* mask, hipStream_t s) {
    const int64_t n4 = (int64_t)L * (C >> 2);
    hipLaunchKernelGGL(k_add_segvec_cl, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, x, L, C, vec, vec_ld, seg_of, mask);
}

// Generator tail on a channels-last plane: leaky-ReLU -> conv_post (C -> 1, k taps, no bias) -> tanh, written de-gapped per utterance.
// A workgroup stages its 256 + k - 1 rows (activated once) and the C x k weights in LDS; read straight from global memory every
// sample re-read its k rows and issued C * k weight loads (0.8 ms for 786 MB = 1 TB/s).
constexpr int kPostMaxC = 32, kPostMaxK = 12;
// CT / KT > 0: channel and tap counts known at compile time (JP-Extra: 16 channels, 7 taps): the loops unroll, the C x k weights are wave-uniform scalar
// loads used as SGPR operands of the fmas (no LDS reads for them: the generic form issued C * k broadcast ds_reads per sample beside its k * C / 4 row
// reads and ran at 2.6 TB/s, LDS 50 % busy).  The compiler contracts the unrolled sums differently: f32 rounding apart from the generic form, which every
// caller of one model takes or does not take alike (batch row == single call, streamed == whole-sequence keep their bits).
template <int CT, int KT>
__global__ __launch_bounds__(256) void k_conv_post_tanh_cl(const float* x, int C_, int64_t L, const float* w, int k_, float slope,
                                                            const int* seg_start, const int* seg_len, const int64_t* pcm_off, int up,
                                                            float* pcm) {
    __shared__ float xs[(256 + kPostMaxK) * (kPostMaxC + 4)];
    __shared__ float ws[kPostMaxC * kPostMaxK];
    const int C = CT > 0 ? CT : C_, k = KT > 0 ? KT : k_;
    const int sg = blockIdx.y;
    const int64_t len = (int64_t)seg_len[sg] * up;
    const int64_t s0 = (int64_t)blockIdx.x * 256;
    if (s0 >= len) return;
    const int tid = threadIdx.x;
    const int half = k / 2;
    const int pitch = C + 4;
    const int c4n = C >> 2;
    const int64_t row0 = (int64_t)seg_start[sg] * up + s0 - half;   // plane row of staged row 0
    for (int idx = tid; idx < (256 + k - 1) * c4n; idx += 256) {
        const int r = idx / c4n, c4 = idx - r * c4n;
        const int64_t q = row0 + r;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q >= 0 && q < L) v = *reinterpret_cast<const float4*>(x + q * C + c4 * 4);
        v.x = fmaxf(v.x, v.x * slope);
        v.y = fmaxf(v.y, v.y * slope);
        v.z = fmaxf(v.z, v.z * slope);
        v.w = fmaxf(v.w, v.w * slope);
        *reinterpret_cast<float4*>(xs + r * pitch + c4 * 4) = v;
    }
    if constexpr (CT == 0) {
        for (int idx = tid; idx < C * k; idx += 256) ws[idx] = w[idx];   // [c][j]
    }
    __syncthreads();
    const int64_t sidx = s0 + tid;
    if (sidx >= len) return;
    float a = 0.f;
    if constexpr (CT > 0) {
#pragma unroll
        for (int j = 0; j < KT; ++j) {
            const float* r = xs + (tid + j) * (CT + 4);
#pragma unroll
            for (int c = 0; c < CT; c += 4) {
                const float4 v = *reinterpret_cast<const float4*>(r + c);
                a += w[c * KT + j] * v.x + w[(c + 1) * KT + j] * v.y + w[(c + 2) * KT + j] * v.z + w[(c + 3) * KT + j] * v.w;
            }
        }
    } else {
        for (int j = 0; j < k; ++j) {
            const float* r = xs + (tid + j) * pitch;
            for (int c = 0; c < C; c += 4) {
                const float4 v = *reinterpret_cast<const float4*>(r + c);
                a += ws[c * k + j] * v.x + ws[(c + 1) * k + j] * v.y + ws[(c + 2) * k + j] * v.z + ws[(c + 3) * k + j] * v.w;
            }
        }
    }
    pcm[pcm_off[sg] + sidx] = tanhf(a);
}
void conv_post_tanh_cl(const float* x, int C, int64_t L, const float* w, int k, float slope, const int* seg_start, const int* seg_len,
                       const int64_t* pcm_off, int nseg, int up, int64_t max_samples, float* pcm, hipStream_t s) {
    SBV2_REQUIRE(C <= kPostMaxC && (C & 3) == 0 && k <= kPostMaxK, "conv_post: more than 32 channels or 12 taps");
    const dim3 grid((unsigned)((max_samples + 255) / 256), nseg);
    if (C == 16 && k == 7) hipLaunchKernelGGL((k_conv_post_tanh_cl<16, 7>), grid, dim3(256), 0, s, x, C, L, w, k, slope, seg_start, seg_len, pcm_off, up, pcm);
    else hipLaunchKernelGGL((k_conv_post_tanh_cl<0, 0>), grid, dim3(256), 0, s, x, C, L, w, k, slope, seg_start, seg_len, pcm_off, up, pcm);
}

void fill_zero(void* p, size_t bytes, hipStream_t s) { HIP_CHECK(hipMemsetAsync(p, 0, bytes, s)); }

// ------------------------------------------------------------------------------------------------
// Streaming decode: out[c][j] = in[c][col0 + j] (0 outside [0, in.L)), mask[j] = column col0 + j exists; out.L columns
// ------------------------------------------------------------------------------------------------
__global__ void k_window_cols(Plane in, int col0, Plane out, unsigned char* mask) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int c = blockIdx.y;
    if (j >= out.L) return;
    const int q = col0 + j;
    const bool ok = q >= 0 && q < in.L;
    out.p[(size_t)c * out.ld + j] = ok ? in.p[(size_t)c * in.ld + q] : 0.f;
    if (c == 0 && mask) mask[j] = ok ? 1 : 0;
}
void window_cols(Plane in, int col0, Plane out, unsigned char* mask, hipStream_t s) {
    hipLaunchKernelGGL(k_window_cols, dim3((out.L + 255) / 256, out.C), dim3(256), 0, s, in, col0, out, mask);
}

// ------------------------------------------------------------------------------------------------
// Segment permutation: dst[tab[i].dst + e] = src[tab[i].src + e], e < tab[i].len  (PCM blocks received per device -> utterance order)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_copy_segments(const float* src, float* dst, const int64_t* tab) {
    const int64_t so = tab[3 * blockIdx.y], dof = tab[3 * blockIdx.y + 1], len = tab[3 * blockIdx.y + 2];
    for (int64_t e = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; e < len; e += (int64_t)gridDim.x * 1024) {
        if (e + 4 <= len && ((so + e) & 3) == 0 && ((dof + e) & 3) == 0) {
            *reinterpret_cast<float4*>(dst + dof + e) = *reinterpret_cast<const float4*>(src + so + e);
        } else {
            for (int64_t q = e; q < min(e + 4, len); ++q) dst[dof + q] = src[so + q];
        }
    }
}
void copy_segments(const float* src, float* dst, const int64_t* d_table, int n, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_copy_segments, dim3(64, n), dim3(256), 0, s, src, dst, d_table);
}

}  // namespace sbv2
