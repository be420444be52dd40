// Launchers for the non-GEMM kernels of the hot path (ops.hip).  All activations are channel-major planes.
#pragma once
#include "common.h"

namespace sbv2 {

// one attention problem = (utterance, head)
struct AttnGroup {
    int64_t qk_off;  // element offset of this head's first row / this utterance's first column in Q, K, ctx planes
    int64_t s_off;   // offset of the [T][lds] score block
    int64_t aux_off; // offset of the per-group auxiliary block (DeBERTa: c2p^T; VITS: P window [2w+1][T])
    int64_t aux2_off;  // DeBERTa: p2c block
    int T;           // sequence length of the utterance
    int lds;         // pitch of the score block
    int col0;        // first column of the utterance in the packed plane (for masks)
    int head;
};

void deberta_embed_ln(const int* ids, const float* emb, int H, const float* gamma, const float* beta, float eps,
                      Plane out, hipStream_t s);
// out = mask * ( act(LN_c(in)) + res )   (act: ACT_NONE / ACT_GELU; res optional; in == out allowed)
// split != null: the result is also written as bf16 parts (the operand format of gemm_bfs.hip)
void layernorm_ch(Plane in, Plane out, const float* gamma, const float* beta, float eps, int act, const float* res,
                  int ldr, const unsigned char* mask, hipStream_t s, const SplitPlanes* split = nullptr);
// out = mask * gelu(LN_c(depthwise_conv_k3(in, dilation) + b))
void dds_dw_ln_gelu(Plane in, Plane out, const float* w, const float* b, int dil, const float* gamma,
                    const float* beta, const unsigned char* mask, hipStream_t s);

// fused disentangled attention for short sequences (attn_deberta.hip); deberta_attention_fits says whether a batch qualifies
bool deberta_attention_fits(int maxT, int wlen, int dh);
// the tiled variant for 65 .. 128 tokens (same arguments; the window is the one reachable by the longest sequence of the batch)
bool deberta_attention_long_fits(int T, int dh);   // > 128 tokens: key-tile loop with an online softmax, one workgroup per 32-query tile
void deberta_attention_long(const AttnGroup* groups, int ngroups, int maxT, const float* Q, const float* K, int ld, const float* V,
                            const float* posk, const float* posq, int ldp, int win_lo, int wlen, const int* tab, int tab_center, int span,
                            float inv_scale, const unsigned char* tok_mask, int dh, float* ctx, int ldc, hipStream_t s);
bool deberta_attention128_fits(int T, int dh);
void deberta_attention128(const AttnGroup* groups, int ngroups, const float* Q, const float* K, int ld, const float* V, const float* posk,
                          const float* posq, int ldp, int win_lo, int wlen, const int* tab, int tab_center, int span, float inv_scale,
                          const unsigned char* tok_mask, int dh, float* ctx, int ldc, hipStream_t s);
void deberta_attention(const AttnGroup* groups, int ngroups, const float* Q, const float* K, int ld, const float* V, const float* posk,
                       const float* posq, int ldp, int win_lo, int wlen, const int* tab, int tab_center, int span, float inv_scale,
                       const unsigned char* tok_mask, int dh, float* ctx, int ldc, hipStream_t s);
void deberta_softmax(const AttnGroup* groups, int ngroups, int maxT, float* S, const float* c2pT, const float* p2c,
                     const int* bucket_tab, int tab_center, int span, int win_lo, int win_ld, float inv_scale,
                     const unsigned char* tok_mask, hipStream_t s);
void vits_softmax(const AttnGroup* groups, int ngroups, int maxT, float* S, const float* Q, int ldq, int dk,
                  const float* erk, int window, float qscale, float* pwin, hipStream_t s);
// fused QK^T + relative-key term + online softmax + PV + relative-value term (attn_flash.hip); Q, K, V, ctx: k-major planes
bool flash_pipelined_usable(int dk);   // attn_flash.hip: k_vits_flash_x3q takes this head dimension 
void vits_flash_attention_parts(const AttnGroup* groups, int ngroups, int maxT, const float* Q, int ld, const SplitPlanes& kv, int k_row0,
                                int v_row0, float* ctx, int ldc, int dk, const float* erk, const float* erv, int window, float qscale,
                                hipStream_t s, int pipelined = 1);   // pipelined: 0 = k_vits_flash_x3p, 1 = k_vits_flash_x3q (shape by grid), 2 = ... 8-wave shape;   // keys / values pre-split into two bf16 parts (rows of kv); same bits as the split variant below
void vits_flash_attention(const AttnGroup* groups, int ngroups, int maxT, const float* Q, const float* K, const float* V, int ld, float* ctx,
                          int ldc, int dk, const float* erk, const float* erv, int window, float qscale, bool split_bf16, hipStream_t s);
void vits_relv_add(const AttnGroup* groups, int ngroups, int maxT, float* ctx, int ldc, int dk, const float* erv,
                   int window, const float* pwin, hipStream_t s);

void add_segvec(Plane x, const float* vec, int vec_ld, const int* seg_of, int div, const unsigned char* mask,
                hipStream_t s);
void linear_vec(const float* W, const float* bias, int M, int K, const float* v, int ldv, float* out, int ldo, int B,
                hipStream_t s);
void gather_rows(const float* table, int K, const int* idx, float* out, int B, hipStream_t s);
void text_embed(const int* phones, const int* tones, const int* langs, const int* seg_of, const float* emb,
                const float* tone_emb, const float* lang_emb, const float* bertproj, int ldb, const float* styleproj,
                int lds_, float scale, Plane out, hipStream_t s);
void convflow_pre(const float* z0, const float* w, const float* b, Plane cond, Plane out, const unsigned char* mask,
                  hipStream_t s);
void spline_inverse(Plane params, float* z0, float* z1, int nbins, float tail, float inv_sqrt_f,
                    const unsigned char* mask, int L, hipStream_t s);
void swap_rows(float* a, float* b, int L, hipStream_t s);
void flip_channels(Plane in, Plane out, hipStream_t s);
void affine_reverse(float* z0, float* z1, const float* m, const float* logs, const float* scale, const unsigned char* mask, int L,
                    hipStream_t s);   // (x - m) * exp(-logs), or * scale when logs is null (exp(-logs) folded into the weight file)
void durations(const float* sdp, const float* dp, float ratio, float length_scale, const unsigned char* mask, int L,
               float* logw, int* dur, hipStream_t s);
void noise_fill(float* out, int ld, int rows, const int* seg_of, const int* seg_start, const int* seg_len, const int* seg_utt, int L,
                uint64_t seed, int stream_id, float scale, hipStream_t s);
void expand_frames(Plane m_p, Plane logs_p, const int* tok_of_frame, const int* seg_of, const int* seg_start,
                   const int* seg_len, const int* seg_utt, uint64_t seed, float noise_scale, Plane out, hipStream_t s);
// out[c][j] = in[c][col0 + j] or 0 outside the plane; mask[j] = 1 where the column exists (the chunk window of the streaming decoder)
void window_cols(Plane in, int col0, Plane out, unsigned char* mask, hipStream_t s);
// dst[tab[3i + 1] + e] = src[tab[3i] + e] for e < tab[3i + 2], i < n (device table)
void copy_segments(const float* src, float* dst, const int64_t* d_table, int n, hipStream_t s);
void conv_post_tanh(Plane x, const float* w, int k, float slope, const int* seg_start, const int* seg_len,
                    const int64_t* pcm_off, int nseg, int up, int64_t max_samples, float* pcm, hipStream_t s);
void transpose_out(Plane in, int col0, int T, float* out, hipStream_t s);  // out[t][c] = in[c][col0+t]
void gather_cols(Plane in, const int* map, Plane out, hipStream_t s);       // out[c][n] = map[n]>=0 ? in[c][map[n]] : 0
void fill_zero(void* p, size_t bytes, hipStream_t s);
// channels-last helpers: x[L][C]
void add_segvec_cl(float* x, int L, int C, const float* vec, int vec_ld, const int* seg_of, const unsigned char* mask, hipStream_t s);
void conv_post_tanh_cl(const float* x, int C, int64_t L, const float* w, int k, float slope, const int* seg_start, const int* seg_len,
                       const int64_t* pcm_off, int nseg, int up, int64_t max_samples, float* pcm, hipStream_t s);

}  // namespace sbv2
