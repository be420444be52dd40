/* sbv2_hip.h — C ABI of libsbv2_hip.so: the MI355X-native replacement for the two ONNX Runtime sessions of sbv2_core.
 *
 * Drop-in boundary.  Every entry point cites the reference interface it replaces; paths are relative
 * to the reference repository (tuna2134/sbv2-api @ 2025-03-07).
 *
 * Conventions
 *   - return value 0 = ok; non-zero = error, message via sbv2_last_error() (thread local).  The Rust shim maps this to
 *     sbv2_core::error::Error::OtherError(String) (crates/sbv2_core/src/error.rs:29-30); nothing panics across the FFI.
 *   - inputs are borrowed for the duration of the call (the reference passes views: model.rs:68-90, bert.rs:13-14);
 *     outputs are either caller-allocated or owned buffers released with sbv2_pcm_free (the reference returns owned
 *     arrays: model.rs:108, bert.rs:21).
 *   - a handle is used by one caller at a time (`&mut Session` in the reference: bert.rs:7, model.rs:54).
 *   - model bytes: what the reference hands to load_model, i.e. an ONNX ModelProto (deberta.onnx / model_<name>.onnx: the initializers are
 *     read, the graph itself is replaced by the HIP path), or a whole `.sbv2` file (zstd(tar{model.onnx, style_vectors.json})) for the VITS
 *     handle; also the synthetic weight container "SBV2W001" (sbv2-api_amd/synth.py) used by tests and bench.py.  The bytes need not
 *     outlive *_create (the reference drops them unless max_loaded_models is set: tts.rs:171-175).
 */
#ifndef SBV2_HIP_H
#define SBV2_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sbv2_bert sbv2_bert;         /* replaces the `Session` held in TTSModelHolder.bert (tts.rs:40-46) */
typedef struct sbv2_vits sbv2_vits;         /* replaces the `Session` held in TTSModel.vits2 (tts.rs:32-38) */
typedef struct sbv2_pipeline sbv2_pipeline; /* new: batched, device-resident bert -> vits path */

/* Message of the last failing call on this thread (never NULL). */
const char* sbv2_last_error(void);
/* Number of visible HIP devices (0 when no GPU / no driver). */
int sbv2_device_count(void);

/* ---- load_model(model_file, bert = true)  crates/sbv2_core/src/model.rs:6-50 ---------------------------------- */
int sbv2_bert_create(const uint8_t* model, size_t model_len, int device, sbv2_bert** out);
void sbv2_bert_destroy(sbv2_bert* h);
int64_t sbv2_bert_hidden(const sbv2_bert* h); /* 1024 for deberta-v2-large */
/* arithmetic of the handle's Linear products: 0 = exact f32 MFMA, 2 = bf16x3 (two bf16 parts per operand), 3 = bf16x6 (three parts: f32-grade),
   4 = f16x3 (f16 hi + scaled f16 lo: 22 mantissa bits, three MFMAs per product; the default, SBV2_BERT_GEMM=f32|bf16x3|bf16x6|f16x3) */
int sbv2_bert_gemm_parts(const sbv2_bert* h);

/* ---- bert::predict(session, token_ids, attention_masks) -> Array2<f32>[S, 1024]  crates/sbv2_core/src/bert.rs:6-24
 * out: caller-allocated S * hidden floats, row-major [S][hidden]. */
int sbv2_bert_predict(sbv2_bert* h, const int64_t* token_ids, const int64_t* attention_mask, int64_t S, float* out);
/* new: n utterances at once; ids / mask concatenated, lens[n]; out = concatenated [sum S][hidden].
 * Row block i equals what sbv2_bert_predict returns for utterance i alone. */
int sbv2_bert_predict_batch(sbv2_bert* h, int64_t n, const int64_t* token_ids, const int64_t* attention_mask,
                            const int64_t* lens, float* out);

/* ---- load_model(model_file, bert = false)  crates/sbv2_core/src/model.rs:6-50 --------------------------------- */
int sbv2_vits_create(const uint8_t* model, size_t model_len, int device, sbv2_vits** out);
void sbv2_vits_destroy(sbv2_vits* h);
int64_t sbv2_vits_hop(const sbv2_vits* h);       /* samples per frame (512) */
int64_t sbv2_vits_bert_dim(const sbv2_vits* h);  /* 1024 */
int64_t sbv2_vits_style_dim(const sbv2_vits* h); /* 256 */
/* Decoder arithmetic chosen at create time (env SBV2_DECODER = f32 | bf16x3 | bf16 | f16): 0 = exact f32 MFMA, 1 = split-bf16 MFMA
 * (hi/lo operands, f32-grade, default), 2 = plain bf16 MFMA, 3 = fp16 MFMA operands. */
int sbv2_vits_decoder_mode(const sbv2_vits* h);
/* Device workspace (activation arena) the handle currently holds, in bytes.  It is sized by the largest recent batch: after a dozen
 * calls with much smaller batches it is released and re-grown on demand (a long-running server does not accumulate one arena per shape). */
int64_t sbv2_vits_workspace_bytes(const sbv2_vits* h);

/* ---- model::synthesize(session, bert_ori, x_tst, sid, tones, lang_ids, style_vector, sdp_ratio, length_scale,
 *                        noise_scale, noise_scale_w) -> Array3<f32>[1, 1, L]      crates/sbv2_core/src/model.rs:53-111
 * bert: [bert_dim][T] row-major (Array2 [1024, T], model.rs:56); x_tst/tones/lang: i64[T]; style: f32[style_dim].
 * *pcm: owned buffer of *pcm_len samples (release with sbv2_pcm_free).  noise_seed selects the counter-based noise
 * stream that replaces the graph's RandomNormalLike nodes (irrelevant when both noise scales are 0). */
int sbv2_vits_synthesize(sbv2_vits* h, const float* bert, const int64_t* x_tst, const int64_t* tones,
                         const int64_t* lang_ids, int64_t T, int64_t sid, const float* style_vector, float sdp_ratio,
                         float length_scale, float noise_scale, float noise_scale_w, uint64_t noise_seed, float** pcm,
                         int64_t* pcm_len);
void sbv2_pcm_free(float* pcm);

/* new: a batch of utterances in one call.  All per-token arrays are concatenated over utterances (utterance-major). */
typedef struct sbv2_batch {
    int64_t n;                       /* utterances */
    const int64_t* t_lens;           /* [n] T_text of each utterance */
    const int64_t* x_tst;            /* [sum T] phone ids */
    const int64_t* tones;            /* [sum T] */
    const int64_t* lang_ids;         /* [sum T] */
    const int64_t* sids;             /* [n] */
    const float* style_vectors;      /* [n][style_dim] */
    const float* bert;               /* concatenated [bert_dim][T_i] blocks; NULL in sbv2_pipeline_* (features come from DeBERTa) */
    float sdp_ratio, length_scale, noise_scale, noise_scale_w;
    uint64_t noise_seed;
    const int64_t* forced_durations; /* optional [sum T]: teacher-forced w_ceil (benchmark / parity mode), else NULL */
} sbv2_batch;

/* Runs the batch; results stay on the device until fetched.  pcm_lens: caller-allocated [n]. */
int sbv2_vits_synthesize_batch(sbv2_vits* h, const sbv2_batch* batch, int64_t* pcm_lens);
/* Concatenated PCM of the last batch (sum of pcm_lens samples) -> host.  capacity = samples `pcm` can hold; a batch that does not
 * fit is refused (no write). */
int sbv2_vits_fetch_pcm(sbv2_vits* h, float* pcm, int64_t capacity);
/* Device pointer to the concatenated PCM of the last batch (valid until the next call on the handle). */
const float* sbv2_vits_pcm_device(sbv2_vits* h, int64_t* total);
/* Copies the concatenated PCM of the last batch into caller-owned DEVICE memory (e.g. the send buffer of the RCCL gather). */
int sbv2_vits_copy_pcm_device(sbv2_vits* h, void* dst_device, int64_t capacity);
/* Blocks until every kernel the handle has launched is complete (the batch calls are asynchronous up to the PCM). */
int sbv2_sync(sbv2_vits* h);
/* Predicted integer durations w_ceil (before any forcing) and log-durations of the last batch, concatenated [sum T];
 * capacity = entries each non-NULL output can hold. */
int sbv2_vits_fetch_durations(sbv2_vits* h, int64_t* durations, float* logw, int64_t capacity);
/* Debug/parity: keep named intermediates of the next calls (x_emb, x, stats, z_p, z, dec_pre, dec_stage<i>). */
int sbv2_vits_set_trace(sbv2_vits* h, int on);
int sbv2_vits_get_trace(sbv2_vits* h, const char* name, int64_t utt, float* out, int64_t cap, int64_t* rows, int64_t* cols);

/* ---- new: the whole hot path for a batch, device resident:
 *   bert::predict per utterance (tts.rs:210-212) -> feature repeat by word2ph (tts_util.rs:129-154) -> model::synthesize.
 * token_ids / word2ph are concatenated over utterances; s_lens[n] gives S_i; sum(word2ph of utterance i) must equal t_lens[i]. */
int sbv2_pipeline_create(sbv2_bert* bert, sbv2_vits* vits, sbv2_pipeline** out);
void sbv2_pipeline_destroy(sbv2_pipeline* p);
int sbv2_pipeline_run(sbv2_pipeline* p, const sbv2_batch* batch, const int64_t* token_ids, const int64_t* s_lens,
                      const int64_t* word2ph, int64_t* pcm_lens);

/* Calls are pipelined: call n runs on execution context n % SBV2_PIPELINE_DEPTH (default 2; own stream + workspace, shared
 * weights) and returns once its kernels are enqueued, so the latency-bound DeBERTa / text / flow part of the next batch overlaps
 * the HiFi-GAN kernels of this one.  Every run gets a TICKET (1, 2, 3, ...: the call number); its results stay available until the
 * run `depth` calls later reuses its context, after which the ticket is stale and every call with it fails. */
int64_t sbv2_pipeline_last_ticket(sbv2_pipeline* p);
int sbv2_pipeline_wait(sbv2_pipeline* p, int64_t ticket);   /* blocks until that run is complete */
int sbv2_pipeline_sync(sbv2_pipeline* p);                   /* ... until every run is complete */
/* Concatenated PCM of a run in utterance order (waits for it); capacity = samples dst can hold (a longer result is refused);
 * dst_is_device != 0: dst is device memory. */
int sbv2_pipeline_fetch_pcm_ticket(sbv2_pipeline* p, int64_t ticket, float* dst, int64_t capacity, int dst_is_device);
int sbv2_pipeline_fetch_pcm(sbv2_pipeline* p, float* dst, int64_t capacity, int dst_is_device);   /* the most recent run */
/* Pinned (page-locked) host memory for PCM destinations: device -> host copies into it run at the full PCIe rate and overlap compute. */
void* sbv2_host_alloc(size_t bytes);
void sbv2_host_free(void* p);

/* ---- sbv2file.rs:15-37 `parse_sbv2file(bytes) -> (style_vectors, vits2)`: a .sbv2 file is zstd(tar{version.txt, model.onnx,
 * style_vectors.json}) (writer: scripts/convert/convert_model.py:156-175).  Both outputs are owned copies (sbv2_bytes_free).
 * Errors: "model not found: style_vectors" / "model not found: vits2" (Error::ModelNotFoundError, sbv2file.rs:31-36). ------------------- */
int sbv2_parse_sbv2file(const uint8_t* sbv2_bytes, size_t len, uint8_t** style_vectors, size_t* style_len, uint8_t** vits2, size_t* vits2_len);
void sbv2_bytes_free(uint8_t* p);
/* style.rs:11-17 `load_style`: {"shape": [n, dim], "data": [[..], ..]} -> owned f32 [n][dim] (release with sbv2_bytes_free) */
int sbv2_style_load(const uint8_t* json, size_t len, float** data, int64_t* n, int64_t* dim);
/* tts.rs:84-124 `load_aivmx` (cargo feature "aivmx"): the bytes are the VITS ONNX model itself (hand them to sbv2_vits_create); the style
 * table is ModelProto.metadata_props["aivm_style_vectors"] = base64(.npy, 2-D float32, C or Fortran order) -> owned f32 [n][dim]
 * (sbv2_bytes_free).  Errors: key absent; not 2-D ("expected 2D array", the reference's panic); not float32. */
int sbv2_aivmx_style_vectors(const uint8_t* aivmx_bytes, size_t len, float** data, int64_t* n, int64_t* dim);
/* style.rs:19-28 `get_style_vector`: out[dim] = mean + (style_vectors[style_id] - mean) * weight, mean = row 0 */
int sbv2_style_vector(const float* style_vectors, int64_t n, int64_t dim, int64_t style_id, float weight, float* out);

/* ---- new: multi-GPU.  The reference is one process, one device, batch 1; a batch of independent utterances is
 * sharded over the GPUs of a node (sorted by cost, longest-processing-time-first deal, a full weight replica per GPU, no data-path
 * collective) and the PCM is gathered to rank 0 over RCCL / xGMI (one all-gather of the sample counts + grouped send / recv).
 * RCCL is dlopen'ed on first use: single-GPU callers never load it. ------------------------------------------------------------------ */
/* rank_of[i] = rank that synthesises utterance i (host only, deterministic on every rank). */
int sbv2_deal(int64_t n, const int64_t* costs, int world, int32_t* rank_of);
/* Host only: the gather of a dealt batch.  Rank r's message is the PCM of its utterances in ascending caller index; the messages sit in
   rank order in the root's staging buffer.  counts[world] = samples per rank, table[3 n] = {offset in the staging buffer, offset in the
   caller's utterance order, samples} per utterance (the permutation sbv2_node_synthesize applies on the device). */
int sbv2_gather_plan(int64_t n, const int64_t* pcm_lens, const int32_t* rank_of, int world, int64_t* counts, int64_t* table);

/* (a) one process per GPU: rank 0 obtains a 128-byte id (ncclGetUniqueId) and hands it to the other ranks by any side channel. */
typedef struct sbv2_comm sbv2_comm;
int sbv2_comm_unique_id(uint8_t* id128);
int sbv2_comm_create(const uint8_t* id128, int rank, int world, int device, sbv2_comm** out);
void sbv2_comm_destroy(sbv2_comm* c);
int sbv2_comm_rank(const sbv2_comm* c);
int sbv2_comm_world(const sbv2_comm* c);
int sbv2_comm_barrier(sbv2_comm* c);
int sbv2_comm_max_f64(sbv2_comm* c, double* v);   /* in place: max over ranks (also a barrier) */
/* PCM of the run `ticket` of this rank's pipeline -> root: counts[world] = samples per rank (filled on every rank); on the root
 * dst_host (capacity samples; may be sbv2_host_alloc memory) receives the ranks' PCM concatenated in rank order. */
int sbv2_comm_gather_pcm(sbv2_comm* c, sbv2_pipeline* p, int64_t ticket, int root, float* dst_host, int64_t capacity, int64_t* counts);

/* (b) one process, N devices: the batched multi-GPU entry SURVEY.md §8b calls `sbv2_synthesize_batch`.  devices[ndev] are HIP ordinals (the model bytes are
 * the ones sbv2_bert_create / sbv2_vits_create take); one host thread + stream set per device, ncclCommInitAll when ndev > 1. */
typedef struct sbv2_node sbv2_node;
int sbv2_node_create(const uint8_t* bert_model, size_t bert_len, const uint8_t* vits_model, size_t vits_len, const int* devices, int ndev,
                     sbv2_node** out);
void sbv2_node_destroy(sbv2_node* nd);
int sbv2_node_devices(const sbv2_node* nd);
int sbv2_node_uses_rccl(const sbv2_node* nd);
/* Same inputs as sbv2_pipeline_run.  Outputs: pcm_lens[n]; pcm_host = the batch's PCM concatenated in the CALLER's utterance order
 * (capacity samples).  Utterance i's samples equal what a one-GPU call of the whole batch returns for it, bit for bit. */
int sbv2_node_synthesize(sbv2_node* nd, const sbv2_batch* batch, const int64_t* token_ids, const int64_t* s_lens, const int64_t* word2ph,
                         int64_t* pcm_lens, float* pcm_host, int64_t capacity);
int sbv2_node_last_deal(const sbv2_node* nd, int32_t* rank_of, int64_t n);   /* which device ran which utterance in the last call */

/* ---- new: streaming long-form synthesis (BASELINE configs[4]).  The reference synthesises one sentence per session.run and only splits
 * long text on '\n' (tts.rs:290-321).  Here DeBERTa / text encoder / durations / flow run whole-sequence (global attention) and the HiFi-GAN
 * decoder runs on fixed windows of chunk_frames + 2 x 16 halo frames, ONE hipGraph captured per window shape and replayed per chunk;
 * chunked output equals the whole-sequence output (the halo covers the generator's 13.4-frame receptive field per side). ------------- */
typedef struct sbv2_stream sbv2_stream;
/* batch->n must be 1; inputs as for sbv2_pipeline_run.  *total_samples = samples of the whole utterance.  The handles are busy until _end. */
int sbv2_stream_begin(sbv2_bert* bert, sbv2_vits* vits, const sbv2_batch* batch, const int64_t* token_ids, const int64_t* s_lens,
                      const int64_t* word2ph, int64_t chunk_frames, sbv2_stream** out, int64_t* total_samples);
/* next chunk -> dst (host; capacity samples, chunk_frames * hop always suffices); *n = samples written, 0 at the end */
int sbv2_stream_next(sbv2_stream* s, float* dst, int64_t capacity, int64_t* n);
int sbv2_stream_uses_graph(const sbv2_stream* s);
int64_t sbv2_stream_workspace_bytes(const sbv2_stream* s);
void sbv2_stream_end(sbv2_stream* s);

/* ---- test hooks (no reference counterpart) ------------------------------------------------------------------------ */
/* bucket(rel) for rel in [-(max_s-1), max_s-1] (transformers modeling_deberta_v2.py:57-69); host only, no GPU needed. */
int sbv2_debug_bucket_table(int64_t max_s, int64_t buckets, int64_t max_rel, int32_t* out);
/* y[Cout][L] = conv1d(x[Cin][L], w[Cout][Cin][k], bias, dilation, 'same' padding) with optional leaky-ReLU on the input,
 * through the production implicit-GEMM kernel; host buffers. */
int sbv2_debug_conv1d(int device, const float* x, const float* w, const float* bias, int64_t cin, int64_t cout, int64_t k,
                      int64_t L, int64_t dilation, float pre_slope, float* y);
/* y[Cout][L*stride] = conv_transpose1d(x[Cin][L], w[Cin][Cout][k], bias, stride, padding) via the polyphase path. */
int sbv2_debug_conv_transpose1d(int device, const float* x, const float* w, const float* bias, int64_t cin, int64_t cout,
                                int64_t k, int64_t L, int64_t stride, int64_t padding, float pre_slope, float* y);
/* Same contract as sbv2_debug_conv1d but through the channels-last bf16 MFMA kernel (mode 1 = split-bf16, 2 = plain bf16);
 * when iters > 0 also returns the mean time of `iters` further launches in *ms. */
int sbv2_debug_conv1d_cl(int device, const float* x, const float* w, const float* bias, int64_t cin, int64_t cout, int64_t k,
                         int64_t L, int64_t dilation, float pre_slope, int mode, int64_t iters, float* y, float* ms);
/* The named-tensor table an import of `model` (ONNX / .sbv2 / container; kind 1 = DeBERTa, 2 = VITS) produces, written back as an SBV2W001
 * container (owned, sbv2_bytes_free): tests compare it with the container the same weights were packed into.  Host only. */
int sbv2_debug_import_to_container(const uint8_t* model, size_t len, int kind, uint8_t** out, size_t* out_len);
/* Per-launch HIP-event timing of the implicit-GEMM kernel family between begin and end; end writes a JSON array
 * [{"kernel", "launches", "ms", "flop"}] (one entry per tile configuration) into json[cap]. */
int sbv2_prof_begin(void);
int sbv2_prof_end(char* json, int64_t cap);
/* Times `iters` launches of one dilated conv (device buffers, random data) and returns the mean kernel time in ms. */
int sbv2_debug_time_conv1d(int device, int64_t cin, int64_t cout, int64_t k, int64_t L, int64_t dilation, int64_t iters,
                           float* ms);
/* Small-grid threshold of the f32 GEMM (workgroups of the 64 x 64 tiling below which the one-wave 16 x 16 kernel runs; 0 = never; default
 * 128).  Returns the previous value; tests use it to compare both kernels bit for bit in one process. */
int sbv2_debug_set_skinny_max(int workgroups);
/* 1 (default): the ResBlocks of the wide decoder stages run on conv_clx.hip when the launch has >= 128 tiles; 2: always; 0: on conv_cl.hip.
   The two kernels sum in different orders (round 5: conv_clx on 16 x 16 x 32 MFMAs) and agree to f32 rounding; with 0 (and sbv2_debug_set_ksplit(0)) every
   launch size takes the same kernels and a batch row equals its single-utterance call bit for bit.  Returns the previous value. */
int sbv2_debug_set_clx(int on);
/* 1 (default): the small-grid dispatch of a single utterance's launches: gemm_bfs products with a long K loop split it over workgroups (K >= 2048) or over the
   four waves of a 32 x 32 tile (the 1024 x 1024 products) and add the partial sums in group order, and LayerNorm runs few columns per workgroup; other
   summation orders than the batch's launch shapes, so a single call and its batch row agree to f32 rounding; 0: the batch's launch shapes at every size (batch
   row == single call bit for bit; the bit-equality tests run on it).  Returns the previous value. */
int sbv2_debug_set_ksplit(int on);
/* the flow's attention on keys / values pre-split by the q | k | v product: 1 (default) for sequences of >= 4096 frames and launches of <= 64
   workgroups, 2 at every length, 3 at every length on the un-pipelined kernel (k_vits_flash_x3p, the fallback for head dimensions that are no multiple
   of 8), 4 at every length on the pipelined kernel's 8-wave shape (the batch shape, forced for the test), 0 never (converted per key tile inside the
   attention kernel); bit-identical; returns the previous setting */
int sbv2_debug_set_flash_parts(int on);
/* Same contract as sbv2_debug_conv1d_cl (mode 1) through conv_clx.hip: x is split into bf16 parts of lrelu(x, pre_slope) first (split_cl), the
   convolution reads the parts; y = (conv + bias + res) * beta; ys_sum (optional) = hi + lo of the parts of lrelu(y, 0.1) the epilogue emits. */
int sbv2_debug_conv1d_clx(int device, const float* x, const float* w, const float* bias, const float* res, int64_t cin, int64_t cout, int64_t k,
                          int64_t L, int64_t dilation, float pre_slope, float beta, int64_t iters, float* y, float* ys_sum, float* ms);
/* Diagnostics (MI355X_MICROARCH.md "DVFS give-back" item 6): the dominant decoder convolution (C x C, k taps, channels-last, split-bf16,
   128-row workgroups) on random data, `seconds` of back-to-back launches, then out4 = {in-kernel shader clock in MHz = d s_memtime /
   d s_memrealtime x 100 (median over workgroups), ms per launch, shader cycles of a workgroup's chunk loop, workgroups stamped}.
   abl: 0 = the kernel, 1 = without its MFMAs, 2 = its MFMAs only, 3 = staging + barriers only. */
int sbv2_debug_conv_cl_clock(int device, int64_t C, int64_t k, int64_t dilation, int64_t L, int abl, double seconds, double* out4);
/* Diagnostics: the life of a conv_clx workgroup (the ResBlock convolutions of the 128- / 256-channel decoder stages: scripts/convert/convert_model.py:97-110
   exports them; no reference counterpart).  kind 1 = conv1 (parts in, parts out), 2 = conv2 (+ residual in, f32 + parts out), 3 = a branch's last conv2
   (accumulating).  `seconds` of back-to-back launches, then the stamps of one more: 12 words per workgroup {loop start / end in shader cycles and in
   100 MHz ticks, kernel entry, last store issued, stores acknowledged (100 MHz), HW_ID | XCC_ID << 32, epilogue: behind the post-loop barrier, its
   global reads arrived, the first half's stores issued (100 MHz), 0}; *ms_per_launch is of the un-stamped kernel.  `variant` selects a kernel variant under
   test in builder experiments; the library holds only variant 0 (the product kernel), other values run the same kernel. */
int sbv2_debug_clx_timeline(int device, int64_t C, int64_t k, int64_t dilation, int64_t L, int kind, int variant, double seconds, uint64_t* stamps,
                            int64_t capacity_words, int64_t* workgroups, double* ms_per_launch);
/* Diagnostics for the f16x3 operand format (DeBERTa's and the flow's 1x1 products): the split of an activation into the f16 hi / scaled-lo pair clamps
   finite values beyond +-65504 (NaN and infinities propagate).  enable = 1 / 0 switches the device-side counter of clamped values on / off for planes
   allocated from then on (-1: leave as is; the SBV2_F16X3_SATCOUNT=1 environment variable switches it on from the start); *count (optional) receives the
   number of clamped values since the last call, on `device`.  A non-zero count on a real checkpoint means: run with SBV2_BERT_GEMM=bf16x6. */
int sbv2_debug_f16x3_saturation(int device, int enable, uint64_t* count);
/* 1 (default): the fused ResBlock steps of the <= 64-channel decoder stages run on respair_x16.hip where it exists (C = 32 / 64, k = 7 / 11: 16x16x32 MFMAs,
   step pairs in the K dimension; f32 rounding apart) and on respair_clx.hip (split-bf16, k in {3, 7, 11}) otherwise; 2 (SBV2_RESPAIR_X16=0): respair_clx.hip
   at every shape; 0: respair_cl.hip (the bits of mode 2).  Returns the previous value. */
int sbv2_debug_set_respair_clx(int on);
/* One fused ResBlock1 step y' = beta (conv2(lrelu(conv1(lrelu(x), dilation) + b1)) + b2 + x) [+ y when accumulate], masked by mask[n / mask_div] (may be
   null), channels-last x / y [N][C], w [C][C][k], split-bf16, through respair_cl.hip (variant 0), the default dispatch (variant 1: respair_x16.hip /
   respair_clx.hip) or respair_clx.hip at every shape (variant 2).  Test hook. */
int sbv2_debug_respair(int device, const float* x, const float* w1, const float* w2, const float* b1, const float* b2, int64_t C, int64_t N, int64_t k,
                       int64_t dilation, const uint8_t* mask, int64_t mask_div, float beta, int accumulate, int variant, float* y);
/* 1 (default; SBV2_RESBRANCH): the k = 3 branches of the 128- / 64- / 32- / 16-channel decoder stages and the k = 7 / 11 branches of the 16-channel stage
   run their three steps in ONE launch (resbranch_clx.hip: y_1, y_2 stay on the chip, 2 plane passes through HBM per branch instead of 6); 2: the k = 3
   branches only; 0: three fused-step launches (same bits; at 128 channels six conv_clx launches: f32 rounding apart).  Returns the previous value. */
int sbv2_debug_set_resbranch(int on);
/* 1 (default; SBV2_UPX): the ConvTranspose1d of the wide decoder stages (large launches) runs as ONE phased conv_clx.hip launch on pre-split operands
   (rows = (phase, cout), every phase on its own input taps); read when the weights are packed: 2 = rows in plain (phase, channel) order (same bits), 3 = the
   union of all phases' taps with a zero tap per phase (f32 rounding apart); 0: conv_cl.hip's phase groups (f32 rounding apart).  Returns the previous value. */
int sbv2_debug_set_upx(int on);
/* ConvTranspose1d(lrelu(x, pre_slope)) [cin][L] -> y [cout][L * stride] (weight [cin][cout][k], padding (k - stride) / 2) through the phased conv_clx launch;
   mask (may be null): input position n and its `stride` outputs are kept iff mask[n / mask_div]; ys_sum (may be null): hi + lo of the bf16 parts of
   lrelu(y, 0.1) the launch writes for the ResBlocks.  iters > 0: *ms = average duration.  Test / measurement hook (scripts/convert/convert_model.py:97-110's
   `ups` layers). */
int sbv2_debug_conv_transpose1d_clx(int device, const float* x, const float* w, const float* bias, int64_t cin, int64_t cout, int64_t k, int64_t L,
                                    int64_t stride, float pre_slope, const uint8_t* mask, int64_t mask_div, int64_t iters, float* y, float* ys_sum, float* ms);
/* A whole ResBlock1 branch (HifiGanResidualBlock.forward, modeling_vits.py:455-463; the graph of scripts/convert/convert_model.py:97-110): three steps
   y_q = conv2_q(lrelu(conv1_q(lrelu(y_{q-1}), dilations[q]) + b1_q)) + b2_q + y_{q-1}, result beta * y_3 [+ y when accumulate], masked by mask[n / mask_div]
   (a power of two; mask may be null) at every layer; channels-last x / y [N][C], w [6][C][C][k] and bias [6][C] in the order conv1_0, conv2_0, conv1_1, ...,
   split-bf16, C in {16, 32, 64, 128}; variant 0 = three launches of the fused step (respair_clx.hip, C <= 64), 1 = one launch (resbranch_clx.hip), 2 = six
   launches of conv_cl.hip (any C).  iters > 0: *ms = average duration of
   `iters` further runs; stamps (variant 1, may be null): 16 words per workgroup of one more, stamped launch (s_memtime at entry [0], window converted [1], end
   of step 1 / 2 / 3 [2 .. 4], stores issued [6]; s_memrealtime at entry / exit [14, 15]).  Test / measurement hook. */
int sbv2_debug_resbranch(int device, const float* x, const float* w, const float* bias, int64_t C, int64_t N, int64_t k, const int64_t* dilations,
                         const uint8_t* mask, int64_t mask_div, float beta, int accumulate, int variant, int64_t iters, float* y, float* ms,
                         uint64_t* stamps, int64_t stamps_cap);
/* Diagnostics: one fused ResBlock step (respair_cl.hip, split-bf16, C = 16 / 32 / 64) on random data, `seconds` of back-to-back launches,
   then out[0] = in-kernel clock (MHz, median over workgroups), out[1] = ms per launch, out[2] = workgroups stamped, out[2 + i] = median shader
   cycles from a workgroup's entry to phase stamp i (1 = conv1 window staged, 7 / 8 / 9 / 10 = first chunk's MFMAs / barrier / next chunk staged /
   barrier, 2 = conv1 done, 3 = intermediate written, 4 = barrier, 5 = conv2 done, 6 = stores issued).  variant 0 = the stamped instantiation of
   respair_cl (abl bits: 1 cache-hot reads, 2 no stores, 4 no MFMAs, 8 no window conversion, 16 no intermediate epilogue), 1 = the product respair_cl
   (time only), 2 = respair_clx stamped, 3 = the product respair_clx (time only), 4 = respair_x16 stamped (C = 32 / 64, k = 7 / 11; stamps 7 - 9: chunk
   pair 0 done / chunk pair 1 converted / first pair done). */
int sbv2_debug_respair_clock(int device, int64_t C, int64_t k, int64_t dilation, int64_t L, int variant, int abl, double seconds, double* out, int nout);
/* y[M][N] = act(w[M][K] x[K][N] + bias) (+ res) through the split-bf16 1x1 GEMM (gemm_bfs.hip; parts 2 = bf16x3, 3 = bf16x6).  split_out != 0:
   the result is also emitted as that many bf16 parts and y returns their sum.  iters > 0: average launch time in *ms.  Test hook. */
int sbv2_debug_gemm_bfs(int device, const float* x, const float* w, const float* bias, const float* res, int64_t M, int64_t N, int64_t K,
                        int parts, int act, int split_out, int64_t iters, float* y, float* ms);
/* The same product for TWO inputs xa, xb [K][N] launched alternately (iters + 2 launches) on ONE K-split scratch buffer that is never cleared in between;
   ya / yb = the last result of each.  A workgroup that summed a stale partial sum (the other input's, left in its XCD's L2 by the previous launch) shows up as a
   wrong result: the race screen of gemm_bfs.hip's cross-workgroup K split.  parts as above. */
int sbv2_debug_gemm_bfs_alt(int device, const float* xa, const float* xb, const float* w, const float* bias, const float* res, int64_t M, int64_t N, int64_t K,
                            int parts, int64_t iters, float* ya, float* yb);

#ifdef __cplusplus
}
#endif
#endif /* SBV2_HIP_H */
