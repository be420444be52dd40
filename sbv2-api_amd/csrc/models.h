// Host-side model objects behind the C ABI: weights resident in HBM, per-call workspace arena,
// batched forward passes expressed as sequences of kernel launches on one HIP stream.
#pragma once
#include "common.h"
#include "ops.h"

#include <memory>

namespace sbv2 {

// A convolution packed for conv_cl.hip (bf16 MFMA fragment blocks); parts = 2 keeps a bf16 hi and a bf16 lo copy.
struct ClConv {
    void* w = nullptr;
    void* wxp = nullptr;   // split-bf16, M == K in {32, 64}: the same weights as step-pair fragments of the 16x16x32 MFMA (pack_step_pairs; respair_x16.hip), else null
    void* wx = nullptr;    // split-bf16, M % 64 == 0, K % 32 == 0: the same weights as [hi | lo] fragments of the 16x16x32 MFMA (pack_clx16; conv_clx.hip), else null
    void* wp = nullptr;    // 16 x 16 convolutions of the fused ResBlock step: tap-pair fragments [pair][part][64 lanes][8] (pack_cl_pairs), else null
    float* bias = nullptr;
    int M = 0, K = 0, k = 1, nmt = 0, tm = 1, parts = 0;
};

// Weights in the layout gemm_conv expects: [tap][Cin][lda] (k-major, Cout contiguous), plus (optionally) the bf16 fragment
// packing of the same weights: when `cl.parts != 0` conv_plain / linear_tokmajor run on the bf16 matrix cores.
struct PackedConv {
    float* w = nullptr;
    float* bias = nullptr;
    int cout = 0, cin = 0, k = 1, lda = 0;
    ClConv cl;
    BfsWeights bfs;   // 1x1 products only: pre-split bf16 fragments for gemm_bfs.hip (parts = 0: not packed)
    int64_t tap_stride() const { return (int64_t)cin * lda; }
};
// One ConvTranspose1d split into groups of output phases that share the same input taps.
struct PackedUpsample {
    struct Grp {
        float* w = nullptr;
        int lda = 0, ntaps = 0, nph = 0;
        int shift[kMaxTaps];
        int phase_off[kMaxPhases];
    };
    std::vector<Grp> groups;
    float* bias = nullptr;
    int cin = 0, cout = 0, stride = 1;
};

// Packed segments along the time axis: utterance i occupies columns [start[i], start[i]+len[i]); every start is a
// multiple of 4 and at least `gap` zero columns separate neighbours (the zero padding of every 'same' convolution).
struct SegLayout {
    int n = 0, L = 0;
    std::vector<int> start, len;
    int* d_seg_of = nullptr;
    unsigned char* d_mask = nullptr;
    int* d_start = nullptr;
    int* d_len = nullptr;
    int max_len() const {
        int m = 0;
        for (int v : len) m = std::max(m, v);
        return m;
    }
};
SegLayout make_layout(const std::vector<int>& lens, int gap, Arena& arena, hipStream_t stream, const unsigned char* extra_mask = nullptr, int round_to = 4);

class WeightStore;
// w is [M][K][k] (Conv1d layout); K is zero-padded to a multiple of 16
ClConv pack_cl(WeightStore& ws, const float* w, int M, int K, int k, int parts, const float* bias);
void* pack_clx16(WeightStore& ws, const float* w, int M, int K, int k);
// A ConvTranspose1d (weight [cin][cout][k], stride s, padding (k - s) / 2) as ONE phased product for conv_clx.hip (round 6): rows (phase, cout), taps = the
// union of the phases' input taps (zero weights where a phase has none; 2 .. 5 taps); tap t
// reads input position n + shift0 - t.  wx == nullptr: the shape does not fit (decoder_cl.cpp falls back to conv_cl.hip's phase groups).
struct ClUpX {
    void* wx = nullptr;
    float* bias = nullptr;      // per row (phase, cout)
    int M = 0, K = 0, ntaps = 0, shift0 = 0, nph = 0, cout = 0;
    int group = 1;              // ConvClxParams::phase_group of the row order the weights and the bias were packed in
    double alg_macs_per_pos = 0;   // multiply-adds per input position that are not padding (the launch's algorithmic FLOP for the profile)
    int phase_off[kMaxPhases] = {0};
    int phase_tap0[kMaxPhases] = {0};   // ConvClxParams::phase_tap0: where each phase's own taps start in the union of all phases' taps
};
ClUpX build_upx(WeightStore& ws, const float* wt, const float* ub, int cin, int cout, int k, int s, bool parts_out);
void* pack_step_pairs(WeightStore& ws, const float* w, int C, int k);   // w [C][C][k], C in {32, 64} -> step-pair fragments (split-bf16) for respair_x16.hip
void* pack_cl_pairs(WeightStore& ws, const float* w, int k);   // w [16][16][k] -> tap-pair fragments (split-bf16) for respair_clx's 16-channel kernel
// w is [M][K] (Linear / 1x1 conv): bf16 parts (2 = hi + lo, 3 = hi + mid + lo) as MFMA A fragments; K must be a multiple of 16
BfsWeights pack_bfs(WeightStore& ws, const float* w, int M, int K, int parts);

class WeightStore {
  public:
    // cl_parts: 0 = f32 MFMA only, 1 = also pack plain-bf16 fragments, 2 = also pack split-bf16 (hi/lo) fragments, 3 = fp16 fragments
    explicit WeightStore(const Blob& b, int cl_parts = 0) : blob_(b), cl_parts_(cl_parts) {}
    int cl_parts() const { return cl_parts_; }
    void set_cl_parts(int parts) { cl_parts_ = parts; }
    // 1x1 products loaded from now on also get their pre-split fragments for gemm_bfs.hip (0 = none, 2 = bf16x3, 3 = bf16x6)
    int bfs_parts() const { return bfs_parts_; }
    void set_bfs_parts(int parts) { bfs_parts_ = parts; }
    ~WeightStore();
    float* upload(const float* host, size_t n);
    float* tensor(const std::string& name);                      // raw copy
    // raw copy of a tensor whose shape is fixed by the config: a container whose tensor has another shape is rejected here, before
    // any kernel indexes the weights (trailing / leading 1-dims are ignored: [C] == [C, 1] == [1, C])
    float* tensor(const std::string& name, std::initializer_list<int64_t> dims);
    void expect(const PackedConv& c, const std::string& prefix, int cout, int cin, int k) const;
    PackedConv conv(const std::string& prefix, bool bias = true);  // <prefix>.weight [Cout][Cin][k] (+ .bias)
    PackedConv conv_cat(const std::vector<std::string>& prefixes);  // 1x1 convs of one input stacked along Cout
    PackedConv linear(const std::string& prefix);                // <prefix>.weight [Cout][Cin] + .bias
    PackedUpsample upsample(const std::string& prefix, int stride, int padding);  // weight [Cin][Cout][k]
    size_t bytes() const { return bytes_; }
    const Blob& blob() const { return blob_; }

  private:
    const Blob& blob_;
    int cl_parts_ = 0, bfs_parts_ = 0;
    std::vector<void*> allocs_;
    size_t bytes_ = 0;
};

// helpers shared by both models
void conv_plain(const PackedConv& w, Plane x, Plane y, int dil, int pad_l, const unsigned char* mask, int mask_div, hipStream_t s,
                int act = ACT_NONE, float pre_slope = 1.0f, const Plane* res = nullptr, float alpha = 1.0f, float beta = 1.0f,
                int accumulate = 0);
// 1x1 product with pre-split bf16 operands (gemm_bfs.hip): y (f32 plane) and / or ys (bf16 parts) receive act(W x + b) * alpha (+ res) * beta
void conv_bfs(const PackedConv& w, const SplitPlanes& xs, const Plane* y, const SplitPlanes* ys, const unsigned char* mask, int mask_div,
              hipStream_t s, int act = ACT_NONE, const Plane* res = nullptr, float alpha = 1.0f, float beta = 1.0f, int y_rows = -1, int ys_row0 = 0,
              const BfsSplitK* sk = nullptr);
// (y_rows >= 0: only rows < y_rows go to y; ys_row0: only rows >= ys_row0 go to ys)
// y[n][m] (token-major) = x^T W + b : the "weights as B operand" form used for V^T
bool conv_km_to_cl(const PackedConv& w, Plane x, float* y, int ldy, int dil, int pad_l, const unsigned char* mask, int mask_div,
                   hipStream_t s);
bool conv_cl_to_km(const PackedConv& w, const float* x, int ldx, Plane y, int dil, int pad_l, const unsigned char* mask, int mask_div,
                   hipStream_t s, float pre_slope, const Plane* res);
void linear_tokmajor(const PackedConv& w, Plane x, float* y, int ldy, hipStream_t s);

// the library-wide default arithmetic of DeBERTa's 1x1 products (bert.cpp documents the choice): bf16 parts per operand, 0 = exact f32
int default_bert_bfs_parts();
bool flash_parts_enabled();   // vits.cpp: the flow's attention reads pre-split keys / values
int set_flash_parts(int on);
struct BertConfig {
    int vocab, hidden, layers, heads, inter, buckets, max_rel;
    float eps;
    int conv_k = 0;          // DebertaV2Encoder.conv (ConvLayer after layer 0) when conv_kernel_size > 0
    int conv_act = ACT_TANH; // ConvLayer's default activation is tanh (modeling_deberta_v2.py:453)
};

class BertModel {
  public:
    BertModel(const Blob& blob, int device);
    ~BertModel();
    // A second execution context on the same weights: own stream and workspace (micro-batch pipelining)
    BertModel* clone() const;
    int device() const { return device_; }
    const BertConfig& cfg() const { return cfg_; }
    int gemm_parts() const { return bfs_parts_; }   // 0 = exact-f32 products, 2 = bf16x3, 3 = bf16x6, 4 = f16x3
    // ids/mask concatenated over utterances; result stays on the device (out_, layout_)
    void forward(int n, const int64_t* ids, const int64_t* mask, const int64_t* lens);
    void copy_out(float* host);  // [sum S][hidden], utterances concatenated
    const Plane& out() const { return out_; }
    const SegLayout& layout() const { return layout_; }
    hipStream_t stream() const { return stream_; }
    static std::vector<int> bucket_table(int maxS, int buckets, int max_rel);  // bucket(rel) for rel in [-(maxS-1), maxS-1]

  private:
    struct Layer {
        PackedConv q, k, v, o, ffn1, ffn2;
        PackedConv qkv;   // rows q | k | v: one product for the fused-attention path (k-major V)
        float *ln1_g, *ln1_b, *ln2_g, *ln2_b;
        Plane pos_k, pos_q;  // key_proj / query_proj of the LayerNorm'ed relative embeddings: [H][2*span]
    };
    int device_;
    BertConfig cfg_;
    int bfs_parts_ = 0;   // parts code: 0 = exact-f32 products, 2 = bf16x3, 3 = bf16x6, 4 = f16x3 (gemm_bfs.hip)
    std::shared_ptr<WeightStore> ws_;
    float *emb_, *emb_g_, *emb_b_;
    PackedConv conv_;                       // encoder.conv.conv (k = conv_k)
    float *conv_g_ = nullptr, *conv_b_ = nullptr;   // encoder.conv.LayerNorm
    std::vector<Layer> layers_;
    Arena arena_;
    hipStream_t stream_ = nullptr;
    unsigned* sk_counters_ = nullptr;   // this context's arrival counters for gemm_bfs' small-grid K split (kSkCounters, zero between launches)
    static constexpr int kSkCounters = 1024;
    static constexpr size_t kSkWsBytes = (size_t)48 << 20;   // (round 6: the batch's K = 4096 product: 272 tiles x 2 groups x 64 KB)
    SatWatch sat_watch_;                // f16x3 clamp warning of this handle (common.h)
    Plane out_;
    SegLayout layout_;
};

struct VitsConfig {
    int n_vocab, n_tones, n_langs, n_speakers, hidden, inter, filter, heads, enc_layers, enc_kernel, window, gin, style_dim,
        bert_dim, cond_layer_idx, flow_n, flow_layers, flow_kernel, dp_filter, dp_kernel, sdp_kernel, sdp_flows, sdp_bins,
        sdp_dds_layers;
    float sdp_tail;
    std::vector<int> up_rates, up_kernels, res_kernels;
    std::vector<std::vector<int>> res_dilations;
    int up_initial;
    int hop() const {
        int h = 1;
        for (int r : up_rates) h *= r;
        return h;
    }
};

constexpr int kStreamBurst = 8;   // windows per graph replay of the streaming decoder after an utterance's first chunk
struct VitsBatch {
    int n = 0;
    const int64_t* t_lens = nullptr;   // [n]
    const int64_t* phones = nullptr;   // concatenated
    const int64_t* tones = nullptr;
    const int64_t* langs = nullptr;
    const int64_t* sids = nullptr;     // [n]
    const float* styles = nullptr;     // [n][style_dim]
    const float* bert_host = nullptr;  // concatenated [bert_dim][T_i] blocks, or null when bert_dev is given
    const Plane* bert_dev = nullptr;   // device plane [bert_dim][*] ...
    const int* bert_map = nullptr;     // ... with HOST map: text column (packed, no gaps, utterance-major) -> source column
    float sdp_ratio = 0.f, length_scale = 1.f, noise_scale = 0.f, noise_scale_w = 0.f;
    uint64_t seed = 0;
    const int64_t* forced_durations = nullptr;  // concatenated, optional
    int utt0 = 0;                               // index of the first utterance in the caller's batch (noise stream keys) ...
    const int64_t* utt_ids = nullptr;           // ... or, for an arbitrary subset (a shard dealt to this GPU), every utterance's index
    hipStream_t after_stream = nullptr;         // when set, the forward's kernels wait for the work queued on this stream
    bool skip_decoder = false;                  // stop after the flow (streaming: the decoder then runs chunk by chunk, stream_*)
};

class VitsModel {
  public:
    VitsModel(const Blob& blob, int device);
    ~VitsModel();
    // A second execution context on the same weights: own stream and workspace (micro-batch pipelining)
    VitsModel* clone() const;
    int device() const { return device_; }
    const VitsConfig& cfg() const { return cfg_; }
    void forward(const VitsBatch& b);
    // Streaming long-form decode (BASELINE configs[4]) of the utterance of the last forward(skip_decoder = true, n = 1): the HiFi-GAN
    // decoder runs on windows of chunk_frames + 2 * kStreamHalo frames, ONE hipGraph captured for that fixed shape and replayed per
    // chunk; the workspace is bounded by the window.  stream_begin returns the number of frames; stream_chunk decodes frames
    // [f0, f0 + chunk_frames) into dst (host) and returns the number of samples written.
    // halo frames per side = the generator's receptive field (from the config: 13.4 frames -> 16 for JP-Extra, SURVEY.md §5 "Long-context")
    int stream_halo() const;
    int64_t stream_begin(int chunk_frames);
    int64_t stream_chunk(int64_t f0, float* dst_host, int64_t capacity);
    bool stream_graph_captured() const { return chunk_ && chunk_->exec != nullptr; }
    size_t stream_workspace_bytes() const { return (chunk_ ? chunk_->ar.capacity() : 0) + (burst_ ? burst_->ar.capacity() : 0); }
    // results of the last forward
    const std::vector<int64_t>& pcm_lens() const { return pcm_lens_; }
    const std::vector<int64_t>& pcm_offs() const { return pcm_offs_; }
    const float* pcm_device() const { return pcm_; }
    int64_t pcm_total() const { return pcm_total_; }
    void copy_pcm(float* host);  // concatenated
    const std::vector<int>& durations() const { return dur_host_; }   // concatenated predicted w_ceil
    const std::vector<float>& logw() const { return logw_host_; }
    hipStream_t stream() const { return stream_; }
    void set_trace(bool on) { trace_ = on; }
    int decoder_mode() const { return dec_mode_; }
    size_t workspace_bytes() const { return arena_.capacity() + keep_.capacity(); }
    // copies a traced plane of the last forward for utterance `utt`: returns rows/cols
    bool get_trace(const std::string& name, int utt, std::vector<float>& out, int& rows, int& cols);

  private:
    struct Attn {
        PackedConv q, k, v, o;
        PackedConv qkv;   // rows q | k | v: one launch for the three projections (fused attention path)
        float *erk, *erv;
    };
    struct EncLayer {
        Attn attn;
        float *n1g, *n1b, *n2g, *n2b;
        PackedConv ffn1, ffn2;
    };
    struct Encoder {
        float *spk_w = nullptr, *spk_b = nullptr;  // [hidden][gin] row-major (linear_vec)
        std::vector<EncLayer> layers;
    };
    struct DDS {
        std::vector<float*> sep_w, sep_b, n1g, n1b, n2g, n2b;
        std::vector<PackedConv> pw;
    };
    struct ConvFlow {
        float *pre_w, *pre_b;
        DDS dds;
        PackedConv proj;
    };
    struct Coupling {
        PackedConv pre, post;
        Encoder enc;
    };
    struct ResBranch {
        std::vector<PackedConv> c1, c2;
        std::vector<int> dil;
        int k;
    };
    struct Stage {
        PackedUpsample up;
        std::vector<ResBranch> branches;
        int ch, rate;
    };
    struct TraceEntry {
        Plane p;
        const SegLayout* lay;
        int div;  // columns per layout unit (upsampling factor)
    };

    Encoder load_encoder(const std::string& prefix, int n_layers);
    DDS load_dds(const std::string& prefix, int channels);
    void run_encoder(const Encoder& e, Plane x, const SegLayout& lay, const float* spk_vec, Arena& ar);
    void run_dds(const DDS& d, Plane x, const SegLayout& lay, Arena& ar);
    void run_decoder(Arena& ar, Plane z, const SegLayout& fl, const float* cond_vec);
    // channels-last bf16 / split-bf16 MFMA decoder (decoder_cl.cpp)
    struct ClUpGroup {
        ClConv c;
        int ntaps = 0, nph = 0;
        int shift[kMaxTaps];
        int phase_off[kMaxPhases];
    };
    struct ClBranch {
        std::vector<ClConv> c1, c2;
        std::vector<int> dil;
        int k;
    };
    struct ClStage {
        std::vector<ClUpGroup> up;
        ClUpX upx;              // the same transposed convolution as one phased conv_clx launch (large batches of the wide stages; else conv_cl's groups)
        std::vector<ClBranch> branches;
        int cin, ch, rate;
        int mode = 1;   // the stage's arithmetic (dec_mode_ codes 1 .. 3; SBV2_DECODER_STAGES)
    };
    void load_decoder_cl(const Blob& blob);
    void run_decoder_cl(Arena& ar, Plane z, const SegLayout& fl, const float* cond_vec);
    void conv_cl(const ClConv& c, const float* X, int ldx, int NB, float* Y, int ldy, int N, int dil, int pad_l, const unsigned char* mask,
                 int mask_div, float pre_slope, const float* R, int ldr, float beta, int accumulate);
    void trace(const std::string& name, Plane p, const SegLayout& lay, int div = 1);
    // fixed-shape decoder of the streaming path: own arena (never reset while the plan lives), persistent input / conditioning buffers
    // and the captured graph
    struct ChunkPlan {
        int chunk = 0, W = 0, nwin = 1;   // nwin windows of W frames per replay
        Arena ar;
        SegLayout lay;
        Plane zin;
        float* cond = nullptr;
        Arena::Mark mark{0, 0};
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        float* pcm = nullptr;       // window PCM (W * hop samples), valid after a replay
        // chunk c + 1 is enqueued before the host waits for chunk c: two pinned host slots + events
        float* host[2] = {nullptr, nullptr};
        hipEvent_t ev[2] = {nullptr, nullptr};
        int64_t slot_f0[2] = {-1, -1}, slot_n[2] = {0, 0};
        ~ChunkPlan() {
            if (exec) (void)hipGraphExecDestroy(exec);
            if (graph) (void)hipGraphDestroy(graph);
            for (int i = 0; i < 2; ++i) {
                if (host[i]) (void)hipHostFree(host[i]);
                if (ev[i]) (void)hipEventDestroy(ev[i]);
            }
        }
    };
    void ensure_plan(std::shared_ptr<ChunkPlan>& slot, int chunk_frames, int nwin);
    void stream_enqueue(ChunkPlan& c, int64_t f0, int slot);
    bool stream_bursts_ = false;                 // the running stream uses burst_ behind its first chunk
    std::shared_ptr<ChunkPlan> chunk_, burst_;   // one window (an utterance's first chunk) / kStreamBurst windows per replay (every later one)
    Plane z_{};              // flow output of the last forward (frame-rate plane, packed layout fl_)

    int device_;
    VitsConfig cfg_;
    std::shared_ptr<WeightStore> ws_;
    hipStream_t stream_ = nullptr;
    SatWatch sat_watch_;                // f16x3 clamp warning of this handle (common.h)
    hipEvent_t after_ev_ = nullptr;   // orders this context after the producer stream of its DeBERTa features (created on first use)
    Arena arena_, keep_;
    // weights
    float *emb_g_, *emb_, *tone_emb_, *lang_emb_, *style_w_, *style_b_;
    PackedConv bert_proj_, enc_proj_;
    Encoder enc_p_;
    PackedConv dp_c1_, dp_c2_, dp_proj_;
    float *dp_n1g_, *dp_n1b_, *dp_n2g_, *dp_n2b_, *dp_cond_w_, *dp_cond_b_;
    PackedConv sdp_pre_, sdp_proj_;
    float *sdp_cond_w_, *sdp_cond_b_, *sdp_ea_m_, *sdp_ea_logs_ = nullptr, *sdp_ea_scale_ = nullptr;
    DDS sdp_dds_;
    std::vector<ConvFlow> sdp_cf_;  // ConvFlow 2..n (index 0 = flows.3)
    std::vector<Coupling> flows_;
    PackedConv dec_pre_;
    float *dec_cond_w_, *dec_cond_b_, *dec_post_w_;
    float* dec_cond_vec_ = nullptr;  // cond(g) per utterance of the running forward
    int dec_post_k_ = 7;
    std::vector<Stage> stages_;
    bool fuse_pairs_ = true;  // the fused ResBlock step of the <= 64-channel stages (always on in the product; false only in kernel tests)
    int dec_mode_ = 0;  // 0 = exact f32 MFMA (k-major), 1 = split-bf16 (f32-grade), 2 = plain bf16, 3 = fp16 operands
    ClConv cl_pre_;
    std::vector<ClStage> cl_stages_;
    // last-forward results
    float* pcm_ = nullptr;
    int64_t pcm_total_ = 0;
    std::vector<int64_t> pcm_lens_, pcm_offs_;
    std::vector<int> dur_host_;
    std::vector<float> logw_host_;
    bool trace_ = false;
    std::map<std::string, TraceEntry> traces_;
    SegLayout tl_, fl_;
};

}  // namespace sbv2
