// Style-Bert-VITS2 JP-Extra synthesizer (the graph behind `model::synthesize`, crates/sbv2_core/src/model.rs:53-111,
// exported from SynthesizerTrn.infer by scripts/convert/convert_model.py:89-113).
//
// A batch of utterances is PACKED along the time axis (SegLayout): text-rate planes [C][sum T_text + gaps] and frame-rate
// planes [C][sum T_frames + gaps]; the zero gaps play the role of every convolution's zero padding, every kernel re-zeroes
// them through its column mask, and attention is grouped per (utterance, head).  So each utterance's result is the
// reference's batch-1 result (the reference cannot batch: model.rs:66-79) while every launch covers the whole batch.
// Stage by stage this follows oracle/sbv2_oracle.py::vits_forward.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>

#include <atomic>
#include "models.h"

namespace sbv2 {

// keys / values of the flow's attention as pre-split bf16 planes (sbv2_debug_set_flash_parts, for the test that holds the attention kernels to each other:
// 1 = the default policy below, 2: at every length, 3: at every length on the un-pipelined kernel k_vits_flash_x3p, 4: at every length on k_vits_flash_x3q's
// 8-wave shape, 0: never = converted per key tile inside the attention kernel; bit-identical)
static std::atomic<int> g_flash_parts{1};
bool flash_parts_enabled() { return g_flash_parts.load(std::memory_order_relaxed) != 0; }
static int flash_parts_mode() { return g_flash_parts.load(std::memory_order_relaxed); }
int set_flash_parts(int on) { return g_flash_parts.exchange(on); }

static const int kTextGap = 16;   // >= 9: DDSConv depthwise dilation 3^2 with k=3
static const int kFrameGap = 4;   // >= 25/8 frames: widest decoder tap offset (k=11, d=5) at the first upsampled rate

namespace {
struct AttnPlan {
    int ng = 0, maxT = 0, lds = 0;
    AttnGroup* d_ag = nullptr;
    GemmGroup* d_st = nullptr;
    GemmGroup* d_pv = nullptr;
    float* S = nullptr;
    float* PW = nullptr;
    double flops = 0;  // algorithmic FLOP of one grouped product over all (utterance, head) problems
};

AttnPlan make_attn_plan(const SegLayout& lay, int H, int heads, int ld, int window, Arena& ar, hipStream_t stream, bool scores) {
    AttnPlan pl;
    const int dk = H / heads;
    pl.ng = lay.n * heads;
    pl.maxT = lay.max_len();
    pl.lds = round_up(pl.maxT, 4);
    std::vector<AttnGroup> ag(pl.ng);
    std::vector<GemmGroup> st(pl.ng), pv(pl.ng);
    int64_t s_off = 0, w_off = 0;
    for (int u = 0; u < lay.n; ++u)
        for (int h = 0; h < heads; ++h) {
            const int gi = u * heads + h;
            const int T = lay.len[u];
            AttnGroup& a = ag[gi];
            a.qk_off = (int64_t)h * dk * ld + lay.start[u];
            a.s_off = s_off;
            a.aux_off = w_off;
            a.aux2_off = 0;
            a.T = T;
            a.lds = pl.lds;
            a.col0 = lay.start[u];
            a.head = h;
            st[gi] = GemmGroup{a.qk_off, a.qk_off, s_off, 0, T, T, dk, T};                        // S^T = K^T Q
            pv[gi] = GemmGroup{(int64_t)lay.start[u] * H + h * dk, s_off, a.qk_off, 0, dk, T, T, T};  // ctx = V P^T
            pl.flops += 2.0 * (double)T * T * dk;
            s_off += (int64_t)T * pl.lds;
            w_off += (int64_t)(2 * window + 1) * pl.lds;
        }
    if (scores) {   // the unfused path materialises the T x T score block and the band of probabilities
        pl.S = ar.array<float>((size_t)s_off);
        pl.PW = ar.array<float>((size_t)w_off);
    }
    pl.d_ag = ar.array<AttnGroup>(pl.ng);
    pl.d_st = ar.array<GemmGroup>(pl.ng);
    pl.d_pv = ar.array<GemmGroup>(pl.ng);
    {
        UploadBatch ub(ar);
        ar.upload(pl.d_ag, ag.data(), sizeof(AttnGroup) * pl.ng, stream);
        ar.upload(pl.d_st, st.data(), sizeof(GemmGroup) * pl.ng, stream);
        ar.upload(pl.d_pv, pv.data(), sizeof(GemmGroup) * pl.ng, stream);
    }
    return pl;
}

void grouped_gemm(const float* A, int lda, const float* B, int ldb, float* C, int ldc, const GemmGroup* grp, int ng, int maxM, int maxN,
                  float alpha, double flops, hipStream_t s) {
    ConvParams p;
    p.A = A;
    p.lda = lda;
    p.B = B;
    p.ldb = ldb;
    p.C = C;
    p.ldc = ldc;
    p.alpha = alpha;
    p.groups = grp;
    p.ngroups = ng;
    p.maxM = maxM;
    p.maxN = maxN;
    p.flops_hint = flops;
    launch_conv(p, s);
}
}  // namespace

VitsModel::Encoder VitsModel::load_encoder(const std::string& p, int n_layers) {
    Encoder e;
    const int Hh = cfg_.hidden, dkk = cfg_.hidden / cfg_.heads, nwin = 2 * cfg_.window + 1;
    e.spk_w = ws_->tensor(p + "spk_emb_linear.weight", {Hh, cfg_.gin});
    e.spk_b = ws_->tensor(p + "spk_emb_linear.bias", {Hh});
    e.layers.resize(n_layers);
    for (int i = 0; i < n_layers; ++i) {
        EncLayer& L = e.layers[i];
        const std::string a = p + "attn_layers." + std::to_string(i) + ".";
        L.attn.q = ws_->conv(a + "conv_q");
        L.attn.k = ws_->conv(a + "conv_k");
        L.attn.v = ws_->conv(a + "conv_v");
        L.attn.o = ws_->conv(a + "conv_o");
        L.attn.qkv = ws_->conv_cat({a + "conv_q", a + "conv_k", a + "conv_v"});
        L.attn.erk = ws_->tensor(a + "emb_rel_k", {nwin, dkk});     // [1][2w+1][dk]: shared by the heads
        L.attn.erv = ws_->tensor(a + "emb_rel_v", {nwin, dkk});
        L.n1g = ws_->tensor(p + "norm_layers_1." + std::to_string(i) + ".gamma", {Hh});
        L.n1b = ws_->tensor(p + "norm_layers_1." + std::to_string(i) + ".beta", {Hh});
        L.n2g = ws_->tensor(p + "norm_layers_2." + std::to_string(i) + ".gamma", {Hh});
        L.n2b = ws_->tensor(p + "norm_layers_2." + std::to_string(i) + ".beta", {Hh});
        L.ffn1 = ws_->conv(p + "ffn_layers." + std::to_string(i) + ".conv_1");
        L.ffn2 = ws_->conv(p + "ffn_layers." + std::to_string(i) + ".conv_2");
        for (const PackedConv* c : {&L.attn.q, &L.attn.k, &L.attn.v, &L.attn.o}) ws_->expect(*c, a + "conv_*", Hh, Hh, 1);
        ws_->expect(L.ffn1, p + "ffn_layers.conv_1", cfg_.filter, Hh, L.ffn1.k);
        ws_->expect(L.ffn2, p + "ffn_layers.conv_2", Hh, cfg_.filter, L.ffn1.k);
    }
    return e;
}

VitsModel::DDS VitsModel::load_dds(const std::string& p, int) {
    DDS d;
    for (int i = 0; i < cfg_.sdp_dds_layers; ++i) {
        const std::string s = std::to_string(i);
        const HostTensor& w = ws_->blob().get(p + "convs_sep." + s + ".weight");
        SBV2_REQUIRE(w.dims.size() == 3 && w.dims[2] == 3, "DDSConv depthwise kernel must be 3");
        const int Hd = cfg_.hidden;
        d.sep_w.push_back(ws_->tensor(p + "convs_sep." + s + ".weight", {Hd, 3}));
        d.sep_b.push_back(ws_->tensor(p + "convs_sep." + s + ".bias", {Hd}));
        d.pw.push_back(ws_->conv(p + "convs_1x1." + s));
        ws_->expect(d.pw.back(), p + "convs_1x1." + s, Hd, Hd, 1);
        d.n1g.push_back(ws_->tensor(p + "norms_1." + s + ".gamma", {Hd}));
        d.n1b.push_back(ws_->tensor(p + "norms_1." + s + ".beta", {Hd}));
        d.n2g.push_back(ws_->tensor(p + "norms_2." + s + ".gamma", {Hd}));
        d.n2b.push_back(ws_->tensor(p + "norms_2." + s + ".beta", {Hd}));
    }
    return d;
}

VitsModel::VitsModel(const Blob& blob, int device) : device_(device) {
    SBV2_REQUIRE(blob.kind == 2, "weight container is not a VITS (kind 2) model");
    HIP_CHECK(hipSetDevice(device));
    HIP_CHECK(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));  // never serialised against the NULL stream (e.g. RCCL launched by the caller)
    f16x3_sat_prepare();
    sat_watch_.baseline();
    const std::string& js = blob.config_json;
    auto I = [&](const char* k) { return (int)json_number(js, k); };
    cfg_.n_vocab = I("n_vocab"); cfg_.n_tones = I("n_tones"); cfg_.n_langs = I("n_langs"); cfg_.n_speakers = I("n_speakers");
    cfg_.hidden = I("hidden"); cfg_.inter = I("inter"); cfg_.filter = I("filter"); cfg_.heads = I("heads");
    cfg_.enc_layers = I("enc_layers"); cfg_.enc_kernel = I("enc_kernel"); cfg_.window = I("window"); cfg_.gin = I("gin");
    cfg_.style_dim = I("style_dim"); cfg_.bert_dim = I("bert_dim"); cfg_.cond_layer_idx = I("cond_layer_idx");
    cfg_.flow_n = I("flow_n"); cfg_.flow_layers = I("flow_layers"); cfg_.flow_kernel = I("flow_kernel");
    cfg_.dp_filter = I("dp_filter"); cfg_.dp_kernel = I("dp_kernel"); cfg_.sdp_kernel = I("sdp_kernel");
    cfg_.sdp_flows = I("sdp_flows"); cfg_.sdp_bins = I("sdp_bins"); cfg_.sdp_dds_layers = I("sdp_dds_layers");
    cfg_.sdp_tail = (float)json_number(js, "sdp_tail");
    cfg_.up_rates = json_int_array(js, "up_rates");
    cfg_.up_kernels = json_int_array(js, "up_kernels");
    cfg_.res_kernels = json_int_array(js, "res_kernels");
    cfg_.res_dilations = json_int_array2(js, "res_dilations");
    cfg_.up_initial = I("up_initial");
    SBV2_REQUIRE(cfg_.sdp_kernel == 3, "DDSConv kernel must be 3");
    SBV2_REQUIRE(cfg_.hidden % cfg_.heads == 0 && (cfg_.hidden / cfg_.heads) % 4 == 0, "head size must be a multiple of 4");
    SBV2_REQUIRE(cfg_.inter % 8 == 0, "flow channels must be a multiple of 8");
    SBV2_REQUIRE(cfg_.res_dilations.size() == cfg_.res_kernels.size(), "resblock config mismatch");

    // Arithmetic of the k >= 3 convolutions outside the decoder, i.e. the flow's k = 5 FFN convs (SBV2_GEMM): split-bf16 through the
    // k-major variant of conv_cl.hip by default (203 TFLOP/s algorithmic against ~60 for the f32 MFMA kernel on those shapes), or
    // f32 | bf16 | f16.  1x1 products always stay on the f32 MFMA kernel: the k-major variant transposes while staging and a 1x1 has too
    // few MFMAs per chunk to pay for it (~48 TFLOP/s measured).
    int gemm_parts = 2;
    if (const char* m = getenv("SBV2_GEMM")) {
        const std::string v(m);
        if (v == "f32") gemm_parts = 0;
        else if (v == "bf16") gemm_parts = 1;
        else if (v == "f16") gemm_parts = 3;
        else SBV2_REQUIRE(v == "bf16x3" || v.empty(), "SBV2_GEMM must be f32, bf16x3, bf16 or f16");
    }
    ws_.reset(new WeightStore(blob, gemm_parts));
    WeightStore& w = *ws_;
    // The text side decides the INTEGER durations (ceil(exp(logw) * length_scale)): text encoder and both duration predictors run on
    // the exact-f32 kernels whatever SBV2_GEMM says.  (Round 2 measured their k = 3 convolutions on the split-bf16 matrix cores behind a knob: one flipped
    // duration per 102 800 symbols, profiles/r02_flip_rate.json; the knob and its script are gone, the text side is exact f32.)
    w.set_cl_parts(0);
    // every tensor's shape is checked against the config before a kernel indexes it (a container / imported ONNX whose config and
    // weights disagree is refused here)
    const int Hc = cfg_.hidden, Gc = cfg_.gin, Ic = cfg_.inter, Fd = cfg_.dp_filter;
    SBV2_REQUIRE(cfg_.n_vocab >= 1 && cfg_.n_tones >= 1 && cfg_.n_langs >= 1 && cfg_.n_speakers >= 1 && Hc >= 4 && Gc >= 1 && Ic >= 2 && Fd >= 1 &&
                     cfg_.enc_layers >= 1 && cfg_.flow_n >= 1 && cfg_.flow_layers >= 1 && cfg_.sdp_bins >= 2 && cfg_.sdp_bins <= 16 && cfg_.window >= 0,
                 "bad VITS config");
    emb_g_ = w.tensor("emb_g.weight", {cfg_.n_speakers, Gc});
    emb_ = w.tensor("enc_p.emb.weight", {cfg_.n_vocab, Hc});
    tone_emb_ = w.tensor("enc_p.tone_emb.weight", {cfg_.n_tones, Hc});
    lang_emb_ = w.tensor("enc_p.language_emb.weight", {cfg_.n_langs, Hc});
    bert_proj_ = w.conv("enc_p.bert_proj");
    w.expect(bert_proj_, "enc_p.bert_proj", Hc, cfg_.bert_dim, 1);
    style_w_ = w.tensor("enc_p.style_proj.weight", {Hc, cfg_.style_dim});
    style_b_ = w.tensor("enc_p.style_proj.bias", {Hc});
    enc_p_ = load_encoder("enc_p.encoder.", cfg_.enc_layers);
    enc_proj_ = w.conv("enc_p.proj");
    w.expect(enc_proj_, "enc_p.proj", 2 * Ic, Hc, 1);
    dp_c1_ = w.conv("dp.conv_1");
    dp_c2_ = w.conv("dp.conv_2");
    dp_proj_ = w.conv("dp.proj");
    w.expect(dp_c1_, "dp.conv_1", Fd, Hc, cfg_.dp_kernel);
    w.expect(dp_c2_, "dp.conv_2", Fd, Fd, cfg_.dp_kernel);
    w.expect(dp_proj_, "dp.proj", 1, Fd, 1);
    dp_n1g_ = w.tensor("dp.norm_1.gamma", {Fd}); dp_n1b_ = w.tensor("dp.norm_1.beta", {Fd});
    dp_n2g_ = w.tensor("dp.norm_2.gamma", {Fd}); dp_n2b_ = w.tensor("dp.norm_2.beta", {Fd});
    dp_cond_w_ = w.tensor("dp.cond.weight", {Hc, Gc}); dp_cond_b_ = w.tensor("dp.cond.bias", {Hc});
    sdp_pre_ = w.conv("sdp.pre");
    sdp_proj_ = w.conv("sdp.proj");
    w.expect(sdp_pre_, "sdp.pre", Hc, Hc, 1);
    w.expect(sdp_proj_, "sdp.proj", Hc, Hc, 1);
    sdp_cond_w_ = w.tensor("sdp.cond.weight", {Hc, Gc}); sdp_cond_b_ = w.tensor("sdp.cond.bias", {Hc});
    sdp_ea_m_ = w.tensor("sdp.flows.0.m", {2});
    // (an onnxsim-processed export has exp(-logs) folded into a constant: import.cpp rule 7 stores it as flows.0.exp_neg_logs)
    if (blob.has("sdp.flows.0.logs") || !blob.has("sdp.flows.0.exp_neg_logs")) sdp_ea_logs_ = w.tensor("sdp.flows.0.logs", {2});
    else sdp_ea_scale_ = w.tensor("sdp.flows.0.exp_neg_logs", {2});
    sdp_dds_ = load_dds("sdp.convs.", cfg_.hidden);
    for (int i = 2; i <= cfg_.sdp_flows; ++i) {  // ConvFlow 1 is the "useless vflow" dropped in reverse mode
        const std::string p = "sdp.flows." + std::to_string(2 * i - 1) + ".";
        ConvFlow cf;
        cf.pre_w = w.tensor(p + "pre.weight", {Hc});
        cf.pre_b = w.tensor(p + "pre.bias", {Hc});
        cf.dds = load_dds(p + "convs.", cfg_.hidden);
        cf.proj = w.conv(p + "proj");
        SBV2_REQUIRE(cf.proj.cout == 3 * cfg_.sdp_bins - 1, "ConvFlow projection size");
        sdp_cf_.push_back(cf);
    }
    w.set_cl_parts(gemm_parts);   // the flow only shapes the (continuous) latent: split-bf16 for its k = 5 FFN convs
    // ... and for the 1x1 products of its attention layers (q | k | v and the output projection: 48 launches of a batch-32 step) through
    // gemm_bfs.hip with pre-split operands.  SBV2_FLOW_1X1=f32 keeps them on
    // the exact-f32 kernel; SBV2_GEMM=f32 does as well.
    // Operand format: f16x3 (f16 hi + scaled f16 lo: 22 mantissa bits, common.h) costs the same three MFMAs per product as bf16x3 and is
    // 20x closer to the f32 product on these shapes (tools/bfs_probe.py), so it is the default; SBV2_FLOW_1X1=bf16x3 keeps the two-bf16 split.
    int flow_bfs = gemm_parts == 2 && (Hc & 15) == 0 ? kPartsF16x3 : 0;
    if (const char* m = getenv("SBV2_FLOW_1X1")) {
        const std::string v(m);
        SBV2_REQUIRE(v == "f16x3" || v == "bf16x3" || v == "f32" || v.empty(), "SBV2_FLOW_1X1 must be f32, bf16x3 or f16x3");
        if (v == "f32") flow_bfs = 0;
        else if (v == "bf16x3" && flow_bfs) flow_bfs = 2;
    }
    w.set_bfs_parts(flow_bfs);
    for (int i = 0; i < cfg_.flow_n; ++i) {
        const std::string p = "flow.flows." + std::to_string(2 * i) + ".";
        Coupling c;
        c.pre = w.conv(p + "pre");
        c.post = w.conv(p + "post");
        w.expect(c.pre, p + "pre", Hc, Ic / 2, 1);
        w.expect(c.post, p + "post", Ic / 2, Hc, 1);
        c.enc = load_encoder(p + "enc.", cfg_.flow_layers);
        flows_.push_back(c);
    }
    // the k-major decoder weights stay exact f32 (SBV2_DECODER=f32 is the exact reference path); the bf16 decoder has its own packing
    w.set_cl_parts(0);
    w.set_bfs_parts(0);
    dec_pre_ = w.conv("dec.conv_pre");
    w.expect(dec_pre_, "dec.conv_pre", cfg_.up_initial, Ic, dec_pre_.k);
    dec_cond_w_ = w.tensor("dec.cond.weight", {cfg_.up_initial, Gc});
    dec_cond_b_ = w.tensor("dec.cond.bias", {cfg_.up_initial});
    {
        const HostTensor& pw = blob.get("dec.conv_post.weight");
        SBV2_REQUIRE(pw.dims.size() == 3 && pw.dims[0] == 1 && pw.dims[2] >= 1 && pw.dims[2] <= 15 && (pw.dims[2] & 1), "dec.conv_post.weight must be [1][C][odd k]");
        int cl = cfg_.up_initial;
        for (size_t i = 0; i < cfg_.up_rates.size(); ++i) cl /= 2;
        dec_post_w_ = w.tensor("dec.conv_post.weight", {cl, pw.dims[2]});
        dec_post_k_ = (int)pw.dims[2];
    }
    SBV2_REQUIRE(cfg_.up_rates.size() == cfg_.up_kernels.size() && !cfg_.up_rates.empty(), "upsampling config mismatch");
    int C = cfg_.up_initial;
    const int nk = (int)cfg_.res_kernels.size();
    for (size_t i = 0; i < cfg_.up_rates.size(); ++i) {
        Stage st;
        st.rate = cfg_.up_rates[i];
        st.up = w.upsample("dec.ups." + std::to_string(i), st.rate, (cfg_.up_kernels[i] - st.rate) / 2);
        SBV2_REQUIRE(st.up.cin == C && st.up.cout == C / 2, "dec.ups." + std::to_string(i) + ": channel counts do not follow up_initial / 2^i");
        C /= 2;
        st.ch = C;
        for (int j = 0; j < nk; ++j) {
            ResBranch rb;
            rb.k = cfg_.res_kernels[j];
            rb.dil = cfg_.res_dilations[j];
            const std::string p = "dec.resblocks." + std::to_string(i * nk + j) + ".";
            for (size_t n = 0; n < rb.dil.size(); ++n) {
                rb.c1.push_back(w.conv(p + "convs1." + std::to_string(n)));
                rb.c2.push_back(w.conv(p + "convs2." + std::to_string(n)));
                w.expect(rb.c1.back(), p + "convs1", C, C, rb.k);
                w.expect(rb.c2.back(), p + "convs2", C, C, rb.k);
            }
            st.branches.push_back(rb);
        }
        stages_.push_back(st);
    }
    // decoder arithmetic: exact f32 MFMA (default) | split-bf16 MFMA, f32-grade | plain bf16 MFMA
    //   bf16x3 (default when every decoder channel count is a multiple of 16): split-bf16 MFMA, waveform within ~2e-6 of the f32 path
    //   f32: exact f32 MFMA on k-major planes;  bf16: plain bf16 operands (~9e-4 waveform error at a 0.1 peak);
    //   f16: fp16 operands, same speed as bf16, 1.3e-4 (relative to the signal that is ~1e-3: opt-in, not the default)
    bool cl_ok = cfg_.inter % 16 == 0;
    for (const Stage& st : stages_) cl_ok = cl_ok && st.ch % 16 == 0;
    dec_mode_ = cl_ok ? 1 : 0;
    if (const char* m = getenv("SBV2_DECODER")) {
        const std::string v(m);
        if (v == "bf16x3") dec_mode_ = 1;
        else if (v == "bf16") dec_mode_ = 2;
        else if (v == "f16") dec_mode_ = 3;
        else if (v == "f32") dec_mode_ = 0;
        else SBV2_REQUIRE(v.empty(), "SBV2_DECODER must be f32, bf16x3, bf16 or f16");
        SBV2_REQUIRE(dec_mode_ == 0 || cl_ok, "SBV2_DECODER: the bf16 MFMA decoder needs channel counts that are multiples of 16");
    }
    if (dec_mode_) load_decoder_cl(blob);
}

VitsModel* VitsModel::clone() const {
    HIP_CHECK(hipSetDevice(device_));
    VitsModel* c = new VitsModel(*this);   // shares ws_ (device weights); Arena copies are empty
    c->stream_ = nullptr;
    HIP_CHECK(hipStreamCreateWithFlags(&c->stream_, hipStreamNonBlocking));
    c->after_ev_ = nullptr;   // created on first use PER CONTEXT: a copied handle would be destroyed twice (and waited on after its first destruction)
    c->pcm_ = nullptr;
    c->pcm_total_ = 0;
    c->pcm_lens_.clear();
    c->pcm_offs_.clear();
    c->traces_.clear();
    c->chunk_.reset();
    c->burst_.reset();
    c->z_ = Plane{};
    c->trace_ = false;
    return c;
}

VitsModel::~VitsModel() {
    (void)hipSetDevice(device_);
    if (after_ev_) (void)hipEventDestroy(after_ev_);
    if (stream_) (void)hipStreamDestroy(stream_);
}

void VitsModel::trace(const std::string& name, Plane p, const SegLayout& lay, int div) {
    if (!trace_) return;
    Plane c = keep_.plane(p.C, p.L);
    HIP_CHECK(hipMemcpy2DAsync(c.p, sizeof(float) * c.ld, p.p, sizeof(float) * p.ld, sizeof(float) * p.L, p.C, hipMemcpyDeviceToDevice, stream_));
    traces_[name] = TraceEntry{c, &lay, div};
}

bool VitsModel::get_trace(const std::string& name, int utt, std::vector<float>& out, int& rows, int& cols) {
    auto it = traces_.find(name);
    if (it == traces_.end()) return false;
    const TraceEntry& t = it->second;
    if (utt < 0 || utt >= t.lay->n) return false;
    HIP_CHECK(hipSetDevice(device_));
    rows = t.p.C;
    cols = t.lay->len[utt] * t.div;
    out.resize((size_t)rows * cols);
    HIP_CHECK(hipStreamSynchronize(stream_));
    HIP_CHECK(hipMemcpy2D(out.data(), sizeof(float) * cols, t.p.p + (size_t)t.lay->start[utt] * t.div, sizeof(float) * t.p.ld,
                          sizeof(float) * cols, rows, hipMemcpyDeviceToHost));
    return true;
}

// attentions.Encoder (post-LN, window-relative attention, FFN with 'same' convolutions, speaker vector before layer 2)
void VitsModel::run_encoder(const Encoder& e, Plane x, const SegLayout& lay, const float* spk_vec, Arena& ar) {
    const int H = x.C, heads = cfg_.heads, dk = H / heads, N = lay.L;
    const Arena::Mark mk = ar.mark();
    // SBV2_ATTN=unfused keeps the four-launch attention (grouped GEMM, softmax, grouped GEMM, relative-value add) for A/B runs
    static const bool fused = !(getenv("SBV2_ATTN") && std::string(getenv("SBV2_ATTN")) == "unfused");
    // SBV2_ATTN=f32 keeps the fused kernel on the exact-f32 MFMA everywhere; by default an encoder whose convs run on the split-bf16
    // matrix cores (the flow: nothing there feeds the integer durations) takes the split-bf16 attention as well
    static const bool attn_f32 = getenv("SBV2_ATTN") && std::string(getenv("SBV2_ATTN")) == "f32";
    const bool split_attn = !attn_f32 && e.layers[0].ffn1.cl.parts == 2 && (H / heads) % 16 == 0;
    const AttnPlan pl = make_attn_plan(lay, H, heads, x.ld, cfg_.window, ar, stream_, !fused);
    Plane QKV = ar.plane(3 * H, N);
    Plane Q = QKV.rows(0, H), K = QKV.rows(H, H), ctx = ar.plane(H, N), Y = ar.plane(H, N);
    SBV2_REQUIRE(Q.ld == x.ld, "plane pitch mismatch");
    Plane F = ar.plane(e.layers[0].ffn1.cout, N);
    constexpr bool cl_ffn = true;
    const PackedConv& f1 = e.layers[0].ffn1;
    const PackedConv& f2 = e.layers[0].ffn2;
    float* Fcl = (cl_ffn && f1.cl.parts && f2.cl.parts && f1.k >= 3 && f2.k >= 3 && (f1.cout & 15) == 0) ? ar.array<float>((size_t)N * f1.cout) : nullptr;
    float* VT = fused ? nullptr : ar.array<float>((size_t)N * H);
    Plane Vp = QKV.rows(2 * H, H);
    fill_zero(ctx.p, sizeof(float) * (size_t)H * ctx.ld, stream_);
    const float qscale = 1.0f / std::sqrt((float)dk);
    // 1x1 products on pre-split operands (gemm_bfs.hip; the flow by default): the parts of x are written by the LayerNorm that produces x
    // (by split_planes for the encoder's input and after the speaker vector is added), the parts of the attention output by split_planes
    const int SP = !fused ? 0 : (e.layers[0].attn.qkv.bfs.f16 ? kPartsF16x3 : e.layers[0].attn.qkv.bfs.parts);   // parts code
    SplitPlanes Xs, Cs, QKVs;
    if (SP) {
        Xs = alloc_split(ar, SP, H, N);
        Cs = alloc_split(ar, SP, H, N);
        split_planes(x, Xs, stream_);
    }
    // the split-bf16 attention reads its keys / values as bf16 hi / lo planes written by the q | k | v product's epilogue next to the f32
    // plane (attn_flash.hip, k_vits_flash_x3p: no conversion per key tile; the same bits as converting while staging)
    // Which kernel (same bits, so the choice is free).  The software-pipelined kernel on pre-split tiles (k_vits_flash_x3q, attn_flash.hip) at every size when
    // the head dimension fits its DMA blocks: 4-wave workgroups while they leave at most one per CU (a single utterance: 58 us per launch at 897 frames
    // against 84 for k_vits_flash_x3p and ~80 for the converting kernel), 8-wave workgroups beyond (32 x 897 frames: 84 us against 113; the q | k | v product
    // then writes q as f32 and k / v as parts only: the same bytes).  Otherwise the un-pipelined pre-split kernel from 4096 frames (at 897
    // frames x 32 it is slower than converting, 138 against 124 us; at 14 001 it wins) and for launches of <= 64 workgroups; the converting kernel for the rest.
    // set_flash_parts: 2 = parts at every length, 3 = ... on the un-pipelined kernel, 0 = never (the tests).
    constexpr int parts_min_t = 4096, parts_max_wgs = 64;
    const int64_t attn_wgs = (int64_t)((pl.maxT + 127) / 128) * pl.ng;
    const bool kv_parts = SP && split_attn && flash_parts_enabled() &&
                          ((flash_parts_mode() != 3 && flash_pipelined_usable(dk)) || pl.maxT >= parts_min_t || attn_wgs <= parts_max_wgs || flash_parts_mode() >= 2);
    if (kv_parts) QKVs = alloc_split(ar, 2, 3 * H, N);
    // Large batches: the FFN pair on conv_clx.hip (pre-split chunk-major operands by LDS-DMA instead of conv_cl's transposing register staging):
    // x is split once per layer from its k-major plane (split_cl_km), conv_1's epilogue writes relu(.) as conv_2's operand parts, conv_2 writes
    // the k-major result + residual.  Small launches (a single utterance) stay on conv_cl: since round 5 conv_clx multiplies on 16 x 16 x 32 MFMAs in another
    // summation order, so the two agree to f32 rounding (1e-5 kernel against kernel), not bit for bit.
    const bool clx_ffn = Fcl && f1.cl.parts == 2 && f2.cl.parts == 2 && f1.k == 5 && f2.k == 5 && (H & 63) == 0 && (f1.cout & 63) == 0 &&
                         clx_wanted((int64_t)(N / 256) * (H / 64), 192);
    SplitClPlanes XsC, FsC;
    if (clx_ffn) {
        XsC = make_split_cl(ar.alloc(split_cl_bytes(H, N)), H, N, stream_);
        FsC = make_split_cl(ar.alloc(split_cl_bytes(f1.cout, N)), f1.cout, N, stream_);
    }
    for (size_t i = 0; i < e.layers.size(); ++i) {
        const EncLayer& L = e.layers[i];
        if ((int)i == cfg_.cond_layer_idx && spk_vec) {
            add_segvec(x, spk_vec, H, lay.d_seg_of, 1, lay.d_mask, stream_);
            if (SP) split_planes(x, Xs, stream_);
        }
        if (fused) {
            // (keys / values as parts: q goes to the f32 plane, k and v to the parts only: the bytes of the one-format product)
            if (SP && kv_parts) conv_bfs(L.attn.qkv, Xs, &QKV, &QKVs, nullptr, 1, stream_, ACT_NONE, nullptr, 1.0f, 1.0f, H, H);
            else if (SP) conv_bfs(L.attn.qkv, Xs, &QKV, nullptr, nullptr, 1, stream_);
            else conv_plain(L.attn.qkv, x, QKV, 1, 0, nullptr, 1, stream_);
            if (kv_parts)
                vits_flash_attention_parts(pl.d_ag, pl.ng, pl.maxT, Q.p, Q.ld, QKVs, H, 2 * H, ctx.p, ctx.ld, dk, L.attn.erk, L.attn.erv, cfg_.window,
                                           qscale, stream_, flash_parts_mode() == 3 ? 0 : (flash_parts_mode() == 4 ? 2 : 1));
            else
                vits_flash_attention(pl.d_ag, pl.ng, pl.maxT, Q.p, K.p, Vp.p, Q.ld, ctx.p, ctx.ld, dk, L.attn.erk, L.attn.erv, cfg_.window, qscale,
                                     split_attn, stream_);
        } else {
            conv_plain(L.attn.q, x, Q, 1, 0, nullptr, 1, stream_);
            conv_plain(L.attn.k, x, K, 1, 0, nullptr, 1, stream_);
            linear_tokmajor(L.attn.v, x, VT, H, stream_);
            grouped_gemm(K.p, K.ld, Q.p, Q.ld, pl.S, pl.lds, pl.d_st, pl.ng, pl.maxT, pl.maxT, qscale, pl.flops, stream_);
            vits_softmax(pl.d_ag, pl.ng, pl.maxT, pl.S, Q.p, Q.ld, dk, L.attn.erk, cfg_.window, qscale, pl.PW, stream_);
            grouped_gemm(VT, H, pl.S, pl.lds, ctx.p, ctx.ld, pl.d_pv, pl.ng, dk, pl.maxT, 1.0f, pl.flops, stream_);
            vits_relv_add(pl.d_ag, pl.ng, pl.maxT, ctx.p, ctx.ld, dk, L.attn.erv, cfg_.window, pl.PW, stream_);
        }
        if (SP) {
            split_planes(ctx, Cs, stream_);
            conv_bfs(L.attn.o, Cs, &Y, nullptr, nullptr, 1, stream_, ACT_NONE, &x);
        } else {
            conv_plain(L.attn.o, ctx, Y, 1, 0, nullptr, 1, stream_, ACT_NONE, 1.0f, &x);
        }
        layernorm_ch(Y, x, L.n1g, L.n1b, 1e-5f, ACT_NONE, nullptr, 0, lay.d_mask, stream_);
        const int k = L.ffn1.k;
        // FFN: conv_1 -> ReLU -> conv_2 (+ x).  On the matrix-core path the 768-channel intermediate stays channels-last and the ReLU is
        // conv_2's pre-activation (leaky slope 0); otherwise (exact-f32 text side) both convs run on k-major planes.
        const int Fc = L.ffn1.cout;
        if (clx_ffn) {
            split_cl_km(x, 1.0f, XsC, stream_);
            ConvClxParams p1;
            p1.X = XsC;
            p1.W = L.ffn1.cl.wx;
            p1.nmt = L.ffn1.cl.nmt;
            p1.M = Fc;
            p1.N = N;
            p1.K = H;
            p1.ntaps = k;
            p1.shift0 = -(k - 1) / 2;
            p1.shift_step = 1;
            p1.Ys = FsC;
            p1.ys_slope = 0.0f;     // ReLU
            p1.bias = L.ffn1.bias;
            p1.mask = lay.d_mask;
            p1.mask_shift = 0;
            launch_conv_clx(p1, stream_);
            ConvClxParams p2;
            p2.X = FsC;
            p2.W = L.ffn2.cl.wx;
            p2.nmt = L.ffn2.cl.nmt;
            p2.M = H;
            p2.N = N;
            p2.K = Fc;
            p2.ntaps = k;
            p2.shift0 = -(k - 1) / 2;
            p2.shift_step = 1;
            p2.Ykm = Y.p;
            p2.ldykm = Y.ld;
            p2.Rkm = x.p;
            p2.ldrkm = x.ld;
            p2.bias = L.ffn2.bias;
            p2.mask = lay.d_mask;
            p2.mask_shift = 0;
            launch_conv_clx(p2, stream_);
        } else if (Fcl && conv_km_to_cl(L.ffn1, x, Fcl, Fc, 1, (k - 1) / 2, lay.d_mask, 1, stream_)) {
            SBV2_REQUIRE(conv_cl_to_km(L.ffn2, Fcl, Fc, Y, 1, (k - 1) / 2, lay.d_mask, 1, stream_, 0.0f, &x), "FFN: conv_2 has no matrix-core path");
        } else {
            conv_plain(L.ffn1, x, F, 1, (k - 1) / 2, lay.d_mask, 1, stream_, ACT_RELU);
            conv_plain(L.ffn2, F, Y, 1, (k - 1) / 2, lay.d_mask, 1, stream_, ACT_NONE, 1.0f, &x);
        }
        layernorm_ch(Y, x, L.n2g, L.n2b, 1e-5f, ACT_NONE, nullptr, 0, lay.d_mask, stream_, (SP && i + 1 < e.layers.size()) ? &Xs : nullptr);
    }
    ar.rewind(mk);
}

// modules.DDSConv: x += gelu(LN(1x1(gelu(LN(depthwise(x * mask))))))  x 3, dilation 3^i
void VitsModel::run_dds(const DDS& d, Plane x, const SegLayout& lay, Arena& ar) {
    const Arena::Mark mk = ar.mark();
    Plane a = ar.plane(x.C, x.L), b = ar.plane(x.C, x.L);
    int dil = 1;
    for (size_t i = 0; i < d.pw.size(); ++i) {
        dds_dw_ln_gelu(x, a, d.sep_w[i], d.sep_b[i], dil, d.n1g[i], d.n1b[i], lay.d_mask, stream_);
        conv_plain(d.pw[i], a, b, 1, 0, nullptr, 1, stream_);
        layernorm_ch(b, x, d.n2g[i], d.n2b[i], 1e-5f, ACT_GELU, x.p, x.ld, lay.d_mask, stream_);
        dil *= cfg_.sdp_kernel;
    }
    ar.rewind(mk);
}

// models_jp_extra.Generator
void VitsModel::run_decoder(Arena& ar, Plane z, const SegLayout& fl, const float* cond_vec) {
    const int n = fl.n;
    const int Lf = fl.L;
    Plane cur = ar.plane(cfg_.up_initial, Lf);
    conv_plain(dec_pre_, z, cur, 1, dec_pre_.k / 2, fl.d_mask, 1, stream_);
    add_segvec(cur, cond_vec, cfg_.up_initial, fl.d_seg_of, 1, fl.d_mask, stream_);
    trace("dec_pre", cur, fl, 1);
    int U = 1;
    const int nk = (int)cfg_.res_kernels.size();
    for (size_t si = 0; si < stages_.size(); ++si) {
        const Stage& st = stages_[si];
        U *= st.rate;
        const int Lo = Lf * U;
        Plane XS = ar.plane(st.ch, Lo);
        const Arena::Mark mk = ar.mark();
        Plane XU = ar.plane(st.ch, Lo), T1 = ar.plane(st.ch, Lo), YA = ar.plane(st.ch, Lo), YB = ar.plane(st.ch, Lo);
        for (const auto& g : st.up.groups) {
            ConvParams p;
            p.A = g.w;
            p.lda = g.lda;
            p.a_tap_stride = (int64_t)st.up.cin * g.lda;
            p.B = cur.p;
            p.ldb = cur.ld;
            p.nb = cur.L;
            p.C = XU.p;
            p.ldc = XU.ld;
            p.M = g.nph * st.up.cout;
            p.N = cur.L;
            p.K = st.up.cin;
            p.ntaps = g.ntaps;
            for (int t = 0; t < g.ntaps; ++t) p.shift[t] = g.shift[t];
            p.bias = st.up.bias;
            p.bias_mode = BIAS_ROW;
            p.pre_slope = 0.1f;
            p.mask = fl.d_mask;
            p.mask_div = U;
            p.out_stride = st.rate;
            p.phase_rows = st.up.cout;
            for (int q = 0; q < kMaxPhases; ++q) p.phase_off[q] = g.phase_off[q];
            launch_conv(p, stream_);
        }
        for (int j = 0; j < nk; ++j) {
            const ResBranch& rb = st.branches[j];
            Plane y = XU;
            const int nd = (int)rb.dil.size();
            for (int q = 0; q < nd; ++q) {
                const int d = rb.dil[q];
                conv_plain(rb.c1[q], y, T1, d, d * (rb.k - 1) / 2, fl.d_mask, U, stream_, ACT_NONE, 0.1f);
                if (q + 1 < nd) {
                    Plane yn = (y.p == YA.p) ? YB : YA;
                    conv_plain(rb.c2[q], T1, yn, 1, (rb.k - 1) / 2, fl.d_mask, U, stream_, ACT_NONE, 0.1f, &y);
                    y = yn;
                } else {
                    conv_plain(rb.c2[q], T1, XS, 1, (rb.k - 1) / 2, fl.d_mask, U, stream_, ACT_NONE, 0.1f, &y, 1.0f, 1.0f / nk, j > 0);
                }
            }
        }
        ar.rewind(mk);
        cur = XS;
        trace("dec_stage" + std::to_string(si), cur, fl, U);
    }
    // compact PCM, one contiguous block per utterance
    pcm_lens_.assign(n, 0);
    pcm_offs_.assign(n, 0);
    int64_t tot = 0, maxlen = 0;
    for (int u = 0; u < n; ++u) {
        pcm_offs_[u] = tot;
        pcm_lens_[u] = (int64_t)fl.len[u] * U;
        tot += pcm_lens_[u];
        maxlen = std::max(maxlen, pcm_lens_[u]);
    }
    pcm_total_ = tot;
    pcm_ = ar.array<float>((size_t)tot);
    int64_t* d_off = ar.array<int64_t>(n);
    ar.upload(d_off, pcm_offs_.data(), sizeof(int64_t) * n, stream_);
    conv_post_tanh(cur, dec_post_w_, dec_post_k_, 0.01f, fl.d_start, fl.d_len, d_off, n, U, maxlen, pcm_, stream_);
}

void VitsModel::forward(const VitsBatch& b) {
    HIP_CHECK(hipSetDevice(device_));
    SBV2_REQUIRE(b.n >= 1, "empty batch");
    HIP_CHECK(hipStreamSynchronize(stream_));  // pinned staging of the previous call must be drained before reuse
    arena_.reset();
    keep_.reset();
    traces_.clear();
    if (b.after_stream) {
        if (!after_ev_) HIP_CHECK(hipEventCreateWithFlags(&after_ev_, hipEventDisableTiming));
        HIP_CHECK(hipEventRecord(after_ev_, b.after_stream));
        HIP_CHECK(hipStreamWaitEvent(stream_, after_ev_, 0));
    }
    Arena& ar = arena_;
    const int n = b.n, H = cfg_.hidden, I = cfg_.inter;
    const uint64_t seed = b.seed;   // noise streams: noise_key(seed, index of the utterance in the caller's batch, stream)
    std::vector<int> T(n);
    int64_t total_t = 0;
    for (int u = 0; u < n; ++u) {
        SBV2_REQUIRE(b.t_lens[u] >= 1 && b.t_lens[u] < (1 << 20), "bad text length");
        T[u] = (int)b.t_lens[u];
        total_t += T[u];
    }
    tl_ = make_layout(T, kTextGap, ar, stream_);
    const SegLayout& tl = tl_;
    const int Lt = tl.L;

    // ---- inputs -------------------------------------------------------------------------------------
    std::vector<int> ph(Lt, 0), tn(Lt, 0), lg(Lt, 0), sid(n);
    {
        int64_t e = 0;
        for (int u = 0; u < n; ++u) {
            SBV2_REQUIRE(b.sids[u] >= 0 && b.sids[u] < cfg_.n_speakers, "speaker id out of range");
            sid[u] = (int)b.sids[u];
            for (int t = 0; t < T[u]; ++t, ++e) {
                SBV2_REQUIRE(b.phones[e] >= 0 && b.phones[e] < cfg_.n_vocab && b.tones[e] >= 0 && b.tones[e] < cfg_.n_tones &&
                                 b.langs[e] >= 0 && b.langs[e] < cfg_.n_langs,
                             "phone / tone / language id out of range");
                ph[tl.start[u] + t] = (int)b.phones[e];
                tn[tl.start[u] + t] = (int)b.tones[e];
                lg[tl.start[u] + t] = (int)b.langs[e];
            }
        }
    }
    int* d_ph = ar.array<int>(Lt);
    int* d_tn = ar.array<int>(Lt);
    int* d_lg = ar.array<int>(Lt);
    int* d_sid = ar.array<int>(n);
    int* d_uid = ar.array<int>(n);
    float* d_style = ar.array<float>((size_t)n * cfg_.style_dim);
    {
        // six neighbours in the arena, uploaded in allocation order: one copy (Arena::begin_uploads)
        UploadBatch ub(ar);
        std::vector<int> uid(n);
        for (int u = 0; u < n; ++u) uid[u] = b.utt_ids ? (int)b.utt_ids[u] : b.utt0 + u;
        ar.upload(d_ph, ph.data(), sizeof(int) * Lt, stream_);
        ar.upload(d_tn, tn.data(), sizeof(int) * Lt, stream_);
        ar.upload(d_lg, lg.data(), sizeof(int) * Lt, stream_);
        ar.upload(d_sid, sid.data(), sizeof(int) * n, stream_);
        ar.upload(d_uid, uid.data(), sizeof(int) * n, stream_);
        ar.upload(d_style, b.styles, sizeof(float) * (size_t)n * cfg_.style_dim, stream_);
    }

    Plane bert = ar.plane(cfg_.bert_dim, Lt);
    if (b.bert_host) {
        fill_zero(bert.p, sizeof(float) * (size_t)bert.C * bert.ld, stream_);
        HIP_CHECK(hipStreamSynchronize(stream_));
        int64_t off = 0;
        for (int u = 0; u < n; ++u) {
            HIP_CHECK(hipMemcpy2D(bert.p + tl.start[u], sizeof(float) * bert.ld, b.bert_host + off, sizeof(float) * T[u],
                                  sizeof(float) * T[u], cfg_.bert_dim, hipMemcpyHostToDevice));
            off += (int64_t)cfg_.bert_dim * T[u];
        }
    } else {
        SBV2_REQUIRE(b.bert_dev && b.bert_map, "no BERT features given");
        std::vector<int> map(Lt, -1);
        int64_t e = 0;
        for (int u = 0; u < n; ++u)
            for (int t = 0; t < T[u]; ++t, ++e) map[tl.start[u] + t] = b.bert_map[e];
        int* d_map = ar.array<int>(Lt);
        ar.upload(d_map, map.data(), sizeof(int) * Lt, stream_);
        gather_cols(*b.bert_dev, d_map, bert, stream_);
    }

    // ---- per-utterance vectors: g = emb_g(sid) and every Linear / 1x1 conv applied to it ---------------
    float* G = ar.array<float>((size_t)n * cfg_.gin);
    gather_rows(emb_g_, cfg_.gin, d_sid, G, n, stream_);
    auto gvec = [&](const float* w, const float* bias, int M) {
        float* out = ar.array<float>((size_t)n * M);
        linear_vec(w, bias, M, cfg_.gin, G, cfg_.gin, out, M, n, stream_);
        return out;
    };
    float* v_style = ar.array<float>((size_t)n * H);
    linear_vec(style_w_, style_b_, H, cfg_.style_dim, d_style, cfg_.style_dim, v_style, H, n, stream_);
    float* v_encp = gvec(enc_p_.spk_w, enc_p_.spk_b, H);
    float* v_dp = gvec(dp_cond_w_, dp_cond_b_, H);
    float* v_sdp = gvec(sdp_cond_w_, sdp_cond_b_, H);
    dec_cond_vec_ = gvec(dec_cond_w_, dec_cond_b_, cfg_.up_initial);
    std::vector<float*> v_flow;
    for (auto& c : flows_) v_flow.push_back(gvec(c.enc.spk_w, c.enc.spk_b, H));

    // ---- TextEncoder ----------------------------------------------------------------------------------
    std::unique_ptr<TraceRange> tr(new TraceRange("text_encoder"));
    Plane BP = ar.plane(H, Lt);
    conv_plain(bert_proj_, bert, BP, 1, 0, nullptr, 1, stream_);
    Plane X = ar.plane(H, Lt);
    text_embed(d_ph, d_tn, d_lg, tl.d_seg_of, emb_, tone_emb_, lang_emb_, BP.p, BP.ld, v_style, H, std::sqrt((float)H), X, stream_);
    trace("x_emb", X, tl);
    run_encoder(enc_p_, X, tl, v_encp, ar);
    trace("x", X, tl);
    Plane ST = ar.plane(2 * I, Lt);
    conv_plain(enc_proj_, X, ST, 1, 0, tl.d_mask, 1, stream_);
    Plane m_p = ST.rows(0, I), logs_p = ST.rows(I, I);
    trace("stats", ST, tl);

    // ---- DurationPredictor -----------------------------------------------------------------------------
    tr.reset();
    tr.reset(new TraceRange("durations"));
    Plane XD = ar.plane(H, Lt);
    HIP_CHECK(hipMemcpyAsync(XD.p, X.p, sizeof(float) * (size_t)H * X.ld, hipMemcpyDeviceToDevice, stream_));
    add_segvec(XD, v_dp, H, tl.d_seg_of, 1, tl.d_mask, stream_);
    Plane D1 = ar.plane(cfg_.dp_filter, Lt), D2 = ar.plane(cfg_.dp_filter, Lt), DPO = ar.plane(1, Lt);
    conv_plain(dp_c1_, XD, D1, 1, cfg_.dp_kernel / 2, tl.d_mask, 1, stream_, ACT_RELU);
    layernorm_ch(D1, D1, dp_n1g_, dp_n1b_, 1e-5f, ACT_NONE, nullptr, 0, tl.d_mask, stream_);
    conv_plain(dp_c2_, D1, D2, 1, cfg_.dp_kernel / 2, tl.d_mask, 1, stream_, ACT_RELU);
    layernorm_ch(D2, D2, dp_n2g_, dp_n2b_, 1e-5f, ACT_NONE, nullptr, 0, tl.d_mask, stream_);
    conv_plain(dp_proj_, D2, DPO, 1, 0, tl.d_mask, 1, stream_);

    // ---- StochasticDurationPredictor, reverse (executed even when sdp_ratio == 0, like the exported graph) ----
    Plane XSd = ar.plane(H, Lt), COND = ar.plane(H, Lt), HH = ar.plane(H, Lt);
    Plane PR = ar.plane(3 * cfg_.sdp_bins - 1, Lt), Z = ar.plane(2, Lt);
    conv_plain(sdp_pre_, X, XSd, 1, 0, nullptr, 1, stream_);
    add_segvec(XSd, v_sdp, H, tl.d_seg_of, 1, tl.d_mask, stream_);
    run_dds(sdp_dds_, XSd, tl, ar);
    conv_plain(sdp_proj_, XSd, COND, 1, 0, tl.d_mask, 1, stream_);
    float* z0 = Z.p;
    float* z1 = Z.p + Z.ld;
    noise_fill(Z.p, Z.ld, 2, tl.d_seg_of, tl.d_start, tl.d_len, d_uid, Lt, seed, 0, b.noise_scale_w, stream_);
    const float inv_sqrt_f = 1.0f / std::sqrt((float)H);
    for (int i = (int)sdp_cf_.size() - 1; i >= 0; --i) {
        const ConvFlow& cf = sdp_cf_[i];
        swap_rows(z0, z1, Lt, stream_);  // Flip
        convflow_pre(z0, cf.pre_w, cf.pre_b, COND, HH, tl.d_mask, stream_);
        run_dds(cf.dds, HH, tl, ar);
        conv_plain(cf.proj, HH, PR, 1, 0, tl.d_mask, 1, stream_);
        spline_inverse(PR, z0, z1, cfg_.sdp_bins, cfg_.sdp_tail, inv_sqrt_f, tl.d_mask, Lt, stream_);
    }
    swap_rows(z0, z1, Lt, stream_);
    affine_reverse(z0, z1, sdp_ea_m_, sdp_ea_logs_, sdp_ea_scale_, tl.d_mask, Lt, stream_);

    // ---- durations (the one device -> host sync of the batch: T_frames is data dependent) ----------------
    float* d_logw = ar.array<float>(Lt);
    int* d_dur = ar.array<int>(Lt);
    sbv2::durations(z0, DPO.p, b.sdp_ratio, b.length_scale, tl.d_mask, Lt, d_logw, d_dur, stream_);
    std::vector<int> dur_p(Lt);
    std::vector<float> logw_p(Lt);
    HIP_CHECK(hipMemcpyAsync(dur_p.data(), d_dur, sizeof(int) * Lt, hipMemcpyDeviceToHost, stream_));
    HIP_CHECK(hipMemcpyAsync(logw_p.data(), d_logw, sizeof(float) * Lt, hipMemcpyDeviceToHost, stream_));
    // (the f16x3 clamp count of everything queued so far: DeBERTa's products when a pipeline run feeds this call on the same device, the previous call's
    // flow; the one host sync of the batch is here anyway)
    sat_watch_.enqueue(stream_);
    HIP_CHECK(hipStreamSynchronize(stream_));
    sat_watch_.check("DeBERTa / flow products seen by the VITS call");
    dur_host_.assign((size_t)total_t, 0);
    logw_host_.assign((size_t)total_t, 0.f);
    std::vector<int> Tf(n);
    std::vector<std::vector<int>> used(n);
    {
        int64_t e = 0;
        for (int u = 0; u < n; ++u) {
            int64_t sum = 0;
            used[u].resize(T[u]);
            for (int t = 0; t < T[u]; ++t, ++e) {
                dur_host_[e] = dur_p[tl.start[u] + t];
                logw_host_[e] = logw_p[tl.start[u] + t];
                const int64_t dv = b.forced_durations ? b.forced_durations[e] : dur_host_[e];
                SBV2_REQUIRE(dv >= 0 && dv < (1 << 20), "duration out of range");
                used[u][t] = (int)dv;
                sum += dv;
            }
            SBV2_REQUIRE(sum < (1 << 24), "utterance too long");
            Tf[u] = (int)std::max<int64_t>(sum, 1);  // torch.clamp_min(sum, 1)
        }
    }
    // (frame count rounded to 32: every plane of the decoder's wide stages - 8 and 64 positions per frame - is then a whole number of conv_clx's 256-position
    // tiles, and the kernel instance without the guarded last-tile epilogue applies)
    ar.begin_uploads();   // the layout's four arrays and the token map: neighbours, one copy
    fl_ = make_layout(Tf, kFrameGap, ar, stream_, nullptr, 32);
    const SegLayout& fl = fl_;
    const int Lf = fl.L;
    std::vector<int> tok(Lf, -1);
    for (int u = 0; u < n; ++u) {
        int y = fl.start[u];
        for (int t = 0; t < T[u]; ++t)
            for (int q = 0; q < used[u][t]; ++q) tok[y++] = tl.start[u] + t;
    }
    int* d_tok = ar.array<int>(Lf);
    ar.upload(d_tok, tok.data(), sizeof(int) * Lf, stream_);
    ar.end_uploads();

    // ---- alignment expansion + prior sample ---------------------------------------------------------------
    tr.reset();
    tr.reset(new TraceRange("flow"));
    Plane ZA = ar.plane(I, Lf), ZB = ar.plane(I, Lf);
    expand_frames(m_p, logs_p, d_tok, fl.d_seg_of, fl.d_start, fl.d_len, d_uid, seed, b.noise_scale, ZA, stream_);
    trace("z_p", ZA, fl);

    // ---- TransformerCouplingBlock, reverse ----------------------------------------------------------------
    const int half = I / 2;
    Plane Hf = ar.plane(H, Lf);
    for (int i = cfg_.flow_n - 1; i >= 0; --i) {
        const Coupling& c = flows_[i];
        flip_channels(ZA, ZB, stream_);
        std::swap(ZA, ZB);
        conv_plain(c.pre, ZA.rows(0, half), Hf, 1, 0, fl.d_mask, 1, stream_);
        run_encoder(c.enc, Hf, fl, v_flow[i], ar);
        Plane x1 = ZA.rows(half, half);
        conv_plain(c.post, Hf, x1, 1, 0, fl.d_mask, 1, stream_, ACT_NONE, 1.0f, &x1, -1.0f);  // x1 = (x1 - m) * mask
    }
    trace("z", ZA, fl);
    tr.reset();
    z_ = ZA;
    if (b.skip_decoder) {   // streaming: the caller decodes chunk by chunk (stream_begin / stream_chunk)
        pcm_ = nullptr;
        pcm_total_ = 0;
        pcm_lens_.assign(n, 0);
        pcm_offs_.assign(n, 0);
        return;
    }
    TraceRange td("decoder");
    if (dec_mode_) run_decoder_cl(ar, ZA, fl, dec_cond_vec_);
    else run_decoder(ar, ZA, fl, dec_cond_vec_);
    log_line("vits forward: " + std::to_string(n) + " utterance(s), " + std::to_string(Lt) + " text columns, " + std::to_string(Lf) + " frame columns");
}

// ---- streaming long-form decode ---------------------------------------------------------------------------------------------------
// The generator is purely convolutional: frames [f0, f0 + chunk) depend on z[f0 - 13.4, f0 + chunk + 13.4) only, so a window with a
// 16-frame halo on both sides reproduces the whole-sequence result on its centre.  The window's column mask marks which of its frames
// exist (the sequence ends are zero padding at EVERY layer, exactly like the gaps of a packed batch), interior window edges only
// contaminate the halo, which is discarded.  The window shape is fixed, so the ~90 launches of the decoder are captured once into a
// hipGraph and replayed per chunk; all device pointers of the captured launches live in the plan's own arena.
// Receptive field of the generator per side, in frames: conv_pre + per stage {transposed-conv taps at the input rate, the widest MRF branch
// sum_d (d + 1)(k - 1) / 2 at the output rate} + conv_post; + 1 frame of margin, rounded up to a multiple of 4.
int VitsModel::stream_halo() const {
    double rf = (dec_pre_.k - 1) / 2.0;
    double rate = 1.0;
    for (size_t i = 0; i < stages_.size(); ++i) {
        int tmax = 0;
        for (const auto& g : stages_[i].up.groups)
            for (int t = 0; t < g.ntaps; ++t) tmax = std::max(tmax, std::abs(g.shift[t]));
        rf += tmax / rate;
        rate *= stages_[i].rate;
        int mrf = 0;
        for (const auto& rb : stages_[i].branches) {
            int h = 0;
            for (int d : rb.dil) h += (d + 1) * (rb.k - 1) / 2;
            mrf = std::max(mrf, h);
        }
        rf += mrf / rate;
    }
    rf += (dec_post_k_ - 1) / 2.0 / rate;
    return round_up((int)std::ceil(rf) + 1, 4);
}

// A decode plan = nwin windows of chunk + 2 * halo frames packed into one batch (the packed-batch layout: the gaps between the windows are
// the zero padding of every convolution), its persistent input / conditioning buffers, its own arena and the graph captured for that shape.
// chunk_ (one window) decodes the first chunk of an utterance: time to first PCM; burst_ (kStreamBurst windows per replay) every later
// one: a 288-frame window is ~90 launches of mostly < 100 workgroups, i.e. launch-latency bound, and four windows per replay cost little
// more than one (profiles/r03_longform_*.json).
void VitsModel::ensure_plan(std::shared_ptr<ChunkPlan>& slot, int chunk_frames, int nwin) {
    static const bool no_graph = getenv("SBV2_STREAM_GRAPH") && atoi(getenv("SBV2_STREAM_GRAPH")) == 0;   // A/B knob
    if (!slot || slot->chunk != chunk_frames || slot->nwin != nwin) {
        slot.reset(new ChunkPlan);
        ChunkPlan& c = *slot;
        c.chunk = chunk_frames;
        c.nwin = nwin;
        c.W = chunk_frames + 2 * stream_halo();
        c.lay = make_layout(std::vector<int>(nwin, c.W), kFrameGap, c.ar, stream_);
        c.zin = c.ar.plane(cfg_.inter, c.lay.L);
        fill_zero(c.zin.p, sizeof(float) * (size_t)c.zin.C * c.zin.ld, stream_);
        c.cond = c.ar.array<float>((size_t)cfg_.up_initial * nwin);
        c.mark = c.ar.mark();
        for (int i = 0; i < 2; ++i) {
            HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&c.host[i]), sizeof(float) * (size_t)nwin * chunk_frames * cfg_.hop(), hipHostMallocDefault));
            HIP_CHECK(hipEventCreateWithFlags(&c.ev[i], hipEventDisableTiming));
        }
    }
    ChunkPlan& c = *slot;
    // the speaker conditioning vector of THIS utterance goes into the plan's persistent buffer (the captured launches read it there), once per window
    for (int w = 0; w < nwin; ++w)
        HIP_CHECK(hipMemcpyAsync(c.cond + (size_t)w * cfg_.up_initial, dec_cond_vec_, sizeof(float) * (size_t)cfg_.up_initial, hipMemcpyDeviceToDevice, stream_));
    if (!c.exec && !no_graph) {
        // warm-up pass (sizes the arena: no hipMalloc may happen under capture), then the same launch sequence under capture
        auto decode = [&]() {
            c.ar.rewind(c.mark);
            if (dec_mode_) run_decoder_cl(c.ar, c.zin, c.lay, c.cond);
            else run_decoder(c.ar, c.zin, c.lay, c.cond);
            c.pcm = pcm_;
        };
        const bool tr = trace_;
        trace_ = false;
        decode();
        HIP_CHECK(hipStreamSynchronize(stream_));
        HIP_CHECK(hipStreamBeginCapture(stream_, hipStreamCaptureModeThreadLocal));
        try {
            decode();
        } catch (...) {
            hipGraph_t g = nullptr;
            (void)hipStreamEndCapture(stream_, &g);
            if (g) (void)hipGraphDestroy(g);
            trace_ = tr;
            throw;
        }
        HIP_CHECK(hipStreamEndCapture(stream_, &c.graph));
        HIP_CHECK(hipGraphInstantiate(&c.exec, c.graph, nullptr, nullptr, 0));
        trace_ = tr;
    }
    if (!c.exec) {   // eager path: the plan's pinned staging is recycled once per stream
        HIP_CHECK(hipStreamSynchronize(stream_));
        c.ar.reset_pinned();
    }
    c.slot_f0[0] = c.slot_f0[1] = -1;
}

int64_t VitsModel::stream_begin(int chunk_frames) {
    HIP_CHECK(hipSetDevice(device_));
    SBV2_REQUIRE(fl_.n == 1 && z_.p, "stream_begin needs a preceding forward of ONE utterance with skip_decoder");
    SBV2_REQUIRE(chunk_frames >= 16 && chunk_frames <= (1 << 20), "chunk_frames must be in [16, 2^20]");
    constexpr int burst = kStreamBurst;   // (1 / 2 / 4 / 8 / 12 / 16 windows per replay measured in round 3: 2.62 / 1.88 / 1.64 / 1.41 / 1.39 / 1.47 ms per chunk)
    const int64_t Tf = fl_.len[0];
    ensure_plan(chunk_, chunk_frames, 1);
    const bool want_burst = burst > 1 && Tf > chunk_frames;   // an utterance of one chunk never needs it
    if (want_burst) {
        // windows per replay: no more than the chunks that follow the first one (a 2-chunk utterance does not decode 7 all-zero windows, nor pay
        // their workspace: 256 MiB per window at 256-frame chunks), from {2, 4, 8, 16} so that utterances of different lengths share few captures;
        // a larger plan of the same chunk size that this handle already holds is reused as it is
        const int64_t rest = (Tf - 1) / chunk_frames;
        int nw = 2;
        while (nw < burst && nw < rest) nw *= 2;
        nw = std::min(nw, burst);
        if (burst_ && burst_->chunk == chunk_frames && burst_->nwin >= nw) nw = burst_->nwin;
        ensure_plan(burst_, chunk_frames, nw);
    }
    stream_bursts_ = want_burst;
    stream_enqueue(*chunk_, 0, 0);                            // the first chunk starts right behind the flow ...
    if (want_burst) stream_enqueue(*burst_, chunk_frames, 0); // ... and the first burst right behind it
    return Tf;
}

// windows + decoder (graph replay) + device -> pinned-host copies of the windows' centres, all asynchronous on the model's stream.
// The plan's windows are the chunks starting at f0, f0 + chunk, ... (windows past the end of the utterance decode zeros and are not copied).
void VitsModel::stream_enqueue(ChunkPlan& c, int64_t f0, int slot) {
    const int64_t Tf = fl_.len[0];
    const int hop = cfg_.hop(), halo = stream_halo();
    // window w = frames [f0_w - halo, f0_w - halo + W) of this utterance (fl_.start[0] is its first column in the packed plane)
    Plane zu = z_;
    zu.p = z_.p + fl_.start[0];
    zu.L = (int)Tf;
    for (int w = 0; w < c.nwin; ++w) {
        const int64_t fw = std::min<int64_t>(f0 + (int64_t)w * c.chunk, Tf + c.W);   // (past the end: an all-zero window)
        window_cols(zu, (int)(fw - halo), Plane{c.zin.p + c.lay.start[w], c.zin.C, c.W, c.zin.ld}, c.lay.d_mask + c.lay.start[w], stream_);
    }
    if (c.exec) {
        HIP_CHECK(hipGraphLaunch(c.exec, stream_));
    } else {
        c.ar.rewind(c.mark);
        if (dec_mode_) run_decoder_cl(c.ar, c.zin, c.lay, c.cond);
        else run_decoder(c.ar, c.zin, c.lay, c.cond);
        c.pcm = pcm_;
    }
    int64_t total = 0;
    for (int w = 0; w < c.nwin; ++w) {
        const int64_t fw = f0 + (int64_t)w * c.chunk;
        if (fw >= Tf) break;
        const int64_t nsamp = std::min<int64_t>(c.chunk, Tf - fw) * hop;
        HIP_CHECK(hipMemcpyAsync(c.host[slot] + (size_t)w * c.chunk * hop, c.pcm + ((size_t)w * c.W + halo) * hop, sizeof(float) * (size_t)nsamp,
                                 hipMemcpyDeviceToHost, stream_));
        total += nsamp;
    }
    HIP_CHECK(hipEventRecord(c.ev[slot], stream_));
    c.slot_f0[slot] = f0;
    c.slot_n[slot] = total;
}

int64_t VitsModel::stream_chunk(int64_t f0, float* dst_host, int64_t capacity) {
    HIP_CHECK(hipSetDevice(device_));
    SBV2_REQUIRE(chunk_ && z_.p && fl_.n == 1, "stream_chunk without stream_begin");
    ChunkPlan& c1 = *chunk_;
    const int64_t Tf = fl_.len[0];
    SBV2_REQUIRE(f0 >= 0 && f0 < Tf && f0 % c1.chunk == 0, "chunk start out of range");
    const int hop = cfg_.hop();
    const int64_t nsamp = std::min<int64_t>(c1.chunk, Tf - f0) * hop;
    SBV2_REQUIRE(capacity >= nsamp, "PCM buffer too small for the chunk");
    const int64_t ci = f0 / c1.chunk;
    const bool bursts = stream_bursts_ && burst_ && burst_->chunk == c1.chunk;
    if (ci == 0 || !bursts) {
        // the single-window plan: the utterance's first chunk (and everything when bursts are off); the next chunk runs while this one is delivered
        const int slot = (int)(ci & 1);
        if (c1.slot_f0[slot] != f0) stream_enqueue(c1, f0, slot);                                  // (random access: not the streaming order)
        if (!bursts && f0 + c1.chunk < Tf && c1.slot_f0[slot ^ 1] != f0 + c1.chunk) stream_enqueue(c1, f0 + c1.chunk, slot ^ 1);
        HIP_CHECK(hipEventSynchronize(c1.ev[slot]));
        std::memcpy(dst_host, c1.host[slot], sizeof(float) * (size_t)nsamp);
        c1.slot_f0[slot] = -1;
        return nsamp;
    }
    ChunkPlan& cb = *burst_;
    const int64_t bi = (ci - 1) / cb.nwin, within = (ci - 1) % cb.nwin;
    const int64_t bf0 = (1 + bi * cb.nwin) * c1.chunk;      // first frame of this chunk's burst
    const int slot = (int)(bi & 1);
    if (cb.slot_f0[slot] != bf0) stream_enqueue(cb, bf0, slot);                                    // (random access)
    // the following burst is decoded while this one is delivered (its slot was drained one burst ago)
    const int64_t nf0 = bf0 + (int64_t)cb.nwin * c1.chunk;
    if (within == 0 && nf0 < Tf && cb.slot_f0[slot ^ 1] != nf0) stream_enqueue(cb, nf0, slot ^ 1);
    HIP_CHECK(hipEventSynchronize(cb.ev[slot]));
    std::memcpy(dst_host, cb.host[slot] + (size_t)within * c1.chunk * hop, sizeof(float) * (size_t)nsamp);
    return nsamp;
}

void VitsModel::copy_pcm(float* host) {
    HIP_CHECK(hipSetDevice(device_));
    HIP_CHECK(hipMemcpyAsync(host, pcm_, sizeof(float) * (size_t)pcm_total_, hipMemcpyDeviceToHost, stream_));
    HIP_CHECK(hipStreamSynchronize(stream_));
}

}  // namespace sbv2
