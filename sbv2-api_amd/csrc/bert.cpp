// DeBERTa-v2 encoder (the graph behind `bert::predict`, crates/sbv2_core/src/bert.rs:6-24, exported by
// scripts/convert/convert_deberta.py:22-35 as hidden_states[-3] of AutoModelForMaskedLM).
//
// Batched over utterances by PACKING tokens (no padding): every projection / FFN is one GEMM over all tokens of the
// batch; attention runs as grouped GEMMs over (utterance, head).  Activations are channel-major planes [hidden][tokens].
// The relative-position projections key_proj(LN(rel_emb)) / query_proj(LN(rel_emb)) are input independent and are
// computed once at load time.
#include <algorithm>
#include <cmath>
#include <cstring>

#include "models.h"

namespace sbv2 {

// Default arithmetic of DeBERTa's GEMMs: f16x3 (f16 hi + scaled f16 lo per operand, 22 mantissa bits, three MFMAs per product; common.h).
// Like bf16x6 (three bf16 parts, six MFMAs) it changes the integer durations no more often than a re-ordered f32 sum does (1 flip in
// 205 600 symbols each, the f32-vs-f32 control 2: profiles/r03_flip_rate_bert.json) and its products are as close to f64 as numpy's f32
// (profiles/r03e_gemm_bfs_probe.txt), at half of bf16x6's matrix work.  Its exponent range is f16's (+-65504 saturates; DeBERTa's GEMM
// inputs are LayerNorm / GELU / attention outputs, and the reference's own fp16 BERT option, model.rs:11-17, has the same range).
int default_bert_bfs_parts() { return kPartsF16x3; }

// modeling_deberta_v2.py:57-69 (float32 arithmetic like torch)
std::vector<int> BertModel::bucket_table(int maxS, int buckets, int max_rel) {
    std::vector<int> tab(2 * maxS - 1);
    const int mid = buckets / 2;
    for (int rel = -(maxS - 1); rel <= maxS - 1; ++rel) {
        int b = rel;
        if (buckets > 0 && max_rel > 0) {
            const float r = (float)rel;
            const float sign = r > 0 ? 1.f : (r < 0 ? -1.f : 0.f);
            const float abs_pos = (rel < mid && rel > -mid) ? (float)(mid - 1) : std::fabs(r);
            if (abs_pos <= (float)mid) {
                b = rel;
            } else {
                const float lp = std::ceil(std::log(abs_pos / (float)mid) / std::log((float)(max_rel - 1) / (float)mid) * (float)(mid - 1)) +
                                 (float)mid;
                b = (int)(lp * sign);
            }
        }
        tab[rel + maxS - 1] = b;
    }
    return tab;
}

BertModel::BertModel(const Blob& blob, int device) : device_(device) {
    SBV2_REQUIRE(blob.kind == 1, "weight container is not a DeBERTa (kind 1) model");
    HIP_CHECK(hipSetDevice(device));
    HIP_CHECK(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));  // never serialised against the NULL stream (e.g. RCCL launched by the caller)
    HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&sk_counters_), sizeof(unsigned) * kSkCounters));
    HIP_CHECK(hipMemset(sk_counters_, 0, sizeof(unsigned) * kSkCounters));
    f16x3_sat_prepare();
    sat_watch_.baseline();
    const std::string& js = blob.config_json;
    cfg_.vocab = (int)json_number(js, "vocab_size");
    cfg_.hidden = (int)json_number(js, "hidden");
    cfg_.layers = (int)json_number(js, "layers");
    cfg_.heads = (int)json_number(js, "heads");
    cfg_.inter = (int)json_number(js, "intermediate");
    cfg_.buckets = (int)json_number(js, "position_buckets");
    cfg_.max_rel = (int)json_number(js, "max_relative_positions");
    cfg_.eps = (float)json_number(js, "ln_eps");
    // DebertaV2Encoder.conv (modeling_deberta_v2.py:449-470,592,664): present iff the checkpoint's config has conv_kernel_size > 0
    cfg_.conv_k = json_has(js, "conv_kernel_size") ? (int)json_number(js, "conv_kernel_size") : 0;
    if (cfg_.conv_k > 0) {
        const std::string a = json_has(js, "conv_act") ? json_string(js, "conv_act") : "tanh";
        if (a == "gelu") cfg_.conv_act = ACT_GELU;
        else if (a == "tanh") cfg_.conv_act = ACT_TANH;
        else if (a == "relu") cfg_.conv_act = ACT_RELU;
        else throw Error("DeBERTa conv_act '" + a + "' is not supported (gelu, tanh, relu)");
        SBV2_REQUIRE(cfg_.conv_k % 2 == 1 && cfg_.conv_k <= kMaxTaps, "DeBERTa conv_kernel_size must be odd and <= 12");
        SBV2_REQUIRE(!json_has(js, "conv_groups") || (int)json_number(js, "conv_groups") == 1, "DeBERTa conv_groups != 1 is not supported");
    } else {
        // a checkpoint that carries the ConvLayer but whose config does not declare it would silently produce wrong features
        SBV2_REQUIRE(!blob.has("deberta.encoder.conv.conv.weight"),
                     "the container holds deberta.encoder.conv.* but its config has no conv_kernel_size: refusing to ignore the ConvLayer");
    }
    SBV2_REQUIRE(cfg_.hidden % cfg_.heads == 0 && (cfg_.hidden / cfg_.heads) % 4 == 0, "head size must be a multiple of 4");
    // Arithmetic of the 1x1 products (all of DeBERTa's GEMMs): SBV2_BERT_GEMM = f32 (exact f32 MFMA, gemm_conv.hip) | bf16x6 (three bf16
    // parts per operand, six MFMAs per product: f32-grade, the dropped terms are 2^-24) | bf16x3 (two parts, three MFMAs, 2^-16: what
    // BASELINE configs[2] calls "bf16 MFMA for the DeBERTa GEMMs"; the reference itself offers fp16 TensorRT for BERT, model.rs:11-17).
    // Both split modes run on gemm_bfs.hip with the activation parts written by LayerNorm / the previous product / split_planes.
    // (SBV2_GEMM=bf16: the k-major variant of conv_cl.hip for the ConvLayer's k = 3 product, opt-in as before.)
    int gemm_parts = 0;
    if (const char* m = getenv("SBV2_GEMM")) {
        const std::string v(m);
        if (v == "f32") gemm_parts = 0;
        else if (v == "bf16") gemm_parts = 1;
        else if (v == "f16") gemm_parts = 0;   // a VITS-side knob
        else SBV2_REQUIRE(v == "bf16x3" || v.empty(), "SBV2_GEMM must be f32, bf16x3, bf16 or f16");
    }
    bfs_parts_ = default_bert_bfs_parts();
    if (const char* m = getenv("SBV2_BERT_GEMM")) {
        const std::string v(m);
        if (v == "f32") bfs_parts_ = 0;
        else if (v == "bf16x3") bfs_parts_ = 2;
        else if (v == "bf16x6") bfs_parts_ = 3;
        else if (v == "f16x3") bfs_parts_ = kPartsF16x3;
        else SBV2_REQUIRE(v.empty(), "SBV2_BERT_GEMM must be f32, bf16x3, bf16x6 or f16x3");
    }
    if ((cfg_.hidden & 15) || (cfg_.inter & 15)) bfs_parts_ = 0;   // the split kernel wants 16-deep chunks
    ws_.reset(new WeightStore(blob, gemm_parts));
    const int Hc = cfg_.hidden;
    SBV2_REQUIRE(cfg_.vocab >= 1 && Hc >= 4 && cfg_.layers >= 1 && cfg_.inter >= 1, "bad DeBERTa config");
    emb_ = ws_->tensor("deberta.embeddings.word_embeddings.weight", {cfg_.vocab, Hc});
    emb_g_ = ws_->tensor("deberta.embeddings.LayerNorm.weight", {Hc});
    emb_b_ = ws_->tensor("deberta.embeddings.LayerNorm.bias", {Hc});
    if (cfg_.conv_k > 0) {
        conv_ = ws_->conv("deberta.encoder.conv.conv");
        ws_->expect(conv_, "deberta.encoder.conv.conv", Hc, Hc, cfg_.conv_k);
        conv_g_ = ws_->tensor("deberta.encoder.conv.LayerNorm.weight", {Hc});
        conv_b_ = ws_->tensor("deberta.encoder.conv.LayerNorm.bias", {Hc});
    }

    // LayerNorm of the relative embeddings on the host (modeling_deberta_v2.py:595-599), stored as a plane [H][2*span]
    const int H = cfg_.hidden;
    const int span = cfg_.buckets > 0 ? cfg_.buckets : cfg_.max_rel;
    // An onnxsim-processed export (convert_deberta.py:52) has that whole input-independent subgraph folded away: no rel_embeddings / encoder.LayerNorm,
    // but per layer the PROJECTED positions key_proj(LN(rel)) / query_proj(LN(rel)) as constants, which the importer stores as
    // "<layer>.attention.self.pos_key" / ".pos_query" [2 span][H] (csrc/import.cpp, rule 6).  Those are taken as they are.
    const bool folded = !blob.has("deberta.encoder.rel_embeddings.weight");
    Plane rel;
    rel.C = H;
    rel.L = 2 * span;
    rel.ld = round_up(2 * span, 64);
    if (!folded) {
        const HostTensor& re = blob.get("deberta.encoder.rel_embeddings.weight");
        const HostTensor& rg = blob.get("deberta.encoder.LayerNorm.weight");
        const HostTensor& rb = blob.get("deberta.encoder.LayerNorm.bias");
        SBV2_REQUIRE(re.dims.size() == 2 && re.dims[0] >= 2 * span && re.dims[1] == H, "rel_embeddings shape");
        SBV2_REQUIRE(rg.numel() == H && rb.numel() == H, "encoder.LayerNorm shape");
        std::vector<float> hp((size_t)H * rel.ld, 0.f);
        for (int r = 0; r < 2 * span; ++r) {
            const float* row = re.data + (size_t)r * H;
            float mean = 0.f;
            for (int c = 0; c < H; ++c) mean += row[c];
            mean /= H;
            float var = 0.f;
            for (int c = 0; c < H; ++c) var += (row[c] - mean) * (row[c] - mean);
            const float rstd = 1.0f / std::sqrt(var / H + cfg_.eps);
            for (int c = 0; c < H; ++c) hp[(size_t)c * rel.ld + r] = (row[c] - mean) * rstd * rg.data[c] + rb.data[c];
        }
        rel.p = ws_->upload(hp.data(), hp.size());
    }
    auto folded_plane = [&](const std::string& name) {   // [2 span][H] rows -> the k-major plane [H][ld] the attention kernels read
        const HostTensor& t = blob.get(name);
        SBV2_REQUIRE(t.dims.size() >= 2 && t.dims[0] == 2 * span && t.numel() == (int64_t)2 * span * H, name + ": expected [2 * position_buckets][hidden]");
        std::vector<float> hp((size_t)H * rel.ld, 0.f);
        for (int r = 0; r < 2 * span; ++r)
            for (int c = 0; c < H; ++c) hp[(size_t)c * rel.ld + r] = t.data[(size_t)r * H + c];
        Plane pl = rel;
        pl.p = ws_->upload(hp.data(), hp.size());
        return pl;
    };

    layers_.resize(cfg_.layers);
    for (int i = 0; i < cfg_.layers; ++i) {
        const std::string p = "deberta.encoder.layer." + std::to_string(i) + ".";
        Layer& L = layers_[i];
        // (q, k, v alone serve the load-time position projections and the long-sequence path: exact f32 only)
        L.q = ws_->linear(p + "attention.self.query_proj");
        L.k = ws_->linear(p + "attention.self.key_proj");
        L.v = ws_->linear(p + "attention.self.value_proj");
        ws_->set_bfs_parts(bfs_parts_);   // the four products of a layer's forward also get their pre-split bf16 fragments
        L.o = ws_->linear(p + "attention.output.dense");
        L.qkv = ws_->conv_cat({p + "attention.self.query_proj", p + "attention.self.key_proj", p + "attention.self.value_proj"});
        L.ln1_g = ws_->tensor(p + "attention.output.LayerNorm.weight", {H});
        L.ln1_b = ws_->tensor(p + "attention.output.LayerNorm.bias", {H});
        L.ffn1 = ws_->linear(p + "intermediate.dense");
        L.ffn2 = ws_->linear(p + "output.dense");
        ws_->set_bfs_parts(0);
        L.ln2_g = ws_->tensor(p + "output.LayerNorm.weight", {H});
        L.ln2_b = ws_->tensor(p + "output.LayerNorm.bias", {H});
        ws_->expect(L.q, p + "query_proj", H, H, 1);
        ws_->expect(L.k, p + "key_proj", H, H, 1);
        ws_->expect(L.v, p + "value_proj", H, H, 1);
        ws_->expect(L.o, p + "attention.output.dense", H, H, 1);
        ws_->expect(L.ffn1, p + "intermediate.dense", cfg_.inter, H, 1);
        ws_->expect(L.ffn2, p + "output.dense", H, cfg_.inter, 1);
        // share_att_key: positions go through the layer's own key/query projections (:292-299)
        if (folded) {
            L.pos_k = folded_plane(p + "attention.self.pos_key");
            L.pos_q = folded_plane(p + "attention.self.pos_query");
        } else {
            std::vector<float> zero((size_t)H * rel.ld, 0.f);
            L.pos_k = rel;
            L.pos_k.p = ws_->upload(zero.data(), zero.size());
            L.pos_q = rel;
            L.pos_q.p = ws_->upload(zero.data(), zero.size());
            conv_plain(L.k, rel, L.pos_k, 1, 0, nullptr, 1, stream_);
            conv_plain(L.q, rel, L.pos_q, 1, 0, nullptr, 1, stream_);
        }
    }
    HIP_CHECK(hipStreamSynchronize(stream_));
}

BertModel* BertModel::clone() const {
    HIP_CHECK(hipSetDevice(device_));
    BertModel* c = new BertModel(*this);   // shares ws_ (device weights); Arena copies are empty
    c->stream_ = nullptr;
    HIP_CHECK(hipStreamCreateWithFlags(&c->stream_, hipStreamNonBlocking));
    c->sk_counters_ = nullptr;
    HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&c->sk_counters_), sizeof(unsigned) * kSkCounters));
    HIP_CHECK(hipMemset(c->sk_counters_, 0, sizeof(unsigned) * kSkCounters));
    c->out_ = Plane{};
    c->layout_ = SegLayout{};
    return c;
}

BertModel::~BertModel() {
    (void)hipSetDevice(device_);
    if (stream_) (void)hipStreamDestroy(stream_);
    if (sk_counters_) (void)hipFree(sk_counters_);
}

void BertModel::forward(int n, const int64_t* ids, const int64_t* mask, const int64_t* lens) {
    HIP_CHECK(hipSetDevice(device_));
    SBV2_REQUIRE(n >= 1, "empty batch");
    TraceRange tr("deberta");
    HIP_CHECK(hipStreamSynchronize(stream_));  // pinned staging of the previous call must be drained before reuse
    arena_.reset();
    arena_.begin_uploads();   // every table of the pass is uploaded before the first kernel (deberta_embed_ln below): held back, neighbours merged
    const int H = cfg_.hidden, nh = cfg_.heads, d = H / nh;
    const int span = cfg_.buckets > 0 ? cfg_.buckets : cfg_.max_rel;
    std::vector<int> L(n);
    int64_t total = 0;
    for (int i = 0; i < n; ++i) {
        SBV2_REQUIRE(lens[i] >= 1 && lens[i] < (1 << 20), "bad sequence length");
        L[i] = (int)lens[i];
        total += lens[i];
    }
    std::vector<unsigned char> am((size_t)total, 1);
    if (mask)
        for (int64_t e = 0; e < total; ++e) am[e] = mask[e] != 0;
    // packed utterances are separated by >= conv_k / 2 zero columns when the ConvLayer is present: the gap is its zero padding
    layout_ = make_layout(L, cfg_.conv_k / 2, arena_, stream_, am.data());
    const SegLayout& lay = layout_;
    const int N = lay.L;
    const int maxT = lay.max_len();

    // token ids in the packed layout (-1 in alignment gaps)
    std::vector<int> hid(N, -1);
    {
        int64_t e = 0;
        for (int i = 0; i < n; ++i)
            for (int t = 0; t < L[i]; ++t, ++e) {
                SBV2_REQUIRE(ids[e] >= 0 && ids[e] < cfg_.vocab, "token id out of range");
                hid[lay.start[i] + t] = (int)ids[e];
            }
    }
    int* d_ids = arena_.array<int>(N);
    arena_.upload(d_ids, hid.data(), sizeof(int) * N, stream_);
    // valid-token mask (layout) is all that LayerNorm outputs are multiplied with; the attention mask has its own array
    std::vector<unsigned char> valid(N, 0);
    for (int i = 0; i < n; ++i)
        for (int t = 0; t < L[i]; ++t) valid[lay.start[i] + t] = 1;
    unsigned char* d_valid = arena_.array<unsigned char>(N);
    arena_.upload(d_valid, valid.data(), N, stream_);

    // relative-position window used by this batch
    const std::vector<int> tab = bucket_table(maxT, cfg_.buckets, cfg_.max_rel);
    auto clampi = [&](int v) { return std::min(std::max(v, 0), 2 * span - 1); };
    int dmin = 2 * span, dmax = 0;
    for (int v : tab) {
        dmin = std::min(dmin, std::min(clampi(v + span), clampi(-v + span)));
        dmax = std::max(dmax, std::max(clampi(v + span), clampi(-v + span)));
    }
    const int win_lo = dmin / 4 * 4;
    const int wlen = dmax - win_lo + 1;
    const int win_ld = round_up(wlen, 4);
    int* d_tab = arena_.array<int>(tab.size());
    arena_.upload(d_tab, tab.data(), sizeof(int) * tab.size(), stream_);

    // attention problem descriptors.  Utterances of <= 64 tokens (the usual sentence) take the fused kernel of attn_deberta.hip, 65 .. 128
    // tokens its tiled variant (the reference's TensorRT profile allows 100 tokens: model.rs:15), longer ones its key-tile loop (long-form
    // text); a batch may hold all three.  SBV2_BERT_ATTN=unfused sends everything down the grouped-GEMM + softmax path (five launches per
    // layer; also what a head dimension the fused kernels do not cover falls back to), SBV2_BERT_ATTN=nolong only the > 128-token ones.
    static const bool want_fused = !(getenv("SBV2_BERT_ATTN") && std::string(getenv("SBV2_BERT_ATTN")) == "unfused");
    static const bool want_long = !(getenv("SBV2_BERT_ATTN") && std::string(getenv("SBV2_BERT_ATTN")) == "nolong");
    // bucket window reachable by a short utterance (|i - j| <= 63)
    int win_lo_s = 0, wlen_s = 1;
    {
        int lo = 2 * span, hi2 = 0;
        const int c = maxT - 1, r = std::min(maxT - 1, 63);
        for (int dlt = -r; dlt <= r; ++dlt) {
            const int v = tab[c + dlt];
            lo = std::min(lo, std::min(clampi(v + span), clampi(-v + span)));
            hi2 = std::max(hi2, std::max(clampi(v + span), clampi(-v + span)));
        }
        win_lo_s = lo / 4 * 4;
        wlen_s = hi2 - win_lo_s + 1;
    }
    // ... and by a sequence of up to 128 tokens (|i - j| <= 127): the tiled fused kernel
    int win_lo_m = 0, wlen_m = 1;
    {
        int lo = 2 * span, hi2 = 0;
        const int c = maxT - 1, r = std::min(maxT - 1, 127);
        for (int dlt = -r; dlt <= r; ++dlt) {
            const int v = tab[c + dlt];
            lo = std::min(lo, std::min(clampi(v + span), clampi(-v + span)));
            hi2 = std::max(hi2, std::max(clampi(v + span), clampi(-v + span)));
        }
        win_lo_m = lo / 4 * 4;
        wlen_m = hi2 - win_lo_m + 1;
    }
    const int lds = round_up(maxT, 4);
    Plane X = arena_.plane(H, N), QKV = arena_.plane(3 * H, N), ctx = arena_.plane(H, N), A = arena_.plane(H, N);
    Plane Q = QKV.rows(0, H), Kp = QKV.rows(H, H), Vp = QKV.rows(2 * H, H);
    const int SP = bfs_parts_;
    // f32 path: F holds gelu(ffn1); split path: the FFN intermediate exists only as bf16 parts (its one reader is the next product)
    Plane F = SP ? Plane{} : arena_.plane(cfg_.inter, N);
    SplitPlanes Xs, As, Cs, Fs;
    if (SP) {
        Xs = alloc_split(arena_, SP, H, N);
        As = alloc_split(arena_, SP, H, N);
        Cs = alloc_split(arena_, SP, H, N);
        Fs = alloc_split(arena_, SP, cfg_.inter, N);
    }
    Plane E0{};   // the embedding output is ConvLayer's input (modeling_deberta_v2.py:664: self.conv(hidden_states, output_states, input_mask))
    if (cfg_.conv_k > 0) E0 = arena_.plane(H, N);
    std::vector<AttnGroup> ag_s, ag_m, ag_f, ag_l;   // short / mid / long fused, grouped-GEMM path
    std::vector<GemmGroup> g_st, g_c2p, g_p2c, g_pv;
    int64_t s_off = 0, c_off = 0, p_off = 0;
    int maxTL = 0, maxTF = 0;
    const int ldp = layers_[0].pos_k.ld;
    for (int u = 0; u < n; ++u) {
        const int T = L[u];
        const bool is_short = want_fused && deberta_attention_fits(T, wlen_s, d);
        const bool is_mid = want_fused && !is_short && deberta_attention128_fits(T, d);
        const bool is_long = want_fused && want_long && !is_short && !is_mid && deberta_attention_long_fits(T, d);
        if (is_long) maxTF = std::max(maxTF, T);
        else if (!is_short && !is_mid) maxTL = std::max(maxTL, T);
        for (int h = 0; h < nh; ++h) {
            AttnGroup a;
            a.qk_off = (int64_t)h * d * X.ld + lay.start[u];
            a.s_off = s_off;
            a.aux_off = c_off;
            a.aux2_off = p_off;
            a.T = T;
            a.lds = lds;
            a.col0 = lay.start[u];
            a.head = h;
            if (is_short) {
                ag_s.push_back(a);
                continue;
            }
            if (is_mid) {
                ag_m.push_back(a);
                continue;
            }
            if (is_long) {
                ag_f.push_back(a);
                continue;
            }
            ag_l.push_back(a);
            g_st.push_back(GemmGroup{a.qk_off, a.qk_off, s_off, 0, T, T, d, T});                                   // S^T = K^T Q
            g_c2p.push_back(GemmGroup{(int64_t)h * d * ldp + win_lo, a.qk_off, c_off, 0, wlen, T, d, T});          // posK^T Q
            g_p2c.push_back(GemmGroup{a.qk_off, (int64_t)h * d * ldp + win_lo, p_off, 0, T, wlen, d, wlen});       // K^T posQ
            g_pv.push_back(GemmGroup{(int64_t)lay.start[u] * H + h * d, s_off, a.qk_off, 0, d, T, T, T});          // V P^T
            s_off += (int64_t)T * lds;
            c_off += (int64_t)wlen * lds;
            p_off += (int64_t)T * win_ld;
        }
    }
    const int ngS = (int)ag_s.size(), ngM = (int)ag_m.size(), ngF = (int)ag_f.size(), ngL = (int)ag_l.size();
    float *S = nullptr, *C2P = nullptr, *P2C = nullptr, *VT = nullptr;
    AttnGroup *d_agS = nullptr, *d_agM = nullptr, *d_agF = nullptr, *d_agL = nullptr;
    if (ngF) {
        d_agF = arena_.array<AttnGroup>(ngF);
        arena_.upload(d_agF, ag_f.data(), sizeof(AttnGroup) * ngF, stream_);
    }
    if (ngM) {
        d_agM = arena_.array<AttnGroup>(ngM);
        arena_.upload(d_agM, ag_m.data(), sizeof(AttnGroup) * ngM, stream_);
    }
    GemmGroup* d_g = nullptr;
    if (ngS) {
        d_agS = arena_.array<AttnGroup>(ngS);
        arena_.upload(d_agS, ag_s.data(), sizeof(AttnGroup) * ngS, stream_);
    }
    if (ngL) {
        S = arena_.array<float>((size_t)s_off);
        C2P = arena_.array<float>((size_t)c_off);
        P2C = arena_.array<float>((size_t)p_off);
        VT = arena_.array<float>((size_t)N * H);
        d_agL = arena_.array<AttnGroup>(ngL);
        d_g = arena_.array<GemmGroup>((size_t)4 * ngL);
        arena_.upload(d_agL, ag_l.data(), sizeof(AttnGroup) * ngL, stream_);
        arena_.upload(d_g, g_st.data(), sizeof(GemmGroup) * ngL, stream_);
        arena_.upload(d_g + ngL, g_c2p.data(), sizeof(GemmGroup) * ngL, stream_);
        arena_.upload(d_g + 2 * ngL, g_p2c.data(), sizeof(GemmGroup) * ngL, stream_);
        arena_.upload(d_g + 3 * ngL, g_pv.data(), sizeof(GemmGroup) * ngL, stream_);
    }

    const float inv_scale = 1.0f / std::sqrt((float)d * 3.0f);  // c2p + p2c => scale_factor 3 (:226-232)

    arena_.end_uploads();
    deberta_embed_ln(d_ids, emb_, H, emb_g_, emb_b_, cfg_.eps, X, stream_);
    fill_zero(ctx.p, sizeof(float) * (size_t)H * ctx.ld, stream_);  // alignment-gap columns are never written by P.V
    if (mask) {
        // embeddings * mask (:550-559): tokens with attention_mask 0 become zero columns (a column gather with -1 entries)
        std::vector<int> map(N, -1);
        int64_t e = 0;
        for (int i = 0; i < n; ++i)
            for (int t = 0; t < L[i]; ++t, ++e)
                if (mask[e]) map[lay.start[i] + t] = lay.start[i] + t;
        int* d_map = arena_.array<int>(N);
        arena_.upload(d_map, map.data(), sizeof(int) * N, stream_);
        gather_cols(X, d_map, A, stream_);
        std::swap(X, A);
    }

    if (cfg_.conv_k > 0) HIP_CHECK(hipMemcpyAsync(E0.p, X.p, sizeof(float) * (size_t)H * X.ld, hipMemcpyDeviceToDevice, stream_));

    auto grouped = [&](const float* Aop, int lda, const float* Bop, int ldb, float* Cop, int ldc, const GemmGroup* grp, int maxM, int maxN,
                       float alpha, double flops) {
        ConvParams p;
        p.A = Aop;
        p.lda = lda;
        p.B = Bop;
        p.ldb = ldb;
        p.C = Cop;
        p.ldc = ldc;
        p.alpha = alpha;
        p.groups = grp;
        p.ngroups = ngL;
        p.maxM = maxM;
        p.maxN = maxN;
        p.flops_hint = flops;
        launch_conv(p, stream_);
    };
    double fl_tt = 0, fl_tw = 0;  // algorithmic FLOP of the grouped products (profiling only)
    for (const AttnGroup& a : ag_l) {
        fl_tt += 2.0 * (double)a.T * a.T * d;
        fl_tw += 2.0 * (double)a.T * wlen * d;
    }

    // scratch for the K split of small grids (gemm_bfs.hip: a single utterance's products; larger grids ignore it)
    BfsSplitK sk;
    if (SP) {
        sk.ws = static_cast<float*>(arena_.alloc(kSkWsBytes));
        sk.ws_bytes = kSkWsBytes;
        sk.counters = sk_counters_;
        sk.ncounters = kSkCounters;
    }
    if (SP) split_planes(X, Xs, stream_);
    for (int li = 0; li < cfg_.layers; ++li) {
        const Layer& Ly = layers_[li];
        // q | k | v in one product (k-major planes)
        if (SP) conv_bfs(Ly.qkv, Xs, &QKV, nullptr, nullptr, 1, stream_, ACT_NONE, nullptr, 1.0f, 1.0f, -1, 0, &sk);
        else conv_plain(Ly.qkv, X, QKV, 1, 0, nullptr, 1, stream_);
        if (ngS)
            deberta_attention(d_agS, ngS, Q.p, Kp.p, QKV.ld, Vp.p, Ly.pos_k.p, Ly.pos_q.p, ldp, win_lo_s, wlen_s, d_tab, maxT - 1, span, inv_scale,
                              lay.d_mask, d, ctx.p, ctx.ld, stream_);
        if (ngM)
            deberta_attention128(d_agM, ngM, Q.p, Kp.p, QKV.ld, Vp.p, Ly.pos_k.p, Ly.pos_q.p, ldp, win_lo_m, wlen_m, d_tab, maxT - 1, span, inv_scale,
                                 lay.d_mask, d, ctx.p, ctx.ld, stream_);
        if (ngF)
            deberta_attention_long(d_agF, ngF, maxTF, Q.p, Kp.p, QKV.ld, Vp.p, Ly.pos_k.p, Ly.pos_q.p, ldp, win_lo, wlen, d_tab, maxT - 1, span,
                                   inv_scale, lay.d_mask, d, ctx.p, ctx.ld, stream_);
        if (ngL) {
            linear_tokmajor(Ly.v, X, VT, H, stream_);   // the grouped V P^T product wants V token-major
            grouped(Kp.p, Kp.ld, Q.p, Q.ld, S, lds, d_g, maxTL, maxTL, inv_scale, fl_tt);
            grouped(Ly.pos_k.p, ldp, Q.p, Q.ld, C2P, lds, d_g + ngL, wlen, maxTL, 1.0f, fl_tw);
            grouped(Kp.p, Kp.ld, Ly.pos_q.p, ldp, P2C, win_ld, d_g + 2 * ngL, maxTL, wlen, 1.0f, fl_tw);
            deberta_softmax(d_agL, ngL, maxTL, S, C2P, P2C, d_tab, maxT - 1, span, win_lo, win_ld, inv_scale, lay.d_mask, stream_);
            grouped(VT, H, S, lds, ctx.p, ctx.ld, d_g + 3 * ngL, d, maxTL, 1.0f, fl_tt);
        }
        if (SP) {
            split_planes(ctx, Cs, stream_);
            conv_bfs(Ly.o, Cs, &A, nullptr, nullptr, 1, stream_, ACT_NONE, &X, 1.0f, 1.0f, -1, 0, &sk);
            layernorm_ch(A, A, Ly.ln1_g, Ly.ln1_b, cfg_.eps, ACT_NONE, nullptr, 0, d_valid, stream_, &As);
            conv_bfs(Ly.ffn1, As, nullptr, &Fs, nullptr, 1, stream_, ACT_GELU, nullptr, 1.0f, 1.0f, -1, 0, &sk);
            conv_bfs(Ly.ffn2, Fs, &X, nullptr, nullptr, 1, stream_, ACT_NONE, &A, 1.0f, 1.0f, -1, 0, &sk);
        } else {
            conv_plain(Ly.o, ctx, A, 1, 0, nullptr, 1, stream_, ACT_NONE, 1.0f, &X);
            layernorm_ch(A, A, Ly.ln1_g, Ly.ln1_b, cfg_.eps, ACT_NONE, nullptr, 0, d_valid, stream_);
            conv_plain(Ly.ffn1, A, F, 1, 0, nullptr, 1, stream_, ACT_GELU);
            conv_plain(Ly.ffn2, F, X, 1, 0, nullptr, 1, stream_, ACT_NONE, 1.0f, &A);
        }
        const bool conv_next = li == 0 && cfg_.conv_k > 0;
        layernorm_ch(X, X, Ly.ln2_g, Ly.ln2_b, cfg_.eps, ACT_NONE, nullptr, 0, d_valid, stream_, (SP && !conv_next) ? &Xs : nullptr);
        if (conv_next) {
            // ConvLayer (:461-470): out = act(conv(embeddings), zeroed where attention_mask == 0); x = LayerNorm(layer0 + out) * mask.
            // act(0) == 0 for gelu / tanh / relu, so zeroing the sum's masked columns before the LayerNorm and again after it (lay.d_mask
            // = attention mask) gives the same zeros the reference's output * input_mask produces.  (k = 3: stays on the f32 / conv_cl path.)
            conv_plain(conv_, E0, A, 1, cfg_.conv_k / 2, nullptr, 1, stream_, cfg_.conv_act, 1.0f, &X);
            layernorm_ch(A, X, conv_g_, conv_b_, cfg_.eps, ACT_NONE, nullptr, 0, lay.d_mask, stream_, SP ? &Xs : nullptr);
        }
    }
    out_ = X;
}

void BertModel::copy_out(float* host) {
    HIP_CHECK(hipSetDevice(device_));
    const int H = cfg_.hidden;
    int64_t total = 0;
    for (int v : layout_.len) total += v;
    float* d_out = arena_.array<float>((size_t)total * H);
    int64_t off = 0;
    for (int i = 0; i < layout_.n; ++i) {
        transpose_out(out_, layout_.start[i], layout_.len[i], d_out + off * H, stream_);
        off += layout_.len[i];
    }
    HIP_CHECK(hipMemcpyAsync(host, d_out, sizeof(float) * (size_t)total * H, hipMemcpyDeviceToHost, stream_));
    sat_watch_.enqueue(stream_);
    HIP_CHECK(hipStreamSynchronize(stream_));
    sat_watch_.check("DeBERTa, sbv2_bert_predict");
}

}  // namespace sbv2
