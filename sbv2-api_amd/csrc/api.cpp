// extern "C" boundary of libsbv2_hip.so (see include/sbv2_hip.h for what each entry point replaces).
#include <cstdlib>
#include <cstring>
#include <new>


#include "api_internal.h"

VitsBatch to_batch(const sbv2_batch* b) {
    SBV2_REQUIRE(b && b->n >= 1 && b->t_lens && b->x_tst && b->tones && b->lang_ids && b->sids && b->style_vectors,
                 "sbv2_batch has null fields");
    VitsBatch v;
    v.n = (int)b->n;
    v.t_lens = b->t_lens;
    v.phones = b->x_tst;
    v.tones = b->tones;
    v.langs = b->lang_ids;
    v.sids = b->sids;
    v.styles = b->style_vectors;
    v.bert_host = b->bert;
    v.sdp_ratio = b->sdp_ratio;
    v.length_scale = b->length_scale;
    v.noise_scale = b->noise_scale;
    v.noise_scale_w = b->noise_scale_w;
    v.seed = b->noise_seed;
    v.forced_durations = b->forced_durations;
    return v;
}

namespace {
struct DevBuf {
    float* p = nullptr;
    explicit DevBuf(size_t n) { HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&p), sizeof(float) * std::max<size_t>(n, 4))); }
    ~DevBuf() { (void)hipFree(p); }
};
Blob one_conv_blob(const float* w, const float* bias, std::vector<int64_t> dims, int64_t nbias) {
    Blob b;
    b.kind = 0;
    HostTensor t;
    t.dims = std::move(dims);
    t.data = w;
    b.tensors.emplace("c.weight", t);
    if (bias) {
        HostTensor tb;
        tb.dims = {nbias};
        tb.data = bias;
        b.tensors.emplace("c.bias", tb);
    }
    return b;
}
}  // namespace

extern "C" {

const char* sbv2_last_error(void) { return last_error_cstr(); }

int sbv2_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int sbv2_bert_create(const uint8_t* model, size_t model_len, int device, sbv2_bert** out) {
    API_BEGIN
    SBV2_REQUIRE(out, "null output handle");
    Blob blob = load_model_bytes(model, model_len, 1);
    std::unique_ptr<sbv2_bert> h(new sbv2_bert);
    h->m.reset(new BertModel(blob, device));
    *out = h.release();
    API_END
}
void sbv2_bert_destroy(sbv2_bert* h) { delete h; }
int64_t sbv2_bert_hidden(const sbv2_bert* h) { return h ? h->m->cfg().hidden : 0; }
int sbv2_bert_gemm_parts(const sbv2_bert* h) { return h ? h->m->gemm_parts() : -1; }

int sbv2_bert_predict_batch(sbv2_bert* h, int64_t n, const int64_t* token_ids, const int64_t* attention_mask, const int64_t* lens,
                            float* out) {
    API_BEGIN
    SBV2_REQUIRE(h && token_ids && lens && out && n >= 1, "bad arguments");
    h->m->forward((int)n, token_ids, attention_mask, lens);
    h->m->copy_out(out);
    API_END
}
int sbv2_bert_predict(sbv2_bert* h, const int64_t* token_ids, const int64_t* attention_mask, int64_t S, float* out) {
    return sbv2_bert_predict_batch(h, 1, token_ids, attention_mask, &S, out);
}

int sbv2_vits_create(const uint8_t* model, size_t model_len, int device, sbv2_vits** out) {
    API_BEGIN
    SBV2_REQUIRE(out, "null output handle");
    Blob blob = load_model_bytes(model, model_len, 2);
    std::unique_ptr<sbv2_vits> h(new sbv2_vits);
    h->m.reset(new VitsModel(blob, device));
    *out = h.release();
    API_END
}
void sbv2_vits_destroy(sbv2_vits* h) { delete h; }
int64_t sbv2_vits_hop(const sbv2_vits* h) { return h ? h->m->cfg().hop() : 0; }
int64_t sbv2_vits_bert_dim(const sbv2_vits* h) { return h ? h->m->cfg().bert_dim : 0; }
int64_t sbv2_vits_style_dim(const sbv2_vits* h) { return h ? h->m->cfg().style_dim : 0; }
int sbv2_vits_decoder_mode(const sbv2_vits* h) { return h ? h->m->decoder_mode() : -1; }
int64_t sbv2_vits_workspace_bytes(const sbv2_vits* h) { return h ? (int64_t)h->m->workspace_bytes() : -1; }

int sbv2_vits_synthesize_batch(sbv2_vits* h, const sbv2_batch* batch, int64_t* pcm_lens) {
    API_BEGIN
    SBV2_REQUIRE(h && pcm_lens, "bad arguments");
    VitsBatch v = to_batch(batch);
    SBV2_REQUIRE(v.bert_host, "sbv2_batch.bert is required here");
    h->m->forward(v);
    for (int i = 0; i < v.n; ++i) pcm_lens[i] = h->m->pcm_lens()[i];
    API_END
}
int sbv2_vits_fetch_pcm(sbv2_vits* h, float* pcm, int64_t capacity) {
    API_BEGIN
    SBV2_REQUIRE(h && pcm, "bad arguments");
    SBV2_REQUIRE(capacity >= h->m->pcm_total(), "PCM buffer too small: " + std::to_string(capacity) + " < " + std::to_string(h->m->pcm_total()));
    h->m->copy_pcm(pcm);
    API_END
}
const float* sbv2_vits_pcm_device(sbv2_vits* h, int64_t* total) {
    if (!h) return nullptr;
    if (total) *total = h->m->pcm_total();
    return h->m->pcm_device();
}
int sbv2_vits_copy_pcm_device(sbv2_vits* h, void* dst_device, int64_t capacity) {
    API_BEGIN
    SBV2_REQUIRE(h && dst_device, "bad arguments");
    SBV2_REQUIRE(capacity >= h->m->pcm_total(), "PCM buffer too small");
    HIP_CHECK(hipSetDevice(h->m->device()));
    HIP_CHECK(hipMemcpyAsync(dst_device, h->m->pcm_device(), sizeof(float) * (size_t)h->m->pcm_total(), hipMemcpyDeviceToDevice,
                             h->m->stream()));
    HIP_CHECK(hipStreamSynchronize(h->m->stream()));
    API_END
}
int sbv2_sync(sbv2_vits* h) {
    API_BEGIN
    SBV2_REQUIRE(h, "bad arguments");
    HIP_CHECK(hipSetDevice(h->m->device()));
    HIP_CHECK(hipStreamSynchronize(h->m->stream()));
    API_END
}
int sbv2_prof_begin(void) {
    API_BEGIN
    conv_prof_begin();
    API_END
}
int sbv2_prof_end(char* json, int64_t cap) {
    API_BEGIN
    const std::string s = conv_prof_end();
    SBV2_REQUIRE(json && (int64_t)s.size() + 1 <= cap, "profile buffer too small");
    std::memcpy(json, s.c_str(), s.size() + 1);
    API_END
}
int sbv2_vits_fetch_durations(sbv2_vits* h, int64_t* durations, float* logw, int64_t capacity) {
    API_BEGIN
    SBV2_REQUIRE(h, "bad arguments");
    const auto& d = h->m->durations();
    const auto& l = h->m->logw();
    SBV2_REQUIRE(capacity >= (int64_t)d.size(), "duration buffer too small");
    if (durations)
        for (size_t i = 0; i < d.size(); ++i) durations[i] = d[i];
    if (logw) std::memcpy(logw, l.data(), sizeof(float) * l.size());
    API_END
}
int sbv2_vits_set_trace(sbv2_vits* h, int on) {
    API_BEGIN
    SBV2_REQUIRE(h, "bad arguments");
    h->m->set_trace(on != 0);
    API_END
}
int sbv2_vits_get_trace(sbv2_vits* h, const char* name, int64_t utt, float* out, int64_t cap, int64_t* rows, int64_t* cols) {
    API_BEGIN
    SBV2_REQUIRE(h && name && rows && cols, "bad arguments");
    std::vector<float> v;
    int r = 0, c = 0;
    SBV2_REQUIRE(h->m->get_trace(name, (int)utt, v, r, c), std::string("no trace named ") + name);
    *rows = r;
    *cols = c;
    if (out) {
        SBV2_REQUIRE((int64_t)v.size() <= cap, "trace buffer too small");
        std::memcpy(out, v.data(), sizeof(float) * v.size());
    }
    API_END
}

int sbv2_vits_synthesize(sbv2_vits* h, const float* bert, const int64_t* x_tst, const int64_t* tones, const int64_t* lang_ids, int64_t T,
                         int64_t sid, const float* style_vector, float sdp_ratio, float length_scale, float noise_scale,
                         float noise_scale_w, uint64_t noise_seed, float** pcm, int64_t* pcm_len) {
    API_BEGIN
    SBV2_REQUIRE(h && bert && pcm && pcm_len, "bad arguments");
    sbv2_batch b;
    std::memset(&b, 0, sizeof(b));
    b.n = 1;
    b.t_lens = &T;
    b.x_tst = x_tst;
    b.tones = tones;
    b.lang_ids = lang_ids;
    b.sids = &sid;
    b.style_vectors = style_vector;
    b.bert = bert;
    b.sdp_ratio = sdp_ratio;
    b.length_scale = length_scale;
    b.noise_scale = noise_scale;
    b.noise_scale_w = noise_scale_w;
    b.noise_seed = noise_seed;
    h->m->forward(to_batch(&b));
    const int64_t n = h->m->pcm_total();
    float* buf = static_cast<float*>(std::malloc(sizeof(float) * (size_t)std::max<int64_t>(n, 1)));
    SBV2_REQUIRE(buf, "out of host memory");
    try {
        h->m->copy_pcm(buf);
    } catch (...) {
        std::free(buf);
        throw;
    }
    *pcm = buf;
    *pcm_len = n;
    API_END
}
void sbv2_pcm_free(float* pcm) { std::free(pcm); }

int sbv2_pipeline_create(sbv2_bert* bert, sbv2_vits* vits, sbv2_pipeline** out) {
    API_BEGIN
    SBV2_REQUIRE(bert && vits && out, "bad arguments");
    SBV2_REQUIRE(bert->m->device() == vits->m->device(), "bert and vits handles live on different devices");
    SBV2_REQUIRE(bert->m->cfg().hidden == vits->m->cfg().bert_dim, "DeBERTa hidden size does not match the VITS bert_proj input");
    std::unique_ptr<sbv2_pipeline> p(new sbv2_pipeline);
    p->bert = bert;
    p->vits = vits;
    // execution contexts = pipeline depth across calls (SBV2_PIPELINE_DEPTH, default 2).  (Cutting ONE batch into micro-batches was
    // measured too: the latency-bound chains then repeat per micro-batch and the step gets slower, 220 vs 186 ms with 4 cuts.)
    int k = 2;
    if (const char* e = getenv("SBV2_PIPELINE_DEPTH")) k = std::max(1, std::min(8, atoi(e)));
    for (int i = 1; i < k; ++i) {
        p->bclones.emplace_back(bert->m->clone());
        p->vclones.emplace_back(vits->m->clone());
    }
    *out = p.release();
    API_END
}
void sbv2_pipeline_destroy(sbv2_pipeline* p) { delete p; }

}  // extern "C"

// One micro-batch on one context: bert::predict -> word2ph repeat (tts_util.rs:129-154) -> model::synthesize
void pipeline_run_one(BertModel& bm, VitsModel& vm, VitsBatch v, const int64_t* token_ids, const int64_t* s_lens,
                      const int64_t* word2ph) {
    HIP_CHECK(hipStreamSynchronize(vm.stream()));  // this context's previous batch may still be reading the DeBERTa output plane
    bm.forward(v.n, token_ids, nullptr, s_lens);
    const SegLayout& bl = bm.layout();
    std::vector<int> map;
    int64_t e = 0;
    for (int u = 0; u < v.n; ++u) {
        SBV2_REQUIRE(s_lens[u] == bl.len[u], "internal: layout mismatch");
        int64_t cnt = 0;
        for (int64_t i = 0; i < s_lens[u]; ++i, ++e) {
            SBV2_REQUIRE(word2ph[e] >= 0, "negative word2ph");
            for (int64_t r = 0; r < word2ph[e]; ++r) map.push_back(bl.start[u] + (int)i);
            cnt += word2ph[e];
        }
        SBV2_REQUIRE(cnt == v.t_lens[u], "sum(word2ph) must equal the text length (tts_util.rs:122-127)");
    }
    v.bert_host = nullptr;
    v.bert_dev = &bm.out();
    v.bert_map = map.data();
    // the two models run on their own streams: an event dependency inside forward(), not a host wait
    v.after_stream = bm.stream();
    vm.forward(v);
}

extern "C" {

// Batches are PIPELINED ACROSS CALLS: call n runs on execution context n % depth (own HIP stream + workspace, shared weights), so the
// latency-bound part of batch n+1 (DeBERTa, text encoder, duration predictors, flow: ~1300 small launches, ~55 ms whatever the
// batch size) executes beside the throughput-bound HiFi-GAN kernels of batch n.  A call returns once its kernels are enqueued (the
// host only waits for the batch's own integer durations); results are collected with sbv2_pipeline_wait / _fetch_pcm by ticket.
int sbv2_pipeline_run(sbv2_pipeline* p, const sbv2_batch* batch, const int64_t* token_ids, const int64_t* s_lens, const int64_t* word2ph,
                      int64_t* pcm_lens) {
    API_BEGIN
    SBV2_REQUIRE(p && token_ids && s_lens && word2ph && pcm_lens, "bad arguments");
    const VitsBatch v = to_batch(batch);
    const int ctx = (int)(p->calls % p->contexts());
    ++p->calls;
    pipeline_run_one(p->bm(ctx), p->vm(ctx), v, token_ids, s_lens, word2ph);
    for (int i = 0; i < v.n; ++i) pcm_lens[i] = p->vm(ctx).pcm_lens()[i];
    API_END
}

int64_t sbv2_pipeline_last_ticket(sbv2_pipeline* p) { return p ? p->calls : -1; }

int sbv2_pipeline_wait(sbv2_pipeline* p, int64_t ticket) {
    API_BEGIN
    SBV2_REQUIRE(p, "bad arguments");
    const int ctx = p->ctx_of(ticket);
    HIP_CHECK(hipSetDevice(p->vits->m->device()));
    HIP_CHECK(hipStreamSynchronize(p->vm(ctx).stream()));
    API_END
}

int sbv2_pipeline_sync(sbv2_pipeline* p) {
    API_BEGIN
    SBV2_REQUIRE(p, "bad arguments");
    HIP_CHECK(hipSetDevice(p->vits->m->device()));
    for (int j = 0; j < p->contexts(); ++j) HIP_CHECK(hipStreamSynchronize(p->vm(j).stream()));
    API_END
}

// Concatenated PCM of the run with this ticket (utterance order); dst_is_device != 0: dst is device memory (the RCCL send buffer).
// capacity = samples dst can hold: a run whose PCM is longer is refused instead of overflowing the buffer.
int sbv2_pipeline_fetch_pcm_ticket(sbv2_pipeline* p, int64_t ticket, float* dst, int64_t capacity, int dst_is_device) {
    API_BEGIN
    SBV2_REQUIRE(p && dst, "bad arguments");
    VitsModel& vm = p->vm(p->ctx_of(ticket));
    SBV2_REQUIRE(capacity >= vm.pcm_total(), "PCM buffer too small: " + std::to_string(capacity) + " < " + std::to_string(vm.pcm_total()));
    HIP_CHECK(hipSetDevice(vm.device()));
    HIP_CHECK(hipMemcpyAsync(dst, vm.pcm_device(), sizeof(float) * (size_t)vm.pcm_total(),
                             dst_is_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, vm.stream()));
    HIP_CHECK(hipStreamSynchronize(vm.stream()));
    API_END
}
int sbv2_pipeline_fetch_pcm(sbv2_pipeline* p, float* dst, int64_t capacity, int dst_is_device) {
    return sbv2_pipeline_fetch_pcm_ticket(p, p ? p->calls : -1, dst, capacity, dst_is_device);
}

// Pinned host memory for PCM destinations: a device -> host copy into pageable memory is staged by the runtime at a fraction of the
// PCIe rate; into these buffers it is one DMA that overlaps the other execution context's kernels.
void* sbv2_host_alloc(size_t bytes) {
    void* p = nullptr;
    if (hipHostMalloc(&p, std::max<size_t>(bytes, 64), hipHostMallocDefault) != hipSuccess) {
        set_last_error("hipHostMalloc failed");
        return nullptr;
    }
    return p;
}
void sbv2_host_free(void* p) {
    if (p) (void)hipHostFree(p);
}

int sbv2_debug_bucket_table(int64_t max_s, int64_t buckets, int64_t max_rel, int32_t* out) {
    API_BEGIN
    SBV2_REQUIRE(max_s >= 1 && out, "bad arguments");
    const std::vector<int> t = BertModel::bucket_table((int)max_s, (int)buckets, (int)max_rel);
    for (size_t i = 0; i < t.size(); ++i) out[i] = t[i];
    API_END
}


int sbv2_debug_conv1d(int device, const float* x, const float* w, const float* bias, int64_t cin, int64_t cout, int64_t k, int64_t L,
                      int64_t dilation, float pre_slope, float* y) {
    API_BEGIN
    HIP_CHECK(hipSetDevice(device));
    Blob b = one_conv_blob(w, bias, {cout, cin, k}, cout);
    WeightStore ws(b);
    PackedConv pc = ws.conv("c");
    Plane X{nullptr, (int)cin, (int)L, round_up((int)L, 64)}, Y{nullptr, (int)cout, (int)L, round_up((int)L, 64)};
    DevBuf dx((size_t)cin * X.ld), dy((size_t)cout * Y.ld);
    X.p = dx.p;
    Y.p = dy.p;
    HIP_CHECK(hipMemcpy2D(X.p, sizeof(float) * X.ld, x, sizeof(float) * L, sizeof(float) * L, cin, hipMemcpyHostToDevice));
    conv_plain(pc, X, Y, (int)dilation, (int)(dilation * (k - 1) / 2), nullptr, 1, nullptr, ACT_NONE, pre_slope);
    HIP_CHECK(hipDeviceSynchronize());
    HIP_CHECK(hipMemcpy2D(y, sizeof(float) * L, Y.p, sizeof(float) * Y.ld, sizeof(float) * L, cout, hipMemcpyDeviceToHost));
    API_END
}

int sbv2_debug_conv_transpose1d(int device, const float* x, const float* w, const float* bias, int64_t cin, int64_t cout, int64_t k,
                                int64_t L, int64_t stride, int64_t padding, float pre_slope, float* y) {
    API_BEGIN
    HIP_CHECK(hipSetDevice(device));
    SBV2_REQUIRE(bias, "bias required");
    Blob b = one_conv_blob(w, bias, {cin, cout, k}, cout);
    WeightStore ws(b);
    PackedUpsample up = ws.upsample("c", (int)stride, (int)padding);
    const int Lo = (int)(L * stride);
    Plane X{nullptr, (int)cin, (int)L, round_up((int)L, 64)}, Y{nullptr, (int)cout, Lo, round_up(Lo, 64)};
    DevBuf dx((size_t)cin * X.ld), dy((size_t)cout * Y.ld);
    X.p = dx.p;
    Y.p = dy.p;
    HIP_CHECK(hipMemcpy2D(X.p, sizeof(float) * X.ld, x, sizeof(float) * L, sizeof(float) * L, cin, hipMemcpyHostToDevice));
    for (const auto& g : up.groups) {
        ConvParams p;
        p.A = g.w;
        p.lda = g.lda;
        p.a_tap_stride = (int64_t)up.cin * g.lda;
        p.B = X.p;
        p.ldb = X.ld;
        p.nb = X.L;
        p.C = Y.p;
        p.ldc = Y.ld;
        p.M = g.nph * up.cout;
        p.N = X.L;
        p.K = up.cin;
        p.ntaps = g.ntaps;
        for (int t = 0; t < g.ntaps; ++t) p.shift[t] = g.shift[t];
        p.bias = up.bias;
        p.bias_mode = BIAS_ROW;
        p.pre_slope = pre_slope;
        p.out_stride = (int)stride;
        p.phase_rows = up.cout;
        for (int q = 0; q < kMaxPhases; ++q) p.phase_off[q] = g.phase_off[q];
        launch_conv(p, nullptr);
    }
    HIP_CHECK(hipDeviceSynchronize());
    HIP_CHECK(hipMemcpy2D(y, sizeof(float) * Lo, Y.p, sizeof(float) * Y.ld, sizeof(float) * Lo, cout, hipMemcpyDeviceToHost));
    API_END
}

int sbv2_debug_set_upx(int on) { return set_upx(on); }
int sbv2_debug_conv_transpose1d_clx(int device, const float* x, const float* w, const float* bias, int64_t cin, int64_t cout, int64_t k, int64_t L,
                                    int64_t stride, float pre_slope, const uint8_t* mask, int64_t mask_div, int64_t iters, float* y, float* ys_sum, float* ms) {
    API_BEGIN
    HIP_CHECK(hipSetDevice(device));
    SBV2_REQUIRE(x && w && bias && y && mask_div >= 1 && (mask_div & (mask_div - 1)) == 0, "bad arguments");
    Blob b = one_conv_blob(w, bias, {cin, cout, k}, cout);
    WeightStore ws(b);
    ClUpX u = build_upx(ws, w, bias, (int)cin, (int)cout, (int)k, (int)stride, /*parts_out=*/ys_sum != nullptr);   // (sbv2_debug_set_upx(2): plain row order)
    SBV2_REQUIRE(u.wx, "shape not supported by the phased conv_clx transposed convolution");
    const int64_t Lo = L * stride;
    std::vector<float> xt((size_t)L * cin), yt((size_t)Lo * cout);
    for (int64_t ci = 0; ci < cin; ++ci)
        for (int64_t n = 0; n < L; ++n) xt[(size_t)n * cin + ci] = x[(size_t)ci * L + n];
    DevBuf dx(xt.size()), dy(yt.size());
    HIP_CHECK(hipMemcpy(dx.p, xt.data(), sizeof(float) * xt.size(), hipMemcpyHostToDevice));
    HIP_CHECK(hipMemset(dy.p, 0xFF, sizeof(float) * yt.size()));   // (NaN: every output row must be written)
    DevBuf dxs(split_cl_bytes((int)cin, L) / 4 + 4), dys(split_cl_bytes((int)cout, Lo) / 4 + 4);
    SplitClPlanes xs = make_split_cl(dxs.p, (int)cin, L, nullptr), ysp = make_split_cl(dys.p, (int)cout, Lo, nullptr);
    split_cl(dx.p, (int)cin, L, (int)cin, pre_slope, xs, nullptr);
    unsigned char* dm = nullptr;
    int shift = 0;
    while ((1 << shift) < mask_div) ++shift;
    if (mask) {
        const size_t nm = (size_t)((L + mask_div - 1) / mask_div);
        HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&dm), nm));
        HIP_CHECK(hipMemcpy(dm, mask, nm, hipMemcpyHostToDevice));
    }
    ConvClxParams p;
    p.X = xs;
    p.W = u.wx;
    p.nmt = u.M / 32;
    p.M = u.M;
    p.N = (int)L;
    p.K = (int)cin;
    p.ntaps = u.ntaps;
    p.shift0 = u.shift0;
    p.shift_step = -1;
    p.Y = dy.p;
    p.ldy = (int)cout;
    if (ys_sum) {
        p.Ys = ysp;
        p.ys_slope = 0.1f;
    }
    p.bias = u.bias;
    p.mask = dm;                 // indexed by the INPUT position >> shift
    p.mask_shift = shift;
    p.out_stride = (int)stride;
    p.phase_rows = (int)cout;
    p.phase_group = u.group;
    for (int q = 0; q < kMaxPhases; ++q) {
        p.phase_off[q] = u.phase_off[q];
        p.phase_tap0[q] = u.phase_tap0[q];
    }
    try {
        SBV2_REQUIRE(conv_clx_usable(p), "shape not supported by conv_clx");
        launch_conv_clx(p, nullptr);
        HIP_CHECK(hipDeviceSynchronize());
        if (iters > 0 && ms) {
            hipEvent_t e0, e1;
            HIP_CHECK(hipEventCreate(&e0));
            HIP_CHECK(hipEventCreate(&e1));
            HIP_CHECK(hipEventRecord(e0, nullptr));
            for (int i = 0; i < iters; ++i) launch_conv_clx(p, nullptr);
            HIP_CHECK(hipEventRecord(e1, nullptr));
            HIP_CHECK(hipEventSynchronize(e1));
            float t = 0.f;
            HIP_CHECK(hipEventElapsedTime(&t, e0, e1));
            *ms = t / (float)iters;
            (void)hipEventDestroy(e0);
            (void)hipEventDestroy(e1);
        }
    } catch (...) {
        if (dm) (void)hipFree(dm);
        throw;
    }
    if (dm) (void)hipFree(dm);
    HIP_CHECK(hipMemcpy(yt.data(), dy.p, sizeof(float) * yt.size(), hipMemcpyDeviceToHost));
    for (int64_t co = 0; co < cout; ++co)
        for (int64_t n = 0; n < Lo; ++n) y[(size_t)co * Lo + n] = yt[(size_t)n * cout + co];
    if (ys_sum) {
        const int64_t rows = kClxFront + Lo + kClxBack;
        std::vector<uint16_t> hs(split_cl_bytes((int)cout, Lo) / 2);
        HIP_CHECK(hipMemcpy(hs.data(), ysp.p, hs.size() * 2, hipMemcpyDeviceToHost));
        auto f = [](uint16_t h) {
            const uint32_t v = (uint32_t)h << 16;
            float o;
            std::memcpy(&o, &v, 4);
            return o;
        };
        for (int64_t co = 0; co < cout; ++co)
            for (int64_t n = 0; n < Lo; ++n) {
                const size_t base = ((size_t)(co >> 4) * 2 * rows + kClxFront + n) * 16 + (co & 15);
                ys_sum[(size_t)co * Lo + n] = f(hs[base]) + f(hs[base + (size_t)rows * 16]);
            }
    }
    API_END
}

int sbv2_debug_conv1d_cl(int device, const float* x, const float* w, const float* bias, int64_t cin, int64_t cout, int64_t k, int64_t L,
                         int64_t dilation, float pre_slope, int mode, int64_t iters, float* y, float* ms) {
    API_BEGIN
    HIP_CHECK(hipSetDevice(device));
    SBV2_REQUIRE(mode >= 1 && mode <= 3, "mode: 1 = split-bf16, 2 = bf16, 3 = f16");
    Blob b = one_conv_blob(w, bias, {cout, cin, k}, cout);
    WeightStore ws(b);
    ClConv c = pack_cl(ws, w, (int)cout, (int)cin, (int)k, mode == 1 ? 2 : (mode == 2 ? 1 : 3), bias);
    std::vector<float> xt((size_t)L * cin), yt((size_t)L * cout);
    for (int64_t ci = 0; ci < cin; ++ci)
        for (int64_t n = 0; n < L; ++n) xt[(size_t)n * cin + ci] = x[(size_t)ci * L + n];
    DevBuf dx(xt.size()), dy(yt.size());
    HIP_CHECK(hipMemcpy(dx.p, xt.data(), sizeof(float) * xt.size(), hipMemcpyHostToDevice));
    ConvClParams p;
    p.X = dx.p;
    p.ldx = (int)cin;
    p.NB = (int)L;
    p.W = c.w;
    p.nmt = c.nmt;
    p.tm = c.tm;
    p.split = mode == 1;
    p.f16 = mode == 3;
    p.M = (int)cout;
    p.N = (int)L;
    p.K = (int)cin;
    p.ntaps = (int)k;
    for (int j = 0; j < k; ++j) p.shift[j] = (int)(j * dilation - dilation * (k - 1) / 2);
    p.Y = dy.p;
    p.ldy = (int)cout;
    p.bias = c.bias;
    p.pre_slope = pre_slope;
    launch_conv_cl(p, nullptr);
    HIP_CHECK(hipDeviceSynchronize());
    if (iters > 0 && ms) {
        hipEvent_t e0, e1;
        HIP_CHECK(hipEventCreate(&e0));
        HIP_CHECK(hipEventCreate(&e1));
        HIP_CHECK(hipEventRecord(e0, nullptr));
        for (int i = 0; i < iters; ++i) launch_conv_cl(p, nullptr);
        HIP_CHECK(hipEventRecord(e1, nullptr));
        HIP_CHECK(hipEventSynchronize(e1));
        float t = 0.f;
        HIP_CHECK(hipEventElapsedTime(&t, e0, e1));
        *ms = t / (float)iters;
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
    }
    HIP_CHECK(hipMemcpy(yt.data(), dy.p, sizeof(float) * yt.size(), hipMemcpyDeviceToHost));
    for (int64_t co = 0; co < cout; ++co)
        for (int64_t n = 0; n < L; ++n) y[(size_t)co * L + n] = yt[(size_t)n * cout + co];
    API_END
}

int sbv2_debug_set_skinny_max(int workgroups) { return set_skinny_max(workgroups); }
int sbv2_debug_set_clx(int on) { return set_clx(on); }
int sbv2_debug_set_ksplit(int on) { return set_ksplit(on); }
int sbv2_debug_set_flash_parts(int on) { return set_flash_parts(on); }

int sbv2_debug_conv1d_clx(int device, const float* x, const float* w, const float* bias, const float* res, int64_t cin, int64_t cout, int64_t k,
                          int64_t L, int64_t dilation, float pre_slope, float beta, int64_t iters, float* y, float* ys_sum, float* ms) {
    API_BEGIN
    HIP_CHECK(hipSetDevice(device));
    SBV2_REQUIRE(x && w && y, "bad arguments");
    Blob b = one_conv_blob(w, bias, {cout, cin, k}, cout);
    WeightStore ws(b);
    ClConv c = pack_cl(ws, w, (int)cout, (int)cin, (int)k, 2, bias);
    std::vector<float> xt((size_t)L * cin), yt((size_t)L * cout), rt;
    for (int64_t ci = 0; ci < cin; ++ci)
        for (int64_t n = 0; n < L; ++n) xt[(size_t)n * cin + ci] = x[(size_t)ci * L + n];
    DevBuf dx(xt.size()), dy(yt.size()), dr(res ? yt.size() : 4);
    HIP_CHECK(hipMemcpy(dx.p, xt.data(), sizeof(float) * xt.size(), hipMemcpyHostToDevice));
    if (res) {
        rt.resize(yt.size());
        for (int64_t co = 0; co < cout; ++co)
            for (int64_t n = 0; n < L; ++n) rt[(size_t)n * cout + co] = res[(size_t)co * L + n];
        HIP_CHECK(hipMemcpy(dr.p, rt.data(), sizeof(float) * rt.size(), hipMemcpyHostToDevice));
    }
    DevBuf dxs(split_cl_bytes((int)cin, L) / 4 + 4), dys(split_cl_bytes((int)cout, L) / 4 + 4);
    SplitClPlanes xs = make_split_cl(dxs.p, (int)cin, L, nullptr), ysp = make_split_cl(dys.p, (int)cout, L, nullptr);
    split_cl(dx.p, (int)cin, L, (int)cin, pre_slope, xs, nullptr);
    ConvClxParams p;
    p.X = xs;
    p.W = c.wx;
    p.nmt = c.nmt;
    p.M = (int)cout;
    p.N = (int)L;
    p.K = (int)cin;
    p.ntaps = (int)k;
    p.shift0 = (int)(-dilation * (k - 1) / 2);
    p.shift_step = (int)dilation;
    p.Y = dy.p;
    p.ldy = (int)cout;
    if (ys_sum) {
        p.Ys = ysp;
        p.ys_slope = 0.1f;
    }
    p.bias = c.bias;
    if (res) {
        p.R = dr.p;
        p.ldr = (int)cout;
    }
    p.beta = beta;
    SBV2_REQUIRE(conv_clx_usable(p), "shape not supported by conv_clx");
    launch_conv_clx(p, nullptr);
    HIP_CHECK(hipDeviceSynchronize());
    if (iters > 0 && ms) {
        hipEvent_t e0, e1;
        HIP_CHECK(hipEventCreate(&e0));
        HIP_CHECK(hipEventCreate(&e1));
        HIP_CHECK(hipEventRecord(e0, nullptr));
        for (int i = 0; i < iters; ++i) launch_conv_clx(p, nullptr);
        HIP_CHECK(hipEventRecord(e1, nullptr));
        HIP_CHECK(hipEventSynchronize(e1));
        float t = 0.f;
        HIP_CHECK(hipEventElapsedTime(&t, e0, e1));
        *ms = t / (float)iters;
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
    }
    HIP_CHECK(hipMemcpy(yt.data(), dy.p, sizeof(float) * yt.size(), hipMemcpyDeviceToHost));
    for (int64_t co = 0; co < cout; ++co)
        for (int64_t n = 0; n < L; ++n) y[(size_t)co * L + n] = yt[(size_t)n * cout + co];
    if (ys_sum) {
        const int64_t rows = kClxFront + L + kClxBack;
        std::vector<uint16_t> hs(split_cl_bytes((int)cout, L) / 2);
        HIP_CHECK(hipMemcpy(hs.data(), ysp.p, hs.size() * 2, hipMemcpyDeviceToHost));
        auto f = [](uint16_t h) {
            const uint32_t u = (uint32_t)h << 16;
            float v;
            memcpy(&v, &u, 4);
            return v;
        };
        for (int64_t co = 0; co < cout; ++co)
            for (int64_t n = 0; n < L; ++n) {
                const size_t hi = (((size_t)(co >> 4) * 2) * rows + kClxFront + n) * 16 + (co & 15);
                ys_sum[(size_t)co * L + n] = f(hs[hi + (size_t)rows * 16]) + f(hs[hi]);
            }
    }
    API_END
}

int sbv2_debug_conv_cl_clock(int device, int64_t C, int64_t k, int64_t dilation, int64_t L, int abl, double seconds, double* out4) {
    API_BEGIN
    HIP_CHECK(hipSetDevice(device));
    SBV2_REQUIRE(out4 && C >= 128 && (C & 127) == 0 && k >= 1 && k <= kMaxTaps && L >= 256 && ((abl >= 0 && abl <= 3) || abl == 10), "bad arguments");
    std::vector<float> w((size_t)C * C * k), bias((size_t)C, 0.1f), x((size_t)L * C);
    uint64_t st = 0x9E3779B97F4A7C15ull;
    auto rnd = [&]() {   // N(0,1)-ish random data: zero or constant operands would let the chip hold a higher clock (DVFS give-back, items 1 and 7)
        float a = 0.f;
        for (int i = 0; i < 4; ++i) {
            st = st * 6364136223846793005ull + 1442695040888963407ull;
            a += (float)((st >> 40) * (1.0 / 16777216.0)) - 0.5f;
        }
        return a * 1.7320508f;
    };
    for (auto& v : w) v = rnd() / std::sqrt((float)(C * k));
    for (auto& v : x) v = rnd();
    Blob b = one_conv_blob(w.data(), bias.data(), {C, C, k}, C);
    WeightStore ws(b);
    ClConv c = pack_cl(ws, w.data(), (int)C, (int)C, (int)k, 2, bias.data());
    DevBuf dx(x.size()), dy(x.size());
    HIP_CHECK(hipMemcpy(dx.p, x.data(), sizeof(float) * x.size(), hipMemcpyHostToDevice));
    ConvClParams p;
    p.X = dx.p;
    p.ldx = (int)C;
    p.NB = (int)L;
    p.W = c.w;
    p.nmt = c.nmt;
    p.tm = c.tm;
    p.split = 1;
    p.M = (int)C;
    p.N = (int)L;
    p.K = (int)C;
    p.ntaps = (int)k;
    for (int j = 0; j < k; ++j) p.shift[j] = (int)(j * dilation - dilation * (k - 1) / 2);
    p.Y = dy.p;
    p.ldy = (int)C;
    p.bias = c.bias;
    p.pre_slope = 0.1f;
    int nwg = round_up((int)((L + 255) / 256), 8) * (c.nmt / 4);
    // abl 10: the same convolution through conv_clx.hip (operands pre-split)
    DevBuf dxs(abl == 10 ? split_cl_bytes((int)C, L) / 4 + 4 : 4);
    ConvClxParams px;
    if (abl == 10) {
        SplitClPlanes xs = make_split_cl(dxs.p, (int)C, L, nullptr);
        split_cl(dx.p, (int)C, L, (int)C, 0.1f, xs, nullptr);
        px.X = xs;
        px.W = c.wx;
        px.nmt = c.nmt;
        px.M = (int)C;
        px.N = (int)L;
        px.K = (int)C;
        px.ntaps = (int)k;
        px.shift0 = (int)(-dilation * (k - 1) / 2);
        px.shift_step = (int)dilation;
        px.Y = dy.p;
        px.ldy = (int)C;
        px.bias = c.bias;
        SBV2_REQUIRE(conv_clx_usable(px), "conv_clx: shape");
        nwg *= 2;   // (64-row workgroups at most)
    }
    unsigned long long* d_st = nullptr;
    HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&d_st), sizeof(unsigned long long) * kClxStampWords * nwg));
    HIP_CHECK(hipMemset(d_st, 0, sizeof(unsigned long long) * kClxStampWords * nwg));
    hipEvent_t e0, e1;
    HIP_CHECK(hipEventCreate(&e0));
    HIP_CHECK(hipEventCreate(&e1));
    px.stamps = d_st;
    auto launch_conv_cl_diag = [&](const ConvClParams& q, int a, unsigned long long* st, hipStream_t sm) {
        if (a == 10) launch_conv_clx(px, sm);
        else sbv2::launch_conv_cl_diag(q, a, st, sm);
    };
    // >= `seconds` of back-to-back launches so that the power management has settled, then one timed batch whose last launch's stamps are read
    launch_conv_cl_diag(p, abl, d_st, nullptr);
    HIP_CHECK(hipDeviceSynchronize());
    HIP_CHECK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < 20; ++i) launch_conv_cl_diag(p, abl, d_st, nullptr);
    HIP_CHECK(hipEventRecord(e1, nullptr));
    HIP_CHECK(hipEventSynchronize(e1));
    float t20 = 0.f;
    HIP_CHECK(hipEventElapsedTime(&t20, e0, e1));
    const int reps = std::max(20, (int)(seconds * 1e3 / std::max(t20 / 20.f, 1e-3f)));
    for (int i = 0; i < reps; ++i) launch_conv_cl_diag(p, abl, d_st, nullptr);
    HIP_CHECK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < 50; ++i) launch_conv_cl_diag(p, abl, d_st, nullptr);
    HIP_CHECK(hipEventRecord(e1, nullptr));
    HIP_CHECK(hipEventSynchronize(e1));
    float t50 = 0.f;
    HIP_CHECK(hipEventElapsedTime(&t50, e0, e1));
    std::vector<unsigned long long> hs((size_t)kClxStampWords * nwg);
    HIP_CHECK(hipMemcpy(hs.data(), d_st, sizeof(unsigned long long) * hs.size(), hipMemcpyDeviceToHost));
    (void)hipFree(d_st);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    std::vector<double> mhz, cyc;
    for (int g = 0; g < nwg; ++g) {
        const int sg = abl == 10 ? kClxStampWords * g : 4 * g;   // (conv_clx keeps kClxStampWords per workgroup)
        const unsigned long long t0 = hs[sg], r0 = hs[sg + 1], t1 = hs[sg + 2], r1 = hs[sg + 3];
        if (r1 > r0 && t1 > t0) {
            mhz.push_back((double)(t1 - t0) / (double)(r1 - r0) * 100.0);
            cyc.push_back((double)(t1 - t0));
        }
    }
    SBV2_REQUIRE(!mhz.empty(), "no stamps");
    std::sort(mhz.begin(), mhz.end());
    std::sort(cyc.begin(), cyc.end());
    out4[0] = mhz[mhz.size() / 2];           // in-kernel shader clock, MHz (median over workgroups)
    out4[1] = t50 / 50.0;                    // ms per launch
    out4[2] = cyc[cyc.size() / 2];           // shader cycles of one workgroup's chunk loop (median)
    out4[3] = (double)mhz.size();
    API_END
}

// Diagnostics: where a conv_clx workgroup's life goes.  Launches the ResBlock convolution of a wide decoder stage in the form the decoder launches it
// (kind 1 = conv1: parts in, parts out; 2 = conv2: parts in, residual in, f32 + parts out; 3 = a branch's last conv2: residual + accumulate, f32 out)
// back to back for `seconds`, then returns the stamps of the last launch: 12 words per workgroup {loop start (shader clock), loop start (100 MHz),
// loop end (shader clock), loop end (100 MHz), kernel entry (100 MHz), last store issued, stores acknowledged, HW_ID | XCC_ID << 32, epilogue: behind
// the post-loop barrier, its global reads arrived, the first half's stores issued (100 MHz), 0}.
int sbv2_debug_clx_timeline(int device, int64_t C, int64_t k, int64_t dilation, int64_t L, int kind, int variant, double seconds, uint64_t* stamps,
                            int64_t capacity_words, int64_t* workgroups, double* ms_per_launch) {
    API_BEGIN
    HIP_CHECK(hipSetDevice(device));
    SBV2_REQUIRE(stamps && workgroups && ms_per_launch && C >= 64 && (C & 63) == 0 && k >= 1 && k <= kMaxTaps && L >= 256 && kind >= 1 && kind <= 3, "bad arguments");
    std::vector<float> w((size_t)C * C * k), bias((size_t)C, 0.1f), x((size_t)L * C);
    uint64_t st = 0x9E3779B97F4A7C15ull;
    auto rnd = [&]() {
        float a = 0.f;
        for (int i = 0; i < 4; ++i) {
            st = st * 6364136223846793005ull + 1442695040888963407ull;
            a += (float)((st >> 40) * (1.0 / 16777216.0)) - 0.5f;
        }
        return a * 1.7320508f;
    };
    for (auto& v : w) v = rnd() / std::sqrt((float)(C * k));
    for (auto& v : x) v = rnd();
    Blob b = one_conv_blob(w.data(), bias.data(), {C, C, k}, C);
    WeightStore ws(b);
    ClConv c = pack_cl(ws, w.data(), (int)C, (int)C, (int)k, 2, bias.data());
    DevBuf dx(x.size()), dy(x.size()), dr(x.size());
    HIP_CHECK(hipMemcpy(dx.p, x.data(), sizeof(float) * x.size(), hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(dr.p, x.data(), sizeof(float) * x.size(), hipMemcpyHostToDevice));
    HIP_CHECK(hipMemset(dy.p, 0, sizeof(float) * x.size()));
    DevBuf dxs(split_cl_bytes((int)C, L) / 4 + 4), dys(split_cl_bytes((int)C, L) / 4 + 4);
    SplitClPlanes xs = make_split_cl(dxs.p, (int)C, L, nullptr), ysp = make_split_cl(dys.p, (int)C, L, nullptr);
    split_cl(dx.p, (int)C, L, (int)C, 0.1f, xs, nullptr);
    unsigned char* dm = nullptr;
    const size_t nm = (size_t)((L + 63) / 64);
    HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&dm), nm));
    HIP_CHECK(hipMemset(dm, 1, nm));
    ConvClxParams p;
    p.X = xs;
    p.W = c.wx;
    p.nmt = c.nmt;
    p.M = (int)C;
    p.N = (int)L;
    p.K = (int)C;
    p.ntaps = (int)k;
    p.shift0 = (int)(-dilation * (k - 1) / 2);
    p.shift_step = (int)dilation;
    p.bias = c.bias;
    p.mask = dm;
    p.mask_shift = 6;
    p.variant = variant;
    if (kind == 1) {
        p.Ys = ysp;
        p.ys_slope = 0.1f;
    } else {
        p.Y = dy.p;
        p.ldy = (int)C;
        p.R = dr.p;
        p.ldr = (int)C;
        if (kind == 2) {
            p.Ys = ysp;
            p.ys_slope = 0.1f;
        } else {
            p.beta = 1.0f / 3.0f;
            p.accumulate = 1;
        }
    }
    SBV2_REQUIRE(conv_clx_usable(p), "conv_clx: shape");
    const int64_t nwg = clx_grid_workgroups(p);
    SBV2_REQUIRE(capacity_words >= kClxStampWords * nwg, "stamp buffer too small");
    unsigned long long* d_st = nullptr;
    HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&d_st), sizeof(unsigned long long) * kClxStampWords * nwg));
    HIP_CHECK(hipMemset(d_st, 0, sizeof(unsigned long long) * kClxStampWords * nwg));
    hipEvent_t e0, e1;
    HIP_CHECK(hipEventCreate(&e0));
    HIP_CHECK(hipEventCreate(&e1));
    ConvClxParams pq = p;   // timed launches carry no stamps (the product kernel as the decoder runs it)
    launch_conv_clx(pq, nullptr);
    HIP_CHECK(hipDeviceSynchronize());
    HIP_CHECK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < 10; ++i) launch_conv_clx(pq, nullptr);
    HIP_CHECK(hipEventRecord(e1, nullptr));
    HIP_CHECK(hipEventSynchronize(e1));
    float t10 = 0.f;
    HIP_CHECK(hipEventElapsedTime(&t10, e0, e1));
    const int reps = std::max(10, (int)(seconds * 1e3 / std::max(t10 / 10.f, 1e-3f)));
    for (int i = 0; i < reps; ++i) launch_conv_clx(pq, nullptr);
    HIP_CHECK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < 30; ++i) launch_conv_clx(pq, nullptr);
    HIP_CHECK(hipEventRecord(e1, nullptr));
    p.stamps = d_st;
    launch_conv_clx(p, nullptr);
    HIP_CHECK(hipEventSynchronize(e1));
    float t30 = 0.f;
    HIP_CHECK(hipEventElapsedTime(&t30, e0, e1));
    HIP_CHECK(hipDeviceSynchronize());
    HIP_CHECK(hipMemcpy(stamps, d_st, sizeof(unsigned long long) * kClxStampWords * nwg, hipMemcpyDeviceToHost));
    (void)hipFree(d_st);
    (void)hipFree(dm);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *workgroups = nwg;
    *ms_per_launch = t30 / 30.0;
    API_END
}

int sbv2_debug_f16x3_saturation(int device, int enable, uint64_t* count) {
    API_BEGIN
    HIP_CHECK(hipSetDevice(device));
    if (enable >= 0) f16x3_sat_enable(enable);
    if (count) *count = f16x3_sat_read(true);
    API_END
}

int sbv2_debug_set_respair_clx(int on) { return set_respair_clx(on); }

int sbv2_debug_respair(int device, const float* x, const float* w1, const float* w2, const float* b1, const float* b2, int64_t C, int64_t N, int64_t k,
                       int64_t dilation, const uint8_t* mask, int64_t mask_div, float beta, int accumulate, int variant, float* y) {
    API_BEGIN
    HIP_CHECK(hipSetDevice(device));
    SBV2_REQUIRE(x && w1 && w2 && b1 && b2 && y && (C == 16 || C == 32 || C == 64) && N >= 1 && k >= 1 && k <= kMaxTaps && (k & 1) && mask_div >= 1, "bad arguments");
    Blob b = one_conv_blob(w1, b1, {C, C, k}, C);
    WeightStore ws(b);
    ClConv c1 = pack_cl(ws, w1, (int)C, (int)C, (int)k, 2, b1);
    ClConv c2 = pack_cl(ws, w2, (int)C, (int)C, (int)k, 2, b2);
    DevBuf dx((size_t)N * C), dy((size_t)N * C);
    HIP_CHECK(hipMemcpy(dx.p, x, sizeof(float) * N * C, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(dy.p, y, sizeof(float) * N * C, hipMemcpyHostToDevice));   // (the previous contents matter when accumulate is set)
    unsigned char* dm = nullptr;
    const size_t nm = (size_t)((N + mask_div - 1) / mask_div);
    if (mask) {
        HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&dm), nm));
        HIP_CHECK(hipMemcpy(dm, mask, nm, hipMemcpyHostToDevice));
    }
    ResPairParams rp;
    rp.X = dx.p;
    rp.Y = dy.p;
    rp.W1 = c1.w;
    rp.W2 = c2.w;
    rp.W1x = c1.wxp;
    rp.W2x = c2.wxp;
    if (C == 16) {
        rp.W1p = pack_cl_pairs(ws, w1, (int)k);
        rp.W2p = pack_cl_pairs(ws, w2, (int)k);
    }
    rp.b1 = c1.bias;
    rp.b2 = c2.bias;
    rp.C = (int)C;
    rp.N = (int)N;
    rp.k = (int)k;
    rp.dil = (int)dilation;
    rp.split = 1;
    rp.beta = beta;
    rp.accumulate = accumulate;
    rp.mask = dm;
    rp.mask_div = (int)mask_div;
    const int prev = set_respair_clx(variant == 0 ? 0 : (variant == 2 ? 2 : 1));   // 0 = respair_cl, 1 = the default dispatch (respair_clx / respair_x16), 2 = respair_clx at every shape
    try {
        launch_respair_cl(rp, nullptr);
        HIP_CHECK(hipDeviceSynchronize());
    } catch (...) {
        set_respair_clx(prev);
        if (dm) (void)hipFree(dm);
        throw;
    }
    set_respair_clx(prev);
    HIP_CHECK(hipMemcpy(y, dy.p, sizeof(float) * N * C, hipMemcpyDeviceToHost));
    if (dm) (void)hipFree(dm);
    API_END
}

int sbv2_debug_resbranch(int device, const float* x, const float* w, const float* bias, int64_t C, int64_t N, int64_t k, const int64_t* dilations,
                         const uint8_t* mask, int64_t mask_div, float beta, int accumulate, int variant, int64_t iters, float* y, float* ms,
                         uint64_t* stamps, int64_t stamps_cap) {
    API_BEGIN
    HIP_CHECK(hipSetDevice(device));
    SBV2_REQUIRE(x && w && bias && y && dilations && (C == 16 || C == 32 || C == 64 || C == 128) && N >= 1 && k >= 1 && k <= kMaxTaps && (k & 1) && mask_div >= 1 &&
                     (mask_div & (mask_div - 1)) == 0, "bad arguments");
    SBV2_REQUIRE(variant != 0 || C <= 64, "the fused step exists for C <= 64 (variant 2 = the two-launch conv_cl path at any C)");
    const size_t wsz = (size_t)C * C * k;
    Blob b = one_conv_blob(w, bias, {C, C, k}, C);
    WeightStore ws(b);
    ClConv cv[2 * kResBranchSteps];
    const void* wp[2 * kResBranchSteps] = {};
    for (int i = 0; i < 2 * kResBranchSteps; ++i) {
        cv[i] = pack_cl(ws, w + i * wsz, (int)C, (int)C, (int)k, 2, bias + (size_t)i * C);
        if (C == 16) wp[i] = pack_cl_pairs(ws, w + i * wsz, (int)k);
    }
    DevBuf dx((size_t)N * C), dy((size_t)N * C), da((size_t)N * C), db((size_t)N * C), dt((size_t)N * C);
    HIP_CHECK(hipMemcpy(dx.p, x, sizeof(float) * N * C, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(dy.p, y, sizeof(float) * N * C, hipMemcpyHostToDevice));   // (the previous contents matter when accumulate is set)
    DevBuf dy0((size_t)N * C);
    HIP_CHECK(hipMemcpy(dy0.p, y, sizeof(float) * N * C, hipMemcpyHostToDevice));
    unsigned char* dm = nullptr;
    const size_t nm = (size_t)((N + mask_div - 1) / mask_div);
    if (mask) {
        HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&dm), nm));
        HIP_CHECK(hipMemcpy(dm, mask, nm, hipMemcpyHostToDevice));
    }
    int shift = 0;
    while ((1 << shift) < mask_div) ++shift;
    auto run = [&]() {
        if (variant == 1) {     // 1 = the fused branch (resbranch_clx.hip)
            ResBranchParams rb;
            rb.X = dx.p;
            rb.Y = dy.p;
            for (int i = 0; i < 2 * kResBranchSteps; ++i) {
                rb.W[i] = C == 16 ? wp[i] : cv[i].w;
                rb.b[i] = cv[i].bias;
            }
            rb.C = (int)C;
            rb.N = (int)N;
            rb.k = (int)k;
            for (int q = 0; q < kResBranchSteps; ++q) rb.dil[q] = (int)dilations[q];
            rb.beta = beta;
            rb.accumulate = accumulate;
            rb.mask = dm;
            rb.mask_shift = shift;
            launch_resbranch(rb, nullptr);
        } else if (variant == 2) {   // 2 = six launches of conv_cl.hip (conv1 -> T, conv2 + residual): the unfused path, any C
            const float* cur = dx.p;
            for (int q = 0; q < kResBranchSteps; ++q) {
                const bool last = q + 1 == kResBranchSteps;
                float* yn = last ? dy.p : (q & 1 ? db.p : da.p);
                auto conv = [&](const ClConv& c, const float* X, float* Y, int dil, const float* R, float bt, int accum) {
                    ConvClParams p;
                    p.X = X;
                    p.ldx = (int)C;
                    p.NB = (int)N;
                    p.W = c.w;
                    p.nmt = c.nmt;
                    p.tm = c.tm;
                    p.split = 1;
                    p.M = (int)C;
                    p.N = (int)N;
                    p.K = (int)C;
                    p.ntaps = (int)k;
                    for (int t = 0; t < k; ++t) p.shift[t] = t * dil - dil * (int)(k - 1) / 2;
                    p.Y = Y;
                    p.ldy = (int)C;
                    p.bias = c.bias;
                    p.R = R;
                    p.ldr = (int)C;
                    p.pre_slope = 0.1f;
                    p.beta = bt;
                    p.accumulate = accum;
                    p.mask = dm;
                    p.mask_div = (int)mask_div;
                    launch_conv_cl(p, nullptr);
                };
                conv(cv[2 * q], cur, dt.p, (int)dilations[q], nullptr, 1.0f, 0);
                conv(cv[2 * q + 1], dt.p, yn, 1, cur, last ? beta : 1.0f, last ? accumulate : 0);
                cur = yn;
            }
        } else {           // 0 = three launches of the fused step (respair_clx.hip)
            const float* cur = dx.p;
            for (int q = 0; q < kResBranchSteps; ++q) {
                const bool last = q + 1 == kResBranchSteps;
                ResPairParams rp;
                rp.X = cur;
                rp.Y = last ? dy.p : (q & 1 ? db.p : da.p);
                rp.W1 = cv[2 * q].w;
                rp.W2 = cv[2 * q + 1].w;
                rp.W1p = wp[2 * q];
                rp.W2p = wp[2 * q + 1];
                rp.W1x = cv[2 * q].wxp;
                rp.W2x = cv[2 * q + 1].wxp;
                rp.b1 = cv[2 * q].bias;
                rp.b2 = cv[2 * q + 1].bias;
                rp.C = (int)C;
                rp.N = (int)N;
                rp.k = (int)k;
                rp.dil = (int)dilations[q];
                rp.split = 1;
                rp.beta = last ? beta : 1.0f;
                rp.accumulate = last ? accumulate : 0;
                rp.mask = dm;
                rp.mask_div = (int)mask_div;
                launch_respair_cl(rp, nullptr);
                cur = rp.Y;
            }
        }
    };
    try {
        run();
        HIP_CHECK(hipDeviceSynchronize());
        HIP_CHECK(hipMemcpy(y, dy.p, sizeof(float) * N * C, hipMemcpyDeviceToHost));
        if (iters > 0 && ms) {
            hipEvent_t e0, e1;
            HIP_CHECK(hipEventCreate(&e0));
            HIP_CHECK(hipEventCreate(&e1));
            HIP_CHECK(hipEventRecord(e0, nullptr));
            for (int i = 0; i < iters; ++i) run();
            HIP_CHECK(hipEventRecord(e1, nullptr));
            HIP_CHECK(hipEventSynchronize(e1));
            float t = 0.f;
            HIP_CHECK(hipEventElapsedTime(&t, e0, e1));
            *ms = t / (float)iters;
            (void)hipEventDestroy(e0);
            (void)hipEventDestroy(e1);
        }
        if (stamps && stamps_cap >= 16 && variant == 1) {   // one more launch of the stamped instantiation: 16 words per workgroup (resbranch_clx.hip)
            unsigned long long* ds = nullptr;
            HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&ds), sizeof(unsigned long long) * (size_t)stamps_cap));
            HIP_CHECK(hipMemset(ds, 0, sizeof(unsigned long long) * (size_t)stamps_cap));
            ResBranchParams rb;
            rb.X = dx.p;
            rb.Y = dy0.p;
            for (int i = 0; i < 2 * kResBranchSteps; ++i) {
                rb.W[i] = C == 16 ? wp[i] : cv[i].w;
                rb.b[i] = cv[i].bias;
            }
            rb.C = (int)C;
            rb.N = (int)std::min<int64_t>(N, (stamps_cap / 16 - 8) * 100);   // (at least 104 outputs per workgroup: the stamps of every workgroup fit)
            rb.k = (int)k;
            for (int q = 0; q < kResBranchSteps; ++q) rb.dil[q] = (int)dilations[q];
            rb.beta = beta;
            rb.mask = dm;
            rb.mask_shift = shift;
            rb.stamps = ds;
            launch_resbranch(rb, nullptr);
            HIP_CHECK(hipDeviceSynchronize());
            HIP_CHECK(hipMemcpy(stamps, ds, sizeof(unsigned long long) * (size_t)stamps_cap, hipMemcpyDeviceToHost));
            (void)hipFree(ds);
        }
    } catch (...) {
        if (dm) (void)hipFree(dm);
        throw;
    }
    if (dm) (void)hipFree(dm);
    API_END
}
int sbv2_debug_set_resbranch(int on) { return set_resbranch(on); }

int sbv2_debug_respair_clock(int device, int64_t C, int64_t k, int64_t dilation, int64_t L, int variant, int abl, double seconds, double* out, int nout) {
    API_BEGIN
    HIP_CHECK(hipSetDevice(device));
    SBV2_REQUIRE(out && nout >= 17 && (C == 16 || C == 32 || C == 64) && k >= 1 && k <= kMaxTaps && (k & 1) && L >= 256, "bad arguments");
    std::vector<float> w1((size_t)C * C * k), w2((size_t)C * C * k), bias((size_t)C, 0.1f), x((size_t)L * C);
    uint64_t st = 0x9E3779B97F4A7C15ull;
    auto rnd = [&]() {
        float a = 0.f;
        for (int i = 0; i < 4; ++i) {
            st = st * 6364136223846793005ull + 1442695040888963407ull;
            a += (float)((st >> 40) * (1.0 / 16777216.0)) - 0.5f;
        }
        return a * 1.7320508f;
    };
    for (auto& v : w1) v = rnd() / std::sqrt((float)(C * k));
    for (auto& v : w2) v = rnd() / std::sqrt((float)(C * k));
    for (auto& v : x) v = rnd();
    Blob b = one_conv_blob(w1.data(), bias.data(), {C, C, k}, C);
    WeightStore ws(b);
    ClConv c1 = pack_cl(ws, w1.data(), (int)C, (int)C, (int)k, 2, bias.data());
    ClConv c2 = pack_cl(ws, w2.data(), (int)C, (int)C, (int)k, 2, bias.data());
    DevBuf dx(x.size()), dy(x.size());
    HIP_CHECK(hipMemcpy(dx.p, x.data(), sizeof(float) * x.size(), hipMemcpyHostToDevice));
    ResPairParams rp;
    rp.X = dx.p;
    rp.Y = dy.p;
    rp.W1 = c1.w;
    rp.W2 = c2.w;
    rp.W1x = c1.wxp;
    rp.W2x = c2.wxp;
    if (C == 16) {
        rp.W1p = pack_cl_pairs(ws, w1.data(), (int)k);
        rp.W2p = pack_cl_pairs(ws, w2.data(), (int)k);
    }
    rp.b1 = c1.bias;
    rp.b2 = c2.bias;
    rp.C = (int)C;
    rp.N = (int)L;
    rp.k = (int)k;
    rp.dil = (int)dilation;
    rp.split = 1;
    rp.abl = abl;
    const int nto = 128 - (int)(k - 1);          // (the smallest tile any variant uses: respair_clx at C = 64)
    const int nwg = (int)((L + nto - 1) / nto) + 8;
    unsigned long long* d_st = nullptr;
    HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&d_st), sizeof(unsigned long long) * 16 * nwg));
    HIP_CHECK(hipMemset(d_st, 0, sizeof(unsigned long long) * 16 * nwg));
    rp.stamps = d_st;
    const int prev_rpx = set_respair_clx(variant == 3 ? 1 : 0);
    auto launch = [&]() {
        if (variant == 0) launch_respair_cl_diag(rp, nullptr);
        else if (variant == 2) launch_respair_clx_diag(rp, nullptr);
        else if (variant == 4) launch_respair_x16_diag(rp, nullptr);   // respair_x16.hip, stamped
        else launch_respair_cl(rp, nullptr);   // the product kernels (no stamps: clock and phases read 0): 1 = respair_cl, 3 = respair_clx
    };
    hipEvent_t e0, e1;
    HIP_CHECK(hipEventCreate(&e0));
    HIP_CHECK(hipEventCreate(&e1));
    launch();
    HIP_CHECK(hipDeviceSynchronize());
    HIP_CHECK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < 20; ++i) launch();
    HIP_CHECK(hipEventRecord(e1, nullptr));
    HIP_CHECK(hipEventSynchronize(e1));
    float t20 = 0.f;
    HIP_CHECK(hipEventElapsedTime(&t20, e0, e1));
    const int reps = std::max(20, (int)(seconds * 1e3 / std::max(t20 / 20.f, 1e-3f)));
    for (int i = 0; i < reps; ++i) launch();
    HIP_CHECK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < 50; ++i) launch();
    HIP_CHECK(hipEventRecord(e1, nullptr));
    HIP_CHECK(hipEventSynchronize(e1));
    float t50 = 0.f;
    HIP_CHECK(hipEventElapsedTime(&t50, e0, e1));
    std::vector<unsigned long long> hs((size_t)16 * nwg);
    HIP_CHECK(hipMemcpy(hs.data(), d_st, sizeof(unsigned long long) * hs.size(), hipMemcpyDeviceToHost));
    (void)hipFree(d_st);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    set_respair_clx(prev_rpx);
    for (int i = 0; i < nout; ++i) out[i] = 0.0;
    out[1] = t50 / 50.0;
    std::vector<double> mhz;
    std::vector<std::vector<double>> ph(14);
    for (int g = 0; g < nwg; ++g) {
        const unsigned long long* h = hs.data() + (size_t)16 * g;   // low words of the counters: differences modulo 2^32
        const unsigned dr = (unsigned)h[15] - (unsigned)h[14], dt = (unsigned)h[6] - (unsigned)h[0];
        if (h[15] != 0 && dr > 0 && dt > 0) {
            mhz.push_back((double)dt / (double)dr * 100.0);
            for (int i = 1; i < 14; ++i)
                if (h[i] != 0) ph[i].push_back((double)((unsigned)h[i] - (unsigned)h[0]));
        }
    }
    if (!mhz.empty()) {
        std::sort(mhz.begin(), mhz.end());
        out[0] = mhz[mhz.size() / 2];
        out[2] = (double)mhz.size();
        for (int i = 1; i < 14 && 2 + i < nout; ++i)
            if (!ph[i].empty()) {
                std::sort(ph[i].begin(), ph[i].end());
                out[2 + i] = ph[i][ph[i].size() / 2];   // median cycles from the workgroup's entry to stamp i
            }
    }
    API_END
}

// x2 / y2 (sbv2_debug_gemm_bfs_alt): a second input; the launches alternate between the two on ONE scratch buffer that is never cleared in between, so a
// workgroup that read a stale partial sum (the other input's, left in its XCD's L2 by the previous launch) would show up in the result
static int debug_gemm_bfs_impl(int device, const float* x, const float* x2, const float* w, const float* bias, const float* res, int64_t M, int64_t N, int64_t K,
                               int parts, int act, int split_out, int64_t iters, float* y, float* y2, float* ms) {
    API_BEGIN
    HIP_CHECK(hipSetDevice(device));
    SBV2_REQUIRE((parts == 2 || parts == 3 || parts == kPartsF16x3) && x && w && y && M >= 1 && N >= 4 && (N & 3) == 0 && (K & 15) == 0, "bad arguments");
    Blob b = one_conv_blob(w, bias, {M, K, 1}, M);
    WeightStore ws(b);
    ws.set_bfs_parts(parts);
    PackedConv pc = ws.conv("c");
    const int ld = round_up((int)N, 64);
    Plane X{nullptr, (int)K, (int)N, ld}, Y{nullptr, (int)M, (int)N, ld}, R{nullptr, (int)M, (int)N, ld};
    DevBuf dx((size_t)K * ld), dy((size_t)M * ld), dr((size_t)M * ld), dxs((size_t)split_nplanes(parts) * K * ld / 2 + 16), dys((size_t)3 * M * ld / 2 + 16);
    X.p = dx.p;
    Y.p = dy.p;
    R.p = dr.p;
    HIP_CHECK(hipMemset(X.p, 0, sizeof(float) * (size_t)K * ld));
    HIP_CHECK(hipMemcpy2D(X.p, sizeof(float) * ld, x, sizeof(float) * N, sizeof(float) * N, K, hipMemcpyHostToDevice));
    if (res) HIP_CHECK(hipMemcpy2D(R.p, sizeof(float) * ld, res, sizeof(float) * N, sizeof(float) * N, M, hipMemcpyHostToDevice));
    SplitPlanes xs;
    xs.p = dxs.p;
    xs.parts = split_nplanes(parts);
    xs.f16 = parts == kPartsF16x3;
    xs.sat = xs.f16 ? f16x3_sat_counter() : nullptr;
    xs.C = (int)K;
    xs.L = (int)N;
    xs.ld = ld;
    xs.pstride = (int64_t)K * ld;
    split_planes(X, xs, nullptr);
    DevBuf dx2(x2 ? (size_t)K * ld : 4), dxs2(x2 ? (size_t)split_nplanes(parts) * K * ld / 2 + 16 : 4);
    SplitPlanes xs2 = xs;
    if (x2) {
        Plane X2{dx2.p, (int)K, (int)N, ld};
        HIP_CHECK(hipMemset(X2.p, 0, sizeof(float) * (size_t)K * ld));
        HIP_CHECK(hipMemcpy2D(X2.p, sizeof(float) * ld, x2, sizeof(float) * N, sizeof(float) * N, K, hipMemcpyHostToDevice));
        xs2.p = dxs2.p;
        split_planes(X2, xs2, nullptr);
    }
    SplitPlanes ys;
    ys.p = dys.p;
    ys.parts = split_nplanes(split_out);
    ys.f16 = split_out == kPartsF16x3;
    ys.C = (int)M;
    ys.L = (int)N;
    ys.ld = ld;
    ys.pstride = (int64_t)M * ld;
    // split_out: 0 = f32 result only; 2 / 3 = the result is ALSO written as that many bf16 parts, and y returns their sum (what a consumer sees)
    // (scratch for the small-grid K split, as DeBERTa's forward provides it)
    constexpr size_t kWs = (size_t)48 << 20;   // (as BertModel::kSkWsBytes / kSkCounters)
    DevBuf dsk(kWs / sizeof(float) + 1024);
    HIP_CHECK(hipMemset(dsk.p, 0, kWs + 1024 * sizeof(float)));
    BfsSplitK sk;
    sk.ws = dsk.p;
    sk.ws_bytes = kWs;
    sk.counters = reinterpret_cast<unsigned*>(dsk.p + (kWs / sizeof(float)));
    sk.ncounters = 1024;
    int turn = 0;
    auto run = [&]() {
        conv_bfs(pc, (x2 && (turn++ & 1)) ? xs2 : xs, &Y, split_out ? &ys : nullptr, nullptr, 1, nullptr, act, res ? &R : nullptr, 1.0f, 1.0f, -1, 0, &sk);
    };
    run();
    HIP_CHECK(hipDeviceSynchronize());
    if (iters > 0 && ms) {
        hipEvent_t e0, e1;
        HIP_CHECK(hipEventCreate(&e0));
        HIP_CHECK(hipEventCreate(&e1));
        HIP_CHECK(hipEventRecord(e0, nullptr));
        for (int i = 0; i < iters; ++i) run();
        HIP_CHECK(hipEventRecord(e1, nullptr));
        HIP_CHECK(hipEventSynchronize(e1));
        float t = 0.f;
        HIP_CHECK(hipEventElapsedTime(&t, e0, e1));
        *ms = t / (float)iters;
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
    }
    if (x2) {   // the last launch of each input, back to back on the used scratch
        turn = 1;
        run();
        HIP_CHECK(hipDeviceSynchronize());
        HIP_CHECK(hipMemcpy2D(y2, sizeof(float) * N, Y.p, sizeof(float) * ld, sizeof(float) * N, M, hipMemcpyDeviceToHost));
        turn = 0;
        run();
        HIP_CHECK(hipDeviceSynchronize());
    }
    HIP_CHECK(hipMemcpy2D(y, sizeof(float) * N, Y.p, sizeof(float) * ld, sizeof(float) * N, M, hipMemcpyDeviceToHost));
    if (split_out) {
        const int np = ys.parts;
        std::vector<uint16_t> hs((size_t)np * M * ld);
        HIP_CHECK(hipMemcpy(hs.data(), ys.p, hs.size() * 2, hipMemcpyDeviceToHost));
        for (int64_t m = 0; m < M; ++m)
            for (int64_t n = 0; n < N; ++n) {
                float acc = 0.f;
                if (ys.f16) {
                    _Float16 hi, lo;
                    memcpy(&hi, &hs[(size_t)m * ld + n], 2);
                    memcpy(&lo, &hs[((size_t)M + m) * ld + n], 2);
                    y[(size_t)m * N + n] = (float)hi + (float)lo * (1.0f / kF16LoScale);
                    continue;
                }
                for (int pp = np - 1; pp >= 0; --pp) {
                    const uint32_t u = (uint32_t)hs[((size_t)pp * M + m) * ld + n] << 16;
                    float f;
                    memcpy(&f, &u, 4);
                    acc += f;
                }
                y[(size_t)m * N + n] = acc;
            }
    }
    API_END
}

int sbv2_debug_gemm_bfs(int device, const float* x, const float* w, const float* bias, const float* res, int64_t M, int64_t N, int64_t K,
                        int parts, int act, int split_out, int64_t iters, float* y, float* ms) {
    return debug_gemm_bfs_impl(device, x, nullptr, w, bias, res, M, N, K, parts, act, split_out, iters, y, nullptr, ms);
}
int sbv2_debug_gemm_bfs_alt(int device, const float* xa, const float* xb, const float* w, const float* bias, const float* res, int64_t M, int64_t N, int64_t K,
                            int parts, int64_t iters, float* ya, float* yb) {
    if (!xb || !yb) {
        set_last_error("bad arguments");
        return 1;
    }
    float ms = 0.f;
    return debug_gemm_bfs_impl(device, xa, xb, w, bias, res, M, N, K, parts, 0, 0, iters, ya, yb, &ms);
}

int sbv2_debug_time_conv1d(int device, int64_t cin, int64_t cout, int64_t k, int64_t L, int64_t dilation, int64_t iters, float* ms) {
    API_BEGIN
    HIP_CHECK(hipSetDevice(device));
    SBV2_REQUIRE(ms && iters >= 1, "bad arguments");
    std::vector<float> w((size_t)cout * cin * k), bias((size_t)cout, 0.1f);
    uint32_t st = 12345u;
    auto rnd = [&]() {
        st = st * 1664525u + 1013904223u;
        return ((st >> 8) * (1.0f / 16777216.0f) - 0.5f);
    };
    for (auto& v : w) v = rnd() * 0.1f;
    Blob b = one_conv_blob(w.data(), bias.data(), {cout, cin, k}, cout);
    WeightStore ws(b);
    PackedConv pc = ws.conv("c");
    Plane X{nullptr, (int)cin, (int)L, round_up((int)L, 64)}, Y{nullptr, (int)cout, (int)L, round_up((int)L, 64)};
    DevBuf dx((size_t)cin * X.ld), dy((size_t)cout * Y.ld);
    X.p = dx.p;
    Y.p = dy.p;
    {
        std::vector<float> hx((size_t)cin * X.ld);
        for (auto& v : hx) v = rnd() * 2.f;
        HIP_CHECK(hipMemcpy(X.p, hx.data(), sizeof(float) * hx.size(), hipMemcpyHostToDevice));
    }
    hipEvent_t e0, e1;
    HIP_CHECK(hipEventCreate(&e0));
    HIP_CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) conv_plain(pc, X, Y, (int)dilation, (int)(dilation * (k - 1) / 2), nullptr, 1, nullptr, ACT_NONE, 1.0f);
    HIP_CHECK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < iters; ++i) conv_plain(pc, X, Y, (int)dilation, (int)(dilation * (k - 1) / 2), nullptr, 1, nullptr, ACT_NONE, 1.0f);
    HIP_CHECK(hipEventRecord(e1, nullptr));
    HIP_CHECK(hipEventSynchronize(e1));
    float t = 0.f;
    HIP_CHECK(hipEventElapsedTime(&t, e0, e1));
    *ms = t / (float)iters;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    API_END
}

}  // extern "C"
