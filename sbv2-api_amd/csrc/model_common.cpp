// Weight packing and launch helpers shared by the DeBERTa and VITS model objects.
#include <algorithm>
#include <cstring>

#include "models.h"

namespace sbv2 {

WeightStore::~WeightStore() {
    for (void* p : allocs_) (void)hipFree(p);
}

float* WeightStore::upload(const float* host, size_t n) {
    float* d = nullptr;
    HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&d), std::max<size_t>(n, 4) * sizeof(float)));
    allocs_.push_back(d);
    HIP_CHECK(hipMemcpy(d, host, n * sizeof(float), hipMemcpyHostToDevice));
    bytes_ += n * sizeof(float);
    return d;
}

float* WeightStore::tensor(const std::string& name) {
    const HostTensor& t = blob_.get(name);
    return upload(t.data, (size_t)t.numel());
}

float* WeightStore::tensor(const std::string& name, std::initializer_list<int64_t> dims) {
    const HostTensor& t = blob_.get(name);
    std::vector<int64_t> a, b;
    for (int64_t d : t.dims)
        if (d != 1) a.push_back(d);
    for (int64_t d : dims)
        if (d != 1) b.push_back(d);
    if (a != b) {
        std::string got, want;
        for (int64_t d : t.dims) got += (got.empty() ? "" : ",") + std::to_string(d);
        for (int64_t d : dims) want += (want.empty() ? "" : ",") + std::to_string(d);
        throw Error("tensor '" + name + "' has shape [" + got + "], the model config implies [" + want + "]");
    }
    return upload(t.data, (size_t)t.numel());
}

void WeightStore::expect(const PackedConv& c, const std::string& prefix, int cout, int cin, int k) const {
    if (c.cout != cout || c.cin != cin || c.k != k)
        throw Error("weights '" + prefix + "' are [" + std::to_string(c.cout) + "," + std::to_string(c.cin) + "," + std::to_string(c.k) +
                    "], the model config implies [" + std::to_string(cout) + "," + std::to_string(cin) + "," + std::to_string(k) + "]");
}

static inline uint16_t bfs_bf16_rne(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // NaN stays NaN
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
static inline float bfs_bf16_f32(uint16_t h) {
    const uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

BfsWeights pack_bfs(WeightStore& ws, const float* w, int M, int K, int code) {
    SBV2_REQUIRE((code == 2 || code == 3 || code == kPartsF16x3) && (K & 15) == 0 && M >= 1, "pack_bfs: bad shape");
    BfsWeights b;
    b.M = M;
    b.K = K;
    const int parts = split_nplanes(code);
    b.parts = parts;
    b.f16 = code == kPartsF16x3;
    b.nmt = round_up((M + 31) / 32, 4);   // whole 128-row tiles: a tile's fragment blocks of a chunk are one contiguous DMA source
    const int nchunks = K / 16;
    std::vector<uint16_t> h((size_t)nchunks * b.nmt * parts * 512, 0);
    for (int ch = 0; ch < nchunks; ++ch)
        for (int mt = 0; mt < b.nmt; ++mt) {
            uint16_t* blk = h.data() + (((size_t)ch * b.nmt + mt) * parts) * 512;
            for (int l = 0; l < 64; ++l) {
                const int m = mt * 32 + (l & 31);
                if (m >= M) continue;
                for (int j = 0; j < 8; ++j) {
                    float r = w[(size_t)m * K + ch * 16 + 8 * (l >> 5) + j];
                    if (b.f16) {   // hi = f16(x), lo = f16((x - hi) 2^11): common.h
                        const float c = std::min(std::max(r, -65504.f), 65504.f);
                        const _Float16 hi = (_Float16)c, lo = (_Float16)((c - (float)hi) * kF16LoScale);
                        memcpy(&blk[l * 8 + j], &hi, 2);
                        memcpy(&blk[512 + l * 8 + j], &lo, 2);
                        continue;
                    }
                    for (int p = 0; p < parts; ++p) {
                        const uint16_t q = bfs_bf16_rne(r);
                        blk[p * 512 + l * 8 + j] = q;
                        r -= bfs_bf16_f32(q);
                    }
                }
            }
        }
    b.w = ws.upload(reinterpret_cast<const float*>(h.data()), h.size() / 2);
    return b;
}

PackedConv WeightStore::conv(const std::string& prefix, bool bias) {
    const HostTensor& t = blob_.get(prefix + ".weight");
    SBV2_REQUIRE(t.dims.size() == 3, "conv weight must be [Cout][Cin][k]: " + prefix);
    PackedConv pc;
    pc.cout = (int)t.dims[0];
    pc.cin = (int)t.dims[1];
    pc.k = (int)t.dims[2];
    SBV2_REQUIRE(pc.k <= kMaxTaps, "kernel size exceeds the compiled tap limit: " + prefix);
    pc.lda = round_up(pc.cout, 4);
    std::vector<float> h((size_t)pc.k * pc.cin * pc.lda, 0.f);
    for (int co = 0; co < pc.cout; ++co)
        for (int ci = 0; ci < pc.cin; ++ci)
            for (int j = 0; j < pc.k; ++j) h[((size_t)j * pc.cin + ci) * pc.lda + co] = t.data[((size_t)co * pc.cin + ci) * pc.k + j];
    pc.w = upload(h.data(), h.size());
    if (bias && blob_.has(prefix + ".bias")) pc.bias = tensor(prefix + ".bias", {pc.cout});
    if (cl_parts_) {
        pc.cl = pack_cl(*this, t.data, pc.cout, pc.cin, pc.k, cl_parts_, nullptr);
        pc.cl.bias = pc.bias;
    }
    if (bfs_parts_ && pc.k == 1 && (pc.cin & 15) == 0) pc.bfs = pack_bfs(*this, t.data, pc.cout, pc.cin, bfs_parts_);
    return pc;
}

// Several 1x1 convolutions of the same input as ONE product: rows are the concatenated output channels (q | k | v projections of an
// attention layer: one launch with 3x the rows instead of three under-filled ones).  f32 MFMA packing only.
PackedConv WeightStore::conv_cat(const std::vector<std::string>& prefixes) {
    PackedConv pc;
    pc.k = 1;
    std::vector<const HostTensor*> ts;
    for (const auto& pre : prefixes) {
        const HostTensor& t = blob_.get(pre + ".weight");
        SBV2_REQUIRE((t.dims.size() == 3 && t.dims[2] == 1) || t.dims.size() == 2, "conv_cat: 1x1 conv / linear weights only: " + pre);
        SBV2_REQUIRE(ts.empty() || t.dims[1] == ts[0]->dims[1], "conv_cat: input channel mismatch: " + pre);
        ts.push_back(&t);
        pc.cout += (int)t.dims[0];
    }
    pc.cin = (int)ts[0]->dims[1];
    pc.lda = round_up(pc.cout, 4);
    std::vector<float> h((size_t)pc.cin * pc.lda, 0.f), b((size_t)pc.cout, 0.f);
    int row0 = 0;
    for (size_t n = 0; n < ts.size(); ++n) {
        const HostTensor& t = *ts[n];
        const int co_n = (int)t.dims[0];
        for (int co = 0; co < co_n; ++co)
            for (int ci = 0; ci < pc.cin; ++ci) h[(size_t)ci * pc.lda + row0 + co] = t.data[(size_t)co * pc.cin + ci];
        if (blob_.has(prefixes[n] + ".bias")) {
            const HostTensor& bt = blob_.get(prefixes[n] + ".bias");
            SBV2_REQUIRE(bt.numel() == co_n, "bias size mismatch: " + prefixes[n]);
            for (int co = 0; co < co_n; ++co) b[row0 + co] = bt.data[co];
        }
        row0 += co_n;
    }
    pc.w = upload(h.data(), h.size());
    pc.bias = upload(b.data(), b.size());
    if (bfs_parts_ && (pc.cin & 15) == 0) {
        std::vector<float> rows((size_t)pc.cout * pc.cin);
        for (int co = 0; co < pc.cout; ++co)
            for (int ci = 0; ci < pc.cin; ++ci) rows[(size_t)co * pc.cin + ci] = h[(size_t)ci * pc.lda + co];
        pc.bfs = pack_bfs(*this, rows.data(), pc.cout, pc.cin, bfs_parts_);
    }
    return pc;
}

PackedConv WeightStore::linear(const std::string& prefix) {
    const HostTensor& t = blob_.get(prefix + ".weight");
    SBV2_REQUIRE(t.dims.size() == 2, "linear weight must be [out][in]: " + prefix);
    PackedConv pc;
    pc.cout = (int)t.dims[0];
    pc.cin = (int)t.dims[1];
    pc.k = 1;
    pc.lda = round_up(pc.cout, 4);
    std::vector<float> h((size_t)pc.cin * pc.lda, 0.f);
    for (int co = 0; co < pc.cout; ++co)
        for (int ci = 0; ci < pc.cin; ++ci) h[(size_t)ci * pc.lda + co] = t.data[(size_t)co * pc.cin + ci];
    pc.w = upload(h.data(), h.size());
    if (blob_.has(prefix + ".bias")) pc.bias = tensor(prefix + ".bias", {pc.cout});
    if (cl_parts_) {
        pc.cl = pack_cl(*this, t.data, pc.cout, pc.cin, 1, cl_parts_, nullptr);
        pc.cl.bias = pc.bias;
    }
    if (bfs_parts_ && (pc.cin & 15) == 0) pc.bfs = pack_bfs(*this, t.data, pc.cout, pc.cin, bfs_parts_);
    return pc;
}

// ConvTranspose1d y[co][s*q + r] = b[co] + sum_ci sum_t W[ci][co][s*t + r + p] * x[ci][q - t]; output phases r that use the
// same set of t are computed by one launch whose rows are (phase, co).
PackedUpsample WeightStore::upsample(const std::string& prefix, int stride, int padding) {
    const HostTensor& t = blob_.get(prefix + ".weight");
    SBV2_REQUIRE(t.dims.size() == 3, "transposed conv weight must be [Cin][Cout][k]: " + prefix);
    PackedUpsample u;
    u.cin = (int)t.dims[0];
    u.cout = (int)t.dims[1];
    u.stride = stride;
    const int k = (int)t.dims[2];
    SBV2_REQUIRE(k - 2 * padding == stride, "transposed conv must upsample exactly by its stride (k - 2p == s): " + prefix);
    SBV2_REQUIRE(stride <= 64, "stride too large");
    std::vector<std::vector<int>> tsets(stride);
    for (int r = 0; r < stride; ++r)
        for (int tt = -k; tt <= k; ++tt) {
            const int j = stride * tt + r + padding;
            if (j >= 0 && j < k) tsets[r].push_back(tt);
        }
    std::vector<bool> done(stride, false);
    for (int r = 0; r < stride; ++r) {
        if (done[r]) continue;
        std::vector<int> phases;
        for (int r2 = r; r2 < stride && (int)phases.size() < kMaxPhases; ++r2)
            if (!done[r2] && tsets[r2] == tsets[r]) {
                phases.push_back(r2);
                done[r2] = true;
            }
        PackedUpsample::Grp g;
        g.ntaps = (int)tsets[r].size();
        SBV2_REQUIRE(g.ntaps >= 1 && g.ntaps <= kMaxTaps, "bad polyphase tap count");
        g.nph = (int)phases.size();
        const int M = g.nph * u.cout;
        g.lda = round_up(M, 4);
        std::vector<float> h((size_t)g.ntaps * u.cin * g.lda, 0.f);
        for (int ti = 0; ti < g.ntaps; ++ti) {
            g.shift[ti] = -tsets[r][ti];
            for (int pi = 0; pi < g.nph; ++pi) {
                const int j = stride * tsets[r][ti] + phases[pi] + padding;
                for (int ci = 0; ci < u.cin; ++ci)
                    for (int co = 0; co < u.cout; ++co)
                        h[((size_t)ti * u.cin + ci) * g.lda + pi * u.cout + co] = t.data[((size_t)ci * u.cout + co) * k + j];
            }
        }
        for (int pi = 0; pi < kMaxPhases; ++pi) g.phase_off[pi] = pi < g.nph ? phases[pi] : 0;
        g.w = upload(h.data(), h.size());
        u.groups.push_back(g);
    }
    u.bias = tensor(prefix + ".bias", {u.cout});
    return u;
}

SegLayout make_layout(const std::vector<int>& lens, int gap, Arena& arena, hipStream_t stream, const unsigned char* extra_mask, int round_to) {
    SegLayout l;
    l.n = (int)lens.size();
    int pos = 0;
    for (int v : lens) {
        const int st = round_up(pos, 4);
        l.start.push_back(st);
        l.len.push_back(v);
        pos = st + v + gap;
    }
    l.L = round_up(std::max(pos, 4), std::max(round_to, 4));   // (columns behind the last utterance are gap columns: masked, zero)
    std::vector<int> seg(l.L, -1);
    std::vector<unsigned char> mask(l.L, 0);
    size_t e = 0;
    for (int i = 0; i < l.n; ++i)
        for (int t = 0; t < l.len[i]; ++t, ++e) {
            seg[l.start[i] + t] = i;
            mask[l.start[i] + t] = extra_mask ? extra_mask[e] : 1;
        }
    l.d_seg_of = arena.array<int>(l.L);
    l.d_mask = arena.array<unsigned char>(l.L);
    l.d_start = arena.array<int>(l.n);
    l.d_len = arena.array<int>(l.n);
    UploadBatch ub(arena);   // four neighbours: one copy
    arena.upload(l.d_seg_of, seg.data(), sizeof(int) * l.L, stream);
    arena.upload(l.d_mask, mask.data(), l.L, stream);
    arena.upload(l.d_start, l.start.data(), sizeof(int) * l.n, stream);
    arena.upload(l.d_len, l.len.data(), sizeof(int) * l.n, stream);
    return l;
}

void conv_plain(const PackedConv& w, Plane x, Plane y, int dil, int pad_l, const unsigned char* mask, int mask_div, hipStream_t s,
                int act, float pre_slope, const Plane* res, float alpha, float beta, int accumulate) {
    SBV2_REQUIRE(x.C == w.cin && y.C == w.cout, "conv channel mismatch");
    // bf16 / split-bf16 matrix cores on k-major planes; only for k >= 3: a 1x1 conv has 384 MFMA cycles per 16-channel chunk and the
    // two barriers + the transposing stage around it cost more than the f32 MFMA kernel (measured, round 1)
    if (w.cl.parts && w.k >= 3) {
        ConvClParams q;
        q.X = x.p;
        q.ldx = x.ld;
        q.NB = x.L;
        q.W = w.cl.w;
        q.nmt = w.cl.nmt;
        q.tm = w.cl.tm;
        q.split = w.cl.parts == 2;
        q.f16 = w.cl.parts == 3;
        q.M = w.cout;
        q.N = y.L;
        q.K = w.cin;
        q.ntaps = w.k;
        for (int j = 0; j < w.k; ++j) q.shift[j] = j * dil - pad_l;
        q.Y = y.p;
        q.ldy = y.ld;
        q.bias = w.bias;
        if (res) {
            q.R = res->p;
            q.ldr = res->ld;
        }
        q.pre_slope = pre_slope;
        q.act = act;
        q.alpha = alpha;
        q.beta = beta;
        q.accumulate = accumulate;
        q.mask = mask;
        q.mask_div = mask_div;
        q.in_km = 1;
        q.out_km = 1;
        launch_conv_cl(q, s);
        return;
    }
    ConvParams p;
    p.A = w.w;
    p.lda = w.lda;
    p.a_tap_stride = w.tap_stride();
    p.B = x.p;
    p.ldb = x.ld;
    p.nb = x.L;
    p.C = y.p;
    p.ldc = y.ld;
    p.M = w.cout;
    p.N = y.L;
    p.K = w.cin;
    p.ntaps = w.k;
    for (int j = 0; j < w.k; ++j) p.shift[j] = j * dil - pad_l;
    p.bias = w.bias;
    p.bias_mode = w.bias ? BIAS_ROW : BIAS_NONE;
    p.pre_slope = pre_slope;
    p.act = act;
    p.alpha = alpha;
    p.beta = beta;
    if (res) {
        p.R = res->p;
        p.ldr = res->ld;
    }
    p.accumulate = accumulate;
    p.mask = mask;
    p.mask_div = mask_div;
    launch_conv(p, s);
}

// 1x1 product on the bf16 matrix cores with pre-split operands (gemm_bfs.hip); y and / or ys receive the result.
void conv_bfs(const PackedConv& w, const SplitPlanes& xs, const Plane* y, const SplitPlanes* ys, const unsigned char* mask, int mask_div,
              hipStream_t s, int act, const Plane* res, float alpha, float beta, int y_rows, int ys_row0, const BfsSplitK* sk) {
    SBV2_REQUIRE(w.bfs.parts && w.k == 1 && xs.C == w.cin && xs.parts == w.bfs.parts && xs.f16 == w.bfs.f16, "conv_bfs: operands were not prepared for the split-bf16 kernel");
    GemmBfsParams p;
    p.W = w.bfs;
    p.X = xs;
    p.M = w.cout;
    p.N = xs.L;
    p.K = w.cin;
    if (y) {
        SBV2_REQUIRE(y->C == w.cout && y->L == xs.L, "conv_bfs: output shape");
        p.Y = y->p;
        p.ldy = y->ld;
    }
    if (ys) {
        SBV2_REQUIRE(ys->C == w.cout && ys->L == xs.L, "conv_bfs: split output shape");
        p.Ys = *ys;
    }
    p.bias = w.bias;
    p.act = act;
    p.alpha = alpha;
    p.beta = beta;
    if (res) {
        p.R = res->p;
        p.ldr = res->ld;
    }
    p.mask = mask;
    p.mask_div = mask_div;
    if (y_rows >= 0) p.y_rows = y_rows;
    p.ys_row0 = ys_row0;
    if (sk) p.sk = *sk;
    launch_gemm_bfs(p, s);
}

// The two halves of a conv -> activation -> conv pair whose wide intermediate stays channels-last (the encoders' FFN: 192 -> 768 -> 192):
// the k-major <-> fragment transposition of conv_cl.hip's k-major variant is then paid on the narrow side only, the wide tensor is
// written by the full-line channels-last epilogue and read back by the full-line staging.  Both return false (nothing launched) when
// the weights carry no matrix-core fragments or the shapes do not fit, and the caller falls back to conv_plain.
static void fill_cl(ConvClParams& q, const PackedConv& w, int dil, int pad_l, const unsigned char* mask, int mask_div) {
    q.W = w.cl.w;
    q.nmt = w.cl.nmt;
    q.tm = w.cl.tm;
    q.split = w.cl.parts == 2;
    q.f16 = w.cl.parts == 3;
    q.M = w.cout;
    q.K = w.cin;
    q.ntaps = w.k;
    for (int j = 0; j < w.k; ++j) q.shift[j] = j * dil - pad_l;
    q.bias = w.bias;
    q.mask = mask;
    q.mask_div = mask_div;
}
bool conv_km_to_cl(const PackedConv& w, Plane x, float* y, int ldy, int dil, int pad_l, const unsigned char* mask, int mask_div,
                   hipStream_t s) {
    if (!w.cl.parts || w.k < 3 || (w.cout & 15) || (ldy & 3)) return false;
    SBV2_REQUIRE(x.C == w.cin, "conv channel mismatch");
    ConvClParams q;
    fill_cl(q, w, dil, pad_l, mask, mask_div);
    q.X = x.p;
    q.ldx = x.ld;
    q.NB = x.L;
    q.N = x.L;
    q.Y = y;
    q.ldy = ldy;
    q.in_km = 1;
    q.out_km = 0;
    launch_conv_cl(q, s);
    return true;
}
bool conv_cl_to_km(const PackedConv& w, const float* x, int ldx, Plane y, int dil, int pad_l, const unsigned char* mask, int mask_div,
                   hipStream_t s, float pre_slope, const Plane* res) {
    if (!w.cl.parts || w.k < 3 || (w.cin & 15) || (ldx & 3)) return false;
    SBV2_REQUIRE(y.C == w.cout, "conv channel mismatch");
    ConvClParams q;
    fill_cl(q, w, dil, pad_l, mask, mask_div);
    q.X = x;
    q.ldx = ldx;
    q.NB = y.L;
    q.N = y.L;
    q.Y = y.p;
    q.ldy = y.ld;
    if (res) {
        q.R = res->p;
        q.ldr = res->ld;
    }
    q.pre_slope = pre_slope;
    q.in_km = 0;
    q.out_km = 1;
    launch_conv_cl(q, s);
    return true;
}

void linear_tokmajor(const PackedConv& w, Plane x, float* y, int ldy, hipStream_t s) {
    SBV2_REQUIRE(w.k == 1 && x.C == w.cin, "token-major linear: shape mismatch");
    // (the bf16 matrix-core variant of this 1x1 product was measured slower than the f32 kernel: see conv_plain)
    ConvParams p;
    p.A = x.p;  // A[k = cin][m = token]
    p.lda = x.ld;
    p.B = w.w;  // B[k = cin][n = cout]
    p.ldb = w.lda;
    p.nb = w.cout;
    p.C = y;
    p.ldc = ldy;
    p.M = x.L;
    p.N = w.cout;
    p.K = w.cin;
    p.bias = w.bias;
    p.bias_mode = w.bias ? BIAS_COL : BIAS_NONE;
    launch_conv(p, s);
}

}  // namespace sbv2
