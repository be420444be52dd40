// Real-weight import: the bytes the reference hands to `load_model` (crates/sbv2_core/src/model.rs:6) and keeps in
// its `.sbv2` containers (crates/sbv2_core/src/sbv2file.rs:15-37; writer scripts/convert/convert_model.py:156-175) are turned into the same
// named-tensor table (`Blob`) the synthetic SBV2W001 container produces, so the model objects do not care where weights come from.
//
//   bytes -> sniff:  "SBV2W001"            -> parse_blob (common.cpp)
//                    zstd frame            -> ZSTD_decompress via dlopen("libzstd.so.1") (zstd 0.13 crate in the reference) -> tar
//                    tar ("ustar" / valid header checksum)  -> entries `model.onnx`, `style_vectors.json` (others ignored, like sbv2file.rs:25-29)
//                    ONNX ModelProto       -> dependency-free protobuf reader -> initializers + nodes -> names
//
// Naming.  torch.onnx.export keeps the PyTorch state-dict name for every parameter that reaches an op unchanged (Conv weights / biases,
// Embedding tables, LayerNorm gamma / beta, emb_rel_k / emb_rel_v); what it does NOT keep, and what `onnxsim` (convert_model.py:156,
// convert_deberta.py:52) additionally folds, is recovered from the graph:
//   * Linear on a 3-D input exports as MatMul(x, W^T) + Add(bias): the weight is an anonymous transposed constant ("onnx::MatMul_123"),
//     the bias keeps its name -> weight name = bias name with ".bias" -> ".weight", value transposed back.
//   * weight_norm'ed convolutions (every HiFi-GAN conv) export as g * v / ||v||: either the pair <name>.weight_g / <name>.weight_v is still
//     there (folded here), or onnxsim has folded it into an anonymous constant feeding a Conv / ConvTranspose whose bias input keeps its name
//     -> weight name from the bias name; the bias-free conv_post is the one anonymous Conv with a single output channel.
//   * Subgraphs that depend on initializers only are constant-folded by onnxsim.  DeBERTa: rel_embeddings -> encoder.LayerNorm -> every layer's
//     key_proj / query_proj of the positions: what remains is one anonymous constant per layer and kind, the constant operand of the c2p / p2c MatMul
//     (possibly behind Tile / Expand / Transpose); located by following the MatMul's dynamic operand up to the Add of the layer's named query_proj /
//     key_proj bias, stored as "<layer>.attention.self.pos_key" / ".pos_query" [2 * buckets][heads][d] (rule 6 in name_tensors; BertModel takes them
//     instead of projecting the positions itself; the head count, which an unfolded file does not show, is recovered on the way).  VITS: exp(-logs) of
//     the stochastic duration predictor's ElementwiseAffine -> "sdp.flows.0.exp_neg_logs" (rule 7).
// The hyper-parameters (the SBV2W001 container's JSON) are derived from tensor shapes and node attributes (strides, dilations, epsilon).
// PARITY STATUS: no real deberta.onnx / model_*.onnx exists in the build environment; these rules follow the exporters' documented behaviour
// and are exercised on synthetic files written by tests/onnx_writer.py (builder-authored).  A file that does not fit fails loudly with the
// list of tensors that could not be located, it is never silently half-loaded.
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <set>

#include "api_internal.h"

namespace sbv2 {

namespace {

// ---- protobuf wire format ----------------------------------------------------------------------------------------------------------
struct Span {
    const uint8_t* p = nullptr;
    size_t n = 0;
    std::string str() const { return std::string(reinterpret_cast<const char*>(p), n); }
};

struct PB {
    const uint8_t* p;
    const uint8_t* e;
    explicit PB(Span s) : p(s.p), e(s.p + s.n) {}
    bool done() const { return p >= e; }
    uint64_t varint() {
        uint64_t v = 0;
        for (int sh = 0; sh < 64; sh += 7) {
            SBV2_REQUIRE(p < e, "ONNX: truncated varint");
            const uint8_t b = *p++;
            v |= (uint64_t)(b & 0x7F) << sh;
            if (!(b & 0x80)) return v;
        }
        throw Error("ONNX: varint too long");
    }
    void tag(uint32_t& field, uint32_t& wt) {
        const uint64_t t = varint();
        field = (uint32_t)(t >> 3);
        wt = (uint32_t)(t & 7);
    }
    Span bytes() {
        const uint64_t n = varint();
        SBV2_REQUIRE(n <= (uint64_t)(e - p), "ONNX: truncated length-delimited field");
        Span s{p, (size_t)n};
        p += n;
        return s;
    }
    uint32_t fixed32() {
        SBV2_REQUIRE(e - p >= 4, "ONNX: truncated fixed32");
        uint32_t v;
        std::memcpy(&v, p, 4);
        p += 4;
        return v;
    }
    void skip(uint32_t wt) {
        switch (wt) {
            case 0: varint(); break;
            case 1: SBV2_REQUIRE(e - p >= 8, "ONNX: truncated fixed64"); p += 8; break;
            case 2: bytes(); break;
            case 5: fixed32(); break;
            default: throw Error("ONNX: unsupported wire type " + std::to_string(wt));
        }
    }
};

struct OnnxTensor {
    std::string name;
    std::vector<int64_t> dims;
    int dtype = 0;   // 1 = float32, 7 = int64, 10 = float16, 11 = double
    Span raw;
    std::vector<float> fdata;      // float_data (field 4)
    std::vector<int64_t> idata;    // int64_data (field 7)
    // checked product (a crafted initializer such as four dims of 65536 would wrap a plain product to 0 while keeping the huge dims, from
    // which config values are derived): same bound as parse_blob's containers
    int64_t numel() const {
        int64_t n = 1;
        for (int64_t d : dims) {
            SBV2_REQUIRE(d >= 0 && (d == 0 || n <= (1ll << 40) / std::max<int64_t>(d, 1)), "ONNX: tensor too large: " + name);
            n *= d;
        }
        return n;
    }
};

OnnxTensor parse_tensor(Span s) {
    OnnxTensor t;
    PB pb(s);
    while (!pb.done()) {
        uint32_t f, wt;
        pb.tag(f, wt);
        if (f == 1) {   // dims: repeated int64 (packed or not)
            if (wt == 2) {
                PB q(pb.bytes());
                while (!q.done()) t.dims.push_back((int64_t)q.varint());
            } else t.dims.push_back((int64_t)pb.varint());
        } else if (f == 2 && wt == 0) t.dtype = (int)pb.varint();
        else if (f == 4) {   // float_data
            if (wt == 2) {
                Span b = pb.bytes();
                t.fdata.resize(b.n / 4);
                std::memcpy(t.fdata.data(), b.p, t.fdata.size() * 4);
            } else {
                const uint32_t u = pb.fixed32();
                float v;
                std::memcpy(&v, &u, 4);
                t.fdata.push_back(v);
            }
        } else if (f == 7) {   // int64_data
            if (wt == 2) {
                PB q(pb.bytes());
                while (!q.done()) t.idata.push_back((int64_t)q.varint());
            } else t.idata.push_back((int64_t)pb.varint());
        } else if (f == 8 && wt == 2) t.name = pb.bytes().str();
        else if (f == 9 && wt == 2) t.raw = pb.bytes();
        else if (f == 13 || f == 14) throw Error("ONNX: external tensor data is not supported (the reference loads single-file models: model.rs:6)");
        else pb.skip(wt);
    }
    SBV2_REQUIRE(t.dims.size() <= 8, "ONNX: tensor rank out of range: " + t.name);
    for (int64_t d : t.dims) SBV2_REQUIRE(d >= 0 && d < (1ll << 31), "ONNX: tensor dimension out of range: " + t.name);
    (void)t.numel();   // refuses an overflowing shape here, before any config value is read from the dims
    return t;
}

static float half_to_float(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000) << 16;
    uint32_t exp = (h >> 10) & 0x1F, man = h & 0x3FF, u;
    if (exp == 0) {
        if (man == 0) u = sign;
        else {
            exp = 127 - 15 + 1;
            while (!(man & 0x400)) { man <<= 1; --exp; }
            u = sign | (exp << 23) | ((man & 0x3FF) << 13);
        }
    } else if (exp == 31) u = sign | 0x7F800000u | (man << 13);
    else u = sign | ((exp + 127 - 15) << 23) | (man << 13);
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}

// tensor -> f32 values (float32 / float16 / double payloads; raw_data or typed field)
std::vector<float> tensor_f32(const OnnxTensor& t) {
    const int64_t n = t.numel();
    std::vector<float> v((size_t)n);
    if (t.dtype == 1) {
        if (t.raw.n) {
            SBV2_REQUIRE((int64_t)t.raw.n == n * 4, "ONNX: raw_data size mismatch: " + t.name);
            std::memcpy(v.data(), t.raw.p, (size_t)n * 4);
        } else {
            SBV2_REQUIRE((int64_t)t.fdata.size() == n, "ONNX: float_data size mismatch: " + t.name);
            v = t.fdata;
        }
    } else if (t.dtype == 10) {
        SBV2_REQUIRE((int64_t)t.raw.n == n * 2, "ONNX: float16 tensors must use raw_data: " + t.name);
        for (int64_t i = 0; i < n; ++i) {
            uint16_t h;
            std::memcpy(&h, t.raw.p + 2 * i, 2);
            v[(size_t)i] = half_to_float(h);
        }
    } else if (t.dtype == 11) {
        SBV2_REQUIRE((int64_t)t.raw.n == n * 8, "ONNX: double tensors must use raw_data: " + t.name);
        for (int64_t i = 0; i < n; ++i) {
            double d;
            std::memcpy(&d, t.raw.p + 8 * i, 8);
            v[(size_t)i] = (float)d;
        }
    } else throw Error("ONNX: tensor '" + t.name + "' has unsupported data type " + std::to_string(t.dtype));
    return v;
}

struct OnnxAttr {
    std::string name;
    std::vector<int64_t> ints;
    float f = 0.f;
    int64_t i = 0;
    bool has_t = false;
    OnnxTensor t;
};
struct OnnxNode {
    std::string op, name;
    std::vector<std::string> in, out;
    std::vector<OnnxAttr> attrs;
    const OnnxAttr* attr(const std::string& n) const {
        for (const auto& a : attrs)
            if (a.name == n) return &a;
        return nullptr;
    }
};

OnnxNode parse_node(Span s) {
    OnnxNode nd;
    PB pb(s);
    while (!pb.done()) {
        uint32_t f, wt;
        pb.tag(f, wt);
        if (f == 1 && wt == 2) nd.in.push_back(pb.bytes().str());
        else if (f == 2 && wt == 2) nd.out.push_back(pb.bytes().str());
        else if (f == 3 && wt == 2) nd.name = pb.bytes().str();
        else if (f == 4 && wt == 2) nd.op = pb.bytes().str();
        else if (f == 5 && wt == 2) {
            OnnxAttr a;
            PB q(pb.bytes());
            while (!q.done()) {
                uint32_t g, w2;
                q.tag(g, w2);
                if (g == 1 && w2 == 2) a.name = q.bytes().str();
                else if (g == 2 && w2 == 5) {
                    const uint32_t u = q.fixed32();
                    std::memcpy(&a.f, &u, 4);
                } else if (g == 3 && w2 == 0) a.i = (int64_t)q.varint();
                else if (g == 5 && w2 == 2) {
                    a.t = parse_tensor(q.bytes());
                    a.has_t = true;
                } else if (g == 8) {
                    if (w2 == 2) {
                        PB r(q.bytes());
                        while (!r.done()) a.ints.push_back((int64_t)r.varint());
                    } else a.ints.push_back((int64_t)q.varint());
                } else q.skip(w2);
            }
            nd.attrs.push_back(std::move(a));
        } else pb.skip(wt);
    }
    return nd;
}

struct OnnxGraph {
    std::vector<OnnxNode> nodes;
    std::map<std::string, OnnxTensor> init;   // initializers + Constant node outputs
};

OnnxGraph parse_onnx(const uint8_t* b, size_t n) {
    OnnxGraph g;
    PB pb(Span{b, n});
    Span graph;
    while (!pb.done()) {
        uint32_t f, wt;
        pb.tag(f, wt);
        if (f == 7 && wt == 2) graph = pb.bytes();   // ModelProto.graph
        else pb.skip(wt);
    }
    SBV2_REQUIRE(graph.p, "ONNX: the model has no graph");
    PB gp(graph);
    while (!gp.done()) {
        uint32_t f, wt;
        gp.tag(f, wt);
        if (f == 1 && wt == 2) g.nodes.push_back(parse_node(gp.bytes()));
        else if (f == 5 && wt == 2) {
            OnnxTensor t = parse_tensor(gp.bytes());
            g.init[t.name] = std::move(t);
        } else gp.skip(wt);
    }
    for (const auto& nd : g.nodes)
        if (nd.op == "Constant" && nd.out.size() == 1)
            if (const OnnxAttr* a = nd.attr("value"))
                if (a->has_t) {
                    OnnxTensor t = a->t;
                    t.name = nd.out[0];
                    g.init[t.name] = std::move(t);
                }
    return g;
}

// ---- zstd (dlopen) and tar -----------------------------------------------------------------------------------------------------------
std::vector<uint8_t> zstd_decompress(const uint8_t* b, size_t n) {
    void* h = nullptr;
    for (const char* name : {"libzstd.so.1", "libzstd.so"}) {
        h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (h) break;
    }
    if (!h) throw Error("this is a zstd-compressed .sbv2 file but libzstd.so.1 is not available");
    using size_fn = unsigned long long (*)(const void*, size_t);
    using dec_fn = size_t (*)(void*, size_t, const void*, size_t);
    using err_fn = unsigned (*)(size_t);
    auto get_size = reinterpret_cast<size_fn>(dlsym(h, "ZSTD_getFrameContentSize"));
    auto dec = reinterpret_cast<dec_fn>(dlsym(h, "ZSTD_decompress"));
    auto is_err = reinterpret_cast<err_fn>(dlsym(h, "ZSTD_isError"));
    SBV2_REQUIRE(get_size && dec && is_err, "libzstd.so.1 lacks ZSTD_getFrameContentSize / ZSTD_decompress");
    unsigned long long sz = get_size(b, n);
    // 0xFFFF...FF = unknown, ...FE = error (zstd.h): the reference's writer (ZstdCompressor(...).compress(), convert_model.py:171-174)
    // records the content size in the frame header
    SBV2_REQUIRE(sz != 0xFFFFFFFFFFFFFFFEull, ".sbv2: not a valid zstd frame");
    SBV2_REQUIRE(sz != 0xFFFFFFFFFFFFFFFFull, ".sbv2: zstd frame without content size (streaming frames are not supported)");
    SBV2_REQUIRE(sz <= (16ull << 30), ".sbv2: implausible decompressed size");
    std::vector<uint8_t> out((size_t)sz);
    const size_t got = dec(out.data(), out.size(), b, n);
    SBV2_REQUIRE(!is_err(got) && got == out.size(), ".sbv2: zstd decompression failed");
    return out;
}

bool looks_like_tar(const uint8_t* b, size_t n) {
    if (n < 512) return false;
    if (std::memcmp(b + 257, "ustar", 5) == 0) return true;
    // pre-POSIX header: checksum over the 512 bytes with the checksum field as spaces
    unsigned sum = 0;
    for (int i = 0; i < 512; ++i) sum += (i >= 148 && i < 156) ? ' ' : b[i];
    char f[9] = {0};
    std::memcpy(f, b + 148, 8);
    return b[0] != 0 && std::strtoul(f, nullptr, 8) == sum;
}

// entries of an uncompressed tar archive: name -> (offset, size)
std::map<std::string, std::pair<size_t, size_t>> tar_entries(const uint8_t* b, size_t n) {
    std::map<std::string, std::pair<size_t, size_t>> out;
    size_t pos = 0;
    while (pos + 512 <= n) {
        const uint8_t* h = b + pos;
        bool zero = true;
        for (int i = 0; i < 512 && zero; ++i) zero = h[i] == 0;
        if (zero) break;
        char name[101] = {0}, szf[13] = {0};
        std::memcpy(name, h, 100);
        std::memcpy(szf, h + 124, 12);
        uint64_t size = 0;
        if ((uint8_t)szf[0] & 0x80) {   // base-256 size (GNU)
            for (int i = 1; i < 12; ++i) size = (size << 8) | (uint8_t)h[124 + i];
        } else size = std::strtoull(szf, nullptr, 8);
        pos += 512;
        SBV2_REQUIRE(size <= n - pos, ".sbv2: truncated tar entry");
        const char type = (char)h[156];
        std::string nm(name);
        if (h[345]) {   // ustar prefix
            char pre[156] = {0};
            std::memcpy(pre, h + 345, 155);
            nm = std::string(pre) + "/" + nm;
        }
        if (type == '0' || type == 0) out[nm] = {pos, (size_t)size};
        pos += (size_t)((size + 511) / 512 * 512);
    }
    return out;
}

// ---- graph -> named f32 tensors ------------------------------------------------------------------------------------------------------
struct Named {
    std::map<std::string, std::vector<int64_t>> dims;
    std::map<std::string, std::vector<float>> data;
    void put(const std::string& name, std::vector<int64_t> d, std::vector<float> v) {
        dims[name] = std::move(d);
        data[name] = std::move(v);
    }
    bool has(const std::string& n) const { return data.count(n) != 0; }
};

bool ends_with(const std::string& s, const std::string& suf) { return s.size() >= suf.size() && s.compare(s.size() - suf.size(), suf.size(), suf) == 0; }
bool is_float_tensor(const OnnxTensor& t) { return t.dtype == 1 || t.dtype == 10 || t.dtype == 11; }
// names torch.onnx.export generates for constants that lost their parameter name
bool is_anonymous(const std::string& n) { return n.empty() || n.find("onnx::") == 0 || n.find('/') != std::string::npos || n.find("Constant") != std::string::npos; }

Named name_tensors(const OnnxGraph& g) {
    Named out;
    // 1. every float initializer under its own name
    for (const auto& kv : g.init)
        if (is_float_tensor(kv.second) && !is_anonymous(kv.first)) out.put(kv.first, kv.second.dims, tensor_f32(kv.second));
    // 2. weight_norm pairs still in the file: w = g * v / ||v||, the norm over every dim but 0 (torch.nn.utils.weight_norm, dim = 0)
    for (const auto& kv : g.init) {
        if (!ends_with(kv.first, ".weight_v")) continue;
        const std::string base = kv.first.substr(0, kv.first.size() - 9);
        auto gi = g.init.find(base + ".weight_g");
        if (gi == g.init.end() || out.has(base + ".weight")) continue;
        std::vector<float> v = tensor_f32(kv.second), gg = tensor_f32(gi->second);
        const int64_t rows = kv.second.dims.empty() ? 1 : kv.second.dims[0];
        const int64_t per = rows ? kv.second.numel() / rows : 0;
        SBV2_REQUIRE((int64_t)gg.size() == rows, "weight_norm: " + base + ".weight_g does not have one value per output channel");
        for (int64_t r = 0; r < rows; ++r) {
            double nn = 0;
            for (int64_t e = 0; e < per; ++e) nn += (double)v[(size_t)(r * per + e)] * v[(size_t)(r * per + e)];
            const float sc = gg[(size_t)r] / (float)std::sqrt(nn);
            for (int64_t e = 0; e < per; ++e) v[(size_t)(r * per + e)] *= sc;
        }
        out.put(base + ".weight", kv.second.dims, std::move(v));
    }
    // producers / consumers
    std::map<std::string, std::vector<const OnnxNode*>> consumers;
    for (const auto& nd : g.nodes)
        for (const auto& i : nd.in) consumers[i].push_back(&nd);
    auto init_of = [&](const std::string& n) -> const OnnxTensor* {
        auto it = g.init.find(n);
        return it == g.init.end() ? nullptr : &it->second;
    };
    // 3. Conv / ConvTranspose whose weight lost its name but whose bias kept it (weight_norm folded by onnxsim)
    std::vector<const OnnxNode*> unbiased_anon;
    for (const auto& nd : g.nodes) {
        if ((nd.op != "Conv" && nd.op != "ConvTranspose") || nd.in.size() < 2) continue;
        const OnnxTensor* w = init_of(nd.in[1]);
        if (!w || !is_float_tensor(*w) || !is_anonymous(nd.in[1])) continue;
        if (nd.in.size() >= 3 && !nd.in[2].empty() && ends_with(nd.in[2], ".bias") && !is_anonymous(nd.in[2])) {
            const std::string wn = nd.in[2].substr(0, nd.in[2].size() - 5) + ".weight";
            if (!out.has(wn)) out.put(wn, w->dims, tensor_f32(*w));
        } else unbiased_anon.push_back(&nd);
    }
    // ... and the one bias-free convolution of the generator: conv_post (Conv1d(C, 1, 7, bias=False)), a single output channel
    if (!out.has("dec.conv_post.weight"))
        for (const OnnxNode* nd : unbiased_anon) {
            const OnnxTensor* w = init_of(nd->in[1]);
            if (nd->op == "Conv" && w->dims.size() == 3 && w->dims[0] == 1) {
                out.put("dec.conv_post.weight", w->dims, tensor_f32(*w));
                break;
            }
        }
    // 4. MatMul(x, W^T) + Add(named bias): Linear on a 3-D input
    for (const auto& nd : g.nodes) {
        if (nd.op != "MatMul" || nd.in.size() != 2 || nd.out.empty()) continue;
        const OnnxTensor* w = init_of(nd.in[1]);
        if (!w || !is_float_tensor(*w) || w->dims.size() != 2) continue;
        for (const OnnxNode* c : consumers[nd.out[0]]) {
            if (c->op != "Add") continue;
            for (const auto& bi : c->in) {
                if (bi == nd.out[0] || !ends_with(bi, ".bias") || is_anonymous(bi) || !init_of(bi)) continue;
                const std::string wn = bi.substr(0, bi.size() - 5) + ".weight";
                if (out.has(wn)) continue;
                const int64_t K = w->dims[0], N = w->dims[1];
                const std::vector<float> src = tensor_f32(*w);
                std::vector<float> t((size_t)(K * N));
                for (int64_t k = 0; k < K; ++k)
                    for (int64_t m = 0; m < N; ++m) t[(size_t)(m * K + k)] = src[(size_t)(k * N + m)];
                out.put(wn, {N, K}, std::move(t));
            }
        }
    }
    // 5. Gemm(x, W, bias) with transB = 1 keeps the weight name; with transB = 0 the weight is stored transposed under its own name
    for (const auto& nd : g.nodes) {
        if (nd.op != "Gemm" || nd.in.size() < 2) continue;
        const OnnxAttr* tb = nd.attr("transB");
        const OnnxTensor* w = init_of(nd.in[1]);
        if (!w || w->dims.size() != 2 || (tb && tb->i == 1) || is_anonymous(nd.in[1])) continue;
        std::vector<float> src = tensor_f32(*w);
        const int64_t K = w->dims[0], N = w->dims[1];
        std::vector<float> t((size_t)(K * N));
        for (int64_t k = 0; k < K; ++k)
            for (int64_t m = 0; m < N; ++m) t[(size_t)(m * K + k)] = src[(size_t)(k * N + m)];
        out.put(nd.in[1], {N, K}, std::move(t));
    }
    // 6. DeBERTa after onnxsim: the relative-position subgraph rel_embeddings -> encoder.LayerNorm -> key_proj / query_proj of every layer depends on
    //    initializers only and is constant-folded (convert_deberta.py:52); what is left of it is, per layer, one anonymous constant holding the projected
    //    positions as the constant operand of the c2p MatMul (dynamic side: the layer's query) and one for the p2c MatMul (dynamic side: its key),
    //    possibly behind Tile / Expand (the batch repeat) and Transpose nodes.  They are located by topology: the dynamic side is followed up to the
    //    Add of a named query_proj / key_proj bias (that gives layer and kind), the constant side down to its initializer; the constant [heads, R, d]
    //    (any order of the last two axes) is stored as "<layer>.attention.self.pos_key" / ".pos_query" = rows [R][heads * d] (modeling_deberta_v2.py:
    //    292-299, 318-343: pos_key_layer = transpose_for_scores(key_proj(rel_embeddings)), used unscaled).
    {
        std::map<std::string, const OnnxNode*> producer;
        for (const auto& nd : g.nodes)
            for (const auto& o : nd.out) producer[o] = &nd;
        auto pass_through = [](const std::string& op) {
            return op == "Transpose" || op == "Tile" || op == "Expand" || op == "Reshape" || op == "Cast" || op == "Identity" || op == "Unsqueeze" || op == "Squeeze" ||
                   op == "Div" || op == "Mul";
        };
        // the initializer behind `n`; swaps = Transpose nodes on the way that exchange the last two axes
        // On the CONSTANT side only value- and layout-preserving nodes may sit between the initializer and the MatMul: Identity, Unsqueeze / Squeeze (size-1
        // axes), Tile / Expand (the batch repeat on the leading axis) and a Transpose of the last two axes (counted).  Anything else (a Mul / Div by a scale
        // that was not folded, a Reshape that changes the layout, another permutation) would make the stored positions silently wrong: `bad` names it, and
        // the caller refuses the file once the dynamic side has identified the MatMul as a position product.
        auto const_side = [&](std::string n, int& swaps, std::string& bad) -> const OnnxTensor* {
            swaps = 0;
            bad.clear();
            for (int hop = 0; hop < 8; ++hop) {
                if (const OnnxTensor* t = init_of(n)) return is_float_tensor(*t) ? t : nullptr;
                auto it = producer.find(n);
                if (it == producer.end() || !pass_through(it->second->op) || it->second->in.empty()) return nullptr;
                const std::string& op = it->second->op;
                if (op == "Transpose") {
                    const OnnxAttr* pa = it->second->attr("perm");
                    bool last_two = false, identity = pa != nullptr;
                    if (pa) {
                        const size_t r = pa->ints.size();
                        last_two = r >= 2 && pa->ints[r - 1] == (int64_t)r - 2 && pa->ints[r - 2] == (int64_t)r - 1;
                        for (size_t a = 0; a < r; ++a) {
                            const bool fixed = pa->ints[a] == (int64_t)a;
                            if (!fixed) identity = false;
                            if (!fixed && !(last_two && a + 2 >= r)) last_two = false;
                        }
                    }
                    if (last_two) ++swaps;
                    else if (!identity && bad.empty()) bad = "Transpose (not a swap of the last two axes) at " + n;
                } else if (!(op == "Identity" || op == "Unsqueeze" || op == "Squeeze" || op == "Tile" || op == "Expand")) {
                    if (bad.empty()) bad = op + " at " + n;
                }
                n = it->second->in[0];
            }
            return nullptr;
        };
        auto dyn_side = [&](std::string n, std::string& layer_prefix, bool& is_query) -> bool {
            for (int hop = 0; hop < 12; ++hop) {
                auto it = producer.find(n);
                if (it == producer.end()) return false;
                const OnnxNode* nd = it->second;
                if (nd->op == "Add") {
                    for (const auto& bi : nd->in) {
                        if (is_anonymous(bi) || !init_of(bi)) continue;
                        for (const char* kind : {"query_proj.bias", "key_proj.bias"})
                            if (ends_with(bi, std::string(".attention.self.") + kind)) {
                                layer_prefix = bi.substr(0, bi.size() - std::strlen(kind));
                                is_query = kind[0] == 'q';
                                return true;
                            }
                    }
                    return false;
                }
                if (!pass_through(nd->op) || nd->in.empty()) return false;
                n = nd->in[0];
            }
            return false;
        };
        for (const auto& nd : g.nodes) {
            if (nd.op != "MatMul" || nd.in.size() != 2) continue;
            for (int side = 0; side < 2; ++side) {
                int swaps = 0;
                std::string bad;
                const OnnxTensor* c = const_side(nd.in[side], swaps, bad);
                if (!c || c->dims.size() < 3) continue;
                std::vector<int64_t> d3;
                for (int64_t v : c->dims)
                    if (!(d3.empty() && v == 1)) d3.push_back(v);     // leading 1s (a folded batch axis) dropped
                if (d3.size() != 3) continue;
                std::string lp;
                bool isq = false;
                if (!dyn_side(nd.in[1 - side], lp, isq)) continue;
                SBV2_REQUIRE(bad.empty(), "ONNX import: the folded DeBERTa position constant of " + lp + " reaches its MatMul through a node that is not value- and layout-preserving: " + bad);
                const std::string nm = lp + (isq ? "pos_key" : "pos_query");   // the QUERY meets the projected position KEYS (c2p) and vice versa
                if (out.has(nm)) continue;
                auto qb = out.dims.find(lp + "query_proj.bias");
                if (qb == out.dims.end() || qb->second.empty()) continue;
                const int64_t H = qb->second[0], heads = d3[0];
                if (heads < 1 || H % heads) continue;
                const int64_t dh = H / heads, R = c->numel() / H;
                if (R * H != c->numel()) continue;
                // orientation of the constant's last two axes: from the sizes, or (R == d) from the topology: at the MatMul the operand is [.., d, R] as the
                // second factor ([.., S, d] x [.., d, R]) and [.., R, d] as the first; every last-two Transpose on the way flips it
                bool r_first;
                if (R != dh) {
                    r_first = d3[1] == R && d3[2] == dh;
                    if (!r_first && !(d3[1] == dh && d3[2] == R)) continue;
                } else {
                    if (d3[1] != R || d3[2] != R) continue;
                    r_first = (side == 1) == ((swaps & 1) == 1);
                }
                const std::vector<float> src = tensor_f32(*c);
                std::vector<float> P((size_t)(R * H));
                for (int64_t h = 0; h < heads; ++h)
                    for (int64_t r = 0; r < R; ++r)
                        for (int64_t j = 0; j < dh; ++j)
                            P[(size_t)(r * H + h * dh + j)] = r_first ? src[(size_t)((h * R + r) * dh + j)] : src[(size_t)((h * dh + j) * R + r)];
                out.put(nm, {R, heads, dh}, std::move(P));   // (the head count is recoverable from a folded file)
            }
        }
    }
    // 7. VITS after onnxsim: ElementwiseAffine's exp(-logs) of the stochastic duration predictor's first flow is folded into an anonymous constant
    //    (x - m) * exp(-logs) (modeling_vits.py:689-704, reverse branch): the Mul behind the Sub of the named translate vector
    for (const auto& nd : g.nodes) {
        if (nd.op != "Sub" || nd.in.size() != 2 || nd.out.empty()) continue;
        const std::string& mn = nd.in[1];
        if (!ends_with(mn, "flows.0.m") || is_anonymous(mn) || !init_of(mn)) continue;
        const std::string base = mn.substr(0, mn.size() - 1);
        if (out.has(base + "logs") || out.has(base + "exp_neg_logs")) continue;
        for (const OnnxNode* c : consumers[nd.out[0]]) {
            if (c->op != "Mul") continue;
            for (const auto& ci : c->in) {
                const OnnxTensor* e = init_of(ci);
                if (ci == nd.out[0] || !e || !is_float_tensor(*e) || e->numel() != init_of(mn)->numel()) continue;
                out.put(base + "exp_neg_logs", init_of(mn)->dims, tensor_f32(*e));
            }
        }
    }
    return out;
}

int count_indexed(const Named& t, const std::string& prefix, const std::string& suffix, int step = 1, int first = 0) {
    int n = 0;
    while (t.has(prefix + std::to_string(first + n * step) + suffix)) ++n;
    return n;
}

const std::vector<int64_t>& dims_of(const Named& t, const std::string& name) {
    auto it = t.dims.find(name);
    if (it == t.dims.end()) throw Error("ONNX import: tensor '" + name + "' not found in the graph (initializer names after onnxsim differ from what "
                                        "csrc/import.cpp expects; see the naming rules at the top of that file).  The constant-folded forms onnxsim leaves "
                                        "(DeBERTa: projected relative positions as the constant operand of each layer's c2p / p2c MatMul; VITS: "
                                        "exp(-sdp.flows.0.logs) behind the Sub of flows.0.m) are recognised by topology (rules 6 and 7); if this file "
                                        "folds them differently, convert it with the onnxsim step skipped (INTEGRATION.md, 'Real weights')");
    return it->second;
}

Blob to_blob(Named& t, uint32_t kind, const std::string& json) {
    Blob b;
    b.kind = kind;
    b.config_json = json;
    auto store = std::make_shared<std::vector<std::vector<float>>>();
    store->reserve(t.data.size());
    for (auto& kv : t.data) {
        store->push_back(std::move(kv.second));
        HostTensor h;
        h.dims = t.dims[kv.first];
        if (h.dims.empty()) h.dims = {1};
        h.data = store->back().data();
        b.tensors.emplace(kv.first, std::move(h));
    }
    b.owned = store;
    return b;
}

// the Conv / ConvTranspose node of a layer, found by its (named) weight or bias input
const OnnxNode* node_of_layer(const OnnxGraph& g, const char* op, const std::string& prefix) {
    for (const auto& nd : g.nodes) {
        if (nd.op != op) continue;
        for (size_t i = 1; i < nd.in.size(); ++i)
            if (nd.in[i] == prefix + ".weight" || nd.in[i] == prefix + ".bias") return &nd;
    }
    return nullptr;
}

}  // namespace

// ---- public (within the library) -----------------------------------------------------------------------------------------------------
// JP-Extra synthesizer: names are those of style_bert_vits2's SynthesizerTrn state dict (what synth.py generates as well)
Blob import_vits_onnx(const uint8_t* bytes, size_t n) {
    const OnnxGraph g = parse_onnx(bytes, n);
    Named t = name_tensors(g);
    auto D = [&](const std::string& nm) -> const std::vector<int64_t>& { return dims_of(t, nm); };
    const int hidden = (int)D("enc_p.emb.weight")[1], n_vocab = (int)D("enc_p.emb.weight")[0];
    const int n_tones = (int)D("enc_p.tone_emb.weight")[0], n_langs = (int)D("enc_p.language_emb.weight")[0];
    const int n_speakers = (int)D("emb_g.weight")[0], gin = (int)D("emb_g.weight")[1];
    const int inter = (int)D("enc_p.proj.weight")[0] / 2;
    const int enc_layers = count_indexed(t, "enc_p.encoder.attn_layers.", ".conv_q.weight");
    SBV2_REQUIRE(enc_layers >= 1, "ONNX import: no enc_p.encoder.attn_layers.* found");
    const int filter = (int)D("enc_p.encoder.ffn_layers.0.conv_1.weight")[0], enc_kernel = (int)D("enc_p.encoder.ffn_layers.0.conv_1.weight")[2];
    const auto& erk = D("enc_p.encoder.attn_layers.0.emb_rel_k");
    SBV2_REQUIRE(erk.size() == 3, "emb_rel_k must be [1][2w+1][dk]");
    const int window = (int)(erk[1] - 1) / 2, heads = hidden / (int)erk[2];
    const int style_dim = (int)D("enc_p.style_proj.weight")[1], bert_dim = (int)D("enc_p.bert_proj.weight")[1];
    const int flow_n = count_indexed(t, "flow.flows.", ".pre.weight", 2);
    SBV2_REQUIRE(flow_n >= 1, "ONNX import: no flow.flows.* found");
    const int flow_layers = count_indexed(t, "flow.flows.0.enc.attn_layers.", ".conv_q.weight");
    const int flow_kernel = (int)D("flow.flows.0.enc.ffn_layers.0.conv_1.weight")[2];
    const int dp_filter = (int)D("dp.conv_1.weight")[0], dp_kernel = (int)D("dp.conv_1.weight")[2];
    const int sdp_dds = count_indexed(t, "sdp.convs.convs_sep.", ".weight");
    const int sdp_kernel = (int)D("sdp.convs.convs_sep.0.weight")[2];
    int sdp_flows = 1;   // ConvFlow i lives at sdp.flows.(2i - 1); ConvFlow 1 is unused in reverse mode and may have been pruned
    while (t.has("sdp.flows." + std::to_string(2 * (sdp_flows + 1) - 1) + ".proj.weight")) ++sdp_flows;
    SBV2_REQUIRE(sdp_flows >= 2, "ONNX import: no sdp.flows.*.proj found");
    const int sdp_bins = ((int)D("sdp.flows.3.proj.weight")[0] + 1) / 3;
    const int up_initial = (int)D("dec.conv_pre.weight")[0];
    const int n_up = count_indexed(t, "dec.ups.", ".weight");
    SBV2_REQUIRE(n_up >= 1, "ONNX import: no dec.ups.* found");
    const int n_rb = count_indexed(t, "dec.resblocks.", ".convs1.0.weight");
    SBV2_REQUIRE(n_rb % n_up == 0 && n_rb > 0, "ONNX import: resblock count is not a multiple of the upsampling stages");
    const int nk = n_rb / n_up;
    std::string rates = "[", kernels = "[", rks = "[", rds = "[";
    for (int i = 0; i < n_up; ++i) {
        const auto& wd = D("dec.ups." + std::to_string(i) + ".weight");
        const int k = (int)wd[2];
        // stride: the ConvTranspose node's attribute; without it (weight located by name only) k - 2p == s with the generator's p = (k - s) / 2
        // leaves s undetermined, so the JP-Extra pairs (k, s) = (16, 8), (8, 2), (2, 2) are assumed
        int s = 0;
        if (const OnnxNode* nd = node_of_layer(g, "ConvTranspose", "dec.ups." + std::to_string(i)))
            if (const OnnxAttr* a = nd->attr("strides"))
                if (!a->ints.empty()) s = (int)a->ints[0];
        if (!s) s = k == 16 ? 8 : (k >= 4 ? k / 4 : k);
        rates += (i ? "," : "") + std::to_string(s);
        kernels += (i ? "," : "") + std::to_string(k);
    }
    for (int j = 0; j < nk; ++j) {
        const int nd_ = count_indexed(t, "dec.resblocks." + std::to_string(j) + ".convs1.", ".weight");
        const auto& wd = D("dec.resblocks." + std::to_string(j) + ".convs1.0.weight");
        rks += (j ? "," : "") + std::to_string(wd[2]);
        rds += std::string(j ? "," : "") + "[";
        for (int q = 0; q < nd_; ++q) {
            int d = 2 * q + 1;   // ResBlock1's (1, 3, 5) unless the Conv node says otherwise
            if (const OnnxNode* nd = node_of_layer(g, "Conv", "dec.resblocks." + std::to_string(j) + ".convs1." + std::to_string(q)))
                if (const OnnxAttr* a = nd->attr("dilations"))
                    if (!a->ints.empty() && a->ints[0] >= 1 && a->ints[0] <= 64) d = (int)a->ints[0];
            rds += (q ? "," : "") + std::to_string(d);
        }
        rds += "]";
    }
    rates += "]"; kernels += "]"; rks += "]"; rds += "]";
    char js[2048];
    snprintf(js, sizeof js,
             "{\"n_vocab\": %d, \"n_tones\": %d, \"n_langs\": %d, \"n_speakers\": %d, \"hidden\": %d, \"inter\": %d, \"filter\": %d, \"heads\": %d, "
             "\"enc_layers\": %d, \"enc_kernel\": %d, \"window\": %d, \"gin\": %d, \"style_dim\": %d, \"bert_dim\": %d, \"cond_layer_idx\": 2, "
             "\"flow_n\": %d, \"flow_layers\": %d, \"flow_kernel\": %d, \"dp_filter\": %d, \"dp_kernel\": %d, \"sdp_kernel\": %d, \"sdp_flows\": %d, "
             "\"sdp_bins\": %d, \"sdp_tail\": 5.0, \"sdp_dds_layers\": %d, \"up_rates\": %s, \"up_kernels\": %s, \"up_initial\": %d, "
             "\"res_kernels\": %s, \"res_dilations\": %s}",
             n_vocab, n_tones, n_langs, n_speakers, hidden, inter, filter, heads, enc_layers, enc_kernel, window, gin, style_dim, bert_dim, flow_n,
             flow_layers, flow_kernel, dp_filter, dp_kernel, sdp_kernel, sdp_flows, sdp_bins, sdp_dds, rates.c_str(), kernels.c_str(), up_initial,
             rks.c_str(), rds.c_str());
    return to_blob(t, 2, js);
}

// DeBERTa-v2 (AutoModelForMaskedLM, hidden_states[-3]: convert_deberta.py:22-35); names as in transformers' state dict under "deberta."
Blob import_bert_onnx(const uint8_t* bytes, size_t n) {
    const OnnxGraph g = parse_onnx(bytes, n);
    Named t = name_tensors(g);
    // exporters may or may not keep the "model." wrapper prefix of ORTDeberta (convert_deberta.py:22-29): strip it
    {
        Named u;
        for (auto& kv : t.data) {
            std::string nm = kv.first;
            if (nm.find("model.") == 0) nm = nm.substr(6);
            u.put(nm, t.dims[kv.first], std::move(kv.second));
        }
        t = std::move(u);
    }
    auto D = [&](const std::string& nm) -> const std::vector<int64_t>& { return dims_of(t, nm); };
    const int vocab = (int)D("deberta.embeddings.word_embeddings.weight")[0], hidden = (int)D("deberta.embeddings.word_embeddings.weight")[1];
    const int layers = count_indexed(t, "deberta.encoder.layer.", ".attention.self.query_proj.weight");
    SBV2_REQUIRE(layers >= 1, "ONNX import: no deberta.encoder.layer.* found");
    const int inter = (int)D("deberta.encoder.layer.0.intermediate.dense.weight")[0];
    // (an onnxsim-folded file has no rel_embeddings: the projected positions of layer 0 carry the same row count)
    const int span = (int)(t.has("deberta.encoder.rel_embeddings.weight") ? D("deberta.encoder.rel_embeddings.weight")[0]
                                                                            : D("deberta.encoder.layer.0.attention.self.pos_key")[0]) / 2;
    // attention_head_size 64 in every published DeBERTa-v2 config (not recoverable from an unfolded file's weights; a folded one shows the head count)
    const int heads = t.has("deberta.encoder.layer.0.attention.self.pos_key") ? (int)D("deberta.encoder.layer.0.attention.self.pos_key")[1] : hidden / 64;
    int conv_k = 0;
    if (t.has("deberta.encoder.conv.conv.weight")) conv_k = (int)D("deberta.encoder.conv.conv.weight")[2];
    bool has_tanh = false;
    for (const auto& nd : g.nodes) has_tanh = has_tanh || nd.op == "Tanh";
    float eps = 1e-7f;
    for (const auto& nd : g.nodes)
        if (nd.op == "LayerNormalization")
            if (const OnnxAttr* a = nd.attr("epsilon")) {
                eps = a->f;
                break;
            }
    char js[768];
    snprintf(js, sizeof js,
             "{\"vocab_size\": %d, \"hidden\": %d, \"layers\": %d, \"heads\": %d, \"intermediate\": %d, \"position_buckets\": %d, "
             "\"max_relative_positions\": 512, \"ln_eps\": %.9g, \"conv_kernel_size\": %d, \"conv_act\": \"%s\"}",
             vocab, hidden, layers, heads, inter, span, (double)eps, conv_k, has_tanh ? "tanh" : "gelu");
    return to_blob(t, 1, js);
}

// The (style_vectors.json, model.onnx) pair of a `.sbv2` file (sbv2file.rs:15-37); spans point into `storage`
void parse_sbv2file_bytes(const uint8_t* b, size_t n, std::vector<uint8_t>& storage, Span2& style, Span2& onnx) {
    const uint8_t* tarp = b;
    size_t tarn = n;
    if (n >= 4 && b[0] == 0x28 && b[1] == 0xB5 && b[2] == 0x2F && b[3] == 0xFD) {
        storage = zstd_decompress(b, n);
        tarp = storage.data();
        tarn = storage.size();
    } else {
        storage.assign(b, b + n);
        tarp = storage.data();
    }
    SBV2_REQUIRE(looks_like_tar(tarp, tarn), ".sbv2: the decompressed payload is not a tar archive");
    const auto ents = tar_entries(tarp, tarn);
    auto so = ents.find("style_vectors.json"), mo = ents.find("model.onnx");
    if (so == ents.end()) throw Error("model not found: style_vectors");   // Error::ModelNotFoundError (sbv2file.rs:31-33)
    if (mo == ents.end()) throw Error("model not found: vits2");
    style = Span2{tarp + so->second.first, so->second.second};
    onnx = Span2{tarp + mo->second.first, mo->second.second};
}

// Whatever `load_model` may be handed -> named tensors.  want_kind: 1 = DeBERTa, 2 = VITS.
Blob load_model_bytes(const uint8_t* b, size_t n, uint32_t want_kind) {
    SBV2_REQUIRE(b && n >= 1, "model bytes are empty");
    if (n >= 8 && std::memcmp(b, "SBV2W001", 8) == 0) return parse_blob(b, n);
    const bool zst = n >= 4 && b[0] == 0x28 && b[1] == 0xB5 && b[2] == 0x2F && b[3] == 0xFD;
    if (zst || looks_like_tar(b, n)) {
        SBV2_REQUIRE(want_kind == 2, "a .sbv2 container holds a VITS model, not DeBERTa");
        std::vector<uint8_t> storage;
        Span2 style, onnx;
        parse_sbv2file_bytes(b, n, storage, style, onnx);
        return import_vits_onnx(onnx.p, onnx.n);
    }
    // ONNX ModelProto: field 1 (ir_version, varint) comes first in every exporter's output
    if (n >= 16 && b[0] == 0x08) return want_kind == 1 ? import_bert_onnx(b, n) : import_vits_onnx(b, n);
    throw Error("model bytes are neither an SBV2W001 weight container, a .sbv2 (zstd + tar) file nor an ONNX ModelProto");
}

}  // namespace sbv2

// ---- C ABI ---------------------------------------------------------------------------------------------------------------------------
extern "C" {

// sbv2file.rs:15 `parse_sbv2file(bytes) -> (style_vectors, vits2)`: both outputs are owned copies released with sbv2_bytes_free.
int sbv2_parse_sbv2file(const uint8_t* sbv2_bytes, size_t len, uint8_t** style_vectors, size_t* style_len, uint8_t** vits2, size_t* vits2_len) {
    API_BEGIN
    SBV2_REQUIRE(sbv2_bytes && style_vectors && style_len && vits2 && vits2_len, "bad arguments");
    std::vector<uint8_t> storage;
    Span2 st, ox;
    parse_sbv2file_bytes(sbv2_bytes, len, storage, st, ox);
    uint8_t* a = static_cast<uint8_t*>(std::malloc(std::max<size_t>(st.n, 1)));
    uint8_t* b = static_cast<uint8_t*>(std::malloc(std::max<size_t>(ox.n, 1)));
    if (!a || !b) {
        std::free(a);
        std::free(b);
        throw Error("out of host memory");
    }
    std::memcpy(a, st.p, st.n);
    std::memcpy(b, ox.p, ox.n);
    *style_vectors = a;
    *style_len = st.n;
    *vits2 = b;
    *vits2_len = ox.n;
    API_END
}
void sbv2_bytes_free(uint8_t* p) { std::free(p); }

// style.rs:11-17 `load_style`: {"shape": [n, dim], "data": [[...], ...]} -> owned f32 [n][dim] (sbv2_bytes_free)
int sbv2_style_load(const uint8_t* json, size_t len, float** data, int64_t* n, int64_t* dim) {
    API_BEGIN
    SBV2_REQUIRE(json && data && n && dim, "bad arguments");
    const std::string js(reinterpret_cast<const char*>(json), len);
    const std::vector<int> shape = json_int_array(js, "shape");
    SBV2_REQUIRE(shape.size() == 2 && shape[0] >= 1 && shape[1] >= 1, "style vectors: shape must be [n, dim]");
    size_t p = js.find("\"data\"");
    SBV2_REQUIRE(p != std::string::npos, "style vectors: no data");
    p = js.find('[', p);
    std::vector<float> v;
    v.reserve((size_t)shape[0] * shape[1]);
    int depth = 0;
    for (; p < js.size(); ++p) {
        const char c = js[p];
        if (c == '[') ++depth;
        else if (c == ']') {
            if (--depth == 0) break;
        } else if (c == '-' || c == '+' || c == '.' || (c >= '0' && c <= '9')) {
            char* e;
            v.push_back(std::strtof(js.c_str() + p, &e));
            p = (size_t)(e - js.c_str()) - 1;
        } else if (c == 'N' || c == 'I' || c == 'n' || c == 'i') {   // NaN / Infinity as json.dump writes them
            char* e;
            v.push_back(std::strtof(js.c_str() + p, &e));
            if (e == js.c_str() + p) throw Error("style vectors: malformed number");
            p = (size_t)(e - js.c_str()) - 1;
        }
    }
    SBV2_REQUIRE(v.size() == (size_t)shape[0] * shape[1], "style vectors: " + std::to_string(v.size()) + " values do not fill shape [" +
                                                              std::to_string(shape[0]) + ", " + std::to_string(shape[1]) + "]");
    float* o = static_cast<float*>(std::malloc(sizeof(float) * v.size()));
    SBV2_REQUIRE(o, "out of host memory");
    std::memcpy(o, v.data(), sizeof(float) * v.size());
    *data = o;
    *n = shape[0];
    *dim = shape[1];
    API_END
}

// tts.rs:84-124 `load_aivmx`: an .aivmx file IS the VITS ONNX model; its ModelProto.metadata_props (field 14, StringStringEntryProto
// {key = 1, value = 2}) carry "aivm_style_vectors" = base64(.npy of a 2-D float32 array).  The reference reads it through ort's
// ModelMetadata::custom + base64 + npyz (all absent here): restated from the published formats (RFC 4648 base64; NumPy .npy v1 / v2 / v3:
// "\x93NUMPY", version, header length (u16 or u32 LE), a Python dict literal with 'descr', 'fortran_order', 'shape').
namespace {
std::vector<uint8_t> base64_decode(const Span s) {
    std::vector<uint8_t> out;
    out.reserve(s.n / 4 * 3);
    uint32_t acc = 0;
    int bits = 0;
    for (size_t i = 0; i < s.n; ++i) {
        const uint8_t c = s.p[i];
        int v;
        if (c >= 'A' && c <= 'Z') v = c - 'A';
        else if (c >= 'a' && c <= 'z') v = c - 'a' + 26;
        else if (c >= '0' && c <= '9') v = c - '0' + 52;
        else if (c == '+' || c == '-') v = 62;
        else if (c == '/' || c == '_') v = 63;
        else if (c == '=' || c == '\n' || c == '\r' || c == ' ') continue;
        else throw Error("aivmx: invalid base64 in aivm_style_vectors");
        acc = (acc << 6) | (uint32_t)v;
        bits += 6;
        if (bits >= 8) {
            bits -= 8;
            out.push_back((uint8_t)(acc >> bits));
        }
    }
    return out;
}
}  // namespace

int sbv2_aivmx_style_vectors(const uint8_t* aivmx, size_t len, float** data, int64_t* n, int64_t* dim) {
    API_BEGIN
    SBV2_REQUIRE(aivmx && data && n && dim, "bad arguments");
    Span value;
    {
        PB pb(Span{aivmx, len});
        while (!pb.done()) {
            uint32_t f, wt;
            pb.tag(f, wt);
            if (f == 14 && wt == 2) {
                PB e(pb.bytes());
                Span k, v;
                while (!e.done()) {
                    uint32_t g, w2;
                    e.tag(g, w2);
                    if (g == 1 && w2 == 2) k = e.bytes();
                    else if (g == 2 && w2 == 2) v = e.bytes();
                    else e.skip(w2);
                }
                if (k.str() == "aivm_style_vectors") value = v;
            } else {
                pb.skip(wt);
            }
        }
    }
    if (!value.p) throw Error("model not found: aivm_style_vectors (the ONNX metadata has no such key)");
    const std::vector<uint8_t> npy = base64_decode(value);
    SBV2_REQUIRE(npy.size() >= 10 && std::memcmp(npy.data(), "\x93NUMPY", 6) == 0, "aivmx: aivm_style_vectors is not an .npy file");
    const int major = npy[6];
    size_t hlen, hoff;
    if (major == 1) {
        hlen = (size_t)npy[8] | ((size_t)npy[9] << 8);
        hoff = 10;
    } else {
        SBV2_REQUIRE(npy.size() >= 12, "aivmx: truncated .npy header");
        hlen = (size_t)npy[8] | ((size_t)npy[9] << 8) | ((size_t)npy[10] << 16) | ((size_t)npy[11] << 24);
        hoff = 12;
    }
    SBV2_REQUIRE(hoff + hlen <= npy.size(), "aivmx: truncated .npy header");
    const std::string hdr(reinterpret_cast<const char*>(npy.data() + hoff), hlen);
    auto field = [&](const char* key) {
        const size_t p = hdr.find(key);
        if (p == std::string::npos) throw Error(std::string("aivmx: .npy header has no ") + key);
        return hdr.find(':', p) + 1;
    };
    size_t q = field("'descr'");
    const size_t d0 = hdr.find('\'', q), d1 = hdr.find('\'', d0 + 1);
    SBV2_REQUIRE(d0 != std::string::npos && d1 != std::string::npos, "aivmx: malformed .npy descr");
    const std::string descr = hdr.substr(d0 + 1, d1 - d0 - 1);
    SBV2_REQUIRE(descr == "<f4" || descr == "|f4" || descr == "=f4", "aivmx: style vectors must be little-endian float32 (got " + descr + ")");
    q = field("'fortran_order'");
    const bool fortran = hdr.compare(hdr.find_first_not_of(' ', q), 4, "True") == 0;
    q = field("'shape'");
    const size_t s0 = hdr.find('(', q), s1 = hdr.find(')', s0);
    SBV2_REQUIRE(s0 != std::string::npos && s1 != std::string::npos, "aivmx: malformed .npy shape");
    std::vector<int64_t> shape;
    for (size_t i = s0 + 1; i < s1;) {
        if (hdr[i] >= '0' && hdr[i] <= '9') {
            char* e;
            shape.push_back(std::strtoll(hdr.c_str() + i, &e, 10));
            i = (size_t)(e - hdr.c_str());
        } else {
            ++i;
        }
    }
    SBV2_REQUIRE(shape.size() == 2 && shape[0] >= 1 && shape[1] >= 1, "aivmx: expected 2D array");   // the reference panics with this text
    const size_t count = (size_t)shape[0] * (size_t)shape[1];
    SBV2_REQUIRE(count <= (npy.size() - hoff - hlen) / 4, "aivmx: .npy data shorter than its shape");
    float* o = static_cast<float*>(std::malloc(sizeof(float) * count));
    SBV2_REQUIRE(o, "out of host memory");
    const uint8_t* src = npy.data() + hoff + hlen;
    if (!fortran) {
        std::memcpy(o, src, sizeof(float) * count);
    } else {   // column-major on disk -> the row-major [n][dim] table sbv2_style_vector reads
        for (int64_t i = 0; i < shape[0]; ++i)
            for (int64_t j = 0; j < shape[1]; ++j) std::memcpy(o + i * shape[1] + j, src + 4 * ((size_t)j * shape[0] + i), 4);
    }
    *data = o;
    *n = shape[0];
    *dim = shape[1];
    API_END
}

// style.rs:19-28 `get_style_vector`: mean + (style_vectors[style_id] - mean) * weight, mean = row 0
int sbv2_style_vector(const float* style_vectors, int64_t n, int64_t dim, int64_t style_id, float weight, float* out) {
    API_BEGIN
    SBV2_REQUIRE(style_vectors && out && n >= 1 && dim >= 1, "bad arguments");
    SBV2_REQUIRE(style_id >= 0 && style_id < n, "style_id out of range (the reference panics on the slice)");
    for (int64_t i = 0; i < dim; ++i) out[i] = style_vectors[i] + (style_vectors[style_id * dim + i] - style_vectors[i]) * weight;
    API_END
}

// Debug / tests: the named-tensor table an import produces, as an SBV2W001 container (the exact inverse direction of the importer's job:
// lets the tests compare "synthetic ONNX -> import" with the container the same weights were packed into).
int sbv2_debug_import_to_container(const uint8_t* model, size_t len, int kind, uint8_t** out, size_t* out_len) {
    API_BEGIN
    SBV2_REQUIRE(model && out && out_len && (kind == 1 || kind == 2), "bad arguments");
    const Blob b = load_model_bytes(model, len, (uint32_t)kind);
    std::string head("SBV2W001");
    auto put = [&](const void* p, size_t n) { head.append(reinterpret_cast<const char*>(p), n); };
    const uint32_t k = b.kind, nt = (uint32_t)b.tensors.size();
    const uint64_t jl = b.config_json.size();
    put(&k, 4); put(&nt, 4); put(&jl, 8);
    head += b.config_json;
    size_t table = 0;
    for (const auto& kv : b.tensors) table += 2 + kv.first.size() + 4 + 8 * kv.second.dims.size() + 8;
    uint64_t off = (head.size() + table + 63) / 64 * 64;
    std::vector<uint64_t> offs;
    for (const auto& kv : b.tensors) {
        offs.push_back(off);
        off += ((uint64_t)kv.second.numel() * 4 + 63) / 64 * 64;
    }
    size_t i = 0;
    for (const auto& kv : b.tensors) {
        const uint16_t nl = (uint16_t)kv.first.size();
        const uint32_t nd = (uint32_t)kv.second.dims.size();
        put(&nl, 2);
        head += kv.first;
        put(&nd, 4);
        for (int64_t d : kv.second.dims) {
            const uint64_t v = (uint64_t)d;
            put(&v, 8);
        }
        put(&offs[i++], 8);
    }
    uint8_t* buf = static_cast<uint8_t*>(std::calloc(1, (size_t)off));
    SBV2_REQUIRE(buf, "out of host memory");
    std::memcpy(buf, head.data(), head.size());
    i = 0;
    for (const auto& kv : b.tensors) std::memcpy(buf + offs[i++], kv.second.data, (size_t)kv.second.numel() * 4);
    *out = buf;
    *out_len = (size_t)off;
    API_END
}

}  // extern "C"
