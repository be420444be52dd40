// sbv2_core.hpp — C++ host mirror of the reference's operator interface for the hot path, above the C ABI of sbv2_hip.h.
//
// The reference is Rust (crates/sbv2_core); no Rust toolchain exists in the build image, so the host side that a Rust shim would
// provide is mirrored here in C++ with the same names, argument order, shapes and error behaviour:
//
//   reference (file:line)                                              here
//   model::load_model(model_file, bert) -> Result<Session>   model.rs:6   sbv2_core::load_model(bytes, len, bert[, device]) -> Session
//   bert::predict(&mut Session, ids, masks) -> Array2<f32>   bert.rs:6    sbv2_core::predict(Session&, ids, masks) -> Array2f [S, 1024]
//   model::synthesize(&mut Session, bert_ori, x_tst, sid, tones, lang_ids, style_vector, sdp_ratio, length_scale, noise_scale,
//                     noise_scale_w) -> Array3<f32>          model.rs:53  sbv2_core::synthesize(...) -> Array3f [1, 1, L]
//   error::Error (ORT failures -> OrtError, others -> OtherError(String))  error.rs:6-31   sbv2_core::Error (what() = sbv2_last_error())
//
// Header only; link with libsbv2_hip.so.  A Session is move-only and used by one caller at a time (`&mut Session`).
#ifndef SBV2_CORE_HPP
#define SBV2_CORE_HPP

#include <cstdint>
#include <random>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "sbv2_hip.h"

namespace sbv2_core {

// Result<T, Error> of the reference becomes an exception; the message is the library's (error.rs:29-30 OtherError(String)).
struct Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};
inline void check(int rc) {
    if (rc != 0) throw Error(sbv2_last_error());
}

// row-major owned arrays with ndarray's shapes
struct Array2f {
    std::vector<float> data;
    size_t rows = 0, cols = 0;
    float& operator()(size_t r, size_t c) { return data[r * cols + c]; }
    float operator()(size_t r, size_t c) const { return data[r * cols + c]; }
};
struct Array3f {
    std::vector<float> data;
    size_t d0 = 0, d1 = 0, d2 = 0;
};

class Session {
  public:
    Session() = default;
    Session(Session&& o) noexcept { *this = std::move(o); }
    Session& operator=(Session&& o) noexcept {
        if (this != &o) {
            reset();
            bert_ = o.bert_;
            vits_ = o.vits_;
            o.bert_ = nullptr;
            o.vits_ = nullptr;
        }
        return *this;
    }
    Session(const Session&) = delete;
    Session& operator=(const Session&) = delete;
    ~Session() { reset(); }
    bool is_bert() const { return bert_ != nullptr; }
    sbv2_bert* bert() const { return bert_; }
    sbv2_vits* vits() const { return vits_; }

  private:
    friend Session load_model(const uint8_t*, size_t, bool, int);
    void reset() {
        if (bert_) sbv2_bert_destroy(bert_);
        if (vits_) sbv2_vits_destroy(vits_);
        bert_ = nullptr;
        vits_ = nullptr;
    }
    sbv2_bert* bert_ = nullptr;
    sbv2_vits* vits_ = nullptr;
};

// model.rs:6-50.  `bert` selects which of the two graphs the bytes hold, as in the reference (it only changes session options there).
inline Session load_model(const uint8_t* model_file, size_t len, bool bert, int device = 0) {
    Session s;
    if (bert) check(sbv2_bert_create(model_file, len, device, &s.bert_));
    else check(sbv2_vits_create(model_file, len, device, &s.vits_));
    return s;
}
inline Session load_model(const std::vector<uint8_t>& model_file, bool bert, int device = 0) {
    return load_model(model_file.data(), model_file.size(), bert, device);
}

// bert.rs:6-24: token_ids and attention_masks of one sentence -> [S, hidden]
inline Array2f predict(Session& session, const std::vector<int64_t>& token_ids, const std::vector<int64_t>& attention_masks) {
    if (!session.is_bert()) throw Error("predict: the session does not hold the BERT graph");
    if (token_ids.size() != attention_masks.size()) throw Error("predict: token_ids and attention_masks differ in length");
    Array2f out;
    out.rows = token_ids.size();
    out.cols = (size_t)sbv2_bert_hidden(session.bert());
    out.data.resize(out.rows * out.cols);
    check(sbv2_bert_predict(session.bert(), token_ids.data(), attention_masks.data(), (int64_t)token_ids.size(), out.data.data()));
    return out;
}

// model.rs:53-111: bert_ori [1024, T]; x_tst, tones, lang_ids i64[T]; sid i64[1]; style_vector f32[256] -> [1, 1, L].
// noise_seed is not part of the reference signature: it selects the counter-based stream standing in for the graph's RandomNormalLike.
// Like the reference (ONNX Runtime draws fresh noise on every run) the default is a fresh seed per call; pass one for reproducible output.
inline uint64_t fresh_noise_seed() {
    static std::random_device rd;
    return ((uint64_t)rd() << 32) ^ (uint64_t)rd();
}
inline Array3f synthesize(Session& session, const Array2f& bert_ori, const std::vector<int64_t>& x_tst, const std::vector<int64_t>& sid,
                          const std::vector<int64_t>& tones, const std::vector<int64_t>& lang_ids, const std::vector<float>& style_vector,
                          float sdp_ratio, float length_scale, float noise_scale, float noise_scale_w, uint64_t noise_seed = fresh_noise_seed()) {
    if (session.is_bert() || !session.vits()) throw Error("synthesize: the session does not hold the VITS graph");
    const size_t T = x_tst.size();
    if (bert_ori.cols != T || tones.size() != T || lang_ids.size() != T) throw Error("synthesize: sequence lengths differ");
    if (bert_ori.rows != (size_t)sbv2_vits_bert_dim(session.vits())) throw Error("synthesize: bert_ori must be [bert_dim, T]");
    if (style_vector.size() != (size_t)sbv2_vits_style_dim(session.vits())) throw Error("synthesize: style_vector has the wrong length");
    if (sid.size() != 1) throw Error("synthesize: sid must hold one speaker id");
    float* pcm = nullptr;
    int64_t len = 0;
    check(sbv2_vits_synthesize(session.vits(), bert_ori.data.data(), x_tst.data(), tones.data(), lang_ids.data(), (int64_t)T, sid[0],
                               style_vector.data(), sdp_ratio, length_scale, noise_scale, noise_scale_w, noise_seed, &pcm, &len));
    Array3f out;
    out.d0 = out.d1 = 1;
    out.d2 = (size_t)len;
    out.data.assign(pcm, pcm + len);   // the reference copies into an owned Array3 as well (model.rs:108)
    sbv2_pcm_free(pcm);
    return out;
}

}  // namespace sbv2_core

#endif
