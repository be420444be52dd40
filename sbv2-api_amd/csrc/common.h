// Shared host-side declarations for libsbv2_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>

#include <cstdint>
#include <cstdio>
#include <map>
#include <memory>
#include <stdexcept>
#include <exception>
#include <string>
#include <vector>

namespace sbv2 {

void set_last_error(const std::string& msg);

struct Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};

#define HIP_CHECK(expr)                                                                          \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess)                                                                    \
            throw ::sbv2::Error(std::string(#expr) + ": " + hipGetErrorString(_e) + " (" + __FILE__ + ":" + \
                                std::to_string(__LINE__) + ")");                                 \
    } while (0)

#define SBV2_REQUIRE(cond, msg)                                                          \
    do {                                                                                 \
        if (!(cond)) throw ::sbv2::Error(std::string(msg) + " [" #cond "]");             \
    } while (0)

// rocTX ranges around the stages of a forward pass (tracing).  librocprofiler-sdk-roctx.so.1 is dlopen'ed on first use when
// SBV2_ROCTX=1 (rocprofv3 --marker-trace then shows deberta / text_encoder / durations / flow / decoder / gather); otherwise a no-op.
struct TraceRange {
    explicit TraceRange(const char* name);
    ~TraceRange();
    TraceRange(const TraceRange&) = delete;
    TraceRange& operator=(const TraceRange&) = delete;
    bool on;
};
// stderr logging behind SBV2_LOG=1 (the reference logs through env_logger / RUST_LOG: sbv2_api/src/main.rs:84,131-179)
void log_line(const std::string& msg);

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
inline int64_t round_up64(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

// ---------------------------------------------------------------------------------------------
// Device memory: one growable arena per model handle.  Activations are "planes": [C][ld] f32,
// channel-major with the time/token axis contiguous (the layout every kernel here assumes).
// ---------------------------------------------------------------------------------------------
struct Plane {
    float* p = nullptr;
    int C = 0;   // rows (channels)
    int L = 0;   // valid columns
    int ld = 0;  // row pitch in floats (multiple of 4, base 16-byte aligned)
    Plane rows(int c0, int n) const { return Plane{p + (size_t)c0 * ld, n, L, ld}; }
};

class Arena {
  public:
    struct Mark {
        size_t chunk, off;
    };
    Arena() = default;
    Arena(const Arena&) {}   // a copied model context starts with an empty workspace of its own
    Arena& operator=(const Arena&) = delete;
    ~Arena() { release(); }
    // Start a new forward pass: keeps every chunk (same shapes -> same placement, no allocation after the first pass), unless the
    // workspace has grown far beyond what recent passes used (a server sees many shapes: every new maximum appends chunks): then
    // the device chunks are released and the next pass allocates afresh.  Callers have drained the stream before reset().
    void reset();
    size_t used_last_pass() const { return last_used_; }
    void* alloc(size_t bytes);
    Plane plane(int C, int L);
    template <class T>
    T* array(size_t n) { return static_cast<T*>(alloc(n * sizeof(T))); }
    Mark mark() const { return Mark{cur_, chunks_.empty() ? 0 : chunks_[cur_].off}; }
    void rewind(const Mark& m);  // stack discipline: everything allocated after `m` is dead
    // Host -> device copy that never blocks the host: the bytes are staged in pinned memory owned by the arena (valid until
    // the next reset()) and copied asynchronously on `stream`.
    void upload(void* dst, const void* src, size_t bytes, hipStream_t stream);
    // Between begin_uploads() and end_uploads() the copies of upload() are held back and issued together, neighbours merged: buffers that were allocated
    // one after the other from this arena and uploaded in the same order are ONE copy (staging and device placement both step by the 256-byte rounded
    // size; the padding travels along as ZEROS).  PRECONDITION of the deferred mode: an upload covers its whole allocation (up to the 256-byte rounding);
    // an upload of a prefix of a larger buffer followed by an adjacent allocation would have the merged copy zero the buffer's tail inside the rounding
    // gap.  Every deferred upload of the library is `array<T>(n)` + `upload(.., n * sizeof(T))`.  A single-utterance call issued 67 copies of 4 - 1028 bytes, 5 us of GPU time each.  Nothing that reads the
    // buffers may be launched inside the bracket.  (UploadBatch is the scope guard.)
    void begin_uploads() { ++defer_; }
    void end_uploads();
    // recycles the pinned staging only (callers guarantee that every copy issued so far has completed)
    void reset_pinned() {
        for (auto& c : pinned_) c.off = 0;
        pcur_ = 0;
    }
    void release();
    size_t capacity() const;

  private:
    struct Chunk {
        char* base;
        size_t cap, off;
        size_t hi = 0;   // high-water mark of `off` in the current pass
    };
    std::vector<Chunk> chunks_;
    size_t cur_ = 0;
    size_t last_used_ = 0, recent_peak_ = 0;
    std::vector<Chunk> pinned_;
    size_t pcur_ = 0;
    struct Pending {
        char* dst;
        const char* src;
        size_t bytes;
        hipStream_t stream;
    };
    std::vector<Pending> pending_;
    int defer_ = 0;
};
struct UploadBatch {
    Arena& a;
    explicit UploadBatch(Arena& arena) : a(arena) { a.begin_uploads(); }
    ~UploadBatch() noexcept(false) {
        if (std::uncaught_exceptions() == 0) {
            a.end_uploads();
        } else {
            try {
                a.end_uploads();
            } catch (...) {   // already unwinding: the first error is the one reported
            }
        }
    }
    UploadBatch(const UploadBatch&) = delete;
    UploadBatch& operator=(const UploadBatch&) = delete;
};

// ---------------------------------------------------------------------------------------------
// Weight blob ("SBV2W001", see sbv2-api_amd/synth.py)
// ---------------------------------------------------------------------------------------------
struct HostTensor {
    std::vector<int64_t> dims;
    const float* data = nullptr;
    int64_t numel() const {
        int64_t n = 1;
        for (auto d : dims) n *= d;
        return n;
    }
};

struct Blob {
    uint32_t kind = 0;
    std::string config_json;
    std::map<std::string, HostTensor> tensors;
    std::shared_ptr<void> owned;   // storage behind `tensors` when the blob was imported (ONNX / .sbv2) instead of viewing the caller's bytes
    const HostTensor& get(const std::string& name) const;
    bool has(const std::string& name) const { return tensors.count(name) != 0; }
};

Blob parse_blob(const uint8_t* bytes, size_t n);
// import.cpp: whatever the reference's load_model may be handed (SBV2W001 container, .sbv2 = zstd(tar), bare tar, ONNX ModelProto)
struct Span2 {
    const uint8_t* p = nullptr;
    size_t n = 0;
};
Blob load_model_bytes(const uint8_t* bytes, size_t n, uint32_t want_kind);   // 1 = DeBERTa, 2 = VITS
Blob import_vits_onnx(const uint8_t* bytes, size_t n);
Blob import_bert_onnx(const uint8_t* bytes, size_t n);
void parse_sbv2file_bytes(const uint8_t* b, size_t n, std::vector<uint8_t>& storage, Span2& style, Span2& onnx);
// minimal JSON helpers for the flat config object
bool json_has(const std::string& js, const std::string& key);
std::string json_string(const std::string& js, const std::string& key);
double json_number(const std::string& js, const std::string& key);
std::vector<int> json_int_array(const std::string& js, const std::string& key);
std::vector<std::vector<int>> json_int_array2(const std::string& js, const std::string& key);

// ---------------------------------------------------------------------------------------------
// The grouped implicit-GEMM convolution (gemm_conv.hip): C[m][n] (+)= sum_tap sum_k A_tap[k][m] * pre(B[k][n + shift_tap])
// ---------------------------------------------------------------------------------------------
constexpr int kMaxTaps = 12;
constexpr int kMaxPhases = 8;

struct GemmGroup {
    int64_t a_off, b_off, c_off, r_off;  // element offsets added to A, B, C, R
    int M, N, K;
    int nb;  // valid B columns for this group: B[k][j] is read as 0 unless 0 <= j < nb
};

enum Act { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU = 2, ACT_TANH = 3 };
enum BiasMode { BIAS_NONE = 0, BIAS_ROW = 1, BIAS_COL = 2 };

struct ConvParams {
    const float* A = nullptr;  // [tap][K][lda]: k-major, m contiguous
    int lda = 0;
    int64_t a_tap_stride = 0;
    const float* B = nullptr;  // [K][ldb]: k-major, n contiguous
    int ldb = 0;
    float* C = nullptr;
    int ldc = 0;
    int M = 0, N = 0, K = 0;
    int nb = 0;  // valid B columns (zero fill outside)
    int ntaps = 1;
    int shift[kMaxTaps] = {0};
    const float* bias = nullptr;
    int bias_mode = BIAS_NONE;
    float pre_slope = 1.0f;  // leaky-ReLU slope applied to B while staging (1 = identity)
    int act = ACT_NONE;
    float alpha = 1.0f, beta = 1.0f;
    const float* R = nullptr;  // residual, same indexing as C
    int ldr = 0;
    int accumulate = 0;                    // C += result
    const unsigned char* mask = nullptr;   // output column c is kept iff mask[c / mask_div]
    int mask_div = 1;
    int out_stride = 1;      // polyphase transposed conv: column = n * out_stride + phase_off[m / phase_rows]
    int phase_rows = 1 << 30;  // row = m % phase_rows
    int phase_off[kMaxPhases] = {0};
    const GemmGroup* groups = nullptr;  // device pointer; when set, blockIdx.z selects the group
    int ngroups = 1;
    int maxM = 0, maxN = 0;  // grid extents when grouped
    double flops_hint = 0;   // algorithmic FLOP of a grouped launch (profiling only)
};

// hipFuncSetAttribute applies to the CURRENT device's copy of a kernel; a process that drives several GPUs (sbv2_node_*, one host thread per
// device) must raise the dynamic-LDS limit once per (kernel, device), not once per kernel.  `done` is the caller's static per-kernel mask.
int device_cu_count();   // compute units of the current device (hipDeviceProp, cached per device): the small-grid thresholds are in units of it
inline void allow_full_lds(const void* kernel, std::atomic<uint64_t>& done) {
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    const uint64_t bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return;
    HIP_CHECK(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    done.fetch_or(bit, std::memory_order_release);
}

void launch_conv(const ConvParams& p, hipStream_t stream);
int set_skinny_max(int workgroups);   // returns the previous threshold
int small_grid_max();                 // workgroup count under which the small-grid kernels (gemm_skinny, conv_cl_small) take over; 0 = never
bool launch_gemm_skinny(const ConvParams& p, int mask_shift, hipStream_t stream);
bool launch_gemm_skinny_taps(const ConvParams& p, int mask_shift, hipStream_t stream);   // gemm_skinny.hip: small-grid 1x1 products
void conv_prof_begin();
std::string conv_prof_end();
bool conv_prof_active();
void conv_prof_add(int cfg, double flops, hipEvent_t e0, hipEvent_t e1);

// ---------------------------------------------------------------------------------------------
// Channels-last bf16 / split-bf16 convolution (conv_cl.hip): X[pos][C] f32 in HBM, weights pre-packed as MFMA fragments
// ---------------------------------------------------------------------------------------------
struct ConvClParams {
    const float* X = nullptr;  // [NB][ldx]
    int ldx = 0, NB = 0;
    const void* W = nullptr;   // bf16 fragment blocks [chunk][mtile][tap][part][64 lanes][8]
    int nmt = 0, tm = 2, split = 1;
    int f16 = 0;               // fragments and window are fp16 instead of bf16 (exclusive with split)
    int M = 0, N = 0, K = 0, ntaps = 1;
    int shift[kMaxTaps] = {0};
    float* Y = nullptr;
    int ldy = 0;
    const float* bias = nullptr;
    const float* R = nullptr;
    int ldr = 0;
    float pre_slope = 1.0f, beta = 1.0f;
    int accumulate = 0;
    const unsigned char* mask = nullptr;
    int mask_div = 1;
    int out_stride = 1, phase_rows = 1 << 30;
    int phase_off[kMaxPhases] = {0};
    int in_km = 0, out_km = 0;  // operand / result layout: 0 = channels-last [pos][C], 1 = k-major plane [C][ld]
    int act = ACT_NONE;         // k-major output only
    float alpha = 1.0f;         // k-major output only
    // channels-last output only: the result ALSO as chunk-major bf16 hi / lo planes of lrelu(result, ys_slope) (SplitClPlanes below: the operand format of
    // conv_clx.hip), indexed by OUTPUT position: the transposed convolution of a wide decoder stage writes the ResBlocks' first operand itself
    void* ys_p = nullptr;
    int64_t ys_rows = 0;        // front + N + back rows of one (chunk, part) plane
    int ys_front = 0;
    float ys_slope = 1.0f;
};
void launch_conv_cl(const ConvClParams& p, hipStream_t stream);
bool conv_cl_parts_ok(const ConvClParams& p);   // launch_conv_cl(p) would honour p.ys_p
bool launch_conv_cl_small(const ConvClParams& p, int mask_shift, hipStream_t stream);   // conv_cl_small.hip
void launch_conv_cl_diag(const ConvClParams& p, int abl, unsigned long long* stamps, hipStream_t stream);   // diagnostics (clock stamps + ablations)

// ---------------------------------------------------------------------------------------------
// Split-bf16 1x1 products on k-major planes (gemm_bfs.hip): Y[m][n] = epi(sum_k W[m][k] X[k][n]) on the bf16 matrix cores with both
// operands PRE-SPLIT into bf16 parts (2 parts = hi + lo, three MFMAs per product, relative error ~2^-16: "bf16x3"; 3 parts = hi + mid + lo,
// six MFMAs, the dropped terms are 2^-24: f32-grade, "bf16x6").  The activation parts are written by the producer of the plane (LayerNorm,
// the previous product's epilogue, split_planes), so the GEMM loop has no conversion: both tiles go L2 -> LDS by LDS-DMA.
// ---------------------------------------------------------------------------------------------
// A third operand format, "f16x3" (parts code 4): two f16 planes, hi = f16(x) and lo = f16((x - hi) * 2^11) (the residual scaled back into the
// normal range, so the pair carries 22 mantissa bits whatever x's magnitude), three MFMAs per product: hi*hi into the accumulator,
// lo*hi + hi*lo into a second one that the epilogue adds with the factor 2^-11.  Dropped term 2^-22 relative; exponent range of f16
// (values beyond +-65504 saturate: the reference's own fp16 BERT option, model.rs:11-17, has the same range).
constexpr int kPartsF16x3 = 4;                 // parts CODE (set_bfs_parts, alloc_split, SBV2_BERT_GEMM): 2 = bf16x3, 3 = bf16x6, 4 = f16x3
constexpr float kF16LoScale = 2048.0f;
inline int split_nplanes(int code) { return code == kPartsF16x3 ? 2 : code; }
struct SplitPlanes {        // [part][C][ld] 16-bit, the column axis contiguous; same column indexing as the f32 plane it mirrors
    void* p = nullptr;
    int parts = 0, C = 0, L = 0, ld = 0;   // parts = number of planes
    int f16 = 0;            // 0: bf16 parts (each the rounding of what the previous ones left), 1: the f16 hi / scaled-lo pair
    int64_t pstride = 0;    // elements from one part to the next
    unsigned long long* sat = nullptr;   // f16 pair only (f16x3_sat_counter(): on unless SBV2_F16X3_SATCOUNT=0): counts values the split clamped
    int code() const { return f16 ? kPartsF16x3 : parts; }
    SplitPlanes rows(int c0, int n) const {
        SplitPlanes s = *this;
        s.p = static_cast<char*>(p) + (size_t)c0 * ld * 2;
        s.C = n;
        return s;
    }
};
// Device counter of f16x3 saturations for the current device, or null when counting is off (SBV2_F16X3_SATCOUNT=0; ON by default since round 6: the
// atomic only fires on a clamp).  The f16 pair has f16's exponent range: a finite value beyond +-65504 is clamped by the split (NaN / Inf propagate as on
// the f32 path); with synthetic O(1) weights it never happens, a real DeBERTa checkpoint with outlier channels would be clamped silently without it
// (fallback: SBV2_BERT_GEMM=bf16x6, bf16's range).
unsigned long long* f16x3_sat_counter();
// Per model handle: copies the device's count to a pinned word behind the work queued on `stream` (enqueue), and once the caller has synchronised that
// stream prints ONE stderr warning per handle the first time the count is non-zero (check; independent of SBV2_LOG).  Returns the count it saw.
class SatWatch {
public:
    SatWatch() = default;
    SatWatch(const SatWatch&) {}                           // a cloned execution context watches (and warns) on its own
    SatWatch& operator=(const SatWatch&) { return *this; }
    ~SatWatch();
    void baseline();                      // at handle creation: what the device has counted so far belongs to other handles
    void enqueue(hipStream_t stream);
    unsigned long long check(const char* who);
    bool warned() const { return warned_; }
private:
    unsigned long long* host_ = nullptr;
    unsigned long long base_ = 0;
    bool armed_ = false, warned_ = false;
};
int f16x3_sat_enable(int on);                        // returns the previous setting
void f16x3_sat_prepare();   // creates the current device's counter if counting is on (call at model creation, outside any stream capture)
unsigned long long f16x3_sat_read(bool reset);       // current device
inline SplitPlanes alloc_split(Arena& ar, int code, int C, int L) {   // same pitch rule as Arena::plane; code as in set_bfs_parts
    SplitPlanes s;
    const int parts = split_nplanes(code);
    s.f16 = code == kPartsF16x3;
    s.sat = s.f16 ? f16x3_sat_counter() : nullptr;
    s.parts = parts;
    s.C = C;
    s.L = L;
    s.ld = round_up(L, 64);
    s.pstride = (int64_t)C * s.ld;
    // (+ 32 bytes: k_vits_flash_x3q's clamped 16-byte reads of an utterance shorter than 8 frames may start 4 columns before the end of the last row.  Its
    // LDS-DMA sources are 8-byte aligned - segment starts and pitches are multiples of 4 elements - which the hardware's unaligned access mode serves.)
    s.p = ar.alloc((size_t)parts * C * s.ld * 2 + 32);
    return s;
}
struct BfsWeights {         // W as MFMA A fragments: [K / 16][nmt][parts][64 lanes][8] bf16 (lane l: row 32 mt + (l & 31), k = 16 c + 8 (l >> 5) + j)
    void* w = nullptr;
    int nmt = 0, parts = 0, M = 0, K = 0;
    int f16 = 0;            // as SplitPlanes::f16
};
// Scratch of an execution context for gemm_bfs' small-grid K split: `ws` holds the partial accumulators of one launch (any contents), `counters` one arrival
// counter per output tile, ZERO between launches (the kernel leaves them zero).  Launches that share them must be ordered (one stream).
struct BfsSplitK {
    float* ws = nullptr;
    size_t ws_bytes = 0;
    unsigned* counters = nullptr;
    int ncounters = 0;
};
struct GemmBfsParams {
    BfsWeights W;
    SplitPlanes X;
    int M = 0, N = 0, K = 0;
    float* Y = nullptr;     // f32 result plane [M][ldy] (may be null when only the split copy is wanted)
    int ldy = 0;
    SplitPlanes Ys;         // optional split copy of the result (parts = 0: none)
    int y_rows = 0x7fffffff;   // the f32 plane receives rows < y_rows only ...
    int ys_row0 = 0;           // ... and the split copy rows >= ys_row0 (the flow's q | k | v product: q as f32, k and v as parts for the attention)
    const float* bias = nullptr;   // per row
    int act = ACT_NONE;
    float alpha = 1.0f, beta = 1.0f;
    const float* R = nullptr;
    int ldr = 0;
    const unsigned char* mask = nullptr;   // output column c is kept iff mask[c / mask_div]
    int mask_div = 1;
    BfsSplitK sk;                          // optional scratch for the small-grid K split (gemm_bfs.hip)
};
bool gemm_bfs_usable(const GemmBfsParams& p);
void launch_gemm_bfs(const GemmBfsParams& p, hipStream_t stream);
void split_planes(Plane in, SplitPlanes out, hipStream_t stream);   // out parts = split of `in` (same columns)
#if defined(__HIPCC__)
// device side: the parts of four consecutive values of a row -> the planes of sp at element offset off (8-byte stores)
__device__ __forceinline__ void split_store4(const SplitPlanes& sp, int64_t off, const float (&v)[4]) {
    typedef __bf16 sp_bf16x4 __attribute__((ext_vector_type(4)));
    typedef _Float16 sp_f16x4 __attribute__((ext_vector_type(4)));
    if (sp.f16) {
        sp_f16x4 h, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            // finite values beyond f16's range saturate; NaN stays NaN and an infinity stays one (fmaxf / fminf alone would turn a NaN into 65504)
            const float c = (v[e] != v[e] || fabsf(v[e]) == __builtin_inff()) ? v[e] : fminf(fmaxf(v[e], -65504.f), 65504.f);
            h[e] = (_Float16)c;
            l[e] = (_Float16)((c - (float)h[e]) * kF16LoScale);
        }
        if (sp.sat) {   // (uniform branch; the atomic fires only on a clamp)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (fabsf(v[e]) > 65504.f && fabsf(v[e]) != __builtin_inff()) atomicAdd(sp.sat, 1ull);
        }
        *reinterpret_cast<sp_f16x4*>(static_cast<_Float16*>(sp.p) + off) = h;
        *reinterpret_cast<sp_f16x4*>(static_cast<_Float16*>(sp.p) + sp.pstride + off) = l;
    } else {
        float res[4] = {v[0], v[1], v[2], v[3]};
        for (int pp = 0; pp < sp.parts; ++pp) {
            sp_bf16x4 h;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                h[e] = (__bf16)res[e];
                res[e] -= (float)h[e];
            }
            *reinterpret_cast<sp_bf16x4*>(static_cast<__bf16*>(sp.p) + (int64_t)pp * sp.pstride + off) = h;
        }
    }
}
__device__ __forceinline__ void split_store1(const SplitPlanes& sp, int64_t off, float v) {
    if (sp.f16) {
        const float c = (v != v || fabsf(v) == __builtin_inff()) ? v : fminf(fmaxf(v, -65504.f), 65504.f);
        if (sp.sat && fabsf(v) > 65504.f && fabsf(v) != __builtin_inff()) atomicAdd(sp.sat, 1ull);
        const _Float16 h = (_Float16)c;
        static_cast<_Float16*>(sp.p)[off] = h;
        static_cast<_Float16*>(sp.p)[sp.pstride + off] = (_Float16)((c - (float)h) * kF16LoScale);
    } else {
        float r = v;
        for (int pp = 0; pp < sp.parts; ++pp) {
            const __bf16 h = (__bf16)r;
            static_cast<__bf16*>(sp.p)[(int64_t)pp * sp.pstride + off] = h;
            r -= (float)h;
        }
    }
}
#endif

// ---------------------------------------------------------------------------------------------
// conv_clx.hip: the ResBlock convolutions of the wide decoder stages on pre-split, pre-activated operands (LDS-DMA only, no staging registers)
// ---------------------------------------------------------------------------------------------
constexpr int kClxStampWords = 12;   // sbv2_debug_clx_timeline: words per workgroup
constexpr int kClxFront = 64;    // zero rows in front of every (chunk, part) plane: the left zero padding of the first tile's window
constexpr int kClxBack = 384;    // ... and behind it: right padding + the last position tile's overhang + DMA piece rounding
struct SplitClPlanes {           // bf16 hi / lo of a channels-last activation, chunk-major: [C / 16][2 parts][front + N + back][16] bf16
    void* p = nullptr;
    int C = 0;
    int64_t N = 0;
    int front = 0, back = 0;
};
size_t split_cl_bytes(int C, int64_t N);
SplitClPlanes make_split_cl(void* mem, int C, int64_t N, hipStream_t stream);   // adopts `mem` (split_cl_bytes) and zeroes the halo rows
void split_cl(const float* X, int ldx, int64_t N, int C, float slope, const SplitClPlanes& out, hipStream_t stream);   // out = split(lrelu(X))
void split_cl_km(Plane x, float slope, const SplitClPlanes& out, hipStream_t stream);   // the same from a k-major plane [C][ld]
struct ConvClxParams {
    SplitClPlanes X;            // operand: bf16 parts of the ACTIVATED input
    const void* W = nullptr;    // pack_cl fragments (split-bf16: parts = 2), nmt row tiles of 32
    int nmt = 0;
    int M = 0, N = 0, K = 0, ntaps = 1;
    int shift0 = 0, shift_step = 1;   // tap t reads position + shift0 + t * shift_step
    float* Y = nullptr;         // f32 result [N][ldy] (optional)
    int ldy = 0;
    float* Ykm = nullptr;       // ... or a k-major f32 result plane [M][ldykm] (then Y, Ys, R, accumulate are unused; Rkm is its residual plane)
    int ldykm = 0;
    const float* Rkm = nullptr;
    int ldrkm = 0;
    SplitClPlanes Ys;           // bf16 parts of lrelu(result, ys_slope) (optional: p == nullptr)
    float ys_slope = 1.0f;
    const float* bias = nullptr;
    const float* R = nullptr;   // residual [N][ldr]
    int ldr = 0;
    float beta = 1.0f;
    int accumulate = 0;         // Y += result
    const unsigned char* mask = nullptr;   // position n is kept iff mask[n >> mask_shift]
    int mask_shift = -1;
    unsigned long long* stamps = nullptr;  // diagnostics: kClxStampWords per workgroup (sbv2_debug_clx_timeline)
    int variant = 0;            // diagnostics (sbv2_debug_clx_timeline): kernel variant under test in a builder experiment; the library holds variant 0 only
    // Phased output (round 6: the polyphase ConvTranspose1d of the wide decoder stages): row m of the product is output channel m % phase_rows of phase
    // m / phase_rows, and position n of that phase is OUTPUT row n * out_stride + phase_off[phase] of Y / Ys (Ys.C == phase_rows, Ys.N == N * out_stride;
    // phase_rows a multiple of 64).  N, X and the mask stay indexed by the INPUT position: pass mask_shift = (output positions per mask entry) / out_stride.
    // No residual, no accumulate, whole-tile launches (N >= 256).
    int out_stride = 1, phase_rows = 0;    // phase_rows 0: not phased
    int phase_off[kMaxPhases] = {0};
    // phase_group 2: the rows come in 32-row blocks of (two consecutive phases) x (16 channels): block = (phase / 2) * (phase_rows / 16) + channel / 16, row
    // in block = (phase & 1) * 16 + channel % 16; phase_off[2 j + 1] == phase_off[2 j] + 1.  Four lanes of the epilogue then write the 16-channel parts rows
    // of two ADJACENT output rows = 64 contiguous bytes, where the plain order writes 32-byte pieces out_stride rows apart (partial HBM writes: measured 2x
    // the write bytes plus the read-modify-write fetches, profiles/r06q_pmc_hbm_traffic.csv); the f32 rows leave as 64-byte halves of a line.
    int phase_group = 1;
    // phase_tap0[ph]: tap t of phase ph reads position n + shift0 + (phase_tap0[ph] + t) * shift_step: each phase multiplies ITS OWN ntaps input taps (a
    // ConvTranspose1d(k 16, s 8) phase has two, the first four phases one position later than the last four; the union of all phases would be three: a third
    // of the MFMAs on zero weights).  Equal inside a phase_group pair.
    int phase_tap0[kMaxPhases] = {0};
    double prof_flops = 0.0;               // > 0: the launch's algorithmic FLOP for sbv2_prof_* (a phased launch multiplies zero padding taps too)
};
int64_t clx_grid_workgroups(const ConvClxParams& p);   // workgroups launch_conv_clx starts for p
bool conv_clx_usable(const ConvClxParams& p);
bool clx_enabled();   // decoder_cl.cpp: the wide decoder stages take conv_clx (default) or conv_cl
int set_clx(int on);  // returns the previous setting
int set_upx(int on);  // the wide stages' transposed convolutions as phased conv_clx launches (default 1; read when the weights are packed: 2 = their rows in plain (phase, channel) order, 3 = odd tap counts, a zero tap behind the last one); returns the previous setting
int upx_mode();
// gemm_bfs: small grids split their K loop over groups of waves (another summation order than the batch's tiles; 0 = the unsplit, batch-order dispatch)
bool ksplit_enabled();
int set_ksplit(int on);  // returns the previous setting
bool clx_wanted(int64_t tiles, int64_t min_tiles);   // mode 1: launches of >= min_tiles tiles; mode 2: always; mode 0: never
void launch_conv_clx(const ConvClxParams& p, hipStream_t stream);

// One fused ResBlock1 step y' = beta * (conv2(lrelu(conv1(lrelu(y), dil) + b1)) + b2 + y) on a channels-last plane (respair_cl.hip)
struct ResPairParams {
    const float* X = nullptr;   // y  [N][C]
    float* Y = nullptr;         // y' [N][C] (may be a different buffer; += when accumulate)
    const void* W1 = nullptr;   // conv_cl fragment blocks (tm = 1)
    const void* W2 = nullptr;
    const void* W1p = nullptr;  // C = 16 only: the same weights packed as tap pairs for v_mfma_f32_16x16x32_bf16 (pack_cl_pairs; respair_clx.hip)
    const void* W2p = nullptr;
    const void* W1x = nullptr;  // C = 32 / 64: the same weights packed as step pairs for v_mfma_f32_16x16x32_bf16 (pack_step_pairs; respair_x16.hip)
    const void* W2x = nullptr;
    const float* b1 = nullptr;
    const float* b2 = nullptr;
    int C = 0, N = 0, k = 1, dil = 1, split = 1, f16 = 0;
    float slope = 0.1f, beta = 1.0f;
    int accumulate = 0;
    const unsigned char* mask = nullptr;
    int mask_div = 1;
    int mask_shift = -1;   // set by launch_respair_cl
    int alias_x2 = 1;      // set by launch_respair_cl: the intermediate window re-uses the conv1 window's LDS
    int abl = 0;           // diagnostics (wrong results): 1 = every global read hits the same few cache-hot rows, 2 = no global stores; diag kernel only: 4 = no MFMAs, 8 = no conv1 window conversion, 16 = no intermediate epilogue
    unsigned long long* stamps = nullptr;   // diag kernel only: 16 per workgroup
    int nt_store = 0;      // set by launch_respair_clx: the plane does not fit the caches, its stores bypass them
    int rres_late = 0;     // respair_clx diagnostics only (sbv2_debug_respair_clock): request the residual rows before conv2 (rounds 1-3) instead of with the window
};
void launch_respair_cl(const ResPairParams& p, hipStream_t stream);
void launch_respair_cl_diag(const ResPairParams& p, hipStream_t stream);
// respair_clx.hip: the same step, split-bf16, k in {3, 7, 11}, rebuilt around its instruction count (round 4); bit-identical to respair_cl at C = 32 / 64,
// f32-grade (two taps per 16x16x32 MFMA: another summation order; tests hold 1e-5) at C = 16
bool respair_clx_usable(const ResPairParams& p);          // p.mask_shift set
void launch_respair_clx(const ResPairParams& p, hipStream_t stream);
void launch_respair_clx_diag(const ResPairParams& p, hipStream_t stream);
int set_respair_clx(int on);   // returns the previous setting (default 1; 2: respair_clx.hip at every shape, none on respair_x16.hip)
// respair_x16.hip (round 6): the step at C = 32 / 64, k = 7 / 11 on v_mfma_f32_16x16x32_bf16 (conv_clx.hip's operand scheme); f32 rounding apart from respair_clx
bool respair_x16_usable(const ResPairParams& p);          // p.mask_shift set
void launch_respair_x16(const ResPairParams& p, hipStream_t stream);
void launch_respair_x16_diag(const ResPairParams& p, hipStream_t stream);

// resbranch_clx.hip (round 6): the THREE steps of a ResBlock1 branch in one launch (k = 3, C in {16, 32, 64}): y_1 and y_2 never leave the chip, 2 plane passes
// through HBM per branch instead of 6; bit-identical to three respair_clx launches
constexpr int kResBranchSteps = 3;
constexpr int kResBranchMargin = 6;   // window rows a tap may reach beyond the window's ends (>= dilation * (k - 1) / 2)
struct ResBranchParams {
    const float* X = nullptr;   // y_0 [N][C]
    float* Y = nullptr;         // beta * y_3 [N][C] (+= when accumulate)
    const void* W[2 * kResBranchSteps] = {};   // conv1, conv2 of step 0, conv1, conv2 of step 1, ...: pack_cl fragment blocks (C >= 32) / pack_cl_pairs (C = 16)
    const float* b[2 * kResBranchSteps] = {};
    int C = 0, N = 0, k = 3;
    int dil[kResBranchSteps] = {1, 1, 1};      // dilation of each step's conv1 (conv2: 1)
    float slope = 0.1f, beta = 1.0f;
    int accumulate = 0;
    const unsigned char* mask = nullptr;       // position n is kept iff mask[n >> mask_shift]
    int mask_shift = -1;
    int halo = 0;               // set by launch_resbranch: sum over the steps of (dil + 1) * (k - 1) / 2
    int nt_store = 0;           // set by launch_resbranch
    unsigned long long* stamps = nullptr;      // diagnostics: 16 per workgroup (sbv2_debug_resbranch_clock)
};
bool resbranch_usable(const ResBranchParams& p);
bool resbranch_enabled();
bool resbranch_wanted(int C, int k);   // mode 1 (default): every shape resbranch_clx.hip instantiates; 2: its k = 3 branches only
int set_resbranch(int on);      // returns the previous setting (default 1; SBV2_RESBRANCH=0)
void launch_resbranch(const ResBranchParams& p, hipStream_t stream);

}  // namespace sbv2
