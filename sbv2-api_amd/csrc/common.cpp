// Arena, weight-blob parser, error plumbing (host only).
#include "common.h"

#include <mutex>

#include <dlfcn.h>

#include <cstring>

namespace sbv2 {

static thread_local std::string g_last_error;
void set_last_error(const std::string& msg) { g_last_error = msg; }
const char* last_error_cstr() { return g_last_error.c_str(); }

namespace {
struct Roctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    bool enabled = false;
    Roctx() {
        const char* e = getenv("SBV2_ROCTX");
        if (!e || atoi(e) == 0) return;
        void* h = nullptr;
        for (const char* n : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
            h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (h) break;
        }
        if (!h) return;
        push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
        pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
        enabled = push && pop;
    }
};
Roctx& roctx() {
    static Roctx r;
    return r;
}
}  // namespace
TraceRange::TraceRange(const char* name) : on(roctx().enabled) {
    if (on) roctx().push(name);
}
TraceRange::~TraceRange() {
    if (on) roctx().pop();
}
void log_line(const std::string& msg) {
    static const bool on = getenv("SBV2_LOG") && atoi(getenv("SBV2_LOG")) != 0;
    if (on) fprintf(stderr, "[sbv2_hip] %s\n", msg.c_str());
}

size_t Arena::capacity() const {
    size_t t = 0;
    for (auto& c : chunks_) t += c.cap;
    return t;
}
void Arena::release() {
    for (auto& c : chunks_) (void)hipFree(c.base);
    chunks_.clear();
    cur_ = 0;
    for (auto& c : pinned_) (void)hipHostFree(c.base);
    pinned_.clear();
    pcur_ = 0;
}
void Arena::upload(void* dst, const void* src, size_t bytes, hipStream_t stream) {
    if (!bytes) return;
    const size_t need = (bytes + 255) / 256 * 256;   // the device allocator's rounding: neighbours in the arena are neighbours in the staging buffer
    for (; pcur_ < pinned_.size(); ++pcur_)
        if (pinned_[pcur_].off + need <= pinned_[pcur_].cap) break;
    if (pcur_ == pinned_.size()) {
        Chunk c{nullptr, std::max(need, (size_t)8 << 20), 0};
        HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&c.base), c.cap, hipHostMallocDefault));
        pinned_.push_back(c);
    }
    Chunk& c = pinned_[pcur_];
    std::memcpy(c.base + c.off, src, bytes);
    if (need > bytes) std::memset(c.base + c.off + bytes, 0, need - bytes);   // a merged copy (below) carries this padding to the device: zeros, not stale bytes
    if (defer_ > 0) {
        if (!pending_.empty()) {
            Pending& l = pending_.back();
            const size_t lr = (l.bytes + 255) / 256 * 256;
            if (l.stream == stream && l.dst + lr == static_cast<char*>(dst) && l.src + lr == c.base + c.off) {
                l.bytes = lr + bytes;   // (the rounded tail of the previous buffer is padding on both sides)
                c.off += need;
                return;
            }
        }
        pending_.push_back(Pending{static_cast<char*>(dst), c.base + c.off, bytes, stream});
    } else {
        HIP_CHECK(hipMemcpyAsync(dst, c.base + c.off, bytes, hipMemcpyHostToDevice, stream));
    }
    c.off += need;
}
void Arena::end_uploads() {
    if (--defer_ > 0) return;
    defer_ = 0;
    std::vector<Pending> todo;
    todo.swap(pending_);
    for (const Pending& q : todo) HIP_CHECK(hipMemcpyAsync(q.dst, q.src, q.bytes, hipMemcpyHostToDevice, q.stream));
}
void Arena::reset() {
    // The chunk list is normally kept as it is: a pass with the same shapes replays the same allocation sequence and lands on the same
    // chunks, so after the first pass of a context there is no hipMalloc / hipFree on the path (consolidating here on every pass cost a
    // 9 GB free + malloc + device sync inside the second call of every execution context).
    size_t used = 0;
    for (auto& c : chunks_) used += std::max(c.hi, c.off);
    last_used_ = used;
    recent_peak_ = std::max(used, recent_peak_ - recent_peak_ / 8);   // decays over ~a dozen passes
    if (capacity() > 2 * recent_peak_ + ((size_t)1 << 30)) {
        for (auto& c : chunks_) (void)hipFree(c.base);
        chunks_.clear();
    }
    for (auto& c : chunks_) c.off = c.hi = 0;
    cur_ = 0;
    for (auto& c : pinned_) c.off = 0;  // callers guarantee the previous pass's copies are complete
    pcur_ = 0;
    pending_.clear();   // (a pass that threw inside an upload bracket)
    defer_ = 0;
}
void Arena::rewind(const Mark& m) {
    if (chunks_.empty()) return;
    for (auto& c : chunks_) c.hi = std::max(c.hi, c.off);
    for (size_t i = m.chunk + 1; i < chunks_.size(); ++i) chunks_[i].off = 0;
    chunks_[m.chunk].off = m.off;
    cur_ = m.chunk;
}
void* Arena::alloc(size_t bytes) {
    bytes = (bytes + 255) / 256 * 256;
    for (; cur_ < chunks_.size(); ++cur_) {
        Chunk& c = chunks_[cur_];
        if (c.off + bytes <= c.cap) {
            void* p = c.base + c.off;
            c.off += bytes;
            return p;
        }
    }
    Chunk c{nullptr, std::max(bytes, (size_t)256 << 20), 0, 0};
    HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&c.base), c.cap));
    c.off = bytes;
    chunks_.push_back(c);
    cur_ = chunks_.size() - 1;
    return c.base;
}
Plane Arena::plane(int C, int L) {
    Plane pl;
    pl.C = C;
    pl.L = L;
    pl.ld = round_up(L, 64);
    pl.p = static_cast<float*>(alloc(sizeof(float) * (size_t)C * pl.ld));
    return pl;
}

const HostTensor& Blob::get(const std::string& name) const {
    auto it = tensors.find(name);
    if (it == tensors.end()) throw Error("weight blob has no tensor '" + name + "'");
    return it->second;
}

// Every length / offset field comes from the (possibly user-uploaded) file: bounds are checked in subtraction form (a huge value
// cannot wrap the sum), dims must be positive and < 2^31 (later code narrows them to int) and numel is computed with overflow checks.
Blob parse_blob(const uint8_t* b, size_t n) {
    SBV2_REQUIRE(b && n >= 24 && std::memcmp(b, "SBV2W001", 8) == 0, "model bytes are not an SBV2W001 weight container");
    Blob out;
    uint32_t nt;
    uint64_t jl;
    std::memcpy(&out.kind, b + 8, 4);
    std::memcpy(&nt, b + 12, 4);
    std::memcpy(&jl, b + 16, 8);
    size_t pos = 24;
    SBV2_REQUIRE(jl <= n - pos, "truncated weight container");
    out.config_json.assign(reinterpret_cast<const char*>(b + pos), jl);
    pos += jl;
    for (uint32_t i = 0; i < nt; ++i) {
        SBV2_REQUIRE(n - pos >= 2, "truncated weight container");
        uint16_t nl;
        std::memcpy(&nl, b + pos, 2);
        pos += 2;
        SBV2_REQUIRE(n - pos >= (size_t)nl + 4, "truncated weight container");
        std::string name(reinterpret_cast<const char*>(b + pos), nl);
        pos += nl;
        uint32_t nd;
        std::memcpy(&nd, b + pos, 4);
        pos += 4;
        SBV2_REQUIRE(nd <= 8 && n - pos >= 8 * (size_t)nd + 8, "truncated weight container");
        HostTensor t;
        uint64_t numel = 1;
        for (uint32_t d = 0; d < nd; ++d) {
            uint64_t v;
            std::memcpy(&v, b + pos, 8);
            pos += 8;
            SBV2_REQUIRE(v >= 1 && v < (1ull << 31), "tensor '" + name + "': dimension out of range");
            SBV2_REQUIRE(numel <= (1ull << 40) / v, "tensor '" + name + "': too many elements");
            numel *= v;
            t.dims.push_back((int64_t)v);
        }
        uint64_t off;
        std::memcpy(&off, b + pos, 8);
        pos += 8;
        SBV2_REQUIRE(off % 4 == 0 && off <= n && numel <= (n - off) / 4, "tensor '" + name + "': data out of range");
        t.data = reinterpret_cast<const float*>(b + off);
        out.tensors.emplace(std::move(name), std::move(t));
    }
    return out;
}

static size_t find_key(const std::string& js, const std::string& key) {
    const std::string pat = "\"" + key + "\"";
    size_t p = js.find(pat);
    if (p == std::string::npos) throw Error("config has no key '" + key + "'");
    p = js.find(':', p + pat.size());
    SBV2_REQUIRE(p != std::string::npos, "malformed config json");
    return p + 1;
}
bool json_has(const std::string& js, const std::string& key) { return js.find("\"" + key + "\"") != std::string::npos; }
std::string json_string(const std::string& js, const std::string& key) {
    size_t p = js.find('"', find_key(js, key));
    SBV2_REQUIRE(p != std::string::npos, "config value of '" + key + "' is not a string");
    const size_t e = js.find('"', p + 1);
    SBV2_REQUIRE(e != std::string::npos, "malformed config json");
    return js.substr(p + 1, e - p - 1);
}
double json_number(const std::string& js, const std::string& key) { return std::strtod(js.c_str() + find_key(js, key), nullptr); }
std::vector<int> json_int_array(const std::string& js, const std::string& key) {
    size_t p = js.find('[', find_key(js, key));
    std::vector<int> v;
    ++p;
    while (p < js.size() && js[p] != ']') {
        char* end;
        long x = std::strtol(js.c_str() + p, &end, 10);
        if (end == js.c_str() + p) { ++p; continue; }
        v.push_back((int)x);
        p = end - js.c_str();
    }
    return v;
}
std::vector<std::vector<int>> json_int_array2(const std::string& js, const std::string& key) {
    size_t p = js.find('[', find_key(js, key));
    std::vector<std::vector<int>> out;
    ++p;
    int depth = 1;
    while (p < js.size() && depth > 0) {
        if (js[p] == '[') {
            out.emplace_back();
            ++depth; ++p;
        } else if (js[p] == ']') {
            --depth; ++p;
        } else if ((js[p] >= '0' && js[p] <= '9') || js[p] == '-') {
            char* end;
            long x = std::strtol(js.c_str() + p, &end, 10);
            SBV2_REQUIRE(!out.empty(), "malformed nested int array");
            out.back().push_back((int)x);
            p = end - js.c_str();
        } else {
            ++p;
        }
    }
    return out;
}


int device_cu_count() {
    static std::atomic<int> cached[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    int v = cached[dev].load(std::memory_order_relaxed);
    if (v > 0) return v;
    hipDeviceProp_t pr;
    v = hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 256;
    cached[dev].store(v, std::memory_order_relaxed);
    return v;
}

static std::atomic<int> g_ksplit{1};
bool ksplit_enabled() { return g_ksplit.load(std::memory_order_relaxed) != 0; }
int set_ksplit(int on) { return g_ksplit.exchange(on); }

// ---- f16x3 saturation counter (diagnostics; common.h) ----------------------------------------------------------------------------------
// Counting is ON by default since round 6 (the atomic only fires on a clamp; SBV2_F16X3_SATCOUNT=0 turns it off): a real DeBERTa-large checkpoint with an
// outlier activation beyond +-65504 must not be clamped silently (SatWatch below prints the warning).
static std::atomic<int> g_sat_on{getenv("SBV2_F16X3_SATCOUNT") ? atoi(getenv("SBV2_F16X3_SATCOUNT")) : 1};
static std::mutex g_sat_mu;
static std::map<int, unsigned long long*> g_sat_ctr;   // per device
static unsigned long long* sat_ptr(bool create) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(g_sat_mu);
    auto it = g_sat_ctr.find(dev);
    if (it != g_sat_ctr.end()) return it->second;
    if (!create) return nullptr;
    unsigned long long* p = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&p), sizeof(unsigned long long)) != hipSuccess) return nullptr;
    (void)hipMemset(p, 0, sizeof(unsigned long long));
    g_sat_ctr[dev] = p;
    return p;
}
// (the counter is created by f16x3_sat_prepare - model creation, sbv2_debug_f16x3_saturation - never lazily here: alloc_split calls this inside the
// streaming decoder's hipGraph capture, where a hipMalloc is illegal)
unsigned long long* f16x3_sat_counter() { return g_sat_on.load(std::memory_order_relaxed) ? sat_ptr(false) : nullptr; }
void f16x3_sat_prepare() {
    if (g_sat_on.load(std::memory_order_relaxed)) (void)sat_ptr(true);
}
int f16x3_sat_enable(int on) {
    const int prev = g_sat_on.exchange(on);
    if (on) (void)sat_ptr(true);
    return prev;
}
// ---- SatWatch: one stderr warning per model handle the first time the device's clamp count is non-zero (independent of SBV2_LOG) ---------------
SatWatch::~SatWatch() {
    if (host_) (void)hipHostFree(host_);
}
void SatWatch::enqueue(hipStream_t stream) {
    if (warned_) return;
    unsigned long long* ctr = f16x3_sat_counter();
    if (!ctr) return;
    if (!host_) {
        if (hipHostMalloc(reinterpret_cast<void**>(&host_), sizeof(unsigned long long), hipHostMallocDefault) != hipSuccess) {
            host_ = nullptr;
            return;
        }
        *host_ = 0;
    }
    armed_ = hipMemcpyAsync(host_, ctr, sizeof(unsigned long long), hipMemcpyDeviceToHost, stream) == hipSuccess;
}
unsigned long long SatWatch::check(const char* who) {
    if (!armed_ || !host_) return 0;
    armed_ = false;
    const unsigned long long n = *host_;
    if (n < base_) base_ = 0;   // (the device's count was reset in between: sbv2_debug_f16x3_saturation)
    if (n > base_ && !warned_) {
        warned_ = true;
        fprintf(stderr,
                "sbv2_hip WARNING (%s): %llu activation value(s) beyond +-65504 were clamped by the f16x3 operand split (f16 exponent range). "
                "The result is NOT f32-grade for this input. Re-run with SBV2_BERT_GEMM=bf16x6 (bf16 exponent range, twice the matrix work) "
                "and, for the flow, SBV2_FLOW_1X1=bf16x3; sbv2_debug_f16x3_saturation() reads / resets the count.\n",
                who, n - base_);
    }
    return n;
}
void SatWatch::baseline() {
    if (f16x3_sat_counter()) base_ = f16x3_sat_read(false);   // clamps counted on this device before the handle existed are not its own
}

unsigned long long f16x3_sat_read(bool reset) {
    unsigned long long* p = sat_ptr(false);
    if (!p) return 0;
    unsigned long long v = 0;
    HIP_CHECK(hipDeviceSynchronize());
    HIP_CHECK(hipMemcpy(&v, p, sizeof(v), hipMemcpyDeviceToHost));
    if (reset) HIP_CHECK(hipMemset(p, 0, sizeof(v)));
    return v;
}

}  // namespace sbv2
