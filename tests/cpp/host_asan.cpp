// CPU-only sanitizer driver for the host code of libsbv2_hip.so that parses untrusted bytes ("host ASan/UBSan build"):
// common.cpp (SBV2W001 container, config JSON helpers) and import.cpp (ONNX protobuf, tar, zstd, style JSON) are compiled with plain g++
// -fsanitize=address,undefined (no device code, no GPU call is reached) and fed (a) the valid files the Python test wrote and (b) a few
// thousand mutations of them: truncations, byte flips, overwritten length fields.  Any outcome but "parsed" or "sbv2::Error" aborts under the
// sanitizers.   usage: host_asan <dir with container.bin vits.onnx bert.onnx model.sbv2 style.json style.aivmx> <iterations>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "../../sbv2-api_amd/csrc/common.h"
#include "../../include/sbv2_hip.h"

namespace sbv2 {
const char* last_error_cstr();
}
using namespace sbv2;

static std::vector<uint8_t> slurp(const std::string& p) {
    std::ifstream f(p, std::ios::binary | std::ios::ate);
    if (!f) throw std::runtime_error("cannot open " + p);
    std::vector<uint8_t> v((size_t)f.tellg());
    f.seekg(0);
    f.read(reinterpret_cast<char*>(v.data()), (std::streamsize)v.size());
    return v;
}

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() {
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return rng_state;
}

// 0 = parsed, 1 = rejected with an Error; anything else is a bug the sanitizers (or the abort below) report
static int try_load(const std::vector<uint8_t>& b, uint32_t kind) {
    try {
        // a copy with no slack behind it: a one-byte over-read is an ASan error
        std::vector<uint8_t> tight(b);
        Blob blob = load_model_bytes(tight.data(), tight.size(), kind);
        size_t n = 0;
        for (const auto& kv : blob.tensors) {
            n += (size_t)kv.second.numel();
            volatile float first = kv.second.data[0], last = kv.second.data[kv.second.numel() - 1];   // touch both ends
            (void)first;
            (void)last;
        }
        (void)json_has(blob.config_json, "hidden");
        return 0;
    } catch (const Error&) {
        return 1;
    } catch (const std::bad_alloc&) {
        return 1;
    } catch (const std::length_error&) {
        return 1;
    }
}

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    const std::string d = std::string(argv[1]) + "/";
    const int iters = atoi(argv[2]);
    struct Case {
        const char* file;
        uint32_t kind;
    } cases[] = {{"container.bin", 2}, {"vits.onnx", 2}, {"bert.onnx", 1}, {"model.sbv2", 2}};
    int parsed = 0, rejected = 0;
    for (const Case& c : cases) {
        const std::vector<uint8_t> good = slurp(d + c.file);
        if (try_load(good, c.kind) != 0) {
            fprintf(stderr, "valid file %s was rejected: %s\n", c.file, last_error_cstr());
            return 3;
        }
        for (int it = 0; it < iters; ++it) {
            std::vector<uint8_t> m(good);
            switch (rnd() % 4) {
                case 0: m.resize((size_t)(rnd() % (m.size() + 1))); break;                                   // truncate
                case 1: for (int k = 0; k < 1 + (int)(rnd() % 8); ++k) m[(size_t)(rnd() % m.size())] ^= (uint8_t)(1u << (rnd() % 8)); break;
                case 2: {                                                                                      // overwrite a length-like field
                    const size_t pos = (size_t)(rnd() % (m.size() > 64 ? 64 + (m.size() - 64) % 4096 : m.size()));
                    const uint64_t v = (rnd() % 3 == 0) ? ~0ull : (rnd() % 2 ? rnd() : rnd() % 65536);
                    std::memcpy(m.data() + pos, &v, std::min<size_t>(8, m.size() - pos));
                    break;
                }
                default: {                                                                                     // splice two halves
                    const size_t cut = (size_t)(rnd() % m.size());
                    std::rotate(m.begin(), m.begin() + (long)cut, m.end());
                }
            }
            if (m.empty()) m.push_back(0);
            (try_load(m, c.kind) == 0 ? parsed : rejected)++;
        }
    }
    // style.rs mirror + sbv2file.rs mirror on valid and damaged inputs
    {
        const std::vector<uint8_t> js = slurp(d + "style.json"), sb = slurp(d + "model.sbv2");
        float* data = nullptr;
        int64_t n = 0, dim = 0;
        if (sbv2_style_load(js.data(), js.size(), &data, &n, &dim) != 0) return 4;
        std::vector<float> out((size_t)dim);
        if (sbv2_style_vector(data, n, dim, n - 1, 0.5f, out.data()) != 0) return 4;
        if (sbv2_style_vector(data, n, dim, n, 0.5f, out.data()) == 0) return 4;
        sbv2_bytes_free(reinterpret_cast<uint8_t*>(data));
        for (int it = 0; it < iters / 4; ++it) {
            std::vector<uint8_t> m(js);
            m.resize((size_t)(rnd() % (m.size() + 1)));
            if (m.empty()) m.push_back('{');
            if (sbv2_style_load(m.data(), m.size(), &data, &n, &dim) == 0) sbv2_bytes_free(reinterpret_cast<uint8_t*>(data));
        }
        uint8_t *a = nullptr, *b = nullptr;
        size_t an = 0, bn = 0;
        if (sbv2_parse_sbv2file(sb.data(), sb.size(), &a, &an, &b, &bn) != 0) return 5;
        sbv2_bytes_free(a);
        sbv2_bytes_free(b);
    }
    // tts.rs:92-108 mirror: metadata_props -> base64 -> .npy, on a valid file (small ONNX with only metadata) and on mutations of it
    {
        const std::vector<uint8_t> av = slurp(d + "style.aivmx");
        float* data = nullptr;
        int64_t n = 0, dim = 0;
        if (sbv2_aivmx_style_vectors(av.data(), av.size(), &data, &n, &dim) != 0 || n < 1 || dim < 1) return 6;
        sbv2_bytes_free(reinterpret_cast<uint8_t*>(data));
        for (int it = 0; it < iters; ++it) {
            std::vector<uint8_t> m(av);
            switch (rnd() % 3) {
                case 0: m.resize((size_t)(rnd() % (m.size() + 1))); break;
                case 1: for (int k = 0; k < 1 + (int)(rnd() % 8); ++k) m[(size_t)(rnd() % m.size())] ^= (uint8_t)(1u << (rnd() % 8)); break;
                default: {   // damage the .npy header inside the base64 text: overwrite a run with valid base64 characters
                    const size_t pos = (size_t)(rnd() % m.size());
                    for (size_t k = pos; k < std::min(m.size(), pos + 1 + (size_t)(rnd() % 24)); ++k) m[k] = (uint8_t)("AQgw/+9z"[rnd() % 8]);
                }
            }
            if (m.empty()) m.push_back(0);
            if (sbv2_aivmx_style_vectors(m.data(), m.size(), &data, &n, &dim) == 0) {
                sbv2_bytes_free(reinterpret_cast<uint8_t*>(data));
                ++parsed;
            } else {
                ++rejected;
            }
        }
    }
    printf("HOST_ASAN_OK parsed=%d rejected=%d\n", parsed, rejected);
    return 0;
}
