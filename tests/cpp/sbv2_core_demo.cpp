// Test driver for include/sbv2_core.hpp (the C++ mirror of model.rs / bert.rs): reads raw little-endian files written by
// tests/test_gpu_parity.py::test_cpp_host_mirror, runs load_model -> predict -> synthesize, writes the results back.
//   usage: sbv2_core_demo <dir>          files: bert.blob vits.blob ids.i64 mask.i64 bertori.f32 x.i64 tones.i64 langs.i64 style.f32
//   writes: predict.f32 ([S][hidden]) pcm.f32; exit code 0 = ok, 3 = the error-path checks failed.
#include <cstdio>
#include <fstream>
#include <iostream>

#include "sbv2_core.hpp"

template <class T>
static std::vector<T> slurp(const std::string& path) {
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) throw std::runtime_error("cannot open " + path);
    const std::streamsize n = f.tellg();
    f.seekg(0);
    std::vector<T> v((size_t)n / sizeof(T));
    f.read(reinterpret_cast<char*>(v.data()), n);
    return v;
}
template <class T>
static void dump(const std::string& path, const std::vector<T>& v) {
    std::ofstream f(path, std::ios::binary);
    f.write(reinterpret_cast<const char*>(v.data()), (std::streamsize)(v.size() * sizeof(T)));
}

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    const std::string d = std::string(argv[1]) + "/";
    try {
        // error behaviour first: Result::Err in the reference == exception here, with the library's message
        bool threw = false;
        try {
            const std::vector<uint8_t> junk = {'n', 'o', 'p', 'e'};
            sbv2_core::load_model(junk, false);
        } catch (const sbv2_core::Error& e) {
            threw = std::string(e.what()).find("SBV2W001") != std::string::npos;
        }
        if (!threw) return 3;

        sbv2_core::Session bert = sbv2_core::load_model(slurp<uint8_t>(d + "bert.blob"), true);
        sbv2_core::Session vits = sbv2_core::load_model(slurp<uint8_t>(d + "vits.blob"), false);
        const auto ids = slurp<int64_t>(d + "ids.i64"), mask = slurp<int64_t>(d + "mask.i64");
        const sbv2_core::Array2f h = sbv2_core::predict(bert, ids, mask);
        dump(d + "predict.f32", h.data);

        threw = false;
        try {
            sbv2_core::predict(vits, ids, mask);   // wrong graph
        } catch (const sbv2_core::Error&) {
            threw = true;
        }
        if (!threw) return 3;

        const auto x = slurp<int64_t>(d + "x.i64"), tones = slurp<int64_t>(d + "tones.i64"), langs = slurp<int64_t>(d + "langs.i64");
        sbv2_core::Array2f bert_ori;
        bert_ori.data = slurp<float>(d + "bertori.f32");
        bert_ori.cols = x.size();
        bert_ori.rows = bert_ori.data.size() / x.size();
        const auto style = slurp<float>(d + "style.f32");
        const sbv2_core::Array3f pcm = sbv2_core::synthesize(vits, bert_ori, x, {0}, tones, langs, style, 0.0f, 1.0f, 0.0f, 0.0f);
        if (pcm.d0 != 1 || pcm.d1 != 1) return 3;
        dump(d + "pcm.f32", pcm.data);
        std::printf("S=%zu hidden=%zu T=%zu L=%zu\n", h.rows, h.cols, x.size(), pcm.d2);
    } catch (const std::exception& e) {
        std::cerr << "error: " << e.what() << "\n";
        return 1;
    }
    return 0;
}
