// Shared by the translation units behind the C ABI (api.cpp, node.cpp, stream.cpp): handle structs and the exception -> return-code
// convention.  Not installed; include/sbv2_hip.h is the public contract.
#pragma once
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>

#include "../../include/sbv2_hip.h"
#include "models.h"

namespace sbv2 {
const char* last_error_cstr();
}
using namespace sbv2;

struct sbv2_bert {
    std::unique_ptr<BertModel> m;
};
struct sbv2_vits {
    std::unique_ptr<VitsModel> m;
};
struct sbv2_pipeline {
    sbv2_bert* bert;
    sbv2_vits* vits;
    // execution contexts: context 0 is the caller's pair of handles, the others are clones (shared weights, own stream + arena)
    std::vector<std::unique_ptr<BertModel>> bclones;
    std::vector<std::unique_ptr<VitsModel>> vclones;
    int64_t calls = 0;   // tickets are call numbers 1, 2, ...: ticket t ran on context (t - 1) % depth and is valid until that context is reused
    BertModel& bm(int i) { return i == 0 ? *bert->m : *bclones[i - 1]; }
    VitsModel& vm(int i) { return i == 0 ? *vits->m : *vclones[i - 1]; }
    int contexts() const { return 1 + (int)vclones.size(); }
    // context of a ticket; throws for tickets never issued or already overwritten by a later run on the same context
    int ctx_of(int64_t ticket) const {
        SBV2_REQUIRE(ticket >= 1 && ticket <= calls, "unknown pipeline ticket");
        SBV2_REQUIRE(ticket > calls - contexts(), "stale pipeline ticket: its execution context has been reused by a later run");
        return (int)((ticket - 1) % contexts());
    }
};

#define API_BEGIN try {
#define API_END                                   \
    return 0;                                     \
    }                                             \
    catch (const std::exception& e) {             \
        set_last_error(e.what());                 \
        return 1;                                 \
    }                                             \
    catch (...) {                                 \
        set_last_error("unknown error");          \
        return 1;                                 \
    }

VitsBatch to_batch(const sbv2_batch* b);
// One batch on one execution context: bert::predict -> word2ph repeat (tts_util.rs:129-154) -> model::synthesize; returns once enqueued
void pipeline_run_one(sbv2::BertModel& bm, sbv2::VitsModel& vm, sbv2::VitsBatch v, const int64_t* token_ids, const int64_t* s_lens,
                      const int64_t* word2ph);
