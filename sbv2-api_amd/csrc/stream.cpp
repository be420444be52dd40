// Streaming long-form synthesis (BASELINE configs[4]; the reference has no counterpart: it synthesises a sentence per session.run and
// splits long text on '\n' only, crates/sbv2_core/src/tts.rs:290-321).
//
// DeBERTa, the text encoder, the duration predictors and the flow use global attention and run whole-sequence; the HiFi-GAN decoder, 82 %
// of the work and all of the output bytes, is purely convolutional and runs on fixed-size frame windows (chunk + 16-frame halo per side),
// one hipGraph captured per window shape and replayed per chunk (VitsModel::stream_begin / stream_chunk, vits.cpp).  The first PCM is
// available after the flow + ONE chunk instead of after the whole decoder, and the decoder workspace is bounded by the window.
#include "api_internal.h"

struct sbv2_stream {
    sbv2_bert* bert = nullptr;
    sbv2_vits* vits = nullptr;
    int64_t frames = 0, next = 0, chunk = 0;
};

extern "C" {

// One utterance (batch->n must be 1; same inputs as sbv2_pipeline_run).  chunk_frames: frames of PCM per sbv2_stream_next call (hop = 512
// samples per frame; 256 frames = 2.97 s).  *total_samples = length of the whole utterance.  The two handles must not be used for anything
// else until sbv2_stream_end.
int sbv2_stream_begin(sbv2_bert* bert, sbv2_vits* vits, const sbv2_batch* batch, const int64_t* token_ids, const int64_t* s_lens,
                      const int64_t* word2ph, int64_t chunk_frames, sbv2_stream** out, int64_t* total_samples) {
    API_BEGIN
    SBV2_REQUIRE(bert && vits && batch && token_ids && s_lens && word2ph && out, "bad arguments");
    SBV2_REQUIRE(batch->n == 1, "sbv2_stream_begin takes one utterance");
    SBV2_REQUIRE(bert->m->device() == vits->m->device(), "bert and vits handles live on different devices");
    VitsBatch v = to_batch(batch);
    v.skip_decoder = true;
    pipeline_run_one(*bert->m, *vits->m, v, token_ids, s_lens, word2ph);
    std::unique_ptr<sbv2_stream> s(new sbv2_stream);
    s->bert = bert;
    s->vits = vits;
    s->chunk = chunk_frames;
    s->frames = vits->m->stream_begin((int)chunk_frames);
    if (total_samples) *total_samples = s->frames * vits->m->cfg().hop();
    *out = s.release();
    API_END
}

// Next chunk of PCM (utterance order) -> dst (host, capacity samples; >= chunk_frames * hop always suffices).  *n = samples written,
// 0 once the utterance is complete.
int sbv2_stream_next(sbv2_stream* s, float* dst, int64_t capacity, int64_t* n) {
    API_BEGIN
    SBV2_REQUIRE(s && dst && n, "bad arguments");
    *n = 0;
    if (s->next < s->frames) {
        *n = s->vits->m->stream_chunk(s->next, dst, capacity);
        s->next += s->chunk;
    }
    API_END
}

// 1 when the decoder of this stream replays a captured hipGraph (0: eager launches, SBV2_STREAM_GRAPH=0)
int sbv2_stream_uses_graph(const sbv2_stream* s) { return s && s->vits->m->stream_graph_captured() ? 1 : 0; }
// device workspace of the chunk decoder in bytes (bounded by the window, whatever the utterance length)
int64_t sbv2_stream_workspace_bytes(const sbv2_stream* s) { return s ? (int64_t)s->vits->m->stream_workspace_bytes() : -1; }

void sbv2_stream_end(sbv2_stream* s) { delete s; }

}  // extern "C"
