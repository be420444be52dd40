/* sbv2_ref.c — C / OpenMP restatement of the sbv2_core hot path (TEST INFRASTRUCTURE and bench.py's cpu_baseline; never shipped,
 * never on the product path: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg load liboracle_ref.so).
 *
 * What it restates (same citations as oracle/sbv2_oracle.py, which it is checked against in tests/test_oracle_c.py, which in turn is
 * pinned by the transformers fixtures of tests/golden/):
 *   bert::predict      crates/sbv2_core/src/bert.rs:6-24     graph: scripts/convert/convert_deberta.py:22-35
 *                      = transformers DebertaV2Model hidden_states[-3][0] (modeling_deberta_v2.py, v5.15.0 line numbers below)
 *   model::synthesize  crates/sbv2_core/src/model.rs:53-111  graph: scripts/convert/convert_model.py:89-113
 *                      = style_bert_vits2 SynthesizerTrn.infer (JP-Extra); blocks shared with transformers/models/vits/modeling_vits.py
 *
 * PARITY STATUS: the reference holds no golden vector for this path and ONNX Runtime / the model files are absent, so parity with the
 * real ONNX graphs is unpinned; this file agrees with the numpy oracle, which agrees with `transformers`.
 *
 * This is BASELINE.md §3 "B2": a straightforward fp32 CPU implementation, all host cores (OpenMP), `-O3 -march=native`, register-blocked
 * GEMM micro-kernel written with GCC vector extensions (no BLAS in the image).  It is labelled "port (not onnxruntime)" wherever reported.
 *
 * Layout: every activation is a channel-major plane x[C][L] (row pitch = L).  Weights come from the SBV2W001 container
 * (sbv2-api_amd/synth.py) under the upstream PyTorch state-dict names.  One utterance at a time (the reference is batch 1, model.rs:66-79).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------------------------------------------ */
/* container + config                                                                                                              */
/* ------------------------------------------------------------------------------------------------------------------------------ */
typedef struct {
    char* name;
    int nd;
    int64_t dims[8];
    const float* data;
} Tensor;

typedef struct {
    uint32_t kind;
    char* json;
    int nt;
    Tensor* t;
    uint8_t* bytes; /* owned copy */
} Blob;

static char g_err[512];
const char* sbv2c_last_error(void) { return g_err; }
#define FAIL(...)                                   \
    do {                                            \
        snprintf(g_err, sizeof g_err, __VA_ARGS__); \
        return -1;                                  \
    } while (0)

static void blob_free(Blob* b) {
    if (!b) return;
    for (int i = 0; i < b->nt; ++i) free(b->t[i].name);
    free(b->t);
    free(b->json);
    free(b->bytes);
    memset(b, 0, sizeof *b);
}

static int blob_parse(Blob* out, const uint8_t* src, size_t n) {
    memset(out, 0, sizeof *out);
    if (!src || n < 24 || memcmp(src, "SBV2W001", 8)) FAIL("not an SBV2W001 container");
    uint8_t* b = (uint8_t*)malloc(n);
    if (!b) FAIL("out of memory");
    memcpy(b, src, n);
    out->bytes = b;
    uint32_t nt;
    uint64_t jl;
    memcpy(&out->kind, b + 8, 4);
    memcpy(&nt, b + 12, 4);
    memcpy(&jl, b + 16, 8);
    size_t pos = 24;
    if (jl > n - pos) FAIL("truncated container");
    out->json = (char*)malloc(jl + 1);
    memcpy(out->json, b + pos, jl);
    out->json[jl] = 0;
    pos += jl;
    out->t = (Tensor*)calloc(nt ? nt : 1, sizeof(Tensor));
    out->nt = (int)nt;
    for (uint32_t i = 0; i < nt; ++i) {
        uint16_t nl;
        if (n - pos < 2) FAIL("truncated container");
        memcpy(&nl, b + pos, 2);
        pos += 2;
        if (n - pos < (size_t)nl + 4) FAIL("truncated container");
        Tensor* t = &out->t[i];
        t->name = (char*)malloc(nl + 1);
        memcpy(t->name, b + pos, nl);
        t->name[nl] = 0;
        pos += nl;
        uint32_t nd;
        memcpy(&nd, b + pos, 4);
        pos += 4;
        if (nd > 8 || n - pos < 8 * (size_t)nd + 8) FAIL("truncated container");
        t->nd = (int)nd;
        uint64_t numel = 1;
        for (uint32_t d = 0; d < nd; ++d) {
            uint64_t v;
            memcpy(&v, b + pos, 8);
            pos += 8;
            if (v < 1 || v >= (1ull << 31)) FAIL("bad dimension");
            t->dims[d] = (int64_t)v;
            numel *= v;
        }
        uint64_t off;
        memcpy(&off, b + pos, 8);
        pos += 8;
        if (off % 4 || off > n || numel > (n - off) / 4) FAIL("tensor data out of range");
        t->data = (const float*)(b + off);
    }
    return 0;
}

static const Tensor* blob_find(const Blob* b, const char* name) {
    for (int i = 0; i < b->nt; ++i)
        if (!strcmp(b->t[i].name, name)) return &b->t[i];
    return NULL;
}
static const Tensor* TN(const Blob* b, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
#include <stdarg.h>
static const Tensor* TN(const Blob* b, const char* fmt, ...) {
    char name[256];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(name, sizeof name, fmt, ap);
    va_end(ap);
    const Tensor* t = blob_find(b, name);
    if (!t) {
        fprintf(stderr, "sbv2_ref: missing tensor %s\n", name);
        abort();
    }
    return t;
}
#define W(b, ...) (TN(b, __VA_ARGS__)->data)

static const char* json_at(const char* js, const char* key) {
    char pat[96];
    snprintf(pat, sizeof pat, "\"%s\"", key);
    const char* p = strstr(js, pat);
    if (!p) return NULL;
    p = strchr(p + strlen(pat), ':');
    return p ? p + 1 : NULL;
}
static double json_num(const char* js, const char* key, double dflt) {
    const char* p = json_at(js, key);
    return p ? strtod(p, NULL) : dflt;
}
static int json_ints(const char* js, const char* key, int* out, int cap) {
    const char* p = json_at(js, key);
    if (!p) return 0;
    p = strchr(p, '[');
    if (!p) return 0;
    ++p;
    int n = 0, depth = 1;
    while (*p && depth > 0) {
        if (*p == '[') { ++depth; ++p; }
        else if (*p == ']') { --depth; ++p; }
        else if ((*p >= '0' && *p <= '9') || *p == '-') {
            char* e;
            long v = strtol(p, &e, 10);
            if (n < cap) out[n++] = (int)v;
            p = e;
        } else ++p;
    }
    return n;
}
static int json_str_is(const char* js, const char* key, const char* val) {
    const char* p = json_at(js, key);
    if (!p) return 0;
    p = strchr(p, '"');
    return p && !strncmp(p + 1, val, strlen(val)) && p[1 + strlen(val)] == '"';
}

/* ------------------------------------------------------------------------------------------------------------------------------ */
/* GEMM-shaped convolution core                                                                                                    */
/*   Y[m][n * ostride + ooff] = epi( sum_t sum_k Wt[t][m][k] * Xp[k][n + shift[t]] ),  m < M, n < N                                */
/* Xp is a zero-padded plane with pitch ldx (>= N + max shift + NR slack).  One OpenMP task per (row block, column block).        */
/* ------------------------------------------------------------------------------------------------------------------------------ */
#if defined(__AVX512F__)
#define VL 16
#else
#define VL 8
#endif
typedef float vf __attribute__((vector_size(VL * 4), aligned(4)));
#define MR 6
#define NR (2 * VL)
#define MB 48  /* rows per tile (multiple of MR) */
#define NB 256 /* columns per tile (multiple of NR) */

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU = 2, ACT_TANH = 3 };

typedef struct {
    const float* bias; /* [M] or NULL */
    int act;
    float alpha;        /* v = act(acc + bias) * alpha */
    const float* res;   /* + res[m][col] (pitch ldr) */
    int64_t ldr;
    float beta;         /* v *= beta */
    int accumulate;     /* y += v */
    int ostride, ooff;  /* column = n * ostride + ooff */
} Epi;

static inline float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

/* wp: packed weights [tap][mblk][K][MR] (rows beyond M zero).  */
static float* pack_w(const float* w, int M, int K, int ntaps, int64_t sm, int64_t sk, int64_t st) {
    const int nmb = (M + MR - 1) / MR;
    float* wp = (float*)aligned_alloc(64, (((size_t)ntaps * nmb * K * MR * 4) + 63) / 64 * 64);
    for (int t = 0; t < ntaps; ++t)
        for (int mb = 0; mb < nmb; ++mb)
            for (int k = 0; k < K; ++k)
                for (int r = 0; r < MR; ++r) {
                    const int m = mb * MR + r;
                    wp[(((size_t)t * nmb + mb) * K + k) * MR + r] = m < M ? w[m * sm + k * sk + t * st] : 0.f;
                }
    return wp;
}

static void conv_core(const float* wp, int M, int K, int ntaps, const int* shift, const float* xp, int64_t ldx, int64_t N, float* y,
                      int64_t ldy, const Epi* e) {
    const int nmb = (M + MR - 1) / MR;
    const int mtiles = (M + MB - 1) / MB;
    const int64_t ntiles = (N + NB - 1) / NB;
#pragma omp parallel for collapse(2) schedule(dynamic, 1)
    for (int mt = 0; mt < mtiles; ++mt)
        for (int64_t nt = 0; nt < ntiles; ++nt) {
            const int m0 = mt * MB;
            const int64_t n0 = nt * NB;
            const int mrows = (M - m0 < MB) ? M - m0 : MB;
            const int64_t ncols = (N - n0 < NB) ? N - n0 : NB;
            float ct[MB][NB] __attribute__((aligned(64)));
            for (int mi = 0; mi < mrows; mi += MR) {
                const int mb = (m0 + mi) / MR;
                for (int64_t nj = 0; nj < ncols; nj += NR) {
                    vf acc[MR][2];
                    for (int r = 0; r < MR; ++r) acc[r][0] = acc[r][1] = (vf){0};
                    for (int t = 0; t < ntaps; ++t) {
                        const float* a = wp + ((size_t)t * nmb + mb) * K * MR;
                        const float* b = xp + n0 + nj + shift[t];
                        for (int k = 0; k < K; ++k) {
                            const vf b0 = *(const vf*)(b + (size_t)k * ldx);
                            const vf b1 = *(const vf*)(b + (size_t)k * ldx + VL);
                            for (int r = 0; r < MR; ++r) {
                                const float av = a[k * MR + r];
                                acc[r][0] += av * b0;
                                acc[r][1] += av * b1;
                            }
                        }
                    }
                    for (int r = 0; r < MR; ++r) {
                        *(vf*)&ct[mi + r][nj] = acc[r][0];
                        *(vf*)&ct[mi + r][nj + VL] = acc[r][1];
                    }
                }
            }
            for (int mi = 0; mi < mrows; ++mi) {
                const int m = m0 + mi;
                const float bv = e->bias ? e->bias[m] : 0.f;
                float* yr = y + (size_t)m * ldy;
                const float* rr = e->res ? e->res + (size_t)m * e->ldr : NULL;
                for (int64_t j = 0; j < ncols; ++j) {
                    float v = ct[mi][j] + bv;
                    if (e->act == ACT_RELU) v = v > 0.f ? v : 0.f;
                    else if (e->act == ACT_GELU) v = gelu_f(v);
                    else if (e->act == ACT_TANH) v = tanhf(v);
                    v *= e->alpha;
                    const int64_t col = (n0 + j) * e->ostride + e->ooff;
                    if (rr) v += rr[col];
                    v *= e->beta;
                    if (e->accumulate) v += yr[col];
                    yr[col] = v;
                }
            }
        }
}

/* zero-padded copy of x[C][L] with optional leaky-ReLU: out[C][ldp], data at column pad_l */
static float* pad_plane(const float* x, int C, int64_t L, int64_t pad_l, int64_t pad_r, float slope, int64_t* ldp) {
    const int64_t ld = (L + pad_l + pad_r + NR + 15) / 16 * 16;
    float* p = (float*)aligned_alloc(64, (size_t)C * ld * 4);
#pragma omp parallel for schedule(static)
    for (int c = 0; c < C; ++c) {
        float* r = p + (size_t)c * ld;
        memset(r, 0, (size_t)pad_l * 4);
        const float* s = x + (size_t)c * L;
        if (slope == 1.0f) memcpy(r + pad_l, s, (size_t)L * 4);
        else
            for (int64_t i = 0; i < L; ++i) r[pad_l + i] = s[i] >= 0.f ? s[i] : s[i] * slope;
        memset(r + pad_l + L, 0, (size_t)(ld - pad_l - L) * 4);
    }
    *ldp = ld;
    return p;
}

static Epi epi0(const float* bias) {
    Epi e;
    memset(&e, 0, sizeof e);
    e.bias = bias;
    e.alpha = e.beta = 1.0f;
    e.ostride = 1;
    return e;
}

/* torch.nn.Conv1d on x[Cin][L] -> y[Cout][L + pad_l + pad_r - dil (k-1)]; w[Cout][Cin][k] */
static void conv1d(float* y, const float* x, int Cin, int64_t L, const float* w, const float* bias, int Cout, int k, int dil, int64_t pad_l,
                   int64_t pad_r, float pre_slope, Epi e) {
    int64_t ld;
    float* xp = pad_plane(x, Cin, L, pad_l, pad_r, pre_slope, &ld);
    float* wp = pack_w(w, Cout, Cin, k, (int64_t)Cin * k, k, 1);
    int shift[64];
    for (int t = 0; t < k; ++t) shift[t] = t * dil;
    const int64_t Lout = L + pad_l + pad_r - (int64_t)dil * (k - 1);
    e.bias = bias;
    conv_core(wp, Cout, Cin, k, shift, xp, ld, Lout, y, (e.ostride == 1 ? Lout : Lout * e.ostride), &e);
    free(wp);
    free(xp);
}
static void conv_same(float* y, const float* x, int Cin, int64_t L, const float* w, const float* bias, int Cout, int k, int dil, float pre_slope,
                      Epi e) {
    /* attentions.FFN._same_padding / modules.ResBlock1 get_padding: (k-1)//2 left, k//2 right for dilation 1; (k d - d)/2 both sides else */
    if (dil == 1) conv1d(y, x, Cin, L, w, bias, Cout, k, 1, (k - 1) / 2, k / 2, pre_slope, e);
    else conv1d(y, x, Cin, L, w, bias, Cout, k, dil, (int64_t)dil * (k - 1) / 2, (int64_t)dil * (k - 1) / 2, pre_slope, e);
}
/* y[M][N] = act(w[M][K] x[K][N] + b) : Linear / 1x1 conv; wk = 1 for [M][K], or the conv kernel size when w is [M][K][1] */
static void linear(float* y, const float* x, int K, int64_t N, const float* w, const float* bias, int M, Epi e) {
    conv1d(y, x, K, N, w, bias, M, 1, 1, 0, 0, 1.0f, e);
}

/* torch.nn.ConvTranspose1d, w[Cin][Cout][k], stride s, padding p with k - 2p == s: L_out = L * s.  Polyphase: output phase r uses taps
 * j = s t + r + p in [0, k): y[co][s q + r] = b + sum_ci sum_t w[ci][co][s t + r + p] x[ci][q - t]. */
static void conv_transpose1d(float* y, const float* x, int Cin, int64_t L, const float* w, const float* bias, int Cout, int k, int s, int p,
                             float pre_slope) {
    const int pad = k; /* generous: |t| <= k / s + 1 */
    int64_t ld;
    float* xp = pad_plane(x, Cin, L, pad, pad, pre_slope, &ld);
    for (int r = 0; r < s; ++r) {
        int taps[64], nt = 0;
        for (int t = -k; t <= k; ++t) {
            const int j = s * t + r + p;
            if (j >= 0 && j < k) taps[nt++] = t;
        }
        float* wr = (float*)malloc((size_t)Cout * Cin * nt * 4);
        int shift[64];
        for (int ti = 0; ti < nt; ++ti) {
            shift[ti] = pad - taps[ti];
            for (int co = 0; co < Cout; ++co)
                for (int ci = 0; ci < Cin; ++ci) wr[((size_t)co * Cin + ci) * nt + ti] = w[((size_t)ci * Cout + co) * k + (s * taps[ti] + r + p)];
        }
        float* wp = pack_w(wr, Cout, Cin, nt, (int64_t)Cin * nt, nt, 1);
        Epi e = epi0(bias);
        e.ostride = s;
        e.ooff = r;
        conv_core(wp, Cout, Cin, nt, shift, xp, ld, L, y, L * s, &e);
        free(wp);
        free(wr);
    }
    free(xp);
}

/* LayerNorm over the channel axis of x[C][L] (modules.LayerNorm / torch LayerNorm on the transposed view), biased variance.
 * y = act(LN(x (+ add))) (* colmask) ; in place allowed. */
static void layernorm_ch(float* y, const float* x, const float* add, int C, int64_t L, const float* g, const float* b, float eps, int act,
                         const float* colmask) {
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < L; ++n) {
        float mean = 0.f;
        for (int c = 0; c < C; ++c) mean += x[(size_t)c * L + n] + (add ? add[(size_t)c * L + n] : 0.f);
        mean /= (float)C;
        float var = 0.f;
        for (int c = 0; c < C; ++c) {
            const float d = x[(size_t)c * L + n] + (add ? add[(size_t)c * L + n] : 0.f) - mean;
            var += d * d;
        }
        const float rstd = 1.0f / sqrtf(var / (float)C + eps);
        const float cm = colmask ? colmask[n] : 1.0f;
        for (int c = 0; c < C; ++c) {
            float v = (x[(size_t)c * L + n] + (add ? add[(size_t)c * L + n] : 0.f) - mean) * rstd * g[c] + b[c];
            if (act == ACT_GELU) v = gelu_f(v);
            y[(size_t)c * L + n] = v * cm;
        }
    }
}

static float* falloc(size_t n) {
    float* p = (float*)aligned_alloc(64, ((n ? n : 1) * 4 + 63) / 64 * 64);
    if (!p) {
        fprintf(stderr, "sbv2_ref: out of memory\n");
        abort();
    }
    return p;
}

/* ------------------------------------------------------------------------------------------------------------------------------ */
/* model                                                                                                                           */
/* ------------------------------------------------------------------------------------------------------------------------------ */
typedef struct {
    int vocab, hidden, layers, heads, inter, buckets, max_rel, conv_k, conv_act;
    float eps;
} BertCfg;
typedef struct {
    int n_vocab, n_tones, n_langs, n_speakers, hidden, inter, filter, heads, enc_layers, enc_kernel, window, gin, style_dim, bert_dim,
        cond_layer_idx, flow_n, flow_layers, flow_kernel, dp_filter, dp_kernel, sdp_kernel, sdp_flows, sdp_bins, sdp_dds_layers, up_initial;
    float sdp_tail;
    int n_up, up_rates[8], up_kernels[8], n_res, res_kernels[8], res_dil[8][8], res_nd[8];
} VitsCfg;

typedef struct sbv2c_model {
    Blob bert, vits;
    int has_bert, has_vits;
    BertCfg bc;
    VitsCfg vc;
    float** pos_k; /* per layer [H][2 span]: key_proj(LN(rel_emb)) (input independent: precomputed at load, as on the GPU path) */
    float** pos_q;
} sbv2c_model;

void sbv2c_free(sbv2c_model* m) {
    if (!m) return;
    if (m->pos_k)
        for (int i = 0; i < m->bc.layers; ++i) {
            free(m->pos_k[i]);
            free(m->pos_q[i]);
        }
    free(m->pos_k);
    free(m->pos_q);
    blob_free(&m->bert);
    blob_free(&m->vits);
    free(m);
}

int sbv2c_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    (void)n;
    return 1;
#endif
}

sbv2c_model* sbv2c_load(const uint8_t* bert, size_t nb, const uint8_t* vits, size_t nv) {
    sbv2c_model* m = (sbv2c_model*)calloc(1, sizeof *m);
    if (bert) {
        if (blob_parse(&m->bert, bert, nb) || m->bert.kind != 1) {
            if (!g_err[0]) snprintf(g_err, sizeof g_err, "not a DeBERTa container");
            sbv2c_free(m);
            return NULL;
        }
        m->has_bert = 1;
        const char* js = m->bert.json;
        BertCfg* c = &m->bc;
        c->vocab = (int)json_num(js, "vocab_size", 0);
        c->hidden = (int)json_num(js, "hidden", 0);
        c->layers = (int)json_num(js, "layers", 0);
        c->heads = (int)json_num(js, "heads", 0);
        c->inter = (int)json_num(js, "intermediate", 0);
        c->buckets = (int)json_num(js, "position_buckets", 0);
        c->max_rel = (int)json_num(js, "max_relative_positions", 0);
        c->eps = (float)json_num(js, "ln_eps", 1e-7);
        c->conv_k = (int)json_num(js, "conv_kernel_size", 0);
        c->conv_act = json_str_is(js, "conv_act", "gelu") ? ACT_GELU : (json_str_is(js, "conv_act", "relu") ? ACT_RELU : ACT_TANH);
        /* rel_embeddings -> LayerNorm (modeling_deberta_v2.py:595-599) -> per-layer key_proj / query_proj (share_att_key, :292-299) */
        const int H = c->hidden, span = c->buckets > 0 ? c->buckets : c->max_rel, R = 2 * span;
        const float* re = W(&m->bert, "deberta.encoder.rel_embeddings.weight");
        float* rel = falloc((size_t)H * R); /* plane [H][R] */
        for (int r = 0; r < R; ++r)
            for (int h = 0; h < H; ++h) rel[(size_t)h * R + r] = re[(size_t)r * H + h];
        layernorm_ch(rel, rel, NULL, H, R, W(&m->bert, "deberta.encoder.LayerNorm.weight"), W(&m->bert, "deberta.encoder.LayerNorm.bias"), c->eps,
                     ACT_NONE, NULL);
        m->pos_k = (float**)calloc(c->layers, sizeof(float*));
        m->pos_q = (float**)calloc(c->layers, sizeof(float*));
        for (int i = 0; i < c->layers; ++i) {
            m->pos_k[i] = falloc((size_t)H * R);
            m->pos_q[i] = falloc((size_t)H * R);
            linear(m->pos_k[i], rel, H, R, W(&m->bert, "deberta.encoder.layer.%d.attention.self.key_proj.weight", i),
                   W(&m->bert, "deberta.encoder.layer.%d.attention.self.key_proj.bias", i), H, epi0(NULL));
            linear(m->pos_q[i], rel, H, R, W(&m->bert, "deberta.encoder.layer.%d.attention.self.query_proj.weight", i),
                   W(&m->bert, "deberta.encoder.layer.%d.attention.self.query_proj.bias", i), H, epi0(NULL));
        }
        free(rel);
    }
    if (vits) {
        if (blob_parse(&m->vits, vits, nv) || m->vits.kind != 2) {
            if (!g_err[0]) snprintf(g_err, sizeof g_err, "not a VITS container");
            sbv2c_free(m);
            return NULL;
        }
        m->has_vits = 1;
        const char* js = m->vits.json;
        VitsCfg* c = &m->vc;
#define I(f) c->f = (int)json_num(js, #f, 0)
        I(n_vocab); I(n_tones); I(n_langs); I(n_speakers); I(hidden); I(inter); I(filter); I(heads); I(enc_layers); I(enc_kernel); I(window);
        I(gin); I(style_dim); I(bert_dim); I(cond_layer_idx); I(flow_n); I(flow_layers); I(flow_kernel); I(dp_filter); I(dp_kernel);
        I(sdp_kernel); I(sdp_flows); I(sdp_bins); I(sdp_dds_layers); I(up_initial);
#undef I
        c->sdp_tail = (float)json_num(js, "sdp_tail", 5.0);
        c->n_up = json_ints(js, "up_rates", c->up_rates, 8);
        json_ints(js, "up_kernels", c->up_kernels, 8);
        c->n_res = json_ints(js, "res_kernels", c->res_kernels, 8);
        int flat[64];
        const int nf = json_ints(js, "res_dilations", flat, 64);
        const int per = c->n_res ? nf / c->n_res : 0;
        for (int j = 0; j < c->n_res; ++j) {
            c->res_nd[j] = per;
            for (int q = 0; q < per; ++q) c->res_dil[j][q] = flat[j * per + q];
        }
    }
    return m;
}

/* ------------------------------------------------------------------------------------------------------------------------------ */
/* DeBERTa-v2 (modeling_deberta_v2.py)                                                                                             */
/* ------------------------------------------------------------------------------------------------------------------------------ */
/* make_log_bucket_position (:57-69), float32 like torch */
static int log_bucket(int rel, int bucket, int max_pos) {
    if (bucket <= 0 || max_pos <= 0) return rel;
    const int mid = bucket / 2;
    const float r = (float)rel;
    const float sign = r > 0 ? 1.f : (r < 0 ? -1.f : 0.f);
    const float abs_pos = (rel < mid && rel > -mid) ? (float)(mid - 1) : fabsf(r);
    if (abs_pos <= (float)mid) return rel;
    const float lp = ceilf(logf(abs_pos / (float)mid) / logf((float)(max_pos - 1) / (float)mid) * (float)(mid - 1)) + (float)mid;
    return (int)(lp * sign);
}

int sbv2c_bert(const sbv2c_model* m, const int64_t* ids, const int64_t* mask, int S, float* out /* [S][H] */) {
    if (!m || !m->has_bert) FAIL("no DeBERTa model loaded");
    const BertCfg* c = &m->bc;
    const Blob* B = &m->bert;
    const int H = c->hidden, nh = c->heads, d = H / nh, span = c->buckets > 0 ? c->buckets : c->max_rel, R = 2 * span;
    float* cm = falloc(S);
    for (int i = 0; i < S; ++i) cm[i] = (!mask || mask[i]) ? 1.f : 0.f;
    /* embeddings: word embedding -> LayerNorm -> * mask (:518-559; position_biased_input false, no token types) */
    float* x = falloc((size_t)H * S);
    const float* emb = W(B, "deberta.embeddings.word_embeddings.weight");
    for (int i = 0; i < S; ++i) {
        if (ids[i] < 0 || ids[i] >= c->vocab) {
            free(x);
            free(cm);
            FAIL("token id out of range");
        }
        for (int h = 0; h < H; ++h) x[(size_t)h * S + i] = emb[(size_t)ids[i] * H + h];
    }
    layernorm_ch(x, x, NULL, H, S, W(B, "deberta.embeddings.LayerNorm.weight"), W(B, "deberta.embeddings.LayerNorm.bias"), c->eps, ACT_NONE, cm);
    float* emb_out = NULL;
    if (c->conv_k > 0) {
        emb_out = falloc((size_t)H * S);
        memcpy(emb_out, x, (size_t)H * S * 4);
    }
    float *q = falloc((size_t)H * S), *k = falloc((size_t)H * S), *v = falloc((size_t)H * S), *ctx = falloc((size_t)H * S);
    float *a = falloc((size_t)H * S), *f = falloc((size_t)c->inter * S);
    int* bk = (int*)malloc(sizeof(int) * (2 * S - 1));
    for (int r = -(S - 1); r <= S - 1; ++r) bk[r + S - 1] = log_bucket(r, c->buckets, c->max_rel);
    const float scale = sqrtf((float)d * 3.0f); /* c2p + p2c => scale_factor 3 (:226-232) */
    for (int li = 0; li < c->layers; ++li) {
        linear(q, x, H, S, W(B, "deberta.encoder.layer.%d.attention.self.query_proj.weight", li),
               W(B, "deberta.encoder.layer.%d.attention.self.query_proj.bias", li), H, epi0(NULL));
        linear(k, x, H, S, W(B, "deberta.encoder.layer.%d.attention.self.key_proj.weight", li),
               W(B, "deberta.encoder.layer.%d.attention.self.key_proj.bias", li), H, epi0(NULL));
        linear(v, x, H, S, W(B, "deberta.encoder.layer.%d.attention.self.value_proj.weight", li),
               W(B, "deberta.encoder.layer.%d.attention.self.value_proj.bias", li), H, epi0(NULL));
        const float *pk = m->pos_k[li], *pq = m->pos_q[li];
        /* DisentangledSelfAttention (:193-346): scores = (q k^T + c2p + p2c) / scale, masked_fill(finfo.min), softmax, P v */
#pragma omp parallel for collapse(2) schedule(static)
        for (int h = 0; h < nh; ++h)
            for (int i = 0; i < S; ++i) {
                float sc[S];
                const float* qh = q + (size_t)h * d * S;
                const float* kh = k + (size_t)h * d * S;
                float mx = -INFINITY;
                for (int j = 0; j < S; ++j) {
                    /* c2p index = bucket(i - j) + span (:318-326); p2c index = -bucket(j - i) + span, gathered on the [key][query] block
                     * and transposed (:329-343) */
                    int c2p = bk[i - j + S - 1] + span, p2c = -bk[j - i + S - 1] + span;
                    c2p = c2p < 0 ? 0 : (c2p > R - 1 ? R - 1 : c2p);
                    p2c = p2c < 0 ? 0 : (p2c > R - 1 ? R - 1 : p2c);
                    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
                    for (int e = 0; e < d; ++e) {
                        const float qi = qh[(size_t)e * S + i], kj = kh[(size_t)e * S + j];
                        s0 += qi * kj;
                        s1 += qi * pk[(size_t)(h * d + e) * R + c2p]; /* c2p: q_i . pos_k[bucket(i - j)] (:318-326) */
                        s2 += kj * pq[(size_t)(h * d + e) * R + p2c]; /* p2c: k_j . pos_q[bucket(j - i)] (:329-343) */
                    }
                    float s = (s0 + s1 + s2) / scale;
                    if (!(cm[i] > 0.f && cm[j] > 0.f)) s = -3.4028234663852886e38f;
                    sc[j] = s;
                    mx = s > mx ? s : mx;
                }
                float sum = 0.f;
                for (int j = 0; j < S; ++j) {
                    sc[j] = expf(sc[j] - mx);
                    sum += sc[j];
                }
                const float inv = 1.0f / sum;
                const float* vh = v + (size_t)h * d * S;
                for (int e = 0; e < d; ++e) {
                    float o = 0.f;
                    for (int j = 0; j < S; ++j) o += sc[j] * vh[(size_t)e * S + j];
                    ctx[(size_t)(h * d + e) * S + i] = o * inv;
                }
            }
        linear(a, ctx, H, S, W(B, "deberta.encoder.layer.%d.attention.output.dense.weight", li),
               W(B, "deberta.encoder.layer.%d.attention.output.dense.bias", li), H, epi0(NULL));
        layernorm_ch(a, a, x, H, S, W(B, "deberta.encoder.layer.%d.attention.output.LayerNorm.weight", li),
                     W(B, "deberta.encoder.layer.%d.attention.output.LayerNorm.bias", li), c->eps, ACT_NONE, NULL);
        Epi eg = epi0(NULL);
        eg.act = ACT_GELU;
        linear(f, a, H, S, W(B, "deberta.encoder.layer.%d.intermediate.dense.weight", li), W(B, "deberta.encoder.layer.%d.intermediate.dense.bias", li),
               c->inter, eg);
        linear(x, f, c->inter, S, W(B, "deberta.encoder.layer.%d.output.dense.weight", li), W(B, "deberta.encoder.layer.%d.output.dense.bias", li), H,
               epi0(NULL));
        layernorm_ch(x, x, a, H, S, W(B, "deberta.encoder.layer.%d.output.LayerNorm.weight", li),
                     W(B, "deberta.encoder.layer.%d.output.LayerNorm.bias", li), c->eps, ACT_NONE, NULL);
        if (li == 0 && c->conv_k > 0) {
            /* ConvLayer (:449-475, :664): out = act(conv(embeddings) zeroed at masked tokens); x = LayerNorm(x + out) * mask */
            Epi ec = epi0(NULL);
            ec.act = c->conv_act;
            conv1d(a, emb_out, H, S, W(B, "deberta.encoder.conv.conv.weight"), W(B, "deberta.encoder.conv.conv.bias"), H, c->conv_k, 1,
                   (c->conv_k - 1) / 2, (c->conv_k - 1) / 2, 1.0f, ec);
            for (int h = 0; h < H; ++h)
                for (int i = 0; i < S; ++i) a[(size_t)h * S + i] *= cm[i];
            layernorm_ch(x, x, a, H, S, W(B, "deberta.encoder.conv.LayerNorm.weight"), W(B, "deberta.encoder.conv.LayerNorm.bias"), c->eps, ACT_NONE, cm);
        }
    }
    for (int i = 0; i < S; ++i)
        for (int h = 0; h < H; ++h) out[(size_t)i * H + h] = x[(size_t)h * S + i];
    free(x); free(q); free(k); free(v); free(ctx); free(a); free(f); free(bk); free(cm); free(emb_out);
    return 0;
}

/* ------------------------------------------------------------------------------------------------------------------------------ */
/* VITS2 JP-Extra                                                                                                                  */
/* ------------------------------------------------------------------------------------------------------------------------------ */
/* attentions.MultiHeadAttention with window relative positions (= transformers VitsAttention, modeling_vits.py:844-997), one utterance:
 * scores[i][j] = q_i.k_j / sqrt(dk) + [|j-i| <= w] q_i.erk[j-i+w] / sqrt(dk); out_i = sum_j p_ij v_j + sum_{|j-i|<=w} p_ij erv[j-i+w]. */
static void rel_attention(const Blob* B, const char* p, float* y, const float* x, int C, int64_t T, int heads, int window) {
    const int dk = C / heads;
    char nm[256];
#define WN(s) (snprintf(nm, sizeof nm, "%s%s", p, s), W(B, "%s", nm))
    float *q = falloc((size_t)C * T), *k = falloc((size_t)C * T), *v = falloc((size_t)C * T), *o = falloc((size_t)C * T);
    const float* wq = WN("conv_q.weight"); const float* bq = WN("conv_q.bias");
    linear(q, x, C, T, wq, bq, C, epi0(NULL));
    const float* wk = WN("conv_k.weight"); const float* bk = WN("conv_k.bias");
    linear(k, x, C, T, wk, bk, C, epi0(NULL));
    const float* wv = WN("conv_v.weight"); const float* bv = WN("conv_v.bias");
    linear(v, x, C, T, wv, bv, C, epi0(NULL));
    const float* erk = WN("emb_rel_k"); /* [1][2w+1][dk] */
    const float* erv = WN("emb_rel_v");
    const float qs = 1.0f / sqrtf((float)dk);
    const int nw = 2 * window + 1;
    float* st = falloc((size_t)T * T); /* S^T[j][i] */
    for (int h = 0; h < heads; ++h) {
        const float* qh = q + (size_t)h * dk * T;
        const float* kh = k + (size_t)h * dk * T;
        const float* vh = v + (size_t)h * dk * T;
        /* S^T = K^T Q * qs : A[m = j][kk = e] = kh[e][j] (strides 1, T), B = qh */
        float* wp = pack_w(kh, (int)T, dk, 1, 1, T, 0);
        int64_t ld;
        float* xp = pad_plane(qh, dk, T, 0, 0, 1.0f, &ld);
        int sh0 = 0;
        Epi e = epi0(NULL);
        e.alpha = qs;
        conv_core(wp, (int)T, dk, 1, &sh0, xp, ld, T, st, T, &e);
        free(wp);
        free(xp);
        /* relative-key band + softmax over j (column i of S^T), then the relative-value term */
        float* band = falloc((size_t)nw * T); /* p[i][i + r - w] */
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < T; ++i) {
            for (int r = 0; r < nw; ++r) {
                const int64_t j = i + r - window;
                if (j < 0 || j >= T) continue;
                float s = 0.f;
                for (int e2 = 0; e2 < dk; ++e2) s += qh[(size_t)e2 * T + i] * erk[(size_t)r * dk + e2];
                st[(size_t)j * T + i] += s * qs;
            }
            float mx = -INFINITY;
            for (int64_t j = 0; j < T; ++j) mx = st[(size_t)j * T + i] > mx ? st[(size_t)j * T + i] : mx;
            float sum = 0.f;
            for (int64_t j = 0; j < T; ++j) {
                const float ev = expf(st[(size_t)j * T + i] - mx);
                st[(size_t)j * T + i] = ev;
                sum += ev;
            }
            const float inv = 1.0f / sum;
            for (int64_t j = 0; j < T; ++j) st[(size_t)j * T + i] *= inv;
            for (int r = 0; r < nw; ++r) {
                const int64_t j = i + r - window;
                band[(size_t)r * T + i] = (j < 0 || j >= T) ? 0.f : st[(size_t)j * T + i];
            }
        }
        /* out[e][i] = sum_j vh[e][j] P^T[j][i] */
        wp = pack_w(vh, dk, (int)T, 1, T, 1, 0);
        xp = pad_plane(st, (int)T, T, 0, 0, 1.0f, &ld);
        Epi e2 = epi0(NULL);
        conv_core(wp, dk, (int)T, 1, &sh0, xp, ld, T, o + (size_t)h * dk * T, T, &e2);
        free(wp);
        free(xp);
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < T; ++i)
            for (int e3 = 0; e3 < dk; ++e3) {
                float s = 0.f;
                for (int r = 0; r < nw; ++r) s += band[(size_t)r * T + i] * erv[(size_t)r * dk + e3];
                o[(size_t)(h * dk + e3) * T + i] += s;
            }
        free(band);
    }
    const float* wo = WN("conv_o.weight"); const float* bo = WN("conv_o.bias");
    linear(y, o, C, T, wo, bo, C, epi0(NULL));
#undef WN
    free(q); free(k); free(v); free(o); free(st);
}

/* attentions.Encoder (post-LN), speaker vector spk_emb_linear(g) added before layer cond_layer_idx; x[C][T] in place */
static void encoder(const Blob* B, const VitsCfg* c, const char* p, float* x, int64_t T, const float* g, int n_layers, int kernel) {
    const int C = c->hidden, F = c->filter;
    float *y = falloc((size_t)C * T), *h = falloc((size_t)F * T);
    char nm[256];
    for (int i = 0; i < n_layers; ++i) {
        if (i == c->cond_layer_idx && g) {
            const float* w = W(B, "%sspk_emb_linear.weight", p);
            const float* b = W(B, "%sspk_emb_linear.bias", p);
            for (int ch = 0; ch < C; ++ch) {
                float s = b[ch];
                for (int e = 0; e < c->gin; ++e) s += w[(size_t)ch * c->gin + e] * g[e];
                for (int64_t t = 0; t < T; ++t) x[(size_t)ch * T + t] += s;
            }
        }
        snprintf(nm, sizeof nm, "%sattn_layers.%d.", p, i);
        rel_attention(B, nm, y, x, C, T, c->heads, c->window);
        layernorm_ch(x, x, y, C, T, W(B, "%snorm_layers_1.%d.gamma", p, i), W(B, "%snorm_layers_1.%d.beta", p, i), 1e-5f, ACT_NONE, NULL);
        Epi er = epi0(NULL);
        er.act = ACT_RELU;
        conv_same(h, x, C, T, W(B, "%sffn_layers.%d.conv_1.weight", p, i), W(B, "%sffn_layers.%d.conv_1.bias", p, i), F, kernel, 1, 1.0f, er);
        conv_same(y, h, F, T, W(B, "%sffn_layers.%d.conv_2.weight", p, i), W(B, "%sffn_layers.%d.conv_2.bias", p, i), C, kernel, 1, 1.0f, epi0(NULL));
        layernorm_ch(x, x, y, C, T, W(B, "%snorm_layers_2.%d.gamma", p, i), W(B, "%snorm_layers_2.%d.beta", p, i), 1e-5f, ACT_NONE, NULL);
    }
    free(y);
    free(h);
}

/* modules.DDSConv: x += gelu(LN(1x1(gelu(LN(depthwise_k3(x, dil = 3^i)))))) */
static void dds_conv(const Blob* B, const VitsCfg* c, const char* p, float* x, int64_t T) {
    const int C = c->hidden, k = c->sdp_kernel;
    float *a = falloc((size_t)C * T), *b = falloc((size_t)C * T);
    int dil = 1;
    for (int i = 0; i < c->sdp_dds_layers; ++i) {
        const float* sw = W(B, "%sconvs_sep.%d.weight", p, i);
        const float* sb = W(B, "%sconvs_sep.%d.bias", p, i);
        const int pad = (k * dil - dil) / 2;
        for (int ch = 0; ch < C; ++ch)
            for (int64_t t = 0; t < T; ++t) {
                float s = sb[ch];
                for (int j = 0; j < k; ++j) {
                    const int64_t u = t + (int64_t)j * dil - pad;
                    if (u >= 0 && u < T) s += sw[ch * k + j] * x[(size_t)ch * T + u];
                }
                a[(size_t)ch * T + t] = s;
            }
        layernorm_ch(a, a, NULL, C, T, W(B, "%snorms_1.%d.gamma", p, i), W(B, "%snorms_1.%d.beta", p, i), 1e-5f, ACT_GELU, NULL);
        linear(b, a, C, T, W(B, "%sconvs_1x1.%d.weight", p, i), W(B, "%sconvs_1x1.%d.bias", p, i), C, epi0(NULL));
        layernorm_ch(b, b, NULL, C, T, W(B, "%snorms_2.%d.gamma", p, i), W(B, "%snorms_2.%d.beta", p, i), 1e-5f, ACT_GELU, NULL);
        for (size_t e = 0; e < (size_t)C * T; ++e) x[e] += b[e];
        dil *= k;
    }
    free(a);
    free(b);
}

/* transforms.piecewise_rational_quadratic_transform(inverse=True, tails='linear') for one scalar (modeling_vits.py:93-303, reverse) */
static float spline_inverse1(float x, const float* uw, const float* uh, const float* ud, int nb, float tail) {
    if (!(x >= -tail && x <= tail)) return x;
    const float min_w = 1e-3f, min_h = 1e-3f, min_d = 1e-3f;
    float cw[32], ch[32], wd[32], ht[32], dv[33];
    const float cst = logf(expf(1.0f - min_d) - 1.0f);
    /* softmax -> min + (1 - min nb) * w -> cumsum -> scale to [-tail, tail] */
    for (int pass = 0; pass < 2; ++pass) {
        const float* u = pass ? uh : uw;
        float* cum = pass ? ch : cw;
        float* wid = pass ? ht : wd;
        const float mn = pass ? min_h : min_w;
        float mx = u[0];
        for (int i = 1; i < nb; ++i) mx = u[i] > mx ? u[i] : mx;
        float e[32], sum = 0.f;
        for (int i = 0; i < nb; ++i) {
            e[i] = expf(u[i] - mx);
            sum += e[i];
        }
        float run = 0.f;
        cum[0] = -tail;
        for (int i = 0; i < nb; ++i) {
            run += mn + (1.0f - mn * nb) * (e[i] / sum);
            cum[i + 1] = 2.0f * tail * run - tail;
        }
        cum[nb] = tail;
        for (int i = 0; i < nb; ++i) wid[i] = cum[i + 1] - cum[i];
    }
    for (int i = 0; i <= nb; ++i) {
        const float u = (i == 0 || i == nb) ? cst : ud[i - 1];
        const float sp = u > 20.f ? u : log1pf(expf(u < 20.f ? u : 20.f));
        dv[i] = min_d + sp;
    }
    int bi = 0;
    for (int i = 0; i <= nb; ++i) {
        const float loc = ch[i] + (i == nb ? 1e-6f : 0.f);
        if (x >= loc) bi = i;
    }
    if (bi > nb - 1) bi = nb - 1;
    const float in_cw = cw[bi], in_w = wd[bi], in_ch = ch[bi], in_h = ht[bi], delta = ht[bi] / wd[bi], d0 = dv[bi], d1 = dv[bi + 1];
    const float i1 = d0 + d1 - 2.0f * delta, i2 = x - in_ch, i3 = i2 * i1;
    const float a = in_h * (delta - d0) + i3, b = in_h * d0 - i3, cc = -delta * i2;
    float disc = b * b - 4.0f * a * cc;
    if (disc < 0.f) disc = 0.f;
    const float root = (2.0f * cc) / (-b - sqrtf(disc));
    return root * in_w + in_cw;
}

static void leaky_inplace(float* x, size_t n, float s) {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) x[i] = x[i] >= 0.f ? x[i] : x[i] * s;
}

/* model::synthesize for one utterance.  noise_w: [2][T] already scaled randn (NULL = zeros); prior noise is not supported here
 * (noise_scale 0: the deterministic comparison mode of SURVEY.md §7); forced: optional teacher-forced durations.
 * Outputs: *pcm (malloc'ed, sbv2c_free_pcm), *n_pcm; durations_out / logw_out optional [T]; stage dumps optional. */
int sbv2c_vits(const sbv2c_model* m, const float* bert, const int64_t* phones, const int64_t* tones, const int64_t* langs, int T, int sid,
               const float* style, float sdp_ratio, float length_scale, const float* noise_w, const int64_t* forced, float** pcm, int64_t* n_pcm,
               int64_t* durations_out, float* logw_out, float* x_out /* [H][T] */, float* z_out /* [inter][Tf], cap z_cap */, int64_t z_cap) {
    if (!m || !m->has_vits) FAIL("no VITS model loaded");
    const VitsCfg* c = &m->vc;
    const Blob* B = &m->vits;
    const int H = c->hidden, I = c->inter, G = c->gin;
    if (sid < 0 || sid >= c->n_speakers) FAIL("speaker id out of range");
    const float* g = W(B, "emb_g.weight") + (size_t)sid * G;
    /* TextEncoder: (emb + tone_emb + language_emb + bert_proj(bert) + style_proj(style)) * sqrt(H) -> encoder -> proj */
    float* x = falloc((size_t)H * T);
    linear(x, bert, c->bert_dim, T, W(B, "enc_p.bert_proj.weight"), W(B, "enc_p.bert_proj.bias"), H, epi0(NULL));
    {
        const float *e0 = W(B, "enc_p.emb.weight"), *e1 = W(B, "enc_p.tone_emb.weight"), *e2 = W(B, "enc_p.language_emb.weight");
        const float *sw = W(B, "enc_p.style_proj.weight"), *sb = W(B, "enc_p.style_proj.bias");
        const float sq = sqrtf((float)H);
        for (int t = 0; t < T; ++t)
            if (phones[t] < 0 || phones[t] >= c->n_vocab || tones[t] < 0 || tones[t] >= c->n_tones || langs[t] < 0 || langs[t] >= c->n_langs) {
                free(x);
                FAIL("phone / tone / language id out of range");
            }
        for (int ch = 0; ch < H; ++ch) {
            float sv = sb[ch];
            for (int e = 0; e < c->style_dim; ++e) sv += sw[(size_t)ch * c->style_dim + e] * style[e];
            for (int t = 0; t < T; ++t) {
                /* same summation order as the oracle: ((emb + tone) + lang) + bert_proj, + style, * sqrt(H) */
                float s = e0[(size_t)phones[t] * H + ch] + e1[(size_t)tones[t] * H + ch];
                s += e2[(size_t)langs[t] * H + ch];
                s += x[(size_t)ch * T + t];
                s += sv;
                x[(size_t)ch * T + t] = s * sq;
            }
        }
    }
    encoder(B, c, "enc_p.encoder.", x, T, g, c->enc_layers, c->enc_kernel);
    if (x_out) memcpy(x_out, x, (size_t)H * T * 4);
    float* stats = falloc((size_t)2 * I * T);
    linear(stats, x, H, T, W(B, "enc_p.proj.weight"), W(B, "enc_p.proj.bias"), 2 * I, epi0(NULL));
    const float *m_p = stats, *logs_p = stats + (size_t)I * T;
    (void)logs_p; /* exp(logs_p) * noise: prior noise is 0 here */

    /* per-utterance conditioning vectors cond(g) (1x1 conv of a length-1 plane = Linear) */
    float* gv = falloc(H);
    /* DurationPredictor */
    float* logw_dp = falloc(T);
    {
        const int Fd = c->dp_filter, k = c->dp_kernel;
        float *xd = falloc((size_t)H * T), *d1 = falloc((size_t)Fd * T), *d2 = falloc((size_t)Fd * T);
        linear(gv, g, G, 1, W(B, "dp.cond.weight"), W(B, "dp.cond.bias"), H, epi0(NULL));
        for (int ch = 0; ch < H; ++ch)
            for (int t = 0; t < T; ++t) xd[(size_t)ch * T + t] = x[(size_t)ch * T + t] + gv[ch];
        Epi er = epi0(NULL);
        er.act = ACT_RELU;
        conv1d(d1, xd, H, T, W(B, "dp.conv_1.weight"), W(B, "dp.conv_1.bias"), Fd, k, 1, k / 2, k / 2, 1.0f, er);
        layernorm_ch(d1, d1, NULL, Fd, T, W(B, "dp.norm_1.gamma"), W(B, "dp.norm_1.beta"), 1e-5f, ACT_NONE, NULL);
        conv1d(d2, d1, Fd, T, W(B, "dp.conv_2.weight"), W(B, "dp.conv_2.bias"), Fd, k, 1, k / 2, k / 2, 1.0f, er);
        layernorm_ch(d2, d2, NULL, Fd, T, W(B, "dp.norm_2.gamma"), W(B, "dp.norm_2.beta"), 1e-5f, ACT_NONE, NULL);
        linear(logw_dp, d2, Fd, T, W(B, "dp.proj.weight"), W(B, "dp.proj.bias"), 1, epi0(NULL));
        free(xd); free(d1); free(d2);
    }
    /* StochasticDurationPredictor, reverse (flows reversed, the "useless vflow" ConvFlow 1 dropped: modeling_vits.py:791-804) */
    float* logw_sdp = falloc(T);
    {
        float *xs = falloc((size_t)H * T), *cond = falloc((size_t)H * T), *hh = falloc((size_t)H * T);
        const int nb = c->sdp_bins, P = 3 * nb - 1;
        float* pr = falloc((size_t)P * T);
        linear(xs, x, H, T, W(B, "sdp.pre.weight"), W(B, "sdp.pre.bias"), H, epi0(NULL));
        linear(gv, g, G, 1, W(B, "sdp.cond.weight"), W(B, "sdp.cond.bias"), H, epi0(NULL));
        for (int ch = 0; ch < H; ++ch)
            for (int t = 0; t < T; ++t) xs[(size_t)ch * T + t] += gv[ch];
        dds_conv(B, c, "sdp.convs.", xs, T);
        linear(cond, xs, H, T, W(B, "sdp.proj.weight"), W(B, "sdp.proj.bias"), H, epi0(NULL));
        float *z0 = falloc(T), *z1 = falloc(T);
        for (int t = 0; t < T; ++t) {
            z0[t] = noise_w ? noise_w[t] : 0.f;
            z1[t] = noise_w ? noise_w[T + t] : 0.f;
        }
        const float inv_sqrt = sqrtf((float)H);
        for (int i = c->sdp_flows; i >= 2; --i) {
            float* tsw = z0; z0 = z1; z1 = tsw; /* Flip */
            char p[64];
            snprintf(p, sizeof p, "sdp.flows.%d.", 2 * i - 1);
            const float* pw = W(B, "%spre.weight", p);
            const float* pb = W(B, "%spre.bias", p);
            for (int ch = 0; ch < H; ++ch)
                for (int t = 0; t < T; ++t) hh[(size_t)ch * T + t] = pw[ch] * z0[t] + pb[ch] + cond[(size_t)ch * T + t];
            char pc[80];
            snprintf(pc, sizeof pc, "%sconvs.", p);
            dds_conv(B, c, pc, hh, T);
            linear(pr, hh, H, T, W(B, "%sproj.weight", p), W(B, "%sproj.bias", p), P, epi0(NULL));
            for (int t = 0; t < T; ++t) {
                float uw[32], uh[32], ud[32];
                for (int b2 = 0; b2 < nb; ++b2) {
                    uw[b2] = pr[(size_t)b2 * T + t] / inv_sqrt;
                    uh[b2] = pr[(size_t)(nb + b2) * T + t] / inv_sqrt;
                }
                for (int b2 = 0; b2 < nb - 1; ++b2) ud[b2] = pr[(size_t)(2 * nb + b2) * T + t];
                z1[t] = spline_inverse1(z1[t], uw, uh, ud, nb, c->sdp_tail);
            }
        }
        {
            float* tsw = z0; z0 = z1; z1 = tsw; /* Flip */
            const float* em = W(B, "sdp.flows.0.m");
            const float* el = W(B, "sdp.flows.0.logs");
            for (int t = 0; t < T; ++t) logw_sdp[t] = (z0[t] - em[0]) * expf(-el[0]);
        }
        free(xs); free(cond); free(hh); free(pr); free(z0); free(z1);
    }
    /* logw blend, ceil, path (SynthesizerTrn.infer; VitsModel.forward modeling_vits.py:1349-1376) */
    int64_t* dur = (int64_t*)malloc(sizeof(int64_t) * T);
    int64_t Tf = 0;
    for (int t = 0; t < T; ++t) {
        const float lw = logw_sdp[t] * sdp_ratio + logw_dp[t] * (1.0f - sdp_ratio);
        if (logw_out) logw_out[t] = lw;
        const float w = expf(lw) * length_scale;
        int64_t dv = (int64_t)ceilf(w);
        if (durations_out) durations_out[t] = dv;
        if (forced) dv = forced[t];
        dur[t] = dv;
        Tf += dv;
    }
    const int64_t Tf_sum = Tf;
    if (Tf < 1) Tf = 1; /* clamp_min(sum, 1) */
    float* z = falloc((size_t)I * Tf);
    {
        int64_t* tok = (int64_t*)malloc(sizeof(int64_t) * Tf);
        int64_t y = 0;
        for (int t = 0; t < T; ++t)
            for (int64_t r = 0; r < dur[t]; ++r) tok[y++] = t;
        for (; y < Tf; ++y) tok[y] = -1; /* the degenerate all-zero case: one frame of zeros */
        (void)Tf_sum;
        for (int ch = 0; ch < I; ++ch)
            for (int64_t f = 0; f < Tf; ++f) z[(size_t)ch * Tf + f] = tok[f] >= 0 ? m_p[(size_t)ch * T + tok[f]] : 0.f;
        free(tok);
    }
    /* TransformerCouplingBlock reverse: for i = n-1..0: Flip, x1 -= post(enc(pre(x0))) */
    {
        const int half = I / 2;
        float *hf = falloc((size_t)H * Tf), *mm = falloc((size_t)half * Tf), *zf = falloc((size_t)I * Tf);
        for (int i = c->flow_n - 1; i >= 0; --i) {
            for (int ch = 0; ch < I; ++ch) memcpy(zf + (size_t)ch * Tf, z + (size_t)(I - 1 - ch) * Tf, (size_t)Tf * 4);
            float* sw = z; z = zf; zf = sw;
            char p[64];
            snprintf(p, sizeof p, "flow.flows.%d.", 2 * i);
            linear(hf, z, half, Tf, W(B, "%spre.weight", p), W(B, "%spre.bias", p), H, epi0(NULL));
            char pe[80];
            snprintf(pe, sizeof pe, "%senc.", p);
            encoder(B, c, pe, hf, Tf, g, c->flow_layers, c->flow_kernel);
            linear(mm, hf, H, Tf, W(B, "%spost.weight", p), W(B, "%spost.bias", p), half, epi0(NULL));
            for (size_t e = 0; e < (size_t)half * Tf; ++e) z[(size_t)half * Tf + e] -= mm[e];
        }
        free(hf); free(mm); free(zf);
    }
    if (z_out && (int64_t)I * Tf <= z_cap) memcpy(z_out, z, (size_t)I * Tf * 4);
    /* HiFi-GAN generator (models_jp_extra.Generator = transformers VitsHifiGan, modeling_vits.py:466-551) */
    int C = c->up_initial;
    int64_t L = Tf;
    float* cur = falloc((size_t)C * L);
    {
        const int k = (int)TN(B, "dec.conv_pre.weight")->dims[2];
        conv1d(cur, z, I, L, W(B, "dec.conv_pre.weight"), W(B, "dec.conv_pre.bias"), C, k, 1, k / 2, k / 2, 1.0f, epi0(NULL));
        float* cv = falloc(C);
        linear(cv, g, G, 1, W(B, "dec.cond.weight"), W(B, "dec.cond.bias"), C, epi0(NULL));
        for (int ch = 0; ch < C; ++ch)
            for (int64_t t = 0; t < L; ++t) cur[(size_t)ch * L + t] += cv[ch];
        free(cv);
    }
    for (int si = 0; si < c->n_up; ++si) {
        const int r = c->up_rates[si], ku = c->up_kernels[si], Co = C / 2;
        const int64_t Lo = L * r;
        float* xu = falloc((size_t)Co * Lo);
        conv_transpose1d(xu, cur, C, L, W(B, "dec.ups.%d.weight", si), W(B, "dec.ups.%d.bias", si), Co, ku, r, (ku - r) / 2, 0.1f);
        free(cur);
        float *xs = falloc((size_t)Co * Lo), *t1 = falloc((size_t)Co * Lo), *ya = falloc((size_t)Co * Lo), *yb = falloc((size_t)Co * Lo);
        for (int j = 0; j < c->n_res; ++j) {
            const int k = c->res_kernels[j], nd = c->res_nd[j], rb = si * c->n_res + j;
            const float* y = xu;
            for (int q = 0; q < nd; ++q) {
                const int d = c->res_dil[j][q];
                conv_same(t1, y, Co, Lo, W(B, "dec.resblocks.%d.convs1.%d.weight", rb, q), W(B, "dec.resblocks.%d.convs1.%d.bias", rb, q), Co, k, d, 0.1f,
                          epi0(NULL));
                Epi e = epi0(NULL);
                e.res = y;
                e.ldr = Lo;
                float* yn = (y == ya) ? yb : ya;
                if (q + 1 == nd) {
                    yn = xs;
                    e.accumulate = j > 0;
                }
                conv_same(yn, t1, Co, Lo, W(B, "dec.resblocks.%d.convs2.%d.weight", rb, q), W(B, "dec.resblocks.%d.convs2.%d.bias", rb, q), Co, k, 1, 0.1f, e);
                y = yn;
            }
        }
        const float inv = 1.0f / (float)c->n_res;
#pragma omp parallel for schedule(static)
        for (size_t e = 0; e < (size_t)Co * Lo; ++e) xs[e] *= inv;
        free(xu); free(t1); free(ya); free(yb);
        cur = xs;
        C = Co;
        L = Lo;
    }
    {
        const int k = (int)TN(B, "dec.conv_post.weight")->dims[2];
        float* out = (float*)malloc((size_t)L * 4);
        conv1d(out, cur, C, L, W(B, "dec.conv_post.weight"), NULL, 1, k, 1, k / 2, k / 2, 0.01f, epi0(NULL));
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < L; ++i) out[i] = tanhf(out[i]);
        *pcm = out;
        *n_pcm = L;
    }
    free(cur); free(z); free(dur); free(stats); free(x); free(gv); free(logw_dp); free(logw_sdp);
    return 0;
}

void sbv2c_free_pcm(float* p) { free(p); }

/* standalone conv check for the tests: y[Cout][L] = conv1d_same(lrelu(x, slope), w, b, dil) */
int sbv2c_conv1d_same(const float* x, int Cin, int64_t L, const float* w, const float* b, int Cout, int k, int dil, float slope, float* y) {
    conv_same(y, x, Cin, L, w, b, Cout, k, dil, slope, epi0(NULL));
    return 0;
}
int sbv2c_conv_transpose1d(const float* x, int Cin, int64_t L, const float* w, const float* b, int Cout, int k, int s, int p, float slope, float* y) {
    conv_transpose1d(y, x, Cin, L, w, b, Cout, k, s, p, slope);
    return 0;
}
