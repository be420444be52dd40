// HiFi-GAN generator (models_jp_extra.Generator) on channels-last planes with the bf16 / split-bf16 MFMA kernel of conv_cl.hip.
// Same arithmetic graph as VitsModel::run_decoder (vits.cpp), different layout and matrix-core data type:
//   dec_mode_ 1 = split-bf16 (hi/lo operands, 3 MFMAs per product, f32-grade), 2 = plain bf16 operands, 3 = fp16 operands
//   (1 MFMA per product like bf16, 11-bit significands).  pack_cl's precision code: 1 = bf16, 2 = bf16 hi + lo, 3 = fp16.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>

#include "models.h"

namespace sbv2 {

static inline uint16_t f32_to_bf16_rne(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN stays NaN
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
// f32 -> IEEE binary16, round to nearest even (subnormals kept, overflow -> inf)
static inline uint16_t f32_to_f16_rne(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    const uint32_t sign = (u >> 16) & 0x8000u;
    u &= 0x7FFFFFFFu;
    if (u >= 0x7F800000u) return (uint16_t)(sign | 0x7C00u | (u > 0x7F800000u ? 0x200u : 0u));
    if (u >= 0x477FF000u) return (uint16_t)(sign | 0x7C00u);            // >= 65520 rounds to inf
    if (u < 0x33000001u) return (uint16_t)sign;                          // < 2^-25 rounds to zero
    int e = (int)(u >> 23) - 127;
    uint32_t m = (u & 0x7FFFFFu) | 0x800000u;
    int shift = e < -14 ? 13 + (-14 - e) : 13;                           // bits dropped from the 24-bit significand
    uint32_t half = m >> shift, rem = m & ((1u << shift) - 1u), mid = 1u << (shift - 1);
    if (rem > mid || (rem == mid && (half & 1u))) ++half;
    if (e < -14) return (uint16_t)(sign | half);                         // subnormal (a carry into bit 10 gives the smallest normal)
    return (uint16_t)(sign | (((uint32_t)(e + 15) << 10) + (half - 0x400u)));   // mantissa carry propagates into the exponent
}
static inline float bf16_to_f32(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}

// Packs W (either [M][K][k] = Conv1d layout, or already [M][K][k] rows built by the caller) into MFMA fragment blocks
// [chunk][mtile][tap][part][lane][8]: lane l of a block holds W[m = mtile*32 + (l & 31)][k = chunk*16 + 8*(l >> 5) + j][tap].
ClConv pack_cl(WeightStore& ws, const float* w, int M, int K, int k, int parts, const float* bias) {
    ClConv c;
    c.M = M;
    c.K = K;
    c.k = k;
    c.tm = M >= 64 ? 2 : 1;
    c.nmt = round_up((M + 31) / 32, c.tm);
    c.parts = parts;                      // precision code: 1 = bf16, 2 = bf16 hi + lo, 3 = fp16
    const bool f16 = parts == 3;
    if (f16) parts = 1;                   // one fragment block per tap
    const int nchunks = (K + 15) / 16;
    std::vector<uint16_t> h((size_t)nchunks * c.nmt * k * parts * 512, 0);
    for (int ch = 0; ch < nchunks; ++ch)
        for (int mt = 0; mt < c.nmt; ++mt)
            for (int t = 0; t < k; ++t) {
                uint16_t* blk = h.data() + ((((size_t)ch * c.nmt + mt) * k + t) * parts) * 512;
                for (int l = 0; l < 64; ++l) {
                    const int m = mt * 32 + (l & 31);
                    if (m >= M) continue;
                    for (int j = 0; j < 8; ++j) {
                        const int kk = ch * 16 + 8 * (l >> 5) + j;
                        if (kk >= K) continue;
                        const float v = w[((size_t)m * K + kk) * k + t];
                        const uint16_t hi = f16 ? f32_to_f16_rne(v) : f32_to_bf16_rne(v);
                        blk[l * 8 + j] = hi;
                        if (parts == 2) blk[512 + l * 8 + j] = f32_to_bf16_rne(v - bf16_to_f32(hi));
                    }
                }
            }
    c.w = ws.upload(reinterpret_cast<const float*>(h.data()), h.size() / 2);
    if (bias) c.bias = ws.upload(bias, (size_t)M);
    if (c.parts == 2 && (M & 63) == 0 && (K & 31) == 0) c.wx = pack_clx16(ws, w, M, K, k);   // the shapes conv_clx.hip takes
    if (c.parts == 2 && M == K && (M == 32 || M == 64) && (k & 1)) c.wxp = pack_step_pairs(ws, w, M, k);   // ... and respair_x16.hip
    return c;
}

// The same split-bf16 weights as A fragments of v_mfma_f32_16x16x32_bf16 whose 32-deep K dimension carries two consecutive STEPS (step s = tap s % k of chunk
// s / k; K / 16 even: with an odd k the middle pair of two chunks spans them) for conv_clx.hip: 1 KB blocks [row tile of 64][pair][row tile of 16][part],
// lane l of a block (row l & 15, k group g = l >> 4) holds W[m = 64 mt + 16 rt + (l & 15)][16 chunk(s) + 8 (g & 1) + j][tap(s)], s = 2 pair + (g >> 1),
// j = 0 .. 7; part 0 = bf16 hi, 1 = lo.
void* pack_clx16(WeightStore& ws, const float* w, int M, int K, int k) {
    const int nsteps = (K / 16) * k, nmt = M / 64;
    std::vector<uint16_t> h((size_t)nmt * (nsteps / 2) * 8 * 512, 0);
    for (int mt = 0; mt < nmt; ++mt)
        for (int u = 0; u < nsteps / 2; ++u)
            for (int rt = 0; rt < 4; ++rt) {
                uint16_t* blk = h.data() + ((((size_t)mt * (nsteps / 2) + u) * 4 + rt) * 2) * 512;
                for (int l = 0; l < 64; ++l) {
                    const int g = l >> 4, s = 2 * u + (g >> 1), ch = s / k, t = s % k, m = mt * 64 + rt * 16 + (l & 15);
                    for (int j = 0; j < 8; ++j) {
                        const float v = w[((size_t)m * K + ch * 16 + 8 * (g & 1) + j) * k + t];
                        const uint16_t hi = f32_to_bf16_rne(v);
                        blk[l * 8 + j] = hi;
                        blk[512 + l * 8 + j] = f32_to_bf16_rne(v - bf16_to_f32(hi));
                    }
                }
            }
    return ws.upload(reinterpret_cast<const float*>(h.data()), h.size() / 2);
}

// C -> C channel convolution (C = 32 / 64) as A fragments of v_mfma_f32_16x16x32_bf16 whose 32-deep K dimension carries two consecutive STEPS (step s = tap
// s % k of chunk s / k; k odd: the middle pair of two chunks spans them): 1 KB blocks [pair][row tile of 16][part], lane l of a block (row l & 15, k group
// g = l >> 4) holds W[m = 16 rt + (l & 15)][16 chunk(s) + 8 (g & 1) + j][tap(s)], s = 2 pair + (g >> 1), j = 0 .. 7; bf16 hi / lo parts (respair_x16.hip)
void* pack_step_pairs(WeightStore& ws, const float* w, int C, int k) {
    const int nsteps = (C / 16) * k, nrt = C / 16;
    std::vector<uint16_t> h((size_t)(nsteps / 2) * nrt * 2 * 512, 0);
    for (int u = 0; u < nsteps / 2; ++u)
        for (int rt = 0; rt < nrt; ++rt)
            for (int l = 0; l < 64; ++l) {
                const int g = l >> 4, s = 2 * u + (g >> 1), ch = s / k, t = s % k, m = rt * 16 + (l & 15);
                for (int j = 0; j < 8; ++j) {
                    const float v = w[((size_t)m * C + ch * 16 + 8 * (g & 1) + j) * k + t];
                    const uint16_t hi = f32_to_bf16_rne(v);
                    uint16_t* blk = h.data() + (((size_t)u * nrt + rt) * 2) * 512;
                    blk[l * 8 + j] = hi;
                    blk[512 + l * 8 + j] = f32_to_bf16_rne(v - bf16_to_f32(hi));
                }
            }
    return ws.upload(reinterpret_cast<const float*>(h.data()), h.size() / 2);
}

// 16 -> 16 channel convolution as A fragments of v_mfma_f32_16x16x32_bf16 whose 32-deep K dimension carries two taps: block (pair tp, part), lane l
// holds W[row = l & 15][channel 8 (g & 1) + j][tap 2 tp + (g >> 1)], g = l >> 4 (zero for the phantom tap of an odd kernel size); bf16 hi / lo parts.
void* pack_cl_pairs(WeightStore& ws, const float* w, int k) {
    const int np = (k + 1) / 2;
    std::vector<uint16_t> h((size_t)np * 2 * 512, 0);
    for (int tp = 0; tp < np; ++tp)
        for (int l = 0; l < 64; ++l) {
            const int row = l & 15, g = l >> 4, tap = 2 * tp + (g >> 1);
            if (tap >= k) continue;
            for (int j = 0; j < 8; ++j) {
                const float v = w[((size_t)row * 16 + 8 * (g & 1) + j) * k + tap];
                const uint16_t hi = f32_to_bf16_rne(v);
                h[((size_t)tp * 2) * 512 + l * 8 + j] = hi;
                h[((size_t)tp * 2 + 1) * 512 + l * 8 + j] = f32_to_bf16_rne(v - bf16_to_f32(hi));
            }
        }
    return ws.upload(reinterpret_cast<const float*>(h.data()), h.size() / 2);
}

ClUpX build_upx(WeightStore& ws, const float* wt, const float* ub, int cin, int cout, int k, int s, bool parts_out) {
    ClUpX u;
    const int pad = (k - s) / 2;
    if (k - 2 * pad != s || s < 1 || s > kMaxPhases || (cout & 63) || (cin & 31)) return u;
    int tmin = 1 << 30, tmax = -(1 << 30);
    for (int r = 0; r < s; ++r)
        for (int tt = -k; tt <= k; ++tt) {
            const int j = s * tt + r + pad;
            if (j >= 0 && j < k) {
                tmin = std::min(tmin, tt);
                tmax = std::max(tmax, tt);
            }
        }
    // every phase multiplies its OWN input taps (k 16, s 8: two, one position later for the first four phases than for the last four; k 8, s 2: four);
    // SBV2_UPX=3 (A/B runs): the first build's union of all phases' taps (three / five), zero weights where a phase has none
    const bool uni = upx_mode() == 3;
    int t0[kMaxPhases], U = 0;
    for (int r = 0; r < s; ++r) {
        int lo = 1 << 30, hi = -(1 << 30);
        for (int tt = tmin; tt <= tmax; ++tt) {
            const int j = s * tt + r + pad;
            if (j >= 0 && j < k) {
                lo = std::min(lo, tt);
                hi = std::max(hi, tt);
            }
        }
        t0[r] = uni ? tmin : lo;
        U = std::max(U, uni ? tmax - tmin + 1 : hi - lo + 1);
    }
    if (uni && (U & 1) == 0) ++U;
    if (U < 2 || U > 5) return u;
    const int M = s * cout;
    // a launch that also writes the next stage's operand parts packs its rows as (phase pair, 16 channels, phase in pair, channel): common.h, phase_group
    int group = parts_out && (s & 1) == 0 && upx_mode() != 2 ? 2 : 1;
    if (group == 2)
        for (int r = 0; r < s; r += 2)
            if (t0[r] != t0[r + 1]) group = 1;   // (the two phases of a pair share the window rows they read: s = 2 keeps the plain order)
    auto row_of = [&](int r, int co) { return group == 2 ? (((r >> 1) * (cout >> 4) + (co >> 4)) * 2 + (r & 1)) * 16 + (co & 15) : r * cout + co; };
    std::vector<float> w((size_t)M * cin * U, 0.f), bias((size_t)M);
    double macs = 0;
    for (int r = 0; r < s; ++r)
        for (int co = 0; co < cout; ++co) {
            bias[(size_t)row_of(r, co)] = ub ? ub[co] : 0.f;
            for (int ti = 0; ti < U; ++ti) {
                const int j = s * (t0[r] + ti) + r + pad;
                if (j < 0 || j >= k) continue;
                if (co == 0) macs += (double)cin * cout;
                for (int ci = 0; ci < cin; ++ci) w[((size_t)row_of(r, co) * cin + ci) * U + ti] = wt[((size_t)ci * cout + co) * k + j];
            }
        }
    u.wx = pack_clx16(ws, w.data(), M, cin, U);
    u.bias = ws.upload(bias.data(), bias.size());
    u.M = M;
    u.K = cin;
    u.ntaps = U;
    u.shift0 = -tmin;                            // tap ti of phase r reads input position n - (tmin + phase_tap0[r] + ti)
    u.nph = s;
    u.cout = cout;
    u.group = group;
    u.alg_macs_per_pos = macs;
    for (int r = 0; r < s; ++r) {
        u.phase_off[r] = r;
        u.phase_tap0[r] = t0[r] - tmin;
    }
    return u;
}

// pack_cl's precision code of a decoder arithmetic (dec_mode_ / ClStage::mode)
static int cl_parts_of(int mode) { return mode == 1 ? 2 : (mode == 2 ? 1 : 3); }

void VitsModel::load_decoder_cl(const Blob& blob) {
    auto conv = [&](const std::string& prefix, int mode) {
        const HostTensor& t = blob.get(prefix + ".weight");
        const float* b = blob.has(prefix + ".bias") ? blob.get(prefix + ".bias").data : nullptr;
        ClConv c = pack_cl(*ws_, t.data, (int)t.dims[0], (int)t.dims[1], (int)t.dims[2], cl_parts_of(mode), b);
        if (mode == 1 && t.dims[0] == 16 && t.dims[1] == 16) c.wp = pack_cl_pairs(*ws_, t.data, (int)t.dims[2]);
        return c;
    };
    // SBV2_DECODER_STAGES = one arithmetic per upsampling stage ("f16,f16,bf16x3,bf16x3,bf16x3"): the per-stage precision map of profiles/r05_precision_map.json.
    // An experiment knob: the default is SBV2_DECODER's arithmetic for every stage.
    std::vector<int> stage_mode(cfg_.up_rates.size(), dec_mode_);
    if (const char* e = getenv("SBV2_DECODER_STAGES")) {
        std::string v(e);
        size_t pos = 0;
        for (size_t i = 0; i < stage_mode.size() && pos <= v.size(); ++i) {
            const size_t q = v.find(',', pos);
            const std::string tok = v.substr(pos, q == std::string::npos ? std::string::npos : q - pos);
            if (tok == "bf16x3") stage_mode[i] = 1;
            else if (tok == "bf16") stage_mode[i] = 2;
            else if (tok == "f16") stage_mode[i] = 3;
            else SBV2_REQUIRE(tok.empty(), "SBV2_DECODER_STAGES: bf16x3, bf16 or f16 per stage");
            if (q == std::string::npos) break;
            pos = q + 1;
        }
    }
    cl_pre_ = conv("dec.conv_pre", dec_mode_);
    int C = cfg_.up_initial;
    const int nk = (int)cfg_.res_kernels.size();
    for (size_t i = 0; i < cfg_.up_rates.size(); ++i) {
        ClStage st;
        st.mode = stage_mode[i];
        st.rate = cfg_.up_rates[i];
        st.cin = C;
        C /= 2;
        st.ch = C;
        // polyphase groups: rows (phase, cout), taps = input offsets shared by the group's phases (see WeightStore::upsample)
        const HostTensor& t = blob.get("dec.ups." + std::to_string(i) + ".weight");
        const float* ub = blob.get("dec.ups." + std::to_string(i) + ".bias").data;
        const int k = (int)t.dims[2], s = st.rate, pad = (k - s) / 2, cin = (int)t.dims[0], cout = (int)t.dims[1];
        SBV2_REQUIRE(k - 2 * pad == s, "transposed conv must upsample exactly by its stride");
        std::vector<std::vector<int>> tsets(s);
        for (int r = 0; r < s; ++r)
            for (int tt = -k; tt <= k; ++tt) {
                const int j = s * tt + r + pad;
                if (j >= 0 && j < k) tsets[r].push_back(tt);
            }
        std::vector<bool> done(s, false);
        for (int r = 0; r < s; ++r) {
            if (done[r]) continue;
            std::vector<int> phases;
            for (int r2 = r; r2 < s && (int)phases.size() < kMaxPhases; ++r2)
                if (!done[r2] && tsets[r2] == tsets[r]) {
                    phases.push_back(r2);
                    done[r2] = true;
                }
            ClUpGroup g;
            g.ntaps = (int)tsets[r].size();
            g.nph = (int)phases.size();
            const int M = g.nph * cout;
            std::vector<float> w((size_t)M * cin * g.ntaps), bias((size_t)M);
            for (int pi = 0; pi < g.nph; ++pi)
                for (int co = 0; co < cout; ++co) {
                    bias[pi * cout + co] = ub[co];
                    for (int ci = 0; ci < cin; ++ci)
                        for (int ti = 0; ti < g.ntaps; ++ti)
                            w[((size_t)(pi * cout + co) * cin + ci) * g.ntaps + ti] =
                                t.data[((size_t)ci * cout + co) * k + (s * tsets[r][ti] + phases[pi] + pad)];
                }
            for (int ti = 0; ti < g.ntaps; ++ti) g.shift[ti] = -tsets[r][ti];
            for (int pi = 0; pi < kMaxPhases; ++pi) g.phase_off[pi] = pi < g.nph ? phases[pi] : 0;
            g.c = pack_cl(*ws_, w.data(), M, cin, g.ntaps, cl_parts_of(st.mode), bias.data());
            st.up.push_back(g);
        }
        if (st.mode == 1 && st.ch >= 64) st.upx = build_upx(*ws_, t.data, ub, cin, cout, k, s, /*parts_out=*/st.ch >= 128);   // (the <= 64-channel stages read f32 planes)
        for (int j = 0; j < nk; ++j) {
            ClBranch rb;
            rb.k = cfg_.res_kernels[j];
            rb.dil = cfg_.res_dilations[j];
            const std::string p = "dec.resblocks." + std::to_string(i * nk + j) + ".";
            for (size_t n = 0; n < rb.dil.size(); ++n) {
                rb.c1.push_back(conv(p + "convs1." + std::to_string(n), st.mode));
                rb.c2.push_back(conv(p + "convs2." + std::to_string(n), st.mode));
            }
            st.branches.push_back(rb);
        }
        cl_stages_.push_back(st);
    }
}

// conv_clx.hip for the wide stages' ResBlocks (SBV2_CLX=0 / sbv2_debug_set_clx(0): the conv_cl path, bit-identical, for A/B runs and the test)
static std::atomic<int> g_clx{getenv("SBV2_CLX") ? atoi(getenv("SBV2_CLX")) : 1};
bool clx_enabled() { return g_clx.load(std::memory_order_relaxed) != 0; }
static int64_t clx_min_tiles() {
    // 128 tiles: a single 128-phoneme utterance's 128-channel stage (448 tiles) takes conv_clx, its 256-channel stage (112) stays on conv_cl: 11.93 -> 11.79 ms
    // per call (profiles/r05k_b1_clx_min_tiles.txt); rounds 3-4 required 1024 tiles.
    return g_clx.load(std::memory_order_relaxed) == 2 ? 0 : 128;   // set_clx(2): every size (the bit-equality test runs small batches)
}
bool clx_wanted(int64_t tiles, int64_t min_tiles) {
    const int m = g_clx.load(std::memory_order_relaxed);
    return m == 2 || (m == 1 && tiles >= min_tiles);
}
int set_clx(int on) { return g_clx.exchange(on); }
// the transposed convolutions of the wide stages as phased conv_clx launches (round 6; SBV2_UPX=0 / sbv2_debug_set_upx(0): conv_cl's phase groups, for A/B runs)
static std::atomic<int> g_upx{getenv("SBV2_UPX") ? atoi(getenv("SBV2_UPX")) : 1};
int set_upx(int on) { return g_upx.exchange(on); }
int upx_mode() { return g_upx.load(std::memory_order_relaxed); }

void VitsModel::conv_cl(const ClConv& c, const float* X, int ldx, int NB, float* Y, int ldy, int N, int dil, int pad_l,
                        const unsigned char* mask, int mask_div, float pre_slope, const float* R, int ldr, float beta, int accumulate) {
    ConvClParams p;
    p.X = X;
    p.ldx = ldx;
    p.NB = NB;
    p.W = c.w;
    p.nmt = c.nmt;
    p.tm = c.tm;
    p.split = c.parts == 2;   // (the arithmetic the weights were packed for)
    p.f16 = c.parts == 3;
    p.M = c.M;
    p.N = N;
    p.K = c.K;
    p.ntaps = c.k;
    for (int j = 0; j < c.k; ++j) p.shift[j] = j * dil - pad_l;
    p.Y = Y;
    p.ldy = ldy;
    p.bias = c.bias;
    p.R = R;
    p.ldr = ldr;
    p.pre_slope = pre_slope;
    p.beta = beta;
    p.accumulate = accumulate;
    p.mask = mask;
    p.mask_div = mask_div;
    launch_conv_cl(p, stream_);
}

void VitsModel::run_decoder_cl(Arena& ar, Plane z, const SegLayout& fl, const float* cond_vec) {
    const int n = fl.n, Lf = fl.L, I = cfg_.inter;
    SBV2_REQUIRE(I % 16 == 0, "flow channels must be a multiple of 16 for the channels-last decoder");
    // z [inter][Lf] -> channels-last [Lf][inter]
    float* zc = ar.array<float>((size_t)Lf * I);
    transpose_out(z, 0, Lf, zc, stream_);
    int C = cfg_.up_initial;
    float* cur = ar.array<float>((size_t)Lf * C);
    conv_cl(cl_pre_, zc, I, Lf, cur, C, Lf, 1, cl_pre_.k / 2, fl.d_mask, 1, 1.0f, nullptr, 0, 1.0f, 0);
    add_segvec_cl(cur, Lf, C, cond_vec, C, fl.d_seg_of, fl.d_mask, stream_);
    int U = 1;
    int64_t Lcur = Lf;
    const int nk = (int)cfg_.res_kernels.size();
    SplitClPlanes cur_s;         // bf16 parts of lrelu(cur, 0.1) when the previous stage's last launch wrote them (the operand of a conv_clx transposed convolution)
    bool cur_s_ok = false;
    // whether stage si's transposed convolution runs as ONE phased conv_clx launch (round 6; conv_cl's phase groups ran one 8-wave workgroup per CU at 233
    // registers: 1.0 ms per launch at the 128-channel stage for 0.24 ms of bytes and 0.33 ms of MFMA work, profiles/r06d_decoder_kernel_list.txt)
    auto upx_wanted = [&](size_t si, int64_t Lin, int Uin) {
        if (si >= cl_stages_.size()) return false;
        const ClStage& s2 = cl_stages_[si];
        return clx_enabled() && g_upx.load(std::memory_order_relaxed) != 0 && s2.mode == 1 && s2.upx.wx != nullptr && Lin >= 256 && (Uin & (Uin - 1)) == 0 &&
               clx_wanted((Lin / 256) * (s2.upx.M / 64), clx_min_tiles());
    };
    for (size_t si = 0; si < cl_stages_.size(); ++si) {
        const ClStage& st = cl_stages_[si];
        const int Uin = U;
        U *= st.rate;
        const int64_t Lo = (int64_t)Lf * U;
        SBV2_REQUIRE(Lo < (1ll << 31), "batch too long for 32-bit positions");
        C = st.ch;
        float* XS = ar.array<float>((size_t)Lo * C);
        int ushift = 0;
        while ((1 << ushift) < U) ++ushift;
        // (launches of at least clx_min_tiles() tiles: smaller ones do not pay for the extra halo launches.  The two paths agree to f32 rounding,
        // tests/test_gpu_parity.py::test_decoder_clx_path_agrees_with_conv_cl_path.  The 64-channel stage stays on the fused step: unfused on conv_clx it
        // moves three times the bytes, measured in round 3.)
        bool clx = clx_enabled() && st.mode == 1 && C >= 128 && (C & 63) == 0 && (1 << ushift) == U && (Lo / 256) * (C / 64) >= clx_min_tiles();
        for (int j = 0; j < nk && clx; ++j) {
            const ClBranch& rb = st.branches[j];
            if (!(rb.k == 3 || rb.k == 7 || rb.k == 11) || !rb.c1[0].wx) clx = false;
            for (int d : rb.dil)
                if (d * (rb.k - 1) > 64 || d * (rb.k - 1) / 2 > kClxFront) clx = false;
        }
        // the NEXT stage's transposed convolution on conv_clx reads lrelu(XS) as bf16 parts: this stage's last launch writes them next to XS
        const bool next_upx = clx && upx_wanted(si + 1, Lo, U);
        SplitClPlanes XSs;
        if (next_upx) XSs = make_split_cl(ar.alloc(split_cl_bytes(C, Lo)), C, Lo, stream_);
        const Arena::Mark mk = ar.mark();
        float* XU = ar.array<float>((size_t)Lo * C);
        float* T1 = ar.array<float>((size_t)Lo * C);
        float* YA = ar.array<float>((size_t)Lo * C);
        float* YB = ar.array<float>((size_t)Lo * C);
        // Wide stages (>= 128 channels, split-bf16): the ResBlock convolutions read PRE-SPLIT operands (conv_clx.hip: LDS-DMA only, one barrier
        // per tap).  The stage input is split once (split_cl); every other operand is written by the producing convolution's epilogue as the
        // bf16 parts of lrelu(result), next to (conv2) or instead of (conv1) the f32 plane.  Same bits as the conv_cl path.
        SplitClPlanes XUs, T1s, YsA, YsB;
        if (clx) {
            const size_t sb = split_cl_bytes(C, Lo);
            XUs = make_split_cl(ar.alloc(sb), C, Lo, stream_);
            T1s = make_split_cl(ar.alloc(sb), C, Lo, stream_);
            YsA = make_split_cl(ar.alloc(sb), C, Lo, stream_);
            YsB = make_split_cl(ar.alloc(sb), C, Lo, stream_);
        }
        // (the stage input's parts are written by the transposed convolution's own epilogue below: no separate split pass over XU)
        bool parts_done = clx;   // every phase group's launch wrote its share of XUs
        if (upx_wanted(si, Lcur, Uin)) {
            // ONE phased conv_clx launch: rows (phase, cout), the union of the phases' taps; its operand = bf16 parts of lrelu(cur, 0.1)
            if (!cur_s_ok) {
                cur_s = make_split_cl(ar.alloc(split_cl_bytes(st.cin, Lcur)), st.cin, Lcur, stream_);
                split_cl(cur, st.cin, Lcur, st.cin, 0.1f, cur_s, stream_);
            }
            int ushift_in = 0;
            while ((1 << ushift_in) < Uin) ++ushift_in;
            ConvClxParams pu;
            pu.X = cur_s;
            pu.W = st.upx.wx;
            pu.nmt = st.upx.M / 32;
            pu.M = st.upx.M;
            pu.N = (int)Lcur;
            pu.K = st.cin;
            pu.ntaps = st.upx.ntaps;
            pu.shift0 = st.upx.shift0;
            pu.shift_step = -1;
            pu.Y = XU;
            pu.ldy = C;
            if (clx) {
                pu.Ys = XUs;
                pu.ys_slope = 0.1f;
            }
            pu.bias = st.upx.bias;
            pu.mask = fl.d_mask;
            pu.mask_shift = ushift_in;      // (the mask is per frame: output row >> ushift == input position >> ushift_in)
            pu.out_stride = st.rate;
            pu.phase_rows = C;
            pu.phase_group = st.upx.group;
            for (int q = 0; q < kMaxPhases; ++q) {
                pu.phase_off[q] = st.upx.phase_off[q];
                pu.phase_tap0[q] = st.upx.phase_tap0[q];
            }
            pu.prof_flops = 2.0 * st.upx.alg_macs_per_pos * (double)Lcur;
            SBV2_REQUIRE(conv_clx_usable(pu), "decoder: the phased transposed convolution does not fit conv_clx");
            launch_conv_clx(pu, stream_);
        } else {
            for (const auto& g : st.up) {
                ConvClParams p;
                p.X = cur;
                p.ldx = st.cin;
                p.NB = (int)Lcur;
                p.W = g.c.w;
                p.nmt = g.c.nmt;
                p.tm = g.c.tm;
                p.split = st.mode == 1;
                p.f16 = st.mode == 3;
                p.M = g.c.M;
                p.N = (int)Lcur;
                p.K = st.cin;
                p.ntaps = g.ntaps;
                for (int t = 0; t < g.ntaps; ++t) p.shift[t] = g.shift[t];
                p.Y = XU;
                p.ldy = C;
                p.bias = g.c.bias;
                p.pre_slope = 0.1f;
                p.mask = fl.d_mask;
                p.mask_div = U;
                p.out_stride = st.rate;
                p.phase_rows = C;
                for (int q = 0; q < kMaxPhases; ++q) p.phase_off[q] = g.phase_off[q];
                if (clx && conv_cl_parts_ok(p)) {
                    p.ys_p = XUs.p;
                    p.ys_rows = (int64_t)XUs.front + XUs.N + XUs.back;
                    p.ys_front = XUs.front;
                    p.ys_slope = 0.1f;
                } else {
                    parts_done = false;
                }
                launch_conv_cl(p, stream_);
            }
            if (clx && !parts_done) split_cl(XU, C, Lo, C, 0.1f, XUs, stream_);   // (small launches: one pass over the finished plane, same bits)
        }
        for (int j = 0; j < nk; ++j) {
            const ClBranch& rb = st.branches[j];
            const float* y = XU;
            const SplitClPlanes* ys = &XUs;   // bf16 parts of lrelu(y)
            const int nd = (int)rb.dil.size();
            // k = 3 branches of the <= 64-channel stages: all three steps in ONE launch (resbranch_clx.hip: the residual stream stays in registers, the
            // operands in LDS; 2 plane passes through HBM instead of 6; same bits as the three fused steps below)
            // (the same at C = 128, where it replaces the branch's six conv_clx launches: f32 XU in, f32 XS out, no parts planes; C = 256 does not fit: its
            // window alone is 143 KB at 128 rows)
            if (((!clx && C <= 64) || (clx && C == 128)) && fuse_pairs_ && resbranch_wanted(C, rb.k) && st.mode == 1 && nd == kResBranchSteps && (U & (U - 1)) == 0 &&
                !(next_upx && j + 1 == nk)) {
                ResBranchParams bp;
                bp.X = XU;
                bp.Y = XS;
                for (int q = 0; q < nd; ++q) {
                    bp.W[2 * q] = C == 16 ? rb.c1[q].wp : rb.c1[q].w;
                    bp.W[2 * q + 1] = C == 16 ? rb.c2[q].wp : rb.c2[q].w;
                    bp.b[2 * q] = rb.c1[q].bias;
                    bp.b[2 * q + 1] = rb.c2[q].bias;
                    bp.dil[q] = rb.dil[q];
                }
                bp.C = C;
                bp.N = (int)Lo;
                bp.k = rb.k;
                bp.slope = 0.1f;
                bp.beta = 1.0f / nk;
                bp.accumulate = j > 0;
                bp.mask = fl.d_mask;
                bp.mask_shift = ushift;
                if (resbranch_usable(bp)) {
                    launch_resbranch(bp, stream_);
                    continue;
                }
            }
            for (int q = 0; q < nd; ++q) {
                const int d = rb.dil[q];
                const bool last = q + 1 == nd;
                float* yn = last ? XS : ((y == YA) ? YB : YA);
                if (clx) {
                    ConvClxParams p1;
                    p1.X = *ys;
                    p1.W = rb.c1[q].wx;
                    p1.nmt = rb.c1[q].nmt;
                    p1.M = C;
                    p1.N = (int)Lo;
                    p1.K = C;
                    p1.ntaps = rb.k;
                    p1.shift0 = -d * (rb.k - 1) / 2;
                    p1.shift_step = d;
                    p1.Ys = T1s;               // conv1's result is only ever read as conv2's operand
                    p1.ys_slope = 0.1f;
                    p1.bias = rb.c1[q].bias;
                    p1.mask = fl.d_mask;
                    p1.mask_shift = ushift;
                    launch_conv_clx(p1, stream_);
                    ConvClxParams p2;
                    p2.X = T1s;
                    p2.W = rb.c2[q].wx;
                    p2.nmt = rb.c2[q].nmt;
                    p2.M = C;
                    p2.N = (int)Lo;
                    p2.K = C;
                    p2.ntaps = rb.k;
                    p2.shift0 = -(rb.k - 1) / 2;
                    p2.shift_step = 1;
                    p2.Y = yn;
                    p2.ldy = C;
                    const SplitClPlanes* yns = (ys == &YsA) ? &YsB : &YsA;
                    if (!last) {
                        p2.Ys = *yns;
                        p2.ys_slope = 0.1f;
                    } else if (next_upx && j + 1 == nk) {   // the stage's finished sum: lrelu(XS, 0.1) as the next transposed convolution's operand
                        p2.Ys = XSs;
                        p2.ys_slope = 0.1f;
                    }
                    p2.bias = rb.c2[q].bias;
                    p2.R = y;
                    p2.ldr = C;
                    p2.beta = last ? 1.0f / nk : 1.0f;
                    p2.accumulate = last && j > 0;
                    p2.mask = fl.d_mask;
                    p2.mask_shift = ushift;
                    launch_conv_clx(p2, stream_);
                    ys = yns;
                    y = yn;
                    continue;
                }
                if (fuse_pairs_ && C <= 64 && (U & (U - 1)) == 0) {
                    // stages of <= 64 channels are HBM bound: conv1 -> conv2 fused, the intermediate stays in LDS (respair_cl.hip)
                    ResPairParams rp;
                    rp.X = y;
                    rp.Y = yn;
                    rp.W1 = rb.c1[q].w;
                    rp.W2 = rb.c2[q].w;
                    rp.W1p = rb.c1[q].wp;
                    rp.W2p = rb.c2[q].wp;
                    rp.W1x = rb.c1[q].wxp;
                    rp.W2x = rb.c2[q].wxp;
                    rp.b1 = rb.c1[q].bias;
                    rp.b2 = rb.c2[q].bias;
                    rp.C = C;
                    rp.N = (int)Lo;
                    rp.k = rb.k;
                    rp.dil = d;
                    rp.split = st.mode == 1;
                    rp.f16 = st.mode == 3;
                    rp.slope = 0.1f;
                    rp.beta = last ? 1.0f / nk : 1.0f;
                    rp.accumulate = last && j > 0;
                    rp.mask = fl.d_mask;
                    rp.mask_div = U;
                    launch_respair_cl(rp, stream_);
                } else {
                    conv_cl(rb.c1[q], y, C, (int)Lo, T1, C, (int)Lo, d, d * (rb.k - 1) / 2, fl.d_mask, U, 0.1f, nullptr, 0, 1.0f, 0);
                    conv_cl(rb.c2[q], T1, C, (int)Lo, yn, C, (int)Lo, 1, (rb.k - 1) / 2, fl.d_mask, U, 0.1f, y, C, last ? 1.0f / nk : 1.0f,
                            last && j > 0);
                }
                y = yn;
            }
        }
        ar.rewind(mk);
        cur = XS;
        Lcur = Lo;
        cur_s = XSs;
        cur_s_ok = next_upx;
    }
    pcm_lens_.assign(n, 0);
    pcm_offs_.assign(n, 0);
    int64_t tot = 0, maxlen = 0;
    for (int u = 0; u < n; ++u) {
        pcm_offs_[u] = tot;
        pcm_lens_[u] = (int64_t)fl.len[u] * U;
        tot += pcm_lens_[u];
        maxlen = std::max(maxlen, pcm_lens_[u]);
    }
    pcm_total_ = tot;
    pcm_ = ar.array<float>((size_t)tot);
    int64_t* d_off = ar.array<int64_t>(n);
    ar.upload(d_off, pcm_offs_.data(), sizeof(int64_t) * n, stream_);
    conv_post_tanh_cl(cur, C, Lcur, dec_post_w_, dec_post_k_, 0.01f, fl.d_start, fl.d_len, d_off, n, U, maxlen, pcm_, stream_);
}

}  // namespace sbv2
