// Small-grid twin of conv_cl.hip for channels-last input -> k-major output (the second k = 5 convolution of the flow's FFN in a
// single-utterance call: 768 -> 192 channels over 897 frames is 24 workgroups of conv_cl on 256 CUs, and each of them pays a global
// round trip per 16-channel chunk because nothing else runs on its CU: 102 us per launch against a 19 us MFMA chain, 24 launches per call).
//
// Same arithmetic, element for element: per (chunk, tap) the three split-bf16 MFMAs lo*hi, hi*lo, hi*hi of v_mfma_f32_32x32x16_bf16 in the
// order of conv_cl.hip, the same hi / lo conversion, the same epilogue expression, so a batch row (conv_cl) and the single call of the
// same utterance (this kernel) agree bit for bit.  Only the work decomposition differs:
//   * one 32 rows x 32 positions tile per workgroup: 174 workgroups instead of 24, each ONE serial MFMA chain (wave 0) that three helper
//     waves keep fed (wave 1 issues the DMAs, waves 2 and 3 convert the windows), one raw s_barrier per chunk.  Measured per launch (flow FFN
//     conv 2 of a single-utterance call): conv_cl 102 us; one wave doing everything 80; MFMA wave + three loader / converter waves 50;
//     roles split as above 43, where the MFMA wave's own chain (15 dependent MFMAs + the fragment reads of 5 taps per chunk) is what is left;
//   * both operands arrive by LDS-DMA (global_load_lds_dwordx4) into a 4-slot ring, up to 3 chunks ahead, counted vmcnt: the weight
//     fragment blocks are lane-linear 1 KB images already, the activation window lands as raw f32 rows [position][16 channels] and is
//     converted LDS -> registers -> LDS (leaky-ReLU, hi / lo split, the XOR-swizzled [position][8 | 8] image conv_cl reads) one chunk
//     ahead of the MFMAs that use it.
#include <type_traits>

#include "common.h"

namespace sbv2 {

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

constexpr int kSlots = 8;

struct ClSmallParams {
    ConvClParams p;
    int xrows;      // window rows per chunk: 32 * TN + tap span
    int xg;         // DMA instructions of one raw window (16 rows each)
    int wshift0;    // min shift
    int sh0, sh_step;
    int mask_shift;
    int slot_bytes; // ring slot: ntaps * 2 KB of weight fragments, then xg KB of raw window
    int ahead;      // chunks in flight (1 .. 3)
};

__device__ __forceinline__ void wait_vm_dyn(int n) {
    // s_waitcnt takes an immediate: n is wave-uniform, one scalar branch
    switch (n) {
#define W_(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
        W_(0) W_(1) W_(2) W_(3) W_(4) W_(5) W_(6) W_(7) W_(8) W_(9) W_(10) W_(11) W_(12) W_(13) W_(14) W_(15)
        W_(16) W_(17) W_(18) W_(19) W_(20) W_(21) W_(22) W_(23) W_(24) W_(25) W_(26) W_(27) W_(28) W_(29) W_(30) W_(31)
        W_(32) W_(33) W_(34) W_(35) W_(36) W_(37) W_(38) W_(39) W_(40) W_(41) W_(42) W_(43) W_(44) W_(45) W_(46) W_(47)
        W_(48) W_(49) W_(50) W_(51) W_(52) W_(53) W_(54) W_(55) W_(56) W_(57) W_(58) W_(59) W_(60) W_(61) W_(62)
#undef W_
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

constexpr int kLoaders = 3;      // helper waves: wave 1 issues the DMAs, waves 2 and 3 convert the windows; wave 0 only runs the MFMA chain
constexpr int kConverters = 2;

template <int TN>
__global__ __launch_bounds__(64 * (kLoaders + 1)) void conv_cl_small_kernel(const ClSmallParams kp) {
    constexpr int NTW = 32 * TN;                       // positions per wave
    constexpr int NCV = (NTW + 64) / 16;               // float4 per lane of one window: (32 TN + 64 span) rows * 4 / 64 lanes
    const ConvClParams& p = kp.p;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ntaps = p.ntaps;
    const int n0 = blockIdx.x * NTW, m0 = blockIdx.y * 32;
    const int M = p.M, N = p.N, NB = p.NB;
    const int nchunks = p.K >> 4;
    const int wstart = n0 + kp.wshift0;
    const float slope = p.pre_slope;
    const int wbytes = ntaps * 2048;
    char* ring = smem;
    char* cvt = smem + kSlots * kp.slot_bytes;         // two converted windows: [buf][hi | lo][xrows * 32 bytes]
    const int cvt_bytes = kp.xrows * 32;

    // ---- DMA sources ------------------------------------------------------------------------------------------------------------------
    const char* wsrc = reinterpret_cast<const char*>(p.W) + (int64_t)blockIdx.y * wbytes + lane * 16;   // + chunk * nmt * wbytes
    const int64_t wstep = (int64_t)p.nmt * wbytes;
    const int xr = lane >> 2, xq = (lane & 3) * 4;
    // the 2 * ntaps + xg DMA instructions of a chunk are dealt round-robin over the waves
    const int nload = 2 * ntaps + kp.xg;
    // wave 1 issues every DMA of a chunk: 2 * ntaps weight blocks (contiguous KBs) and xg window pieces whose sources are resolved once
    constexpr int kMaxXg = 6;
    const char* xsrc[kMaxXg];
#pragma unroll
    for (int g = 0; g < kMaxXg; ++g) {
        const int pos = min(max(wstart + g * 16 + xr, 0), NB - 1);
        xsrc[g] = reinterpret_cast<const char*>(p.X + (int64_t)pos * p.ldx + xq);
    }
    auto stage = [&](int c) {
        char* dst = ring + (c & (kSlots - 1)) * kp.slot_bytes;
        const char* ws = wsrc + c * wstep;
        for (int i = 0; i < 2 * ntaps; ++i)
            __builtin_amdgcn_global_load_lds((gbl_void_t*)(ws + i * 1024), (lds_void_t*)(dst + i * 1024), 16, 0, 0);
#pragma unroll
        for (int g = 0; g < kMaxXg; ++g)
            if (g < kp.xg) __builtin_amdgcn_global_load_lds((gbl_void_t*)(xsrc[g] + c * 64), (lds_void_t*)(dst + wbytes + g * 1024), 16, 0, 0);
    };
    // raw window of chunk c (slot c & 3) -> converted image (buffer c & 1); conv_cl.hip's store_x, with LDS as the source
    const int nxf4 = kp.xrows * 4;
    auto convert = [&](int c) {
        const char* raw = ring + (c & (kSlots - 1)) * kp.slot_bytes + wbytes;
        char* xs_hi = cvt + (c & 1) * 2 * cvt_bytes;
        char* xs_lo = xs_hi + cvt_bytes;
        constexpr int NCVL = (NCV + kConverters - 1) / kConverters;
        f32x4v rv[NCVL];
#pragma unroll
        for (int i = 0; i < NCVL; ++i)
            if (i == 0 || i * 64 * kConverters < nxf4)
                rv[i] = *reinterpret_cast<const f32x4v*>(raw + min((wave - 2) * 64 + lane + i * 64 * kConverters, nxf4 - 1) * 16);
#pragma unroll
        for (int i = 0; i < NCVL; ++i) {
            if (i > 0 && i * 64 * kConverters >= nxf4) break;
            const int idx = (wave - 2) * 64 + lane + i * 64 * kConverters;
            float4 v = make_float4(rv[i][0], rv[i][1], rv[i][2], rv[i][3]);
            const int row = idx >> 2, q = idx & 3;
            const int pos = wstart + row;
            if (pos < 0 || pos >= NB) v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (slope != 1.0f) {
                v.x = v.x >= 0.f ? v.x : v.x * slope;
                v.y = v.y >= 0.f ? v.y : v.y * slope;
                v.z = v.z >= 0.f ? v.z : v.z * slope;
                v.w = v.w >= 0.f ? v.w : v.w * slope;
            }
            const int off = row * 32 + ((((q >> 1) ^ (row >> 3)) & 1) << 4) + ((q & 1) << 3);
            bf16x4 h, l;
            h[0] = (__bf16)v.x; h[1] = (__bf16)v.y; h[2] = (__bf16)v.z; h[3] = (__bf16)v.w;
            l[0] = (__bf16)(v.x - (float)h[0]); l[1] = (__bf16)(v.y - (float)h[1]);
            l[2] = (__bf16)(v.z - (float)h[2]); l[3] = (__bf16)(v.w - (float)h[3]);
            if (idx < nxf4) {
                *reinterpret_cast<bf16x4*>(xs_hi + off) = h;
                *reinterpret_cast<bf16x4*>(xs_lo + off) = l;
            }
        }
    };

    f32x16 acc[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const int lcol = lane & 31, lh = lane >> 5;
    // tap loop software-pipelined as in conv_cl.hip: the fragments of tap t + 1 are requested before the MFMAs of tap t are issued
    struct Frags {
        bf16x8 ah, al, bh[TN], bl[TN];
    };
    auto mma_chunk = [&](int c) {
        const char* wsm = ring + (c & (kSlots - 1)) * kp.slot_bytes + lane * 16;
        const char* xs_hi = cvt + (c & 1) * 2 * cvt_bytes;
        const char* xs_lo = xs_hi + cvt_bytes;
        auto load_frags = [&](Frags& f, int tap) {
            const int sh = kp.sh0 + tap * kp.sh_step;
            f.ah = *reinterpret_cast<const bf16x8*>(wsm + tap * 2048);
            f.al = *reinterpret_cast<const bf16x8*>(wsm + tap * 2048 + 1024);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int row = j * 32 + lcol + sh;
                const int off = row * 32 + (((lh ^ (row >> 3)) & 1) << 4);
                f.bh[j] = *reinterpret_cast<const bf16x8*>(xs_hi + off);
                f.bl[j] = *reinterpret_cast<const bf16x8*>(xs_lo + off);
            }
        };
        auto mfma_frags = [&](const Frags& f) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.al, f.bh[j], acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah, f.bl[j], acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah, f.bh[j], acc[j], 0, 0, 0);
            }
        };
        Frags fa, fb;
        load_frags(fa, 0);
        int tap = 0;
        for (; tap + 2 <= ntaps; tap += 2) {
            load_frags(fb, tap + 1);
            mfma_frags(fa);
            load_frags(fa, min(tap + 2, ntaps - 1));
            mfma_frags(fb);
        }
        if (tap < ntaps) mfma_frags(fa);
    };

    // ---- pipeline --------------------------------------------------------------------------------------------------------------------------
    // One raw barrier per chunk (__syncthreads would drain the DMAs in flight with vmcnt(0)).  After barrier B(c): the raw chunk c + 1 has
    // landed (every loader waited for ITS loads: they retire in order, `per` of them per chunk and wave) and the window of chunk c is
    // converted.  Between B(c) and B(c + 1) wave 0 runs the MFMAs of chunk c while the loaders convert the window of chunk c + 1 and issue
    // the DMAs of one more chunk, into a slot that holds neither chunk c nor c + 1 (2 <= ahead < kSlots).
    const int per = nload;      // wave 1's loads per chunk
    int issued = 0;
    if (wave == 1) {
        for (; issued < min(kp.ahead, nchunks); ++issued) stage(issued);
        wait_vm_dyn((issued - 1) * per);                       // chunk 0
    }
    __builtin_amdgcn_s_barrier();
    if (wave >= 2) convert(0);
    for (int c = 0; c < nchunks; ++c) {
        if (wave == 1) wait_vm_dyn(max(issued - c - 2, 0) * per);   // chunk c + 1 (or everything, at the tail)
        __builtin_amdgcn_s_barrier();                           // B(c)
        if (wave == 0) {
            mma_chunk(c);
        } else if (wave == 1) {
            if (issued < nchunks) {
                stage(issued);
                ++issued;
            }
        } else {
            if (c + 1 < nchunks) convert(c + 1);
        }
    }
    if (wave != 0) return;

    // ---- epilogue: conv_cl.hip's k-major form ------------------------------------------------------------------------------------------------
    int nn[TN];
    bool nok[TN], keepn[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        nn[j] = n0 + j * 32 + lcol;
        nok[j] = nn[j] < N;
        keepn[j] = true;
        if (p.mask) {
            const int nc = min(nn[j], N - 1);
            keepn[j] = p.mask[kp.mask_shift >= 0 ? (nc >> kp.mask_shift) : (nc / p.mask_div)] != 0;
        }
    }
    float brow[16], rr[16][TN], old[16][TN];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int mc = min(m0 + (r & 3) + 8 * (r >> 2) + 4 * lh, M - 1);
        brow[r] = p.bias ? p.bias[mc] : 0.f;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int nc = min(nn[j], N - 1);
            rr[r][j] = p.R ? p.R[(int64_t)mc * p.ldr + nc] : 0.f;
            old[r][j] = p.accumulate ? p.Y[(int64_t)mc * p.ldy + nc] : 0.f;
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            if (!nok[j]) continue;
            float v = acc[j][r] + brow[r];
            if (p.act == ACT_RELU) v = fmaxf(v, 0.f);
            else if (p.act == ACT_GELU) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
            else if (p.act == ACT_TANH) v = tanhf(v);
            v *= p.alpha;
            if (p.R) v += rr[r][j];
            v *= p.beta;
            if (p.accumulate) v += old[r][j];
            if (!keepn[j]) v = 0.f;
            p.Y[(int64_t)m * p.ldy + nn[j]] = v;
        }
    }
}

}  // namespace

// true = launched.  Channels-last input, k-major output, split-bf16 operands, unphased; the caller has found conv_cl's grid small.
bool launch_conv_cl_small(const ConvClParams& p, int mask_shift, hipStream_t stream) {
    if (!p.split || p.f16 || p.in_km || !p.out_km || p.phase_rows < (1 << 30) || (p.K & 15) != 0 || p.N <= 0) return false;
    const int64_t waves = (int64_t)p.nmt * ((p.N + 31) / 32);
    if (waves > 1024) return false;       // beyond one wave per SIMD the re-read weights cost more than the latency they hide
    ClSmallParams kp;
    kp.p = p;
    int smin = p.shift[0], smax = p.shift[0];
    for (int t = 1; t < p.ntaps; ++t) {
        smin = std::min(smin, p.shift[t]);
        smax = std::max(smax, p.shift[t]);
    }
    if (smax - smin > 64) return false;
    kp.wshift0 = smin;
    kp.sh0 = p.shift[0] - smin;
    kp.sh_step = p.ntaps > 1 ? p.shift[1] - p.shift[0] : 0;
    for (int t = 1; t < p.ntaps; ++t)
        if (p.shift[t] - p.shift[t - 1] != kp.sh_step) return false;
    constexpr int TN = 1;
    kp.xrows = 32 * TN + (smax - smin);
    kp.xg = (kp.xrows + 15) / 16;
    kp.mask_shift = mask_shift;
    kp.slot_bytes = p.ntaps * 2048 + kp.xg * 1024;
    const int per = 2 * p.ntaps + kp.xg;
    kp.ahead = std::max(2, std::min(std::min(kSlots - 2, 4), 62 / per));   // one wave holds every outstanding load: 6-bit vmcnt
    const size_t lds = (size_t)kSlots * kp.slot_bytes + 4 * (size_t)kp.xrows * 32;
    if (lds > 160 * 1024) return false;
    auto kern = conv_cl_small_kernel<TN>;
    static std::atomic<uint64_t> lds_allowed{0};   // per (kernel instantiation, device)
    allow_full_lds(reinterpret_cast<const void*>(kern), lds_allowed);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const bool prof = conv_prof_active();
    if (prof) {
        HIP_CHECK(hipEventCreate(&e0));
        HIP_CHECK(hipEventCreate(&e1));
        HIP_CHECK(hipEventRecord(e0, stream));
    }
    dim3 grid((p.N + 32 * TN - 1) / (32 * TN), p.nmt);
    hipLaunchKernelGGL(kern, grid, dim3(64 * (kLoaders + 1)), lds, stream, kp);
    HIP_CHECK(hipGetLastError());
    if (prof) {
        HIP_CHECK(hipEventRecord(e1, stream));
        conv_prof_add(13, 2.0 * p.M * (double)p.N * p.K * p.ntaps, e0, e1);   // counted with conv_cl_km<1,split-bf16>
    }
    return true;
}

}  // namespace sbv2
