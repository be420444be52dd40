// Channels-last implicit-GEMM convolution on PRE-SPLIT activations (gfx950, v_mfma_f32_32x32x16_bf16), all staging by LDS-DMA.
//
// conv_cl.hip converts its f32 input (leaky-ReLU, bf16 hi/lo split) inside the consumer: every workgroup re-converts the window
// of every 16-channel chunk (gy = Cout/64 times per element), the conversion VALU competes with the MFMAs of the same SIMD and
// the staging registers (68 VGPRs) leave no room to double-buffer fragments; ablating that staging was worth 25 % (k = 7) to
// 44 % (k = 3) of the kernel.  Here the PRODUCER's epilogue writes the activated, split operand once:
//
//   "A tensor" of a plane x[pos][C]:  A[chunk = c / 16][part = hi | lo][front + pos][16 bf16],  4 bytes per element like f32,
//   with the two 16-byte halves (8 channels each) of a 32-byte row swapped when bit 3 of the padded row index is set (the
//   ds_read_b128 bank swizzle, applied where the row is written so that the LDS image is a plain copy), and zero halo rows in
//   front of and behind the plane so that windows never need clamping.
//
// A 16-channel chunk of a 256-position window is then ONE contiguous run per part in HBM, and weights are pre-packed fragment
// blocks as in conv_cl.hip, so both operands go HBM/L2 -> LDS with global_load_lds_dwordx4: no staging VGPRs, no conversion, no
// ds_write.  Pipeline: stage = (chunk, group of <= 4 taps); weights double-buffered per stage, windows double-buffered per chunk,
// one barrier per stage; the DMA for stage s+1 (and for the next chunk's window) is issued before the MFMAs of stage s.
//
// Results: Y raw f32 (residual stream) and/or YA = A tensor of lrelu(result) for the next convolution.  Same arithmetic as
// conv_cl.hip in the same order (hi*hi + hi*lo + lo*hi, f32 accumulate), so outputs are bit-identical to it.
//
// STATUS (round 1): parity-tested through sbv2_debug_conv1d_ps (tests/test_gpu_parity.py::test_conv1d_ps_kernel) and timed by
// tests/perf_ps_one.py, NOT yet wired into the decoder: measured equal to conv_cl.hip (+-5 %) on the decoder's shapes.  The
// ablations made with it show why: on random data the MFMA stream alone (no LDS reads, no DMA, no epilogue)
// already takes 0.82 of the 1.33 ms of a C = 128, k = 7 launch, i.e. 1.29 PFLOP/s executed = the data-dependent (power-limited)
// MFMA ceiling of the part, and the remaining costs are additive with it rather than hidden behind it; with zero-filled
// operands the identical instruction stream runs 28 % faster.
#include <type_traits>

#include "common.h"

namespace sbv2 {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2v __attribute__((ext_vector_type(2)));

constexpr int kPsThreads = 256;
constexpr int kPsNT = 256;
constexpr int kPsMaxSpan = 56;
constexpr int kPsG = 4;   // taps per stage
constexpr int kPsXR = 320; // window rows per chunk in LDS: 256 + tap span (<= 56) + alignment of the window start to 8 rows (<= 7), rounded to 32

struct PsKernelParams {
    ConvClParams p;
    int wshift0;   // min shift
    int nmt;
    int sh0, sh_step;
    int mask_shift;
    int ntg;       // tap groups per chunk
};

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

// 64 lanes x 16 bytes: global (per-lane address) -> LDS (wave-uniform base + lane * 16), no VGPR destination
__device__ __forceinline__ void glds16(const char* g, char* l) {
    __builtin_amdgcn_global_load_lds((gbl_void_t*)g, (lds_void_t*)l, 16, 0, 0);
}

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    bf16x2 h;
    h[0] = (__bf16)a;
    h[1] = (__bf16)b;
    return __builtin_bit_cast(unsigned, h);
}

template <bool SPLIT>
__global__ __launch_bounds__(kPsThreads) void conv_ps_kernel(const PsKernelParams kp) {
    constexpr int PARTS = SPLIT ? 2 : 1;
    constexpr int TM = 2, TN = 2, G = kPsG;
    constexpr int WB = G * TM * PARTS * 1024;   // bytes of one weight stage buffer
    constexpr int XP = kPsXR * 32;              // bytes of one part of a window
    constexpr int XB = XP * PARTS;              // bytes of one window buffer
    constexpr int XPIECES = (kPsXR / 32) * PARTS;
    const ConvClParams& p = kp.p;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    char* wbuf0 = smem + 2 * XB;

    // The vector ALU is the scarce issue port next to the MFMAs (conv_cl.hip: ~5 VALU per MFMA, PMC SQ_INSTS_VALU / SQ_INSTS_MFMA), so
    // everything wave-uniform is kept scalar: the wave index is read into an SGPR, DMA addresses are SGPR base + one constant
    // per-lane offset, and fragment addresses are one VGPR per tap plus instruction offsets.
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn0 = wave * 64;
    // XCD-aware tile order, as in conv_cl.hip
    const int gy = kp.nmt / TM;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;
    const int by = slot % gy;
    const int bx = (slot / gy) * 8 + xcd;
    const int m0 = by * (TM * 32);
    const int n0 = bx * kPsNT;
    if (n0 >= p.N) return;
    const int M = p.M, N = p.N, ntaps = p.ntaps;
    const int nchunks = p.K >> 4;
    const int ntg = kp.ntg;
    // window row 0 = padded row wstart_al (a multiple of 8, so the swizzle bit of a row is bit 3 of its WINDOW row index xor the
    // uniform bit 3 of wstart_al); tap t reads window rows sh_t + [0, 256), sh_t = (start - wstart_al) + sh0 + t * step
    const int start = p.xa_front + n0 + kp.wshift0;
    const int wstart_al = start & ~7;
    const int sh_base = (start - wstart_al) + kp.sh0;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const unsigned lane16 = lane * 16;
    const char* xa = reinterpret_cast<const char*>(p.XA) + (int64_t)wstart_al * 32;   // uniform
    auto issue_x = [&](int chunk, int b) {
        char* dst = smem + b * XB;
        const char* src = xa + (int64_t)chunk * PARTS * p.xa_plane;
        for (int i = wave; i < XPIECES; i += 4) {   // scalar loop
            const int part = (PARTS == 2 && i >= kPsXR / 32) ? 1 : 0;
            const int rb = i - part * (kPsXR / 32);
            glds16(src + (int64_t)part * p.xa_plane + rb * 1024 + lane16, dst + part * XP + rb * 1024);
        }
    };
    const char* wsrc = reinterpret_cast<const char*>(p.W);
    auto issue_w = [&](int chunk, int tg, int b) {
        const int t0 = tg * G;
        const int g = min(G, ntaps - t0);
        char* dst = wbuf0 + b * WB;
        const int per_i = g * PARTS;
        for (int q = wave; q < TM * per_i; q += 4) {   // scalar loop
            const int i = q >= per_i ? 1 : 0;
            const int r = q - i * per_i;   // tl * PARTS + part
            glds16(wsrc + ((((int64_t)chunk * kp.nmt + by * TM + i) * ntaps + t0) * PARTS + r) * 1024 + lane16, dst + (i * G * PARTS + r) * 1024);
        }
    };

    // vmcnt retires in issue order: a stage issues the next stage's weights FIRST and the next chunk's window after them, so that
    // waiting for all but this wave's nx window pieces retires the weights and leaves the window (needed a stage later) in flight
    const int nx = (XPIECES - wave + 3) >> 2;   // window pieces this wave issues
    auto stage_sync = [&](int leave) {
        switch (leave) {
            case 0: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); break;
            case 1: asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory"); break;
            case 5: asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory"); break;
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    issue_x(0, 0);
    issue_w(0, 0, 0);
    stage_sync(0);

    const int lcol = lane & 31, lh = lane >> 5;
    const int lh16 = (lh << 4) ^ ((wstart_al & 8) << 1);
    const int row0 = wn0 + lcol + sh_base;   // window row of this lane's column j = 0 at tap 0
    struct Frags {
        bf16x8 bh[TN], bl[TN], ah[TM], al[TM];
    };
    int s = 0;
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const char* xb = smem + (chunk & 1) * XB;
        for (int tg = 0; tg < ntg; ++tg, ++s) {
            // prefetch: next stage's weights into the buffer read during stage s-1, and (first stage of a chunk) the next chunk's
            // window into the buffer read during chunk-1; both were released by the barrier that ended the previous stage
            const bool last_tg = tg + 1 == ntg;
            if (!last_tg) issue_w(chunk, tg + 1, (s + 1) & 1);
            else if (chunk + 1 < nchunks) issue_w(chunk + 1, 0, (s + 1) & 1);
            const bool xfly = tg == 0 && chunk + 1 < nchunks;
            if (xfly) issue_x(chunk + 1, (chunk + 1) & 1);

            const char* wb = wbuf0 + (s & 1) * WB + lane16;
            const int t0 = tg * G;
            const int g = min(G, ntaps - t0);
            auto load_frags = [&](Frags& f, int tl) {
                // column j = 1 is 32 rows further: same swizzle bit, +1024 bytes; the lo part is +XP: instruction offsets
                const int row = row0 + (t0 + tl) * kp.sh_step;
                const char* bp = xb + (row << 5) + (((row << 1) ^ lh16) & 16);
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    f.bh[j] = *reinterpret_cast<const bf16x8*>(bp + j * 1024);
                    if (SPLIT) f.bl[j] = *reinterpret_cast<const bf16x8*>(bp + j * 1024 + XP);
                }
                const char* ap = wb + tl * (PARTS * 1024);
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    f.ah[i] = *reinterpret_cast<const bf16x8*>(ap + i * (G * PARTS * 1024));
                    if (SPLIT) f.al[i] = *reinterpret_cast<const bf16x8*>(ap + i * (G * PARTS * 1024) + 1024);
                }
            };
            auto mfma_frags = [&](const Frags& f) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        if (SPLIT) {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.al[i], f.bh[j], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[i], f.bl[j], acc[i][j], 0, 0, 0);
                        }
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[i], f.bh[j], acc[i][j], 0, 0, 0);
                    }
            };
            Frags fa, fb;
            load_frags(fa, 0);
            int tl = 0;
            for (; tl + 2 <= g; tl += 2) {
                load_frags(fb, tl + 1);
                mfma_frags(fa);
                load_frags(fa, min(tl + 2, g - 1));
                mfma_frags(fb);
            }
            if (tl < g) mfma_frags(fa);
            // stage s fully read by every wave; stage s+1's weights landed and are published; the window issued in this stage is
            // waited for only when the next stage is the first of its chunk
            stage_sync((xfly && !last_tg) ? nx : 0);
        }
    }

    // ---- epilogue: per-wave LDS transpose to full channels-last lines (conv_cl.hip), plus the A-tensor result ----------------
    // Address arithmetic is incremental (64-bit pointers advanced by wave-uniform strides): the epilogue's VALU count matters too.
    const bool phased = p.phase_rows < (1 << 30);
    float* tile = reinterpret_cast<float*>(smem) + wave * (64 * 36);
    const float beta = p.beta, sl = p.out_slope;
    const bool odd = lane & 1;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4v v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                *reinterpret_cast<f32x4v*>(tile + (j * 32 + lcol) * 36 + 8 * q + 4 * lh) = v;
            }
        const int c4 = (lane & 7) * 4;
        const int m = m0 + i * 32 + c4;
        int co = m, po = 0, ostride = 1;
        if (phased) {
            const int ph = m / p.phase_rows;
            co = m - ph * p.phase_rows;
            ostride = p.out_stride;
#pragma unroll
            for (int t = 0; t < kMaxPhases; ++t) po = (ph == t) ? p.phase_off[t] : po;
        }
        float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.bias && m < M) b4 = *reinterpret_cast<const float4*>(p.bias + co);
        const int nfirst = n0 + wn0 + (lane >> 3);
        int pos = nfirst * ostride + po;              // < 2^31 (checked by the caller)
        const int pstep = 8 * ostride;
        const float* rp = p.R ? p.R + (int64_t)pos * p.ldr + co : nullptr;
        float* yp = p.Y ? p.Y + (int64_t)pos * p.ldy + co : nullptr;
        const int c8 = co & ~7;                       // first of the lane pair's 8 channels
        char* yap = p.YA ? reinterpret_cast<char*>(p.YA) + ((int64_t)(c8 >> 4) * PARTS + (odd ? 1 : 0)) * p.ya_plane + ((int64_t)p.ya_front + pos) * 32 : nullptr;
        const int64_t rstep = (int64_t)pstep * p.ldr, ystep = (int64_t)pstep * p.ldy;
        int prow = p.ya_front + pos;                  // only bit 3 is used
        const float* trow = tile + (lane >> 3) * 36 + c4;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int n = nfirst + it * 8;
            const f32x4v a = *reinterpret_cast<const f32x4v*>(trow + it * 8 * 36);
            const bool valid = n < N && m < M;   // uniform over each lane pair (M % 8 == 0)
            float4 v = make_float4(a[0] + b4.x, a[1] + b4.y, a[2] + b4.z, a[3] + b4.w);
            if (valid && rp) {
                const float4 r = *reinterpret_cast<const float4*>(rp);
                v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
            }
            if (beta != 1.0f) { v.x *= beta; v.y *= beta; v.z *= beta; v.w *= beta; }
            if (valid && p.accumulate) {
                const float4 o = *reinterpret_cast<const float4*>(yp);
                v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
            }
            if (valid && p.mask) {
                const int mi = kp.mask_shift >= 0 ? (pos >> kp.mask_shift) : (pos / p.mask_div);
                if (!p.mask[mi]) v = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            if (valid && yp) *reinterpret_cast<float4*>(yp) = v;
            if (yap) {
                // A tensor of lrelu(v): the even lane of a pair stores 8 channels of the hi part, the odd lane the same 8 channels of
                // the lo part (one 16-byte store each); lane ^ 1 is a DPP quad permutation, not an LDS shuffle
                v.x = v.x >= 0.f ? v.x : v.x * sl;
                v.y = v.y >= 0.f ? v.y : v.y * sl;
                v.z = v.z >= 0.f ? v.z : v.z * sl;
                v.w = v.w >= 0.f ? v.w : v.w * sl;
                bf16x4 h;
                h[0] = (__bf16)v.x; h[1] = (__bf16)v.y; h[2] = (__bf16)v.z; h[3] = (__bf16)v.w;
                const u32x2v hu = __builtin_bit_cast(u32x2v, h);
                u32x2v lu = {0u, 0u};
                if (SPLIT) lu = {pack_bf16(v.x - (float)h[0], v.y - (float)h[1]), pack_bf16(v.z - (float)h[2], v.w - (float)h[3])};
                const u32x2v send = odd ? hu : lu;
                u32x2v recv;
                recv[0] = (unsigned)__builtin_amdgcn_mov_dpp((int)send[0], 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
                recv[1] = (unsigned)__builtin_amdgcn_mov_dpp((int)send[1], 0xB1, 0xF, 0xF, true);
                const u32x4v out = odd ? u32x4v{recv[0], recv[1], lu[0], lu[1]} : u32x4v{hu[0], hu[1], recv[0], recv[1]};
                const int half16 = (((c8 >> 3) ^ (prow >> 3)) & 1) << 4;
                if (valid && (SPLIT || !odd)) *reinterpret_cast<u32x4v*>(yap + half16) = out;
                yap += (int64_t)pstep * 32;
                prow += pstep;
            }
            pos += pstep;
            if (rp) rp += rstep;
            if (yp) yp += ystep;
        }
    }
}

template <bool SPLIT>
static void launch_ps(PsKernelParams kp, hipStream_t stream) {
    constexpr int PARTS = SPLIT ? 2 : 1;
    const ConvClParams& p = kp.p;
    size_t lds = (size_t)2 * kPsXR * 32 * PARTS + (size_t)2 * kPsG * 2 * PARTS * 1024;
    lds = std::max<size_t>(lds, 4 * 64 * 36 * sizeof(float));
    SBV2_REQUIRE(lds <= 160 * 1024, "conv_ps: LDS budget exceeded");
    auto kern = conv_ps_kernel<SPLIT>;
    static bool attr_set = false;
    if (!attr_set) {
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    const int ntx = round_up((p.N + kPsNT - 1) / kPsNT, 8);
    dim3 grid(ntx * (kp.nmt / 2));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const bool prof = conv_prof_active();
    if (prof) {
        HIP_CHECK(hipEventCreate(&e0));
        HIP_CHECK(hipEventCreate(&e1));
        HIP_CHECK(hipEventRecord(e0, stream));
    }
    hipLaunchKernelGGL(kern, grid, dim3(kPsThreads), lds, stream, kp);
    HIP_CHECK(hipGetLastError());
    if (prof) {
        HIP_CHECK(hipEventRecord(e1, stream));
        conv_prof_add(SPLIT ? 17 : 18, 2.0 * p.M * (double)p.N * p.K * p.ntaps, e0, e1);
    }
}

void launch_conv_ps(const ConvClParams& p, hipStream_t stream) {
    SBV2_REQUIRE(p.ntaps >= 1 && p.ntaps <= kMaxTaps, "bad tap count");
    SBV2_REQUIRE(p.XA && (p.K & 15) == 0, "conv_ps: pre-split input needs Cin % 16 == 0");
    SBV2_REQUIRE((p.M & 31) == 0 && p.tm == 2 && (p.nmt & 1) == 0, "conv_ps: Cout must be a multiple of 64 rows tiles (tm = 2)");
    SBV2_REQUIRE(p.Y || p.YA, "conv_ps: no output");
    SBV2_REQUIRE(!p.Y || (p.ldy & 3) == 0, "conv_ps: output pitch must be a multiple of 4 floats");
    SBV2_REQUIRE(!p.accumulate || p.Y, "conv_ps: accumulate needs the raw output");
    if (p.N <= 0) return;
    PsKernelParams kp;
    kp.p = p;
    int smin = p.shift[0], smax = p.shift[0];
    for (int t = 1; t < p.ntaps; ++t) {
        smin = std::min(smin, p.shift[t]);
        smax = std::max(smax, p.shift[t]);
    }
    kp.wshift0 = smin;
    SBV2_REQUIRE(smax - smin <= kPsMaxSpan, "conv_ps: tap span too large");
    // the window of the last tile may run past the plane by up to kPsXR rows, the first one starts up to 15 - wshift0 rows before it
    SBV2_REQUIRE(p.xa_front + smin >= 8 && p.xa_back >= kPsXR + smax, "conv_ps: input halo too small");
    kp.sh0 = p.shift[0] - kp.wshift0;
    kp.sh_step = p.ntaps > 1 ? p.shift[1] - p.shift[0] : 0;
    for (int t = 1; t < p.ntaps; ++t) SBV2_REQUIRE(p.shift[t] - p.shift[t - 1] == kp.sh_step, "conv_ps: tap shifts must be an arithmetic progression");
    kp.nmt = p.nmt;
    kp.ntg = (p.ntaps + kPsG - 1) / kPsG;
    kp.mask_shift = -1;
    if (p.mask && p.mask_div > 0 && (p.mask_div & (p.mask_div - 1)) == 0) {
        int s = 0;
        while ((1 << s) < p.mask_div) ++s;
        kp.mask_shift = s;
    }
    if (p.split) launch_ps<true>(kp, stream);
    else launch_ps<false>(kp, stream);
}

// ---------------------------------------------------------------------------------------------
// raw channels-last plane -> A tensor (used where the producer is not a conv_ps launch), and halo zeroing
// ---------------------------------------------------------------------------------------------
template <bool SPLIT>
__global__ void k_split_cl(const float* __restrict__ X, int ldx, int64_t N, int C, float slope, char* __restrict__ A, int64_t plane, int front) {
    constexpr int PARTS = SPLIT ? 2 : 1;
    const int oct = C >> 3;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * oct) return;
    const int64_t n = idx / oct;
    const int c8 = (int)(idx - n * oct) * 8;
    const float4 a = *reinterpret_cast<const float4*>(X + n * ldx + c8);
    const float4 b = *reinterpret_cast<const float4*>(X + n * ldx + c8 + 4);
    float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    unsigned hu[4], lu[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float x0 = v[2 * j], x1 = v[2 * j + 1];
        x0 = x0 >= 0.f ? x0 : x0 * slope;
        x1 = x1 >= 0.f ? x1 : x1 * slope;
        const __bf16 h0 = (__bf16)x0, h1 = (__bf16)x1;
        hu[j] = pack_bf16(x0, x1);
        lu[j] = pack_bf16(x0 - (float)h0, x1 - (float)h1);
    }
    const int64_t prow = front + n;
    const int half = ((c8 >> 3) ^ (int)(prow >> 3)) & 1;
    char* dst = A + (int64_t)(c8 >> 4) * PARTS * plane + prow * 32 + half * 16;
    *reinterpret_cast<u32x4v*>(dst) = u32x4v{hu[0], hu[1], hu[2], hu[3]};
    if (SPLIT) *reinterpret_cast<u32x4v*>(dst + plane) = u32x4v{lu[0], lu[1], lu[2], lu[3]};
}

void split_cl(const float* X, int ldx, int64_t N, int C, float slope, int split, void* A, int64_t plane, int front, hipStream_t stream) {
    SBV2_REQUIRE((C & 15) == 0 && (ldx & 3) == 0, "split_cl: channels must be a multiple of 16");
    const int64_t total = N * (C >> 3);
    if (total <= 0) return;
    const unsigned blocks = (unsigned)((total + 255) / 256);
    if (split) hipLaunchKernelGGL(k_split_cl<true>, dim3(blocks), dim3(256), 0, stream, X, ldx, N, C, slope, (char*)A, plane, front);
    else hipLaunchKernelGGL(k_split_cl<false>, dim3(blocks), dim3(256), 0, stream, X, ldx, N, C, slope, (char*)A, plane, front);
    HIP_CHECK(hipGetLastError());
}

__global__ void k_ps_zero_halo(char* A, int nplanes, int64_t plane, int front, int64_t rows, int back) {
    // 16 bytes per thread; halo rows [0, front) and [front + rows, front + rows + back) of every plane
    const int64_t per_plane = (int64_t)(front + back) * 2;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= per_plane * nplanes) return;
    const int pl = (int)(idx / per_plane);
    int64_t r = idx - pl * per_plane;
    const int64_t f2 = (int64_t)front * 2;
    const int64_t off = r < f2 ? r * 16 : ((int64_t)front + rows) * 32 + (r - f2) * 16;
    *reinterpret_cast<u32x4v*>(A + pl * plane + off) = u32x4v{0u, 0u, 0u, 0u};
}

void ps_zero_halo(void* A, int nplanes, int64_t plane, int front, int64_t rows, int back, hipStream_t stream) {
    const int64_t total = (int64_t)(front + back) * 2 * nplanes;
    const unsigned blocks = (unsigned)((total + 255) / 256);
    hipLaunchKernelGGL(k_ps_zero_halo, dim3(blocks), dim3(256), 0, stream, (char*)A, nplanes, plane, front, rows, back);
    HIP_CHECK(hipGetLastError());
}

}  // namespace sbv2
