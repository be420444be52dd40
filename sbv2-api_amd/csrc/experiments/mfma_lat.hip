// Experiment (not built into the library): time per MFMA for dependent chains and for 2 / 4 independent accumulators, one wave per CU or
// one wave per SIMD, f32 16x16x4 / 32x32x2 and bf16 32x32x16.  hipcc --offload-arch=gfx950 -O3 mfma_lat.hip -o /tmp/mfma_lat && /tmp/mfma_lat
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC>
__global__ void k16(float* out, int iters, float a, float b) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0];
    if (s == 1234.5f) out[0] = s;
}
template <int NACC>
__global__ void k32(float* out, int iters, float a, float b) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0];
    if (s == 1234.5f) out[0] = s;
}
template <int NACC>
__global__ void kb(float* out, int iters, float a) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0;
    bf16x8 x, y;
    for (int t = 0; t < 8; ++t) { x[t] = (__bf16)(a + t); y[t] = (__bf16)(a - t); }
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[i], 0, 0, 0);
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0];
    if (s == 1234.5f) out[0] = s;
}

// full-chip rate with operands that change every instruction (random mantissas): the power-managed ceiling of each pipe
template <int NACC>
__global__ void k32r(float* out, int iters, const float* rnd) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0;
    float a[8], b[8];
    for (int t = 0; t < 8; ++t) { a[t] = rnd[(threadIdx.x * 8 + t) & 4095]; b[t] = rnd[(threadIdx.x * 8 + t + 2048) & 4095]; }
    for (int it = 0; it < iters; it += 8)
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[(t + i) & 7], acc[i], 0, 0, 0);
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0];
    if (s == 1234.5f) out[0] = s;
}

// the f32 GEMM's inner loop in isolation: per chunk 8 A + 16 B fragment words from LDS (ds_read_b32), then 16 MFMAs on two accumulators
__global__ void k32lds(float* out, int chunks, const float* rnd) {
    extern __shared__ float sm[];
    for (int i = threadIdx.x; i < 16 * 192 * 2; i += blockDim.x) sm[i] = rnd[i & 4095];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * 64, lcol = lane & 31, lrow = lane >> 5;
    f32x16 acc0, acc1;
    for (int r = 0; r < 16; ++r) { acc0[r] = 0; acc1[r] = 0; }
    for (int c = 0; c < chunks; ++c) {
        const float* wsb = sm + (c & 1) * (16 * 192) + wm0 + lcol;
        const float* xsb = sm + (c & 1) * (16 * 192) + 16 * 64 + wn0 + lcol;
        float a[8], b0[8], b1[8];
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const int kr = kk * 2 + lrow;
            a[kk] = wsb[kr * 64];
            b0[kk] = xsb[kr * 128];
            b1[kk] = xsb[kr * 128 + 32];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], b0[kk], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], b1[kk], acc1, 0, 0, 0);
        }
    }
    if (acc0[0] + acc1[0] == 1234.5f) out[0] = acc0[0];
}

template <int NACC>
__global__ void kbr(float* out, int iters, const float* rnd) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0;
    bf16x8 x[8], y[8];
    for (int t = 0; t < 8; ++t)
        for (int e = 0; e < 8; ++e) {
            x[t][e] = (__bf16)rnd[(threadIdx.x * 64 + t * 8 + e) & 4095];
            y[t][e] = (__bf16)rnd[(threadIdx.x * 64 + t * 8 + e + 1777) & 4095];
        }
    for (int it = 0; it < iters; it += 8)
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[t], y[(t + i) & 7], acc[i], 0, 0, 0);
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0];
    if (s == 1234.5f) out[0] = s;
}

// the 64 x 64 tile's inner loop: per chunk 8 A + 8 B fragment words, 8 MFMAs on ONE accumulator; 32 KB of LDS -> five workgroups per CU
__global__ void k32lds1(float* out, int chunks, const float* rnd) {
    extern __shared__ float sm[];
    for (int i = threadIdx.x; i < 16 * 128 * 2; i += blockDim.x) sm[i] = rnd[i & 4095];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * 32, lcol = lane & 31, lrow = lane >> 5;
    f32x16 acc0;
    for (int r = 0; r < 16; ++r) acc0[r] = 0;
    for (int c = 0; c < chunks; ++c) {
        const float* wsb = sm + (c & 1) * (16 * 128) + wm0 + lcol;
        const float* xsb = sm + (c & 1) * (16 * 128) + 16 * 64 + wn0 + lcol;
        float a[8], b0[8];
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const int kr = kk * 2 + lrow;
            a[kk] = wsb[kr * 64];
            b0[kk] = xsb[kr * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], b0[kk], acc0, 0, 0, 0);
    }
    if (acc0[0] == 1234.5f) out[0] = acc0[0];
}
template <class F>
static float time_us(F f) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < 3; ++i) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f / 3;
}
int main() {
    float* out; hipMalloc(&out, 64);
    const int iters = 4096;
    for (int wg : {32, 256, 1024}) {
        for (int threads : {64, 256}) {
            printf("---- %d workgroups x %d threads, %d iterations ----\n", wg, threads, iters);
#define RUN(name, kern, nacc) { float us = time_us([&] { kern<<<wg, threads>>>(out, iters, 1.5f, 0.5f); }); \
            printf("%-14s %d independent accumulators: %7.1f us  -> %6.2f ns per MFMA per wave\n", name, nacc, us, us * 1e3 / (iters * nacc)); }
#define RUNB(name, kern, nacc) { float us = time_us([&] { kern<<<wg, threads>>>(out, iters, 1.5f); }); \
            printf("%-14s %d independent accumulators: %7.1f us  -> %6.2f ns per MFMA per wave\n", name, nacc, us, us * 1e3 / (iters * nacc)); }
            RUN("f32 16x16x4", k16<1>, 1) RUN("f32 16x16x4", k16<2>, 2) RUN("f32 16x16x4", k16<4>, 4)
            RUN("f32 32x32x2", k32<1>, 1) RUN("f32 32x32x2", k32<2>, 2)
            RUNB("bf16 32x32x16", kb<1>, 1) RUNB("bf16 32x32x16", kb<2>, 2) RUNB("bf16 32x32x16", kb<4>, 4)
        }
    }
    {   // random-data ceiling of the f32 pipe: 1024 workgroups x 256 threads (4 waves per SIMD), 2 accumulators per wave
        float* rnd; hipMalloc(&rnd, 4096 * 4);
        float h[4096]; srand(1); for (int i = 0; i < 4096; ++i) h[i] = (rand() / (float)RAND_MAX - 0.5f) * 2.f;
        hipMemcpy(rnd, h, sizeof(h), hipMemcpyHostToDevice);
        const int it2 = 16384;
        float us = time_us([&] { k32r<2><<<1024, 256>>>(out, it2, rnd); });
        double flops = 1024.0 * 4 * it2 * 2 * 4096.0;
        printf("f32 32x32x2 random operands, full chip: %.1f us -> %.1f TFLOP/s (datasheet 157.3)\n", us, flops / us / 1e6);
        for (int wgs : {768, 1024, 1088}) {
            const int chunks = 1024;
            float u2 = time_us([&] { k32lds<<<wgs, 256, 2 * 16 * 192 * 4 * 2>>>(out, chunks, rnd); });
            printf("GEMM inner loop (LDS fragments + 16 MFMAs per chunk), %d workgroups x 4 waves, 48 KB LDS each: %.1f us -> %.1f TFLOP/s\n", wgs, u2,
                   (double)wgs * 4 * chunks * 16 * 4096.0 / u2 / 1e6);
        }
        for (int wgs : {528, 1280, 2112}) {
            const int chunks = 1024;
            float u2 = time_us([&] { k32lds1<<<wgs, 256, 2 * 16 * 128 * 4 * 2>>>(out, chunks, rnd); });
            printf("64 x 64 inner loop (8 + 8 LDS words, 8 dependent MFMAs per chunk), %d workgroups x 4 waves, 32 KB LDS each: %.1f us -> %.1f TFLOP/s\n", wgs, u2,
                   (double)wgs * 4 * chunks * 8 * 4096.0 / u2 / 1e6);
        }
        {
            const int it3 = 32768;
            float ub = time_us([&] { kbr<2><<<1024, 256>>>(out, it3, rnd); });
            double fb = 1024.0 * 4 * it3 * 2 * 32768.0;
            printf("bf16 32x32x16 random operands, full chip: %.1f us -> %.1f TFLOP/s (datasheet 2500)\n", ub, fb / ub / 1e6);
            ub = time_us([&] { kb<2><<<1024, 256>>>(out, it3, 1.5f); });
            printf("bf16 32x32x16 constant operands, full chip: %.1f us -> %.1f TFLOP/s\n", ub, fb / ub / 1e6);
        }
        us = time_us([&] { k32<2><<<1024, 256>>>(out, it2, 1.5f, 0.5f); });
        printf("f32 32x32x2 constant operands, full chip: %.1f us -> %.1f TFLOP/s\n", us, flops / us / 1e6);
    }
    return 0;
}
