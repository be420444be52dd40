// Experiment (not built into the library): time per MFMA for dependent chains and for 2 / 4 independent accumulators, one wave per CU or
// one wave per SIMD, f32 16x16x4 / 32x32x2 and bf16 32x32x16.  hipcc --offload-arch=gfx950 -O3 mfma_lat.hip -o /tmp/mfma_lat && /tmp/mfma_lat
#include <hip/hip_runtime.h>
#include <cstdio>
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
    return 0;
}
