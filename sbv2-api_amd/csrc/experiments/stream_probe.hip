// Experiment (not built into the library): what one wave / one CU can stream, as a function of loads in flight, with plain
// global_load_dwordx4 (to registers) and with global_load_lds_dwordx4 (LDS-DMA).  Every workgroup streams `bytes` of its own region
// (HBM) or of one shared region (L2).   hipcc --offload-arch=gfx950 -O3 stream_probe.hip -o /tmp/stream_probe && /tmp/stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

template <int U>
__global__ void k_plain(const char* base, size_t region_stride, size_t bytes, float* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const char* src = base + blockIdx.x * region_stride + (size_t)wave * 1024 + lane * 16;
    const size_t step = (size_t)nw * 1024;
    f32x4 acc = {0, 0, 0, 0};
    const size_t n = bytes / step;   // KB-steps per wave
    for (size_t i = 0; i + U <= n; i += U) {
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = *reinterpret_cast<const f32x4*>(src + (i + u) * step);
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = acc[0];
}

template <int D>   // D KB in flight per wave
__global__ void k_glds(const char* base, size_t region_stride, size_t bytes, float* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const char* src = base + blockIdx.x * region_stride + (size_t)wave * 1024 + lane * 16;
    const size_t step = (size_t)nw * 1024;
    char* ring = smem + wave * (2 * D * 1024);
    const size_t n = bytes / step;
    float acc = 0.f;
    for (int u = 0; u < D; ++u) __builtin_amdgcn_global_load_lds((gbl_void_t*)(src + u * step), (lds_void_t*)(ring + u * 1024), 16, 0, 0);
    for (size_t i = 0; i + 2 * D <= n; i += D) {
        char* nxt = ring + (((i / D) + 1) & 1) * D * 1024;
#pragma unroll
        for (int u = 0; u < D; ++u) __builtin_amdgcn_global_load_lds((gbl_void_t*)(src + (i + D + u) * step), (lds_void_t*)(nxt + u * 1024), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D) : "memory");
        const char* cur = ring + ((i / D) & 1) * D * 1024;
        acc += *reinterpret_cast<const float*>(cur + lane * 4);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc == 12345.678f) out[0] = acc;
}

template <class F>
static float time_ms(F f) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    f();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < 5; ++i) f();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return ms / 5;
}

int main() {
    const size_t region = 2u << 20;   // 2 MB per workgroup
    const int nwg = 256;
    char* buf;
    float* out;
    hipMalloc(&buf, region * nwg);
    hipMalloc(&out, 64);
    hipMemset(buf, 0, region * nwg);
    for (int shared = 0; shared < 2; ++shared) {
        const size_t stride = shared ? 0 : region;
        printf("---- %s, 256 workgroups x 2 MB ----\n", shared ? "one shared 2 MB region (L2 / MALL hits)" : "distinct regions (HBM)");
        for (int nw : {1, 2, 4, 8}) {
#define PLAIN(U) { float ms = time_ms([&] { k_plain<U><<<nwg, 64 * nw>>>(buf, stride, region, out); }); \
            printf("plain  waves/WG %d  %2d KB in flight per wave: %7.1f us  %6.1f GB/s per CU  %6.2f TB/s total\n", nw, U, ms * 1e3, region / ms / 1e6, region * nwg / ms / 1e9); }
            PLAIN(4) PLAIN(16) PLAIN(32)
#define GLDS(D) { float ms = time_ms([&] { k_glds<D><<<nwg, 64 * nw, nw * 2 * D * 1024>>>(buf, stride, region, out); }); \
            printf("glds   waves/WG %d  %2d KB in flight per wave: %7.1f us  %6.1f GB/s per CU  %6.2f TB/s total\n", nw, D, ms * 1e3, region / ms / 1e6, region * nwg / ms / 1e9); }
            GLDS(4) GLDS(8)
            if (nw <= 4) GLDS(16)
        }
    }
    // few workgroups: does a lone CU get more?
    printf("---- 32 workgroups only, distinct regions ----\n");
    for (int nw : {1, 4}) {
        { float ms = time_ms([&] { k_plain<16><<<32, 64 * nw>>>(buf, region, region, out); });
          printf("plain  waves/WG %d 16 KB: %7.1f us  %6.1f GB/s per CU\n", nw, ms * 1e3, region / ms / 1e6); }
        { float ms = time_ms([&] { k_glds<8><<<32, 64 * nw, nw * 16 * 1024>>>(buf, region, region, out); });
          printf("glds   waves/WG %d  8 KB: %7.1f us  %6.1f GB/s per CU\n", nw, ms * 1e3, region / ms / 1e6); }
    }
    return 0;
}
