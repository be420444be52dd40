// Probe: does global_load_lds_dwordx4 (LDS-DMA, 16 bytes per lane) accept source addresses that are only 8- or 4-byte aligned?
// Each lane requests 16 bytes at src + shift + 16 lane; the LDS image is copied out and compared with the bytes a plain load sees.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef __attribute__((address_space(3))) void lds_t;
typedef const __attribute__((address_space(1))) void gbl_t;
__global__ void k(const char* src, int shift, unsigned* out) {
    __shared__ __attribute__((aligned(16))) char buf[1024];
    const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)buf);
    __builtin_amdgcn_global_load_lds((gbl_t*)(src + shift + threadIdx.x * 16), (lds_t*)(uintptr_t)lds0, 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) out[i] = reinterpret_cast<unsigned*>(buf)[i];
}
int main() {
    std::vector<unsigned char> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = (unsigned char)(i * 7 + (i >> 8));
    char* d; unsigned* o;
    hipMalloc(&d, 4096); hipMalloc(&o, 1024);
    hipMemcpy(d, h.data(), 4096, hipMemcpyHostToDevice);
    for (int shift : {0, 4, 8, 12, 2}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, shift, o);
        std::vector<unsigned char> r(1024);
        hipMemcpy(r.data(), o, 1024, hipMemcpyDeviceToHost);
        int bad = 0, first = -1;
        for (int i = 0; i < 1024; ++i) if (r[i] != h[shift + i]) { if (first < 0) first = i; ++bad; }
        printf("shift %2d: %d mismatching bytes of 1024 (first at %d)\n", shift, bad, first);
    }
    return 0;
}
