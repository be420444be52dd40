// Experiment (not built into the library): do the f32 MFMA shapes of gfx950 produce the SAME bits for a K-long dot product?
//   (a) v_mfma_f32_32x32x2_f32 chained over k,  (b) v_mfma_f32_16x16x4_f32,  (c) v_mfma_f32_4x4x1_16B_f32,  (d) scalar fmaf chain in k order
// If they do, the tile shape of the f32 GEMM can follow the problem size (small grids) without changing any output bit.
// build + run on a GPU box:  hipcc --offload-arch=gfx950 -O2 mfma_order.hip -o /tmp/mfma_order && /tmp/mfma_order
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// A: [32][K] row-major, B: [K][32]; C: [32][32]
__global__ void k32(const float* A, const float* B, float* C, int K) {
    const int lane = threadIdx.x;
    f32x16 acc = {0};
    for (int k = 0; k < K; k += 2) {
        const float a = A[(lane & 31) * K + k + (lane >> 5)];
        const float b = B[(k + (lane >> 5)) * 32 + (lane & 31)];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        C[row * 32 + (lane & 31)] = acc[r];
    }
}
__global__ void k16(const float* A, const float* B, float* C, int K) {
    const int lane = threadIdx.x;
    for (int ti = 0; ti < 2; ++ti)
        for (int tj = 0; tj < 2; ++tj) {
            f32x4 acc = {0};
            for (int k = 0; k < K; k += 4) {
                const float a = A[(ti * 16 + (lane & 15)) * K + k + (lane >> 4)];
                const float b = B[(k + (lane >> 4)) * 32 + tj * 16 + (lane & 15)];
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
            }
            for (int r = 0; r < 4; ++r) C[(ti * 16 + (lane >> 4) * 4 + r) * 32 + tj * 16 + (lane & 15)] = acc[r];
        }
}
// 4x4x1: 16 blocks; block b = lane / 4; a = A_b[i = lane % 4], b = B_b[j = lane % 4]; D_b[i = r][j = lane % 4]
__global__ void k4(const float* A, const float* B, float* C, int K) {
    const int lane = threadIdx.x;
    // 64 sub-tiles of 4x4 in a 32x32 output: 4 rounds of 16 blocks; block -> (bi, bj)
    for (int round = 0; round < 4; ++round) {
        const int blk = round * 16 + (lane >> 2);
        const int bi = blk >> 3, bj = blk & 7;
        f32x4 acc = {0};
        for (int k = 0; k < K; ++k) {
            const float a = A[(bi * 4 + (lane & 3)) * K + k];
            const float b = B[k * 32 + bj * 4 + (lane & 3)];
            acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc, 0, 0, 0);
        }
        for (int r = 0; r < 4; ++r) C[(bi * 4 + r) * 32 + bj * 4 + (lane & 3)] = acc[r];
    }
}
__global__ void kfma(const float* A, const float* B, float* C, int K) {
    const int t = threadIdx.x + blockIdx.x * blockDim.x;
    const int i = t >> 5, j = t & 31;
    float s = 0.f;
    for (int k = 0; k < K; ++k) s = __builtin_fmaf(A[i * K + k], B[k * 32 + j], s);
    C[i * 32 + j] = s;
}
// pairwise inside an instruction: (a0*b0 + a1*b1) computed first, then added
__global__ void kpair(const float* A, const float* B, float* C, int K) {
    const int t = threadIdx.x + blockIdx.x * blockDim.x;
    const int i = t >> 5, j = t & 31;
    float s = 0.f;
    for (int k = 0; k < K; k += 2) {
        float p = __builtin_fmaf(A[i * K + k + 1], B[(k + 1) * 32 + j], A[i * K + k] * B[k * 32 + j]);
        s += p;
    }
    C[i * 32 + j] = s;
}

static int diff(const std::vector<float>& x, const std::vector<float>& y) {
    int n = 0;
    for (size_t i = 0; i < x.size(); ++i) n += memcmp(&x[i], &y[i], 4) != 0;
    return n;
}

int main() {
    for (int K : {64, 1024, 4096}) {
        std::vector<float> A(32 * K), B(K * 32);
        srand(K);
        for (auto& v : A) v = (rand() / (float)RAND_MAX - 0.5f) * 2.f;
        for (auto& v : B) v = (rand() / (float)RAND_MAX - 0.5f) * 2.f;
        if (K == 64) {  // denormal probe
            A[3] = 1e-30f; B[3 * 32] = 1e-12f;
        }
        float *dA, *dB, *dC;
        hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 1024 * 4);
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        std::vector<float> c32(1024), c16(1024), c4(1024), cf(1024), cp(1024);
        k32<<<1, 64>>>(dA, dB, dC, K); hipMemcpy(c32.data(), dC, 4096, hipMemcpyDeviceToHost);
        k16<<<1, 64>>>(dA, dB, dC, K); hipMemcpy(c16.data(), dC, 4096, hipMemcpyDeviceToHost);
        k4<<<1, 64>>>(dA, dB, dC, K); hipMemcpy(c4.data(), dC, 4096, hipMemcpyDeviceToHost);
        kfma<<<4, 256>>>(dA, dB, dC, K); hipMemcpy(cf.data(), dC, 4096, hipMemcpyDeviceToHost);
        kpair<<<4, 256>>>(dA, dB, dC, K); hipMemcpy(cp.data(), dC, 4096, hipMemcpyDeviceToHost);
        printf("K=%d  differing of 1024:  32x32x2 vs 16x16x4: %d   vs 4x4x1: %d   vs fma chain: %d   vs pairwise: %d   16x16x4 vs fma: %d\n", K,
               diff(c32, c16), diff(c32, c4), diff(c32, cf), diff(c32, cp), diff(c16, cf));
        hipFree(dA); hipFree(dB); hipFree(dC);
    }
    return 0;
}
