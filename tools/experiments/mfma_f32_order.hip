// Is v_mfma_f32_16x16x4_f32 / v_mfma_f32_32x32x2_f32 bit-identical to a sequential fmaf chain over ascending k?
//   hipcc --offload-arch=gfx950 -O2 tools/experiments/mfma_f32_order.hip -o /tmp/mfma_order && /tmp/mfma_order
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// D[16][16] = sum_k A[i][k] B[j][k], K = 64: A, B row-major [16][64]
__global__ void k16(const float* A, const float* B, float* D) {
    const int l = threadIdx.x, i = l & 15, g = l >> 4;
    f32x4 a2 = {0.f, 0.f, 0.f, 0.f};           // A operand: row i = l % 16, k-slot g = l / 16; B operand: column j = l % 16, k-slot g
    for (int k = 0; k < 64; k += 4) a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[i * 64 + k + g], B[i * 64 + k + g], a2, 0, 0, 0);
    // lane l holds D[4 g + e][j = l & 15]
    for (int e = 0; e < 4; ++e) D[(4 * g + e) * 16 + (l & 15)] = a2[e];
}
__global__ void k32(const float* A, const float* B, float* D) {     // A, B [32][64]
    const int l = threadIdx.x, i = l & 31, g = l >> 5;
    f32x16 acc;
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    for (int k = 0; k < 64; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[i * 64 + k + g], B[i * 64 + k + g], acc, 0, 0, 0);
    for (int e = 0; e < 16; ++e) D[((e & 3) + 8 * (e >> 2) + 4 * g) * 32 + i] = acc[e];
}
int main() {
    const int K = 64;
    std::vector<float> A(32 * K), B(32 * K), D16(256), D32(1024);
    srand(1);
    for (auto& v : A) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    for (auto& v : B) v = ((float)rand() / RAND_MAX * 2.f - 1.f) * 3.7f;
    float *dA, *dB, *dD;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, 1024 * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k16, dim3(1), dim3(64), 0, 0, dA, dB, dD); hipMemcpy(D16.data(), dD, 256 * 4, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(k32, dim3(1), dim3(64), 0, 0, dA, dB, dD); hipMemcpy(D32.data(), dD, 1024 * 4, hipMemcpyDeviceToHost);
    int bad16 = 0, bad32 = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        float s = 0.f;
        for (int k = 0; k < K; ++k) s = __builtin_fmaf(A[i * K + k], B[j * K + k], s);
        bad16 += memcmp(&s, &D16[i * 16 + j], 4) != 0;
    }
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
        float s = 0.f;
        for (int k = 0; k < K; ++k) s = __builtin_fmaf(A[i * K + k], B[j * K + k], s);
        bad32 += memcmp(&s, &D32[i * 32 + j], 4) != 0;
    }
    printf("v_mfma_f32_16x16x4_f32 vs ascending fmaf chain: %d of 256 elements differ\n", bad16);
    printf("v_mfma_f32_32x32x2_f32 vs ascending fmaf chain: %d of 1024 elements differ\n", bad32);
    return 0;
}
