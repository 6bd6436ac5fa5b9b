// Operand layout of v_mfma_f32_4x4x4_16b_bf16 (16 independent 4x4x4 blocks per wave), checked against a host loop:
//   block b = lanes 4b .. 4b + 3;  A: lane 4b + i holds A_b[i][0..3];  B: lane 4b + j holds B_b[0..3][j];  D: lane 4b + j holds D_b[0..3][j]
// hipcc --offload-arch=gfx950 -O3 -o mfma444_layout_probe mfma444_layout_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const s16x4* a, const s16x4* b, f32x4* d) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0);
    d[threadIdx.x] = acc;
}
static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)(u >> 16); }
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
int main() {
    uint16_t ha[64][4], hb[64][4];
    float A[16][4][4], B[16][4][4];
    for (int b = 0; b < 16; ++b)
        for (int i = 0; i < 4; ++i)
            for (int kk = 0; kk < 4; ++kk) {
                A[b][i][kk] = bf2f(f2bf(0.5f + 0.25f * b + 1.0f * i + 0.125f * kk));
                B[b][kk][i] = bf2f(f2bf(1.0f - 0.0625f * b + 2.0f * i - 0.5f * kk));        // B[b][k][j = i]
            }
    for (int b = 0; b < 16; ++b)
        for (int x = 0; x < 4; ++x)
            for (int kk = 0; kk < 4; ++kk) { ha[4 * b + x][kk] = f2bf(A[b][x][kk]); hb[4 * b + x][kk] = f2bf(B[b][kk][x]); }
    void *da, *db, *dd;
    hipMalloc(&da, sizeof(ha)); hipMalloc(&db, sizeof(hb)); hipMalloc(&dd, 64 * 16);
    hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, (const s16x4*)da, (const s16x4*)db, (f32x4*)dd);
    float hd[64][4];
    hipMemcpy(hd, dd, sizeof(hd), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int b = 0; b < 16; ++b)
        for (int j = 0; j < 4; ++j)
            for (int i = 0; i < 4; ++i) {
                float ref = 0.f;
                for (int kk = 0; kk < 4; ++kk) ref += A[b][i][kk] * B[b][kk][j];
                if (hd[4 * b + j][i] != ref) { if (bad < 8) printf("mismatch block %d D[%d][%d]: got %g want %g\n", b, i, j, hd[4 * b + j][i], ref); ++bad; }
            }
    printf(bad ? "LAYOUT MISMATCH (%d)\n" : "layout confirmed: A lane 4b+i = row i, B lane 4b+j = column j, D lane 4b+j regs = rows i (%d mismatches)\n", bad);
    return 0;
}
