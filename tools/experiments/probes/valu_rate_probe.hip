// VALU issue rates on gfx950, measured: full-rate fp32 (v_fma_f32) against the transcendental unit (v_exp_f32, v_rcp_f32) and the Swish sequence
// of the library (common.h sigmoidf_: v_mul, v_exp, v_add, v_rcp, v_mul).  Standalone:  hipcc --offload-arch=gfx950 -O3 -o valu_rate_probe valu_rate_probe.hip
// Output: wave-instructions per nanosecond over the chip, and cycles per wave-instruction and SIMD at the measured clock estimate (fma = 4 cycles).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(256) void probe(float* out, int iters, float seed) {
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = seed + 0.001f * (threadIdx.x + i);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) v[i] = __builtin_fmaf(v[i], 0.999f, 0.001f);
            if (MODE == 1) v[i] = __builtin_amdgcn_exp2f(v[i]) * 0.25f;                  // exp + mul
            if (MODE == 2) v[i] = __builtin_amdgcn_rcpf(v[i]) + 0.5f;                    // rcp + add
            if (MODE == 3) v[i] = v[i] * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504f * v[i])) + 0.7f;   // swish + add
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
double run(float* d, int wgs, int iters) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(probe<MODE>, dim3(wgs), dim3(256), 0, 0, d, iters, 0.9f);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(probe<MODE>, dim3(wgs), dim3(256), 0, 0, d, iters, 0.9f);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0.f;
    hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main() {
    const int wgs = 256 * 8, iters = 20000;                     // 8 workgroups of 4 waves per CU: 8 waves per SIMD
    float* d;
    hipMalloc(&d, (size_t)wgs * 256 * 4);
    const double waves = (double)wgs * 4, simds = 1024.0;
    const int per_it[4] = {8, 16, 16, 48};                       // VALU instructions per loop iteration and lane (mode 3: 5 swish + 1 add, x 8)
    const char* names[4] = {"v_fma_f32", "v_exp_f32 + v_mul_f32", "v_rcp_f32 + v_add_f32", "swish (mul exp add rcp mul) + add"};
    double ms[4] = {run<0>(d, wgs, iters), run<1>(d, wgs, iters), run<2>(d, wgs, iters), run<3>(d, wgs, iters)};
    const double fma_ns_per_instr = ms[0] * 1e6 / (waves * iters * per_it[0] / simds);          // ns per wave-instruction and SIMD = 4 cycles
    const double ghz = 4.0 / fma_ns_per_instr;
    printf("clock estimate from v_fma_f32 at 4 cycles per wave-instruction: %.3f GHz\n", ghz);
    for (int m = 0; m < 4; ++m) {
        const double ns = ms[m] * 1e6 / (waves * iters / simds);                                 // ns per loop iteration (8 chains) and SIMD-resident wave slot
        printf("%-36s %8.3f ms   %6.1f cycles per iteration of 8 chains (%d instructions) = %.2f cycles per instruction\n", names[m], ms[m], ns * ghz, per_it[m],
               ns * ghz / per_it[m]);
    }
    printf("per 8 chains: exp alone = %.1f cycles, rcp alone = %.1f cycles, swish = %.1f cycles per element-vector of 64 lanes\n",
           (ms[1] * 1e6 / (waves * iters / simds) * ghz - 32) / 8, (ms[2] * 1e6 / (waves * iters / simds) * ghz - 32) / 8, (ms[3] * 1e6 / (waves * iters / simds) * ghz - 32) / 8);
    return 0;
}
