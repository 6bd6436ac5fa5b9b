// LayerNorm (eps 1e-6, reference modules.py:377, 447, 500; blocks.py:97, 135) and row casts.
// HBM-bound: one 64-lane wave per row, float4 loads, two-pass fp32 statistics held in registers,
// wavefront shuffles for the row reductions; emits the bf16 A-operand of the following GEMM directly
// (and, for the block-final norm, also the fp32 residual stream and the *next* block's pre-normed A).
#include "kernels.h"
#include <algorithm>

namespace {

constexpr float LN_EPS = 1e-6f;
constexpr int MAXV = 8;   // float4 per lane: D <= 64 * 4 * 8 = 2048

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__device__ __forceinline__ void ln_apply(float4 (&x)[MAXV], int nv, int lane, int D, const float* g, const float* b) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
        if (i < nv && (lane + 64 * i) * 4 < D) s += (x[i].x + x[i].y) + (x[i].z + x[i].w);
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
        if (i < nv && (lane + 64 * i) * 4 < D) {
            float a = x[i].x - mean, bb = x[i].y - mean, c = x[i].z - mean, d = x[i].w - mean;
            q += (a * a + bb * bb) + (c * c + d * d);
        }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + LN_EPS);
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
        if (i < nv && (lane + 64 * i) * 4 < D) {
            const int c = (lane + 64 * i) * 4;
            const float4 gg = *reinterpret_cast<const float4*>(g + c);
            const float4 bb = *reinterpret_cast<const float4*>(b + c);
            x[i].x = (x[i].x - mean) * rstd * gg.x + bb.x;
            x[i].y = (x[i].y - mean) * rstd * gg.y + bb.y;
            x[i].z = (x[i].z - mean) * rstd * gg.z + bb.z;
            x[i].w = (x[i].w - mean) * rstd * gg.w + bb.w;
        }
}

__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, int M, int D,
                                                        const float* g1, const float* b1,
                                                        float* out_f32, bf16_t* out_bf16, int ld_bf16,
                                                        const float* g2, const float* b2) {
    const int lane = threadIdx.x & 63;
    const int nv = (D / 4 + 63) / 64;
    // grid-stride over rows: a wave handles many rows so that the launch is not dominated by wave start-up
    for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < M; row += gridDim.x * 4) {
    float4 v[MAXV];
    const float* xr = x + (size_t)row * D;
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
        if (i < nv && (lane + 64 * i) * 4 < D) v[i] = *reinterpret_cast<const float4*>(xr + (lane + 64 * i) * 4);
    ln_apply(v, nv, lane, D, g1, b1);
    if (out_f32) {
#pragma unroll
        for (int i = 0; i < MAXV; ++i)
            if (i < nv && (lane + 64 * i) * 4 < D)
                *reinterpret_cast<float4*>(out_f32 + (size_t)row * D + (lane + 64 * i) * 4) = v[i];
    }
    if (out_bf16) {
        if (g2) ln_apply(v, nv, lane, D, g2, b2);
        bf16_t* o = out_bf16 + (size_t)row * ld_bf16;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = (lane + 64 * i) * 4;
            if (i < nv && c < ld_bf16) {
                uint2 w = make_uint2(0u, 0u);                   // pad columns [D, ld) are written as zeros
                if (c < D) w = make_uint2(pack_bf2(v[i].x, v[i].y), pack_bf2(v[i].z, v[i].w));
                *reinterpret_cast<uint2*>(o + c) = w;
            }
        }
    }
    }
}

__global__ __launch_bounds__(256) void cast_rows_kernel(const float* __restrict__ x, int D, int rows_per_batch, int stride,
                                                        int out_rows_per_batch, int total_out_rows, bf16_t* out, int ld_out) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= total_out_rows) return;
    const int b = row / out_rows_per_batch, r = row - b * out_rows_per_batch;
    const float* xr = x + ((size_t)b * rows_per_batch + (size_t)r * stride) * D;
    bf16_t* o = out + (size_t)row * ld_out;
    for (int c = lane * 4; c < ld_out; c += 256) {
        uint2 w = make_uint2(0u, 0u);
        if (c < D) {
            const float4 v = *reinterpret_cast<const float4*>(xr + c);
            w = make_uint2(pack_bf2(v.x, v.y), pack_bf2(v.z, v.w));
        }
        *reinterpret_cast<uint2*>(o + c) = w;
    }
}

}  // namespace

int launch_layernorm(const float* x, int M, int D, const float* gamma, const float* beta,
                     float* out_f32, bf16_t* out_bf16, int ld_bf16,
                     const float* gamma2, const float* beta2, hipStream_t s) {
    if (M <= 0) return 0;
    if (D % 4 || D > 64 * 4 * MAXV || (out_bf16 && (ld_bf16 % 4 || ld_bf16 < D))) return -2;
    const int blocks = std::min((M + 3) / 4, 256 * 8);
    hipLaunchKernelGGL(layernorm_kernel, dim3(blocks), dim3(256), 0, s, x, M, D, gamma, beta,
                       out_f32, out_bf16, ld_bf16, gamma2, beta2);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_cast_rows(const float* x, int D, int rows_per_batch, int stride, int out_rows_per_batch, int batch,
                     bf16_t* out, int ld_out, hipStream_t s) {
    const int total = out_rows_per_batch * batch;
    if (total <= 0) return 0;
    if (D % 4 || ld_out % 4) return -2;
    hipLaunchKernelGGL(cast_rows_kernel, dim3((total + 3) / 4), dim3(256), 0, s, x, D, rows_per_batch, stride,
                       out_rows_per_batch, total, out, ld_out);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
