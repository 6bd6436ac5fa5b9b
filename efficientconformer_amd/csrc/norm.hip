// LayerNorm (eps 1e-6, reference modules.py:377, 447, 500; blocks.py:97, 135) and row casts.
// HBM-bound: one 64-lane wave per row, float4 loads, two-pass fp32 statistics held in registers,
// wavefront shuffles for the row reductions; emits the bf16 A-operand of the following GEMM directly
// (and, for the block-final norm, also the fp32 residual stream and the *next* block's pre-normed A).
#include "kernels.h"
#include <algorithm>

namespace {

constexpr float LN_EPS = 1e-6f;

// LPR lanes cooperate on one row (64/LPR rows per wave): for D = 120..256 a 16-lane group holds the row in <= 4 float4
// per lane, so every lane of the wave carries loads and a wave keeps 4 rows in flight.
template <int LPR> __device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

template <int LPR, int NV>
__device__ __forceinline__ void ln_apply(float4 (&x)[NV], int sub, int D, const float* g, const float* b) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if ((sub + LPR * i) * 4 < D) s += (x[i].x + x[i].y) + (x[i].z + x[i].w);
    const float mean = group_sum<LPR>(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if ((sub + LPR * i) * 4 < D) {
            const float a = x[i].x - mean, bb = x[i].y - mean, c = x[i].z - mean, d = x[i].w - mean;
            q += (a * a + bb * bb) + (c * c + d * d);
        }
    const float rstd = rsqrtf(group_sum<LPR>(q) / (float)D + LN_EPS);
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if ((sub + LPR * i) * 4 < D) {
            const int c = (sub + LPR * i) * 4;
            const float4 gg = *reinterpret_cast<const float4*>(g + c);
            const float4 bb = *reinterpret_cast<const float4*>(b + c);
            x[i].x = (x[i].x - mean) * rstd * gg.x + bb.x;
            x[i].y = (x[i].y - mean) * rstd * gg.y + bb.y;
            x[i].z = (x[i].z - mean) * rstd * gg.z + bb.z;
            x[i].w = (x[i].w - mean) * rstd * gg.w + bb.w;
        }
}

template <int LPR, int NV>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, int M, int D,
                                                        const float* g1, const float* b1,
                                                        float* out_f32, bf16_t* out_bf16, int ld_bf16,
                                                        const float* g2, const float* b2) {
    constexpr int RPW = 64 / LPR;                     // rows per wave
    const int lane = threadIdx.x & 63, sub = lane % LPR;
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + lane / LPR;
    for (int row = row0; row < M; row += gridDim.x * 4 * RPW) {
        float4 v[NV];
        const float* xr = x + (size_t)row * D;
#pragma unroll
        for (int i = 0; i < NV; ++i) {     // unconditional clamped loads (lanes past D re-read the last float4; never used)
            const int c = (sub + LPR * i) * 4;
            v[i] = *reinterpret_cast<const float4*>(xr + (c < D - 4 ? c : D - 4));
        }
        ln_apply<LPR, NV>(v, sub, D, g1, b1);
        if (out_f32) {
#pragma unroll
            for (int i = 0; i < NV; ++i)
                if ((sub + LPR * i) * 4 < D)
                    *reinterpret_cast<float4*>(out_f32 + (size_t)row * D + (sub + LPR * i) * 4) = v[i];
        }
        if (out_bf16) {
            if (g2) ln_apply<LPR, NV>(v, sub, D, g2, b2);
            bf16_t* o = out_bf16 + (size_t)row * ld_bf16;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = (sub + LPR * i) * 4;
                if (c < ld_bf16) {
                    uint2 w = make_uint2(0u, 0u);               // pad columns [D, ld) are written as zeros
                    if (c < D) w = make_uint2(pack_bf2(v[i].x, v[i].y), pack_bf2(v[i].z, v[i].w));
                    *reinterpret_cast<uint2*>(o + c) = w;
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void cast_rows_kernel(const float* __restrict__ x, int D, int rows_per_batch, int stride,
                                                        int out_rows_per_batch, int total_out_rows, bf16_t* out, int ld_out, RaggedConv rc) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= total_out_rows) return;
    size_t src;
    bool valid = true;
    if (rc.out_off) {        // ragged batch: output row -> (utterance, frame); rows behind the utterance's last frame (group padding) are zeros
        const int b = ragged_find(rc.out_off, rc.n, row), r = row - rc.out_off[b];
        valid = r < rc.out_len[b];
        src = (size_t)rc.in_off[b] + (size_t)(valid ? r : 0) * stride;
    } else {
        const int b = row / out_rows_per_batch, r = row - b * out_rows_per_batch;
        src = (size_t)b * rows_per_batch + (size_t)r * stride;
    }
    const float* xr = x + src * D;
    bf16_t* o = out + (size_t)row * ld_out;
    for (int c = lane * 4; c < ld_out; c += 256) {
        uint2 w = make_uint2(0u, 0u);
        if (c < D && valid) {
            const float4 v = *reinterpret_cast<const float4*>(xr + c);
            w = make_uint2(pack_bf2(v.x, v.y), pack_bf2(v.z, v.w));
        }
        *reinterpret_cast<uint2*>(o + c) = w;
    }
}

// ragged rows [off[b], off[b] + len[b]) of x -> out[b][0 .. len[b]), zeros up to t_out: one wave per output row
__global__ __launch_bounds__(256) void emit_rows_kernel(const float* __restrict__ x, int D, const int* __restrict__ off, const int* __restrict__ len,
                                                        int batch, int t_out, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= batch * t_out) return;
    const int b = row / t_out, t = row - b * t_out;
    const bool valid = t < len[b];
    const float* xr = x + ((size_t)off[b] + (valid ? t : 0)) * D;
    float* o = out + (size_t)row * D;
    for (int c = lane * 4; c < D; c += 256) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (valid) v = *reinterpret_cast<const float4*>(xr + c);
        *reinterpret_cast<float4*>(o + c) = v;
    }
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ x, int D, int t_pitch, const int* __restrict__ off,
                                                          const int* __restrict__ len, int n, int rows, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int b = ragged_find(off, n, row), t = row - off[b];
    const bool valid = t < len[b];
    const float* xr = x + ((size_t)b * t_pitch + (valid ? t : 0)) * D;
    float* o = out + (size_t)row * D;
    for (int c = lane * 4; c < D; c += 256) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (valid) v = *reinterpret_cast<const float4*>(xr + c);
        *reinterpret_cast<float4*>(o + c) = v;
    }
}

// y = LayerNorm(x + alpha * r): the residual + norm glue of ConformerBlock.forward as ONE unit-testable kernel (per-kernel C ABI
// entry effconf_layernorm_residual; the forward itself keeps this inside the chain / GEMM epilogues).  One wave per row.
__global__ __launch_bounds__(256) void layernorm_residual_kernel(const float* __restrict__ x, const float* __restrict__ r, float alpha, int M, int D,
                                                                 const float* __restrict__ g, const float* __restrict__ b, float* __restrict__ y) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* xr = x + (size_t)row * D;
    const float* rr = r ? r + (size_t)row * D : nullptr;
    float s = 0.f;
    for (int c = lane * 4; c < D; c += 256) {
        float4 v = *reinterpret_cast<const float4*>(xr + c);
        if (rr) { const float4 w = *reinterpret_cast<const float4*>(rr + c); v.x += alpha * w.x; v.y += alpha * w.y; v.z += alpha * w.z; v.w += alpha * w.w; }
        s += (v.x + v.y) + (v.z + v.w);
    }
    const float mean = group_sum<64>(s) / (float)D;
    float q = 0.f;
    for (int c = lane * 4; c < D; c += 256) {
        float4 v = *reinterpret_cast<const float4*>(xr + c);
        if (rr) { const float4 w = *reinterpret_cast<const float4*>(rr + c); v.x += alpha * w.x; v.y += alpha * w.y; v.z += alpha * w.z; v.w += alpha * w.w; }
        const float a0 = v.x - mean, a1 = v.y - mean, a2 = v.z - mean, a3 = v.w - mean;
        q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
    }
    const float rstd = rsqrtf(group_sum<64>(q) / (float)D + LN_EPS);
    for (int c = lane * 4; c < D; c += 256) {
        float4 v = *reinterpret_cast<const float4*>(xr + c);
        if (rr) { const float4 w = *reinterpret_cast<const float4*>(rr + c); v.x += alpha * w.x; v.y += alpha * w.y; v.z += alpha * w.z; v.w += alpha * w.w; }
        const float4 gg = *reinterpret_cast<const float4*>(g + c), bb = *reinterpret_cast<const float4*>(b + c);
        *reinterpret_cast<float4*>(y + (size_t)row * D + c) = make_float4((v.x - mean) * rstd * gg.x + bb.x, (v.y - mean) * rstd * gg.y + bb.y,
                                                                          (v.z - mean) * rstd * gg.z + bb.z, (v.w - mean) * rstd * gg.w + bb.w);
    }
}

}  // namespace

int launch_layernorm_residual(const float* x, const float* r, float alpha, int M, int D, const float* gamma, const float* beta, float* y, hipStream_t s) {
    if (M <= 0) return 0;
    if (D % 4) return -2;
    hipLaunchKernelGGL(layernorm_residual_kernel, dim3((M + 3) / 4), dim3(256), 0, s, x, r, alpha, M, D, gamma, beta, y);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_layernorm(const float* x, int M, int D, const float* gamma, const float* beta,
                     float* out_f32, bf16_t* out_bf16, int ld_bf16,
                     const float* gamma2, const float* beta2, hipStream_t s) {
    if (M <= 0) return 0;
    if (D % 4 || D > 2048 || (out_bf16 && (ld_bf16 % 4 || ld_bf16 < D || ld_bf16 > D + 8))) return -2;
    // NV float4 per lane must also cover the (<= 8) bf16 pad columns: (LPR * NV) * 4 >= ld
    auto go = [&](auto kern, int rpw) {
        const int blocks = std::min((M + 4 * rpw - 1) / (4 * rpw), 256 * 8);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, s, x, M, D, gamma, beta, out_f32, out_bf16, ld_bf16, gamma2, beta2);
    };
    if (D <= 120) go(layernorm_kernel<16, 2>, 4);
    else if (D <= 248) go(layernorm_kernel<16, 4>, 4);
    else if (D <= 504) go(layernorm_kernel<32, 4>, 2);
    else if (D <= 1016) go(layernorm_kernel<64, 4>, 1);
    else go(layernorm_kernel<64, 8>, 1);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_cast_rows(const float* x, int D, int rows_per_batch, int stride, int out_rows_per_batch, int batch,
                     bf16_t* out, int ld_out, hipStream_t s, const RaggedConv* rcp) {
    RaggedConv rc{};
    if (rcp) rc = *rcp;
    const int total = rcp ? rc.out_rows : out_rows_per_batch * batch;
    if (total <= 0) return 0;
    if (D % 4 || ld_out % 4) return -2;
    hipLaunchKernelGGL(cast_rows_kernel, dim3((total + 3) / 4), dim3(256), 0, s, x, D, rows_per_batch, stride,
                       out_rows_per_batch, total, out, ld_out, rc);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_gather_rows(const float* x, int D, int t_pitch, const RaggedRows& rg, float* out, hipStream_t s) {
    if (rg.rows <= 0) return 0;
    if (D % 4) return -2;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((rg.rows + 3) / 4), dim3(256), 0, s, x, D, t_pitch, rg.off, rg.len, rg.n, rg.rows, out);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_emit_rows(const float* x, int D, const int* off, const int* len, int batch, int t_out, float* out, hipStream_t s) {
    if (batch <= 0 || t_out <= 0) return 0;
    if (D % 4) return -2;
    hipLaunchKernelGGL(emit_rows_kernel, dim3((batch * t_out + 3) / 4), dim3(256), 0, s, x, D, off, len, batch, t_out, out);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
