// Shared device helpers of the split-precision kernels (split.hip, sxf.hip): fp32 values as fp16 operand pairs x = h + l / 2048 for the fp16 matrix pipe
// (three MFMAs per product, ~2^-21 relative), accurate exp / reciprocal on the transcendental unit, span loads of odd head widths.
#pragma once
#include "common.h"

namespace sx {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr float LO_SCALE = 2048.0f, LO_INV = 1.0f / 2048.0f;

__device__ __forceinline__ uint32_t pack_h2(_Float16 a, _Float16 b) {
    union { _Float16 h[2]; uint32_t u; } c;
    c.h[0] = a; c.h[1] = b;
    return c.u;
}

// two fp32 values -> their (h, l) fp16 pairs.  h by v_cvt_pkrtz_f16_f32 (one instruction for the pair; truncation is as good as rounding
// here - any h within 2^-10 of x leaves a remainder the second half represents - and it saturates instead of overflowing), l rounded to
// nearest.  Values beyond the fp16 range saturate (operands of this path are LayerNorm-ed / gated activations and weights, far inside it).
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split2(float x0, float x1, uint32_t& hi, uint32_t& lo) {
    typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));
    union { fp16x2 p; f16x2 h; uint32_t u; } ch;
    union { f16x2 h; uint32_t u; } cl;
    ch.p = __builtin_amdgcn_cvt_pkrtz(x0, x1);
    const float r0 = __builtin_amdgcn_fmed3f((x0 - (float)ch.h[0]) * LO_SCALE, -65000.f, 65000.f);
    const float r1 = __builtin_amdgcn_fmed3f((x1 - (float)ch.h[1]) * LO_SCALE, -65000.f, 65000.f);
    cl.h = __builtin_convertvector(f32x2v{r0, r1}, f16x2);
    hi = ch.u;
    lo = cl.u;
}

// Same-scale variant (round 6, sxf_ffn.hip): x S = h + l with BOTH halves at the scale S (a power of two that puts the operand's usual magnitude around
// 2^5 .. 2^12, so l - 2^-11 of h - is a normal fp16 number for every element that matters: an element so small that its l underflows contributes < 2^-25 S^-1
// absolutely).  The three products h h' + h l' + l h' then share ONE fp32 accumulator at the scale S S' (undone by an exact multiply at the end): no correction
// accumulator, no fold - the sum rounds like the fp32 accumulation of the reference's own GEMM.  Inputs are the already scaled values.
__device__ __forceinline__ void split2s(float x0, float x1, uint32_t& hi, uint32_t& lo) {
    typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));
    union { fp16x2 p; f16x2 h; uint32_t u; } ch;
    union { f16x2 h; uint32_t u; } cl;
    ch.p = __builtin_amdgcn_cvt_pkrtz(x0, x1);
    cl.h = __builtin_convertvector(f32x2v{x0 - (float)ch.h[0], x1 - (float)ch.h[1]}, f16x2);
    hi = ch.u;
    lo = cl.u;
}

// exp(x) for x <= ~0 .. 88 to ~1 ulp on v_exp_f32: the product x log2(e) in two parts (fma residual + the constant's low part), first-order
// correction of the result; 1 / x by v_rcp_f32 + one Newton step.  (libm's expf and the IEEE division sequence are ~60 instructions per element,
// which made the epilogues and the softmax of this file VALU-bound.)
__device__ __forceinline__ float sx_expf(float x) {
    const float L2E = 1.44269502162933349609375f, L2E_LO = 1.925963033500011e-08f;
    const float t = x * L2E;
    const float c = fmaf(x, L2E, -t) + x * L2E_LO;
    const float e = __builtin_amdgcn_exp2f(t);
    return fmaf(e, c * 0.693147180559945f, e);
}
__device__ __forceinline__ float sx_rcp(float d) {
    const float r = __builtin_amdgcn_rcpf(d);
    return fmaf(r, fmaf(-d, r, 1.0f), r);
}

__device__ __forceinline__ f16x8 as_f16x8(uint4 v) {
    union { uint4 u; f16x8 h; } c;
    c.u = v;
    return c.h;
}

__device__ __forceinline__ float4 ld4u(const float* p) {          // 16-byte global load from a 4-byte aligned address (head spans of odd width)
    typedef float f32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
    const f32x4_a4 v = *reinterpret_cast<const f32x4_a4*>(p);
    return make_float4(v[0], v[1], v[2], v[3]);
}

// rows of a head span: 4 fp32 at element x of a span of d valid elements (zero beyond d), from a 4-byte aligned address
__device__ __forceinline__ float4 ld_span4(const float* row, int x, int d) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (x + 3 < d) v = ld4u(row + x);
    else {
        if (x < d) v.x = row[x];
        if (x + 1 < d) v.y = row[x + 1];
        if (x + 2 < d) v.z = row[x + 2];
    }
    return v;
}


}  // namespace sx
