// Shared device helpers for the gfx950 (CDNA4 / MI355X) kernels of libeffconf.
// Wavefront = 64 lanes everywhere in this code base; no other target is supported.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;  // storage type for bfloat16 activations / weights
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define EC_WAVE 64

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
// float -> bf16, round-to-nearest-even in hardware (v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
    union { bf16x2_t b; uint32_t u; } c;
    c.b = bf16x2_t{(__bf16)lo, (__bf16)hi};
    return c.u;
}
__device__ __forceinline__ uint16_t f2bf(float f) { return (uint16_t)(pack_bf2(f, 0.f) & 0xFFFFu); }
__device__ __forceinline__ float bf2f(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }

// sigmoid via v_exp_f32 (2^x) + v_rcp_f32: ~1 ulp, no IEEE divide sequence
__device__ __forceinline__ float sigmoidf_(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896f * x));
}
__device__ __forceinline__ float swishf_(float x) { return x * sigmoidf_(x); }

// Zero the bf16 elements with index >= valid of a 16-byte chunk of 8 bf16 (valid may be <= 0 or >= 8).
// Written without a local array: an indexed temporary made the compiler park whole staging arrays in scratch memory.
__device__ __forceinline__ uint32_t mask_dw(uint32_t w, int rem) { return rem <= 0 ? 0u : (rem == 1 ? (w & 0xFFFFu) : w); }
__device__ __forceinline__ uint4 mask_chunk(uint4 v, int valid) {
    return make_uint4(mask_dw(v.x, valid), mask_dw(v.y, valid - 2), mask_dw(v.z, valid - 4), mask_dw(v.w, valid - 6));
}

__device__ __forceinline__ bf16x8 as_bf16x8(uint4 v) {
    union { uint4 u; bf16x8 b; } c;
    c.u = v;
    return c.b;
}

// Raise a kernel's dynamic-LDS limit when a launch needs more than any earlier launch of that instantiation ON THAT DEVICE (the
// attribute is per device; one process may hold handles on several).  One static LdsAttr per launch site.
struct LdsAttr { int bytes[16] = {}; };
inline void ensure_dynamic_lds(const void* fn, int bytes, LdsAttr& st) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    dev &= 15;
    if (st.bytes[dev] < bytes) {
        (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        st.bytes[dev] = bytes;
    }
}

// XCD-aware, bijective remap of a linear workgroup id: hardware places block b on XCD b % 8
// (speed only, never correctness); give every XCD a contiguous chunk of the logical tile space so
// neighbouring tiles (which share operand panels) hit the same private L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int nx = 8;
    int xcd = bid % nx, idx = bid / nx;
    int q = nwg / nx, r = nwg % nx;
    int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// Ragged batches: utterance b owns the rows [off[b], off[b + 1]) of a concatenated row space (off has n + 1 entries, non-decreasing).
// Largest b in [0, n) with off[b] <= m  (m < off[n]); ~log2(n) L1-resident loads.
__device__ __forceinline__ int ragged_find(const int* __restrict__ off, int n, int m) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (off[mid] <= m) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// The same search by a whole wave (all 64 lanes active, m wave-uniform): every lane loads 4 entries per 256 (coalesced dwords) and the count of entries
// <= m comes from ballots - ONE memory round trip for n <= 256 instead of log2(n) dependent ones at the head of every workgroup of the ragged attention and
// depthwise-convolution launches (round 5; the search was 4 % of the attention kernel, profiles/r5_17_attention_ablations.txt).
__device__ __forceinline__ int ragged_find_wave(const int* __restrict__ off, int n, int m) {
    const int lane = threadIdx.x & 63;
    int cnt = 0;
    for (int base = 0; base < n; base += 256) {
        int e[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int i = base + 64 * k + lane; e[k] = i < n ? off[i] : 0x7fffffff; }
#pragma unroll
        for (int k = 0; k < 4; ++k) cnt += __popcll(__ballot(e[k] <= m));
    }
    return __builtin_amdgcn_readfirstlane(cnt) - 1;
}

static inline int ec_round_up(int x, int m) { return (x + m - 1) / m * m; }
static inline int ec_cdiv(int a, int b) { return (a + b - 1) / b; }
