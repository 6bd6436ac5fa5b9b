// Split-precision row-local chains (round 6): everything of a Conformer block between two non-row-local kernels (attention, depthwise convolution) as ONE kernel,
// in the arithmetic of sxf_ffn.hip (fp32 tensors; every product on the fp16 matrix pipe with same-scale operand halves x S = h + l, three MFMAs per product, one fp32
// accumulator).  Reference: blocks.py:119-137, modules.py:385-395, 511-522, attentions.py:651-686, 716.
//
//   chain B    : x += O Wo^T + bo (attentions.py:716, blocks.py:126);  g = GLU(LN(x) Wp1^T + bp1)  (modules.py:511-515)
//   chain A    : tail  x = xres + C Wp2^T + bp2 (modules.py:519-522, blocks.py:129);  x += 1/2 FFN2(LN(x));  y = LN(x)  (blocks.py:132-135)
//                head  y += 1/2 FFN1(LN(y)) (next block, blocks.py:122);  Q | K | V = LN(y) Wqkv^T + b  (attentions.py:651-653)
//                (tail alone: the last block; head alone: the first block)
//
// Before: nine launches per block for this work (LayerNorm x 2, GEMM x 4, GLU, FFN x 2), each paying its own prologue / epilogue over the fp32 rows at one wave per
// SIMD - `tools/sxf_ffn_probe.py`: 40 us of a 98 us FFN launch at D = 120 are there with every product removed.  Here a wave keeps its 32 rows TRANSPOSED in the
// accumulator layout from the first load to the last store:
//   * a row tile enters through `load_acc` (the accumulator layout: lane (lr, kh) holds features 32 t + 8 rq + 4 kh .. + 3 of row lr) and becomes split B fragments by
//     `acc_to_frags` - so does every accumulator of a finished product, which makes ONE k order for every weight image: inside a 16-block, position 8 kh + e holds
//     feature 8 (e >> 2) + 4 kh + (e & 3) (pack_sxc_* in encoder.hip);
//   * two product forms.  F1 (`stage_f1`, chunked OUTPUTS: 32 outputs x all of K per chunk, result consumed per chunk: GLU / Q K V stores / the Swish of an FFN) and
//     F2 (`stage_f2`, chunked K: all outputs x 32 inputs per chunk, accumulated in registers: out-proj, pointwise-2, the second FFN product); an FFN is F1 + F2 per chunk;
//   * LayerNorms are lane-local sums + one xor-32 shuffle; gamma / beta of a pre-norm are folded into the following F1 image, its bias rides in column D of the image
//     against a constant in the fragments; biases of F2 products are added in fp32;
//   * one two-stage LDS ring per product (register-staged 16-byte pieces, one LDS-only barrier per chunk); the first chunk of the NEXT product is requested before the
//     epilogue of the current one;
//   * residuals a later product needs again are written to the output buffer and re-read by the SAME lane at the SAME address (no cross-lane visibility involved).
#include "kernels.h"
#include "sx_common.h"

namespace {

using namespace sx;

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

constexpr float SA = 256.0f, SW = 1024.0f, UNS = 1.0f / (256.0f * 1024.0f);      // LayerNorm-ed operands | weights (the scales of sxf_ffn.hip)
constexpr float SR = 64.0f, UNS_R = 1.0f / (64.0f * 1024.0f);                    // operands that are NOT LayerNorm-ed (attention output, conv-module activations): range 1023
constexpr int ROW2 = 80;                                                         // bytes per F2 image row in LDS: 32 inputs (64 B) + 16
enum { MODE_F1 = 1, MODE_F2 = 2, MODE_FFN = 3 };

typedef uint32_t u4v __attribute__((ext_vector_type(4)));

// NO LANE-DIVERGENT BRANCH in these kernels (round 6, profiles/r6_35_side_bisect.txt): hipcc (ROCm 7.2) places VGPR -> AGPR live-range-split copies
// (v_accvgpr_write_b32) inside exec-masked regions; a value defined under the full mask, copied while `if (row < M)` or `if (column < D)` had narrowed it and
// read back after reconvergence is garbage in the lanes that were off - here the Q / K / V row offset of live lanes whose last column quad lies behind D, i.e. a
// store address beyond the aperture (every ragged forward of the D = 120 stage faulted in the builds that had such a copy; the rectangular ones recomputed the
// offset under the full mask).  So masked LOADS read a 16-byte block of zeros, masked STORES are buffer stores at an out-of-range offset, partial ring pieces re-read a
// valid piece and land in a per-thread LDS dump slot: selects on addresses / offsets, no branches (the listing has no s_and_saveexec left).
__device__ __attribute__((aligned(16))) float sxc_zero4[4] = {0.f, 0.f, 0.f, 0.f};
// masked stores: raw buffer stores whose offset lies behind the buffer's byte count are dropped by the address unit - no branch, no memory traffic (a shared sink
// line cost 4 % of the step: every workgroup's masked lanes wrote the same 4 KB)
struct OutBuf { __amdgpu_buffer_rsrc_t rsrc; uint32_t bytes; };
__device__ __forceinline__ OutBuf out_buf(float* base, size_t floats) {
    OutBuf b; b.bytes = (uint32_t)(floats * 4); b.rsrc = __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)b.bytes, 0x00020000); return b;
}
__device__ __forceinline__ void bstore4(const OutBuf& b, uint32_t byte_off, bool on, float x, float y, float z, float w) {
    u4v v; v[0] = __float_as_uint(x); v[1] = __float_as_uint(y); v[2] = __float_as_uint(z); v[3] = __float_as_uint(w);
    __builtin_amdgcn_raw_buffer_store_b128(v, b.rsrc, on ? byte_off : b.bytes, 0, 0);
}

// Which form of the FFN stage a width runs (measured per launch inside the split step, profiles/r6_26 against r6_21): the software pipeline wins where the Swish is
// long against the products and registers allow its second set of hidden-unit fragments (D = 144 .. 180: 250 -> 226 us); at D >= 240 it loses (376 -> 401 us: 500
// registers, the extra copies land between the MFMAs); at D <= 120 the plain loop fits 256 registers, so TWO workgroups share a CU and fill each other's stalls
#ifndef SXC_PIPE_MAX
#define SXC_PIPE_MAX 12
#endif
#ifndef SXC_EARLY_MAX
#define SXC_EARLY_MAX 7
#endif
__host__ __device__ constexpr bool ffn_pipelined(int KS) { return KS <= SXC_PIPE_MAX; }
__host__ __device__ constexpr int waves_per_simd(int KS) { return KS <= 8 ? 2 : 1; }

template <int KS, int NT>
struct CL {
    static constexpr int DP1 = 16 * KS, DP2 = 32 * NT, ROW1 = DP1 * 2 + 16;
    static constexpr int W1B = 2 * 32 * ROW1, W2B = 2 * DP2 * ROW2, STAGE = W1B + W2B;
};

// ---- the weight ring of one product: chunk c of the image = PIECES 16-byte pieces, contiguous in memory, padded rows in LDS (F1 part at 0, F2 part at W1B)
template <int KS, int NT, int MODE>
struct Ring {
    using L = CL<KS, NT>;
    static constexpr int P1 = (MODE & 1) ? 8 * L::DP1 : 0, P2 = (MODE & 2) ? 8 * L::DP2 : 0, PIECES = P1 + P2, NPC = (PIECES + 255) / 256;
    uint32_t loff[NPC];
    u4v wreg[NPC];
    const char* src;
    char* dump;                                                  // this thread's 16-byte LDS slot behind the ring (pieces past the end of a chunk)
    int n, tid;
    __device__ __forceinline__ bool piece_ok(int it) const { return PIECES % 256 == 0 || it < NPC - 1 || tid + 256 * it < PIECES; }
    __device__ __forceinline__ void init(const uint16_t* img, int nchunk, int tid_, char* sm) {
        tid = tid_; n = nchunk;
        dump = sm + 2 * L::STAGE + 16 * tid_;
        src = reinterpret_cast<const char*>(img) + (size_t)tid * 16;
#pragma unroll
        for (int it = 0; it < NPC; ++it) {
            const int q = tid + 256 * it;
            // both layouts computed, one selected (no branch: see the note at the top; which one is a compile-time fact unless a 256-piece group straddles P1)
            const int qa = q < P1 ? q : 0, qb = q >= P1 ? q - P1 : 0;
            const int pla = qa / (4 * L::DP1), rema = qa - pla * 4 * L::DP1, ra = rema / (L::DP1 / 8), cha = rema - ra * (L::DP1 / 8);
            const int oa = pla * 32 * L::ROW1 + ra * L::ROW1 + cha * 16;
            const int plb = qb / (4 * L::DP2), remb = qb - plb * 4 * L::DP2, nb = remb >> 2, chb = remb & 3;
            const int ob = L::W1B + plb * L::DP2 * ROW2 + nb * ROW2 + chb * 16;
            loff[it] = (uint32_t)(P1 == 0 ? ob : (P2 == 0 ? oa : (q < P1 ? oa : ob)));
        }
    }
    __device__ __forceinline__ void fetch(int c) {
        c = c < n ? c : n - 1;                                    // past the end: the last chunk again (never published)
#pragma unroll
        for (int it = 0; it < NPC; ++it)
            wreg[it] = *reinterpret_cast<const u4v*>(src + (size_t)c * ((size_t)PIECES * 16) + (piece_ok(it) ? (size_t)it * 4096 : (size_t)0));
    }
    __device__ __forceinline__ void publish(char* st) {
#pragma unroll
        for (int it = 0; it < NPC; ++it)
            *reinterpret_cast<u4v*>(piece_ok(it) ? st + loff[it] : dump) = wreg[it];
    }
    // first chunk published, second requested; every wave is past the previous product's last barrier when it gets here
    __device__ __forceinline__ void prime(char* sm) { publish(sm); fetch(1); lds_barrier(); }
    // ---- the pipelined FFN stage moves its two weight parts in different phases: unit u = F1 part of chunk u + 1 (ring stage (u + 1) & 1) and F2 part of chunk u
    //      (ring stage u & 1); piece `it` of this thread belongs to the F2 part iff is_f2(it)
    __device__ __forceinline__ bool is_f2(int it) const { return P1 % 256 == 0 ? 256 * it >= P1 : (tid + 256 * it >= P1); }      // a select on the address, not a branch
    __device__ __forceinline__ void fetch_unit(int u) {
        const int c1 = min(max(u + 1, 0), n - 1), c2 = min(max(u, 0), n - 1);
#pragma unroll
        for (int it = 0; it < NPC; ++it)
            wreg[it] = *reinterpret_cast<const u4v*>(src + (size_t)(is_f2(it) ? c2 : c1) * ((size_t)PIECES * 16) + (piece_ok(it) ? (size_t)it * 4096 : (size_t)0));
    }
    __device__ __forceinline__ void publish_unit(char* sm, int u) {
        char *s1 = sm + ((u + 1) & 1) * L::STAGE, *s2 = sm + (u & 1) * L::STAGE;
#pragma unroll
        for (int it = 0; it < NPC; ++it)
            *reinterpret_cast<u4v*>(piece_ok(it) ? (is_f2(it) ? s2 : s1) + loff[it] : dump) = wreg[it];
    }
    // pieces [i0, i1) of unit u / of a whole chunk published (spread over the k-steps of an F1 product: a ds_write_b128 occupies the LDS path for ~13 cycles, sixteen
    // of them in a row were exposed; the REQUESTS stay together behind the product - interleaved with the MFMAs they faulted, profiles/r6_35_side_bisect.txt)
    __device__ __forceinline__ void publish_unit_pieces(char* sm, int u, int i0, int i1) {
        char *s1 = sm + ((u + 1) & 1) * L::STAGE, *s2 = sm + (u & 1) * L::STAGE;
#pragma unroll
        for (int it = 0; it < NPC; ++it)
            if (it >= i0 && it < i1) *reinterpret_cast<u4v*>(piece_ok(it) ? (is_f2(it) ? s2 : s1) + loff[it] : dump) = wreg[it];
    }
    __device__ __forceinline__ void publish_pieces(char* st, int i0, int i1) {
#pragma unroll
        for (int it = 0; it < NPC; ++it)
            if (it >= i0 && it < i1) *reinterpret_cast<u4v*>(piece_ok(it) ? st + loff[it] : dump) = wreg[it];
    }
    // the first request of a product (issued early by the caller)
    __device__ __forceinline__ void fetch0() { if (MODE == MODE_FFN && ffn_pipelined(KS)) fetch_unit(-1); else fetch(0); }
};

// ---- rows <-> accumulator layout
template <int NT>
__device__ __forceinline__ void load_acc(const float* xr, int D, int kh, f32x16 (&acc)[NT]) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            const int f = 32 * t + 8 * rq + 4 * kh;
            const float4 q = *reinterpret_cast<const float4*>(t < NT - 1 || f < D ? xr + f : sxc_zero4);      // quads behind D: zeros (D % 4 == 0; only the last tile has any)
            acc[t][4 * rq + 0] = q.x; acc[t][4 * rq + 1] = q.y; acc[t][4 * rq + 2] = q.z; acc[t][4 * rq + 3] = q.w;
        }
}

// fragments of (acc - sub) mul; column D carries `one` (the bias column of an F1 image), 0 behind it
template <int KS, int NT>
__device__ __forceinline__ void acc_to_frags(const f32x16 (&acc)[NT], float sub, float mul, float one, int D, int kh, f16x8 (&ah)[KS], f16x8 (&al)[KS]) {
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int t = s >> 1, sp = s & 1;
        float v[8];
        uint32_t hh[4], ll[4];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int f = 16 * s + 8 * (e >> 2) + 4 * kh + (e & 3);
            const float a = t < NT ? acc[t < NT ? t : 0][8 * sp + e] : 0.f;
            v[e] = (t < NT - 1 || f < D) ? (a - sub) * mul : (f == D ? one : 0.f);      // only the last tile's k-steps can reach D
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) split2s(v[2 * e], v[2 * e + 1], hh[e], ll[e]);
        ah[s] = as_f16x8(make_uint4(hh[0], hh[1], hh[2], hh[3])); al[s] = as_f16x8(make_uint4(ll[0], ll[1], ll[2], ll[3]));
    }
}

// LayerNorm statistics of a row held in the accumulator layout (entries behind D are zero): two passes, fp32, eps 1e-6 (blocks.py / modules.py LayerNorms)
template <int NT>
__device__ __forceinline__ void row_stats(const f32x16 (&acc)[NT], int D, int kh, float& mean, float& rstd) {
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) s += (acc[t][4 * rq] + acc[t][4 * rq + 1]) + (acc[t][4 * rq + 2] + acc[t][4 * rq + 3]);
    s += __shfl_xor(s, 32);
    mean = s / (float)D;
    float q2 = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int f = 32 * t + 8 * (r >> 2) + 4 * kh + (r & 3);
            const float dlt = (t < NT - 1 || f < D) ? acc[t][r] - mean : 0.f;
            q2 = fmaf(dlt, dlt, q2);
        }
    q2 += __shfl_xor(q2, 32);
    rstd = rsqrtf(q2 / (float)D + 1e-6f);
}

// acc = acc uns + res + bias (entries behind D: 0); res / bias read at this lane's features
template <int NT>
__device__ __forceinline__ void add_residual(f32x16 (&acc)[NT], float uns, const float* res, const float* bias, int D, int kh) {
    // a tile's eight 16-byte loads at a time (all tiles at once = 256 registers of loads beside the accumulators: spills at width 256); quads behind D read zeros
    // and their accumulators are zero (zero image rows): they stay 0
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        float4 xq[4], bq[4];
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            const int f = 32 * t + 8 * rq + 4 * kh;
            const bool ok = t < NT - 1 || f < D;
            xq[rq] = *reinterpret_cast<const float4*>(ok ? res + f : sxc_zero4);
            bq[rq] = *reinterpret_cast<const float4*>(ok ? bias + f : sxc_zero4);
        }
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            const float4 x4 = xq[rq], b4 = bq[rq];
            acc[t][4 * rq + 0] = fmaf(acc[t][4 * rq + 0], uns, x4.x + b4.x); acc[t][4 * rq + 1] = fmaf(acc[t][4 * rq + 1], uns, x4.y + b4.y);
            acc[t][4 * rq + 2] = fmaf(acc[t][4 * rq + 2], uns, x4.z + b4.z); acc[t][4 * rq + 3] = fmaf(acc[t][4 * rq + 3], uns, x4.w + b4.w);
        }
    }
}

template <int NT>
__device__ __forceinline__ void store_acc(const OutBuf& ob, uint32_t row_off, const f32x16 (&acc)[NT], int D, int kh, bool live) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            const int f = 32 * t + 8 * rq + 4 * kh;
            bstore4(ob, row_off + 4 * f, live && (t < NT - 1 || f < D), acc[t][4 * rq], acc[t][4 * rq + 1], acc[t][4 * rq + 2], acc[t][4 * rq + 3]);
        }
}

// acc = (acc - mean) rstd gamma + beta  (entries behind D stay 0)
template <int NT>
__device__ __forceinline__ void apply_ln(f32x16 (&acc)[NT], float mean, float rstd, const float* g, const float* b, int D, int kh) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            const int f = 32 * t + 8 * rq + 4 * kh;
            const bool ok = t < NT - 1 || f < D;                 // quads behind D: gamma = beta = 0 -> 0 (no select on the result: the compiler sinks such loads into exec-masked regions)
            const float4 g4 = *reinterpret_cast<const float4*>(ok ? g + f : sxc_zero4), b4 = *reinterpret_cast<const float4*>(ok ? b + f : sxc_zero4);
            acc[t][4 * rq + 0] = (acc[t][4 * rq + 0] - mean) * rstd * g4.x + b4.x; acc[t][4 * rq + 1] = (acc[t][4 * rq + 1] - mean) * rstd * g4.y + b4.y;
            acc[t][4 * rq + 2] = (acc[t][4 * rq + 2] - mean) * rstd * g4.z + b4.z; acc[t][4 * rq + 3] = (acc[t][4 * rq + 3] - mean) * rstd * g4.w + b4.w;
        }
}

// ---- F1 core: (32 outputs x 32 rows) = W_c a^T over all KS k-steps; three accumulators (one per product kind: an MFMA on the previous instruction's accumulator waits
//      for it), summed by the caller
template <int KS, int NT>
__device__ __forceinline__ void g1(const char* st, int lr, int kh, const f16x8 (&ah)[KS], const f16x8 (&al)[KS], f32x16& h1, f32x16& h2, f32x16& h3) {
    using L = CL<KS, NT>;
#pragma unroll
    for (int r = 0; r < 16; ++r) { h1[r] = 0.f; h2[r] = 0.f; h3[r] = 0.f; }
    const char* w1 = st + lr * L::ROW1 + 16 * kh;
    f16x8 wh[KS], wl[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) { wh[s] = *reinterpret_cast<const f16x8*>(w1 + 32 * s); wl[s] = *reinterpret_cast<const f16x8*>(w1 + 32 * L::ROW1 + 32 * s); }
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        h1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[s], ah[s], h1, 0, 0, 0);
        h2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[s], al[s], h2, 0, 0, 0);
        h3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[s], ah[s], h3, 0, 0, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x100, KS >= 3 ? 6 : 2 * KS, 0);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
        if (s + 3 < KS) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    }
}

// ---- F2 core: oacc[t] += W_c[32 t .. + 31][32 inputs] b^T, b = two k-steps of split fragments; units = (group of up to four tiles, k-step), kind-major inside a unit
template <int KS, int NT>
__device__ __forceinline__ void g2(const char* st, int lr, int kh, const f16x8 (&bh)[2], const f16x8 (&bl)[2], f32x16 (&oacc)[NT]) {
    using L = CL<KS, NT>;
    __builtin_amdgcn_sched_barrier(0);
    const char* w2 = st + L::W1B + lr * ROW2 + 16 * kh;
    constexpr int GS = NT < 4 ? NT : 4, NG = (NT + GS - 1) / GS, NU = 2 * NG;
    f16x8 vh[NU][GS], vl[NU][GS];
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
        for (int i = 0; i < GS; ++i) {
            const int t = (u >> 1) * GS + i, s2 = u & 1;
            if (t < NT) { vh[u][i] = *reinterpret_cast<const f16x8*>(w2 + 32 * t * ROW2 + 32 * s2); vl[u][i] = *reinterpret_cast<const f16x8*>(w2 + L::DP2 * ROW2 + 32 * t * ROW2 + 32 * s2); }
        }
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int s2 = u & 1;
#pragma unroll
        for (int i = 0; i < GS; ++i) { const int t = (u >> 1) * GS + i; if (t < NT) oacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh[u][i], bh[s2], oacc[t], 0, 0, 0); }
#pragma unroll
        for (int i = 0; i < GS; ++i) { const int t = (u >> 1) * GS + i; if (t < NT) oacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh[u][i], bl[s2], oacc[t], 0, 0, 0); }
#pragma unroll
        for (int i = 0; i < GS; ++i) { const int t = (u >> 1) * GS + i; if (t < NT) oacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl[u][i], bh[s2], oacc[t], 0, 0, 0); }
    }
    {
        constexpr int LASTG = NT - (NG - 1) * GS;
        if (NG == 1) __builtin_amdgcn_sched_group_barrier(0x100, 2 * LASTG, 0); else __builtin_amdgcn_sched_group_barrier(0x100, 2 * GS, 0);
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            if (u + 1 < NU) { if (((u + 1) >> 1) == NG - 1) __builtin_amdgcn_sched_group_barrier(0x100, 2 * LASTG, 0); else __builtin_amdgcn_sched_group_barrier(0x100, 2 * GS, 0); }
            if ((u >> 1) == NG - 1) __builtin_amdgcn_sched_group_barrier(0x008, 3 * LASTG, 0); else __builtin_amdgcn_sched_group_barrier(0x008, 3 * GS, 0);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
}

template <int NT>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[NT]) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
}

// ---- product stages.  Every stage ends with all waves behind a barrier (the ring is free)
// F2 over the resident fragments: chunk c = k-steps 2 c, 2 c + 1 (a k-step the operand does not have is a zero fragment)
template <int KS, int NT>
__device__ __forceinline__ void stage_f2(Ring<KS, NT, MODE_F2>& ring, char* sm, int lr, int kh, const f16x8 (&ah)[KS], const f16x8 (&al)[KS], f32x16 (&oacc)[NT]) {
    using L = CL<KS, NT>;
    zero_acc<NT>(oacc);
    ring.prime(sm);
#pragma unroll
    for (int c = 0; c < NT; ++c) {
        const char* st = sm + (c & 1) * L::STAGE;
        if (c + 1 < NT) ring.publish(sm + ((c + 1) & 1) * L::STAGE);
        if (c + 2 < NT) ring.fetch(c + 2);
        f16x8 bh[2], bl[2];
        const f16x8 zf = as_f16x8(make_uint4(0u, 0u, 0u, 0u));
        bh[0] = ah[2 * c < KS ? 2 * c : 0]; bl[0] = al[2 * c < KS ? 2 * c : 0];
        if (2 * c + 1 < KS) { bh[1] = ah[2 * c + 1 < KS ? 2 * c + 1 : 0]; bl[1] = al[2 * c + 1 < KS ? 2 * c + 1 : 0]; } else { bh[1] = zf; bl[1] = zf; }
        g2<KS, NT>(st, lr, kh, bh, bl, oacc);
        lds_barrier();
    }
}

// F1 core with the schedule fixed in the source (a fence per k-step: the machine scheduler, left alone at 500 registers, reads every fragment right before its use
// and parks other work behind the last MFMA): fragment reads two k-steps ahead, `side(s)` = the caller's share of other work for k-step s (weight-ring pieces)
template <int KS, int NT, class Side>
__device__ __forceinline__ void g1x(const char* st, int lr, int kh, const f16x8 (&ah)[KS], const f16x8 (&al)[KS], f32x16& h1, f32x16& h2, f32x16& h3, Side side) {
    using L = CL<KS, NT>;
#pragma unroll
    for (int r = 0; r < 16; ++r) { h1[r] = 0.f; h2[r] = 0.f; h3[r] = 0.f; }
    const char* w1 = st + lr * L::ROW1 + 16 * kh;
    f16x8 wh[KS], wl[KS];
    constexpr int PD = 2;
#pragma unroll
    for (int s = 0; s < PD && s < KS; ++s) { wh[s] = *reinterpret_cast<const f16x8*>(w1 + 32 * s); wl[s] = *reinterpret_cast<const f16x8*>(w1 + 32 * L::ROW1 + 32 * s); }
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        __builtin_amdgcn_sched_barrier(0);
        if (s + PD < KS) { wh[s + PD] = *reinterpret_cast<const f16x8*>(w1 + 32 * (s + PD)); wl[s + PD] = *reinterpret_cast<const f16x8*>(w1 + 32 * L::ROW1 + 32 * (s + PD)); }
        h1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[s], ah[s], h1, 0, 0, 0);
        side(s);
        h2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[s], al[s], h2, 0, 0, 0);
        h3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[s], ah[s], h3, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
}

// Swish (modules.py:389) of a finished F1 chunk -> split B fragments of the F2 product, as 216 single-instruction steps (exp through v_exp_f32 with the rounding of
// its argument corrected to first order, 1 / x by v_rcp_f32: 1 ulp each).  Step order: four values at a time, stage-major inside the four (independent neighbours);
// value r = register r of the accumulators <-> k position 8 kh + (r & 7) of k-step r >> 3.
struct SwishState { float x[16], y[16], w[16]; uint32_t hh[8], ll[8]; };
constexpr int SWISH_STEPS = 16 * 13 + 8;
// Every step ends in an empty volatile asm on its result: the steps are pure arithmetic, which instruction selection otherwise sinks to their only user (the pack
// behind the last MFMA) whatever fences stand in the source; the asm statements keep their order among the fences.
#ifdef SXC_NO_PIN
#define SWISH_PIN(v)
#else
#define SWISH_PIN(v) asm volatile("" : "+v"(v))
#endif
__device__ __forceinline__ void swish_step(int idx, const f32x16& h1, const f32x16& h2, const f32x16& h3, SwishState& q) {
    const int g = idx / 54, o = idx - 54 * g;
    if (o >= 52) { const int pr = 2 * g + (o - 52); split2s(q.x[2 * pr], q.x[2 * pr + 1], q.hh[pr], q.ll[pr]); SWISH_PIN(q.hh[pr]); SWISH_PIN(q.ll[pr]); return; }
    const int stage = o >> 2, r = 4 * g + (o & 3);
    switch (stage) {
        case 0: q.x[r] = h1[r] + h2[r]; SWISH_PIN(q.x[r]); break;
        case 1: q.x[r] = q.x[r] + h3[r]; SWISH_PIN(q.x[r]); break;
        case 2: q.x[r] = q.x[r] * UNS; SWISH_PIN(q.x[r]); break;
        case 3: q.y[r] = fminf(-q.x[r], 87.0f); SWISH_PIN(q.y[r]); break;
        case 4: q.w[r] = q.y[r] * 1.44269502162933349609375f; SWISH_PIN(q.w[r]); break;
        case 5: q.y[r] = fmaf(q.y[r], 1.44269502162933349609375f, -q.w[r]); SWISH_PIN(q.y[r]); break;
        case 6: q.w[r] = __builtin_amdgcn_exp2f(q.w[r]); SWISH_PIN(q.w[r]); break;
        case 7: q.y[r] = q.y[r] * 0.693147180559945f; SWISH_PIN(q.y[r]); break;
        case 8: q.w[r] = fmaf(q.w[r], q.y[r], q.w[r]); SWISH_PIN(q.w[r]); break;
        case 9: q.w[r] = 1.0f + q.w[r]; SWISH_PIN(q.w[r]); break;
        case 10: q.w[r] = __builtin_amdgcn_rcpf(q.w[r]); SWISH_PIN(q.w[r]); break;
        case 11: q.x[r] = q.x[r] * SA; SWISH_PIN(q.x[r]); break;
        default: q.x[r] = q.x[r] * q.w[r]; SWISH_PIN(q.x[r]); break;
    }
}
__device__ __forceinline__ void swish_pack(const SwishState& q, f16x8 (&nh)[2], f16x8 (&nl)[2]) {
#pragma unroll
    for (int s = 0; s < 2; ++s) { nh[s] = as_f16x8(make_uint4(q.hh[4 * s], q.hh[4 * s + 1], q.hh[4 * s + 2], q.hh[4 * s + 3])); nl[s] = as_f16x8(make_uint4(q.ll[4 * s], q.ll[4 * s + 1], q.ll[4 * s + 2], q.ll[4 * s + 3])); }
}
__device__ __forceinline__ void swish_frags(const f32x16& h1, const f32x16& h2, const f32x16& h3, f16x8 (&nh)[2], f16x8 (&nl)[2]) {      // the same arithmetic, value by value
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        uint32_t hh[4], ll[4];
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float z = ((h1[8 * s + e] + h2[8 * s + e]) + h3[8 * s + e]) * UNS;
            const float nz = fminf(-z, 87.0f);
            const float t = nz * 1.44269502162933349609375f;
            const float cc = fmaf(nz, 1.44269502162933349609375f, -t) * 0.693147180559945f;
            float ex = __builtin_amdgcn_exp2f(t);
            ex = fmaf(ex, cc, ex);
            v[e] = (z * SA) * __builtin_amdgcn_rcpf(1.0f + ex);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) split2s(v[2 * e], v[2 * e + 1], hh[e], ll[e]);
        nh[s] = as_f16x8(make_uint4(hh[0], hh[1], hh[2], hh[3])); nl[s] = as_f16x8(make_uint4(ll[0], ll[1], ll[2], ll[3]));
    }
}

// F2 of chunk c - 1 (fragments bh / bl, weights at `st`) with the Swish of chunk c's F1 accumulators BETWEEN its MFMAs, the order fixed in the source (a fence per
// MFMA, its quota of Swish steps behind it): the two are independent, a wave is alone on its SIMD, and nothing else would fill the matrix pipe while the Swish
// issues - left to the scheduler it stayed serial (at D = 120 longer than both products together).  Fragment reads run one unit (group of <= 4 tiles, k-step) ahead.
template <int KS, int NT>
__device__ __forceinline__ void g2_swish(const char* st, int lr, int kh, const f16x8 (&bh)[2], const f16x8 (&bl)[2], f32x16 (&oacc)[NT],
                                         const f32x16& h1, const f32x16& h2, const f32x16& h3, f16x8 (&nh)[2], f16x8 (&nl)[2]) {
    using L = CL<KS, NT>;
    const char* w2 = st + L::W1B + lr * ROW2 + 16 * kh;
    constexpr int GS = NT < 4 ? NT : 4, NG = (NT + GS - 1) / GS, NU = 2 * NG, NM = 6 * NT, Q = (SWISH_STEPS + NM - 1) / NM;
    f16x8 vh[NU][GS], vl[NU][GS];
    auto load_unit = [&](int u) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < GS; ++i) {
            const int t = (u >> 1) * GS + i, s2 = u & 1;
            if (t < NT) { vh[u][i] = *reinterpret_cast<const f16x8*>(w2 + 32 * t * ROW2 + 32 * s2); vl[u][i] = *reinterpret_cast<const f16x8*>(w2 + L::DP2 * ROW2 + 32 * t * ROW2 + 32 * s2); }
        }
    };
    SwishState q;
    constexpr int LASTG = NT - (NG - 1) * GS;
    __builtin_amdgcn_sched_barrier(0);
    load_unit(0);
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int s2 = u & 1, gq = u >> 1, gsz = gq == NG - 1 ? LASTG : GS;
        const int base = 3 * (2 * GS * gq + s2 * gsz);             // MFMAs before this unit (a function of the loop indices only: every Swish step index below is a constant)
#pragma unroll
        for (int kind = 0; kind < 3; ++kind)
#pragma unroll
            for (int i = 0; i < GS; ++i) {
                const int t = gq * GS + i;
                if (t < NT) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (kind == 0 && i == 0 && u + 1 < NU) load_unit(u + 1);
                    oacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kind == 2 ? vl[u][i] : vh[u][i], kind == 1 ? bl[s2] : bh[s2], oacc[t], 0, 0, 0);
                    const int m0 = (base + kind * gsz + i) * Q;
#pragma unroll
                    for (int j = 0; j < Q; ++j) if (m0 + j < SWISH_STEPS) swish_step(m0 + j, h1, h2, h3, q);
                }
            }
    }
    static_assert(NM * Q >= SWISH_STEPS, "every Swish step has its MFMA");
    __builtin_amdgcn_sched_barrier(0);
    swish_pack(q, nh, nl);
}

// FeedForwardModule body, software-pipelined over the chunks of 32 hidden units: iteration c = F1 of chunk c (the weight-ring pieces between its k-steps), then F2
// of chunk c - 1 with the Swish of chunk c between its MFMAs.  The weight parts move in units (Ring::fetch_unit): F1 part of chunk c + 1 and F2 part of chunk c
// are published in iteration c - the F1 slot of ring stage (c + 1) & 1 was last read in iteration c - 1, the F2 slot of stage c & 1 (by the F2 of chunk c - 2) likewise.
// FeedForwardModule body, plain form: per chunk of 32 hidden units F1, Swish, F2 (sxf_ffn.hip's loop)
template <int KS, int NT>
__device__ __forceinline__ void stage_ffn_plain(Ring<KS, NT, MODE_FFN>& ring, char* sm, int lr, int kh, const f16x8 (&ah)[KS], const f16x8 (&al)[KS], f32x16 (&oacc)[NT]) {
    using L = CL<KS, NT>;
    using R = Ring<KS, NT, MODE_FFN>;
    zero_acc<NT>(oacc);
    ring.prime(sm);
    for (int c = 0; c < ring.n; ++c) {
        const char* st = sm + (c & 1) * L::STAGE;
        char* nx = sm + ((c + 1) & 1) * L::STAGE;                // chunk c + 1's slot: last read in iteration c - 1 (a publish past the end lands in a slot nobody reads)
        f32x16 h1, h2, h3;
        g1x<KS, NT>(st, lr, kh, ah, al, h1, h2, h3, [&](int s) __attribute__((always_inline)) { ring.publish_pieces(nx, s * R::NPC / KS, (s + 1) * R::NPC / KS); });
        ring.fetch(c + 2);
        f16x8 hbh[2], hbl[2];
        swish_frags(h1, h2, h3, hbh, hbl);
        g2<KS, NT>(st, lr, kh, hbh, hbl, oacc);
        lds_barrier();
    }
}

template <int KS, int NT>
__device__ __forceinline__ void stage_ffn_pipe(Ring<KS, NT, MODE_FFN>& ring, char* sm, int lr, int kh, const f16x8 (&ah)[KS], const f16x8 (&al)[KS], f32x16 (&oacc)[NT]) {
    using L = CL<KS, NT>;
    using R = Ring<KS, NT, MODE_FFN>;
    zero_acc<NT>(oacc);
    ring.publish_unit(sm, -1);                                   // F1 part of chunk 0 (the F2 half of this unit lands in a slot nobody reads before it is rewritten)
    ring.fetch_unit(0);
    lds_barrier();
    f16x8 hbh[2], hbl[2];
    {
        f32x16 h1, h2, h3;
        g1x<KS, NT>(sm, lr, kh, ah, al, h1, h2, h3, [&](int s) __attribute__((always_inline)) { ring.publish_unit_pieces(sm, 0, s * R::NPC / KS, (s + 1) * R::NPC / KS); });
        ring.fetch_unit(1);
        swish_frags(h1, h2, h3, hbh, hbl);
        lds_barrier();
    }
    for (int c = 1; c < ring.n; ++c) {
        f32x16 h1, h2, h3;
        g1x<KS, NT>(sm + (c & 1) * L::STAGE, lr, kh, ah, al, h1, h2, h3, [&](int s) __attribute__((always_inline)) { ring.publish_unit_pieces(sm, c, s * R::NPC / KS, (s + 1) * R::NPC / KS); });
        ring.fetch_unit(c + 1);
        f16x8 nh[2], nl[2];
        g2_swish<KS, NT>(sm + ((c - 1) & 1) * L::STAGE, lr, kh, hbh, hbl, oacc, h1, h2, h3, nh, nl);
        hbh[0] = nh[0]; hbh[1] = nh[1]; hbl[0] = nl[0]; hbl[1] = nl[1];
        lds_barrier();
    }
    g2<KS, NT>(sm + ((ring.n - 1) & 1) * L::STAGE, lr, kh, hbh, hbl, oacc);
    lds_barrier();
}

template <int KS, int NT>
__device__ __forceinline__ void stage_ffn(Ring<KS, NT, MODE_FFN>& ring, char* sm, int lr, int kh, const f16x8 (&ah)[KS], const f16x8 (&al)[KS], f32x16 (&oacc)[NT]) {
    if (ffn_pipelined(KS)) stage_ffn_pipe<KS, NT>(ring, sm, lr, kh, ah, al, oacc); else stage_ffn_plain<KS, NT>(ring, sm, lr, kh, ah, al, oacc);
}

// F1 with a per-chunk consumer: epi(c, z[16]) - z = the chunk's 32 outputs of this lane's row, register r <-> output 32 c + 8 (r >> 2) + 4 kh + (r & 3)
template <int KS, int NT, class Epi>
__device__ __forceinline__ void stage_f1(Ring<KS, NT, MODE_F1>& ring, char* sm, int lr, int kh, const f16x8 (&ah)[KS], const f16x8 (&al)[KS], float uns, Epi epi) {
    using L = CL<KS, NT>;
    ring.prime(sm);
    for (int c = 0; c < ring.n; ++c) {
        const char* st = sm + (c & 1) * L::STAGE;
        if (c + 1 < ring.n) ring.publish(sm + ((c + 1) & 1) * L::STAGE);
        ring.fetch(c + 2);
        f32x16 h1, h2, h3;
        g1<KS, NT>(st, lr, kh, ah, al, h1, h2, h3);
        float z[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) z[r] = ((h1[r] + h2[r]) + h3[r]) * uns;
        epi(c, z);
        lds_barrier();
    }
}

__device__ __forceinline__ long long remap_row(int m, int rows, int pitch) { return rows > 0 ? (long long)(m / rows) * pitch + m % rows : (long long)m; }

// ---- chain B
template <int KS, int NT>
__global__ __launch_bounds__(256, waves_per_simd(KS)) void sxc_b_kernel(const SxcBParams p) {
    using L = CL<KS, NT>;
    extern __shared__ __attribute__((aligned(16))) char sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, kh = lane >> 5;
    const int D = p.D, De = p.De;
    constexpr bool EARLY = KS <= SXC_EARLY_MAX;                               // the next product's first chunk requested BEFORE the epilogue of the current one (wider: 64 more live registers = spills)
    const int m = blockIdx.x * 128 + wave * 32 + lr;
    const int row = m < p.M ? m : p.M - 1;
    Ring<KS, NT, MODE_F2> ra;
    ra.init(p.w_o, NT, tid, sm);
    ra.fetch0();
    f16x8 ah[KS], al[KS];
    f32x16 acc[NT];
    load_acc<NT>(p.o + remap_row(row, p.o_rows, p.o_pitch) * D, D, kh, acc);
    acc_to_frags<KS, NT>(acc, 0.f, SR, SR, D, kh, ah, al);
    stage_f2<KS, NT>(ra, sm, lr, kh, ah, al, acc);
    Ring<KS, NT, MODE_F1> rb;
    if (EARLY) { rb.init(p.w_p1, p.nch_p1, tid, sm); rb.fetch0(); }
    float* xr = p.x + (size_t)row * D;
    add_residual<NT>(acc, UNS_R, xr, p.b_o, D, kh);               // x += O Wo^T + bo
    const OutBuf obx = out_buf(p.x, (size_t)p.M * D);
    store_acc<NT>(obx, (uint32_t)row * D * 4, acc, D, kh, m < p.M);
    float mean, rstd;
    row_stats<NT>(acc, D, kh, mean, rstd);
    acc_to_frags<KS, NT>(acc, mean, rstd * SA, SA, D, kh, ah, al);
    if (!EARLY) { rb.init(p.w_p1, p.nch_p1, tid, sm); rb.fetch0(); }
    const OutBuf obg = out_buf(p.g, (size_t)p.M * De);
    const uint32_t g_off = (uint32_t)row * De * 4;
    float za[16];
    stage_f1<KS, NT>(rb, sm, lr, kh, ah, al, UNS, [&](int c, const float (&z)[16]) __attribute__((always_inline)) {
        if (!(c & 1)) {
#pragma unroll
            for (int r = 0; r < 16; ++r) za[r] = z[r];
        } else {                                                 // GLU (modules.py:514): value half x sigmoid(gate half)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int f = 32 * (c >> 1) + 8 * rq + 4 * kh;
                float4 o4;
                o4.x = za[4 * rq + 0] * sx_rcp(1.0f + sx_expf(fminf(-z[4 * rq + 0], 87.0f))); o4.y = za[4 * rq + 1] * sx_rcp(1.0f + sx_expf(fminf(-z[4 * rq + 1], 87.0f)));
                o4.z = za[4 * rq + 2] * sx_rcp(1.0f + sx_expf(fminf(-z[4 * rq + 2], 87.0f))); o4.w = za[4 * rq + 3] * sx_rcp(1.0f + sx_expf(fminf(-z[4 * rq + 3], 87.0f)));
                bstore4(obg, g_off + 4 * f, m < p.M && f < De, o4.x, o4.y, o4.z, o4.w);
            }
        }
    });
}

// ---- chain A: TAIL and / or HEAD
template <int KS, int NT, bool TAIL, bool HEAD>
__global__ __launch_bounds__(256) void sxc_a_kernel(const SxcAParams p) {
    using L = CL<KS, NT>;
    extern __shared__ __attribute__((aligned(16))) char sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, kh = lane >> 5;
    const int D = p.D;
    constexpr bool EARLY = KS <= SXC_EARLY_MAX;
    const int m = blockIdx.x * 128 + wave * 32 + lr;
    const int row = m < p.M ? m : p.M - 1;
    const bool live = m < p.M;
    float* yr = p.y + (size_t)row * D;
    const OutBuf oby = out_buf(p.y, (size_t)p.M * D);
    const uint32_t y_off = (uint32_t)row * D * 4;
    f16x8 ah[KS], al[KS];
    f32x16 acc[NT];
    float mean, rstd;
    if (TAIL) {
        Ring<KS, NT, MODE_F2> ra;
        ra.init(p.w_p2, NT, tid, sm);
        ra.fetch0();
        load_acc<NT>(p.c + (size_t)row * D, D, kh, acc);
        acc_to_frags<KS, NT>(acc, 0.f, SR, SR, D, kh, ah, al);
        stage_f2<KS, NT>(ra, sm, lr, kh, ah, al, acc);
        Ring<KS, NT, MODE_FFN> rf;
        if (EARLY) { rf.init(p.w_f2, p.nch_f2, tid, sm); rf.fetch0(); }
        add_residual<NT>(acc, UNS_R, p.xres + (size_t)row * D, p.b_p2, D, kh);      // x = xres + C Wp2^T + bp2
        store_acc<NT>(oby, y_off, acc, D, kh, live);
        row_stats<NT>(acc, D, kh, mean, rstd);
        acc_to_frags<KS, NT>(acc, mean, rstd * SA, SA, D, kh, ah, al);
        if (!EARLY) { rf.init(p.w_f2, p.nch_f2, tid, sm); rf.fetch0(); }
        stage_ffn<KS, NT>(rf, sm, lr, kh, ah, al, acc);
        Ring<KS, NT, MODE_FFN> rh;
        if (HEAD && EARLY) { rh.init(p.w_f1, p.nch_f1, tid, sm); rh.fetch0(); }
        add_residual<NT>(acc, UNS, yr, p.b_f2, D, kh);            // x += 1/2 FFN2(LN(x))  (W2, b2 pre-scaled)
        row_stats<NT>(acc, D, kh, mean, rstd);
        apply_ln<NT>(acc, mean, rstd, p.ln_g, p.ln_b, D, kh);     // y = LN(x): the block's output
        store_acc<NT>(oby, y_off, acc, D, kh, live);
        if (HEAD) {
            row_stats<NT>(acc, D, kh, mean, rstd);
            acc_to_frags<KS, NT>(acc, mean, rstd * SA, SA, D, kh, ah, al);
            if (!EARLY) { rh.init(p.w_f1, p.nch_f1, tid, sm); rh.fetch0(); }
            stage_ffn<KS, NT>(rh, sm, lr, kh, ah, al, acc);
        }
    } else {
        Ring<KS, NT, MODE_FFN> rh;
        rh.init(p.w_f1, p.nch_f1, tid, sm);
        rh.fetch0();
        load_acc<NT>(yr, D, kh, acc);
        row_stats<NT>(acc, D, kh, mean, rstd);
        acc_to_frags<KS, NT>(acc, mean, rstd * SA, SA, D, kh, ah, al);
        stage_ffn<KS, NT>(rh, sm, lr, kh, ah, al, acc);
    }
    if (HEAD) {
        Ring<KS, NT, MODE_F1> rq;
        if (EARLY) { rq.init(p.w_qkv, 3 * NT, tid, sm); rq.fetch0(); }
        add_residual<NT>(acc, UNS, yr, p.b_f1, D, kh);            // y += 1/2 FFN1(LN(y))
        store_acc<NT>(oby, y_off, acc, D, kh, live);
        row_stats<NT>(acc, D, kh, mean, rstd);
        acc_to_frags<KS, NT>(acc, mean, rstd * SA, SA, D, kh, ah, al);
        if (!EARLY) { rq.init(p.w_qkv, 3 * NT, tid, sm); rq.fetch0(); }
        const OutBuf obq = out_buf(p.q, p.qkv_bytes / 4);
        const uint32_t q_off = (uint32_t)(remap_row(row, p.q_rows, p.q_pitch) * D * 4);
        const uint32_t q_step = (uint32_t)(p.qkv_stride * 4);
        stage_f1<KS, NT>(rq, sm, lr, kh, ah, al, UNS, [&](int c, const float (&z)[16]) __attribute__((always_inline)) {
            const int which = c / NT, cc = c - which * NT;
            const uint32_t dst = q_off + (uint32_t)which * q_step;
#pragma unroll
            for (int rq4 = 0; rq4 < 4; ++rq4) {
                const int f = 32 * cc + 8 * rq4 + 4 * kh;
                bstore4(obq, dst + 4 * f, live && (cc < NT - 1 || f < D), z[4 * rq4], z[4 * rq4 + 1], z[4 * rq4 + 2], z[4 * rq4 + 3]);
            }
        });
    }
}

template <int KS, int NT>
int launch_b(const SxcBParams& p, hipStream_t s) {
    using L = CL<KS, NT>;
    static_assert(2 * L::STAGE + 4096 <= 160 * 1024, "weight ring of the split chains + the dump slots");
    static LdsAttr attr;
    ensure_dynamic_lds(reinterpret_cast<const void*>(&sxc_b_kernel<KS, NT>), 2 * L::STAGE + 4096, attr);
    hipLaunchKernelGGL((sxc_b_kernel<KS, NT>), dim3((p.M + 127) / 128), dim3(256), 2 * L::STAGE + 4096, s, p);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

template <int KS, int NT, bool TAIL, bool HEAD>
int launch_a2(const SxcAParams& p, hipStream_t s) {
    using L = CL<KS, NT>;
    static LdsAttr attr;
    ensure_dynamic_lds(reinterpret_cast<const void*>(&sxc_a_kernel<KS, NT, TAIL, HEAD>), 2 * L::STAGE + 4096, attr);
    hipLaunchKernelGGL((sxc_a_kernel<KS, NT, TAIL, HEAD>), dim3((p.M + 127) / 128), dim3(256), 2 * L::STAGE + 4096, s, p);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

template <int KS, int NT>
int launch_a(const SxcAParams& p, hipStream_t s) {
    if (p.tail && p.head) return launch_a2<KS, NT, true, true>(p, s);
    if (p.tail) return launch_a2<KS, NT, true, false>(p, s);
    return launch_a2<KS, NT, false, true>(p, s);
}

}  // namespace

// widths of the shipped configurations (the Tiny test config included); everything else stays on the per-module kernels
bool sxc_supported(int D) {
    switch (D) { case 24: case 32: case 48: case 100: case 120: case 140: case 144: case 168: case 176: case 180: case 200: case 240: case 256: return true; default: return false; }
}

#define SXC_DISPATCH(FN, P)                                  \
    switch ((P).D) {                                         \
        case 24: return FN<2, 1>(P, s);                      \
        case 32: return FN<3, 1>(P, s);                      \
        case 48: return FN<4, 2>(P, s);                      \
        case 100: return FN<7, 4>(P, s);                     \
        case 120: return FN<8, 4>(P, s);                     \
        case 140: return FN<9, 5>(P, s);                     \
        case 144: return FN<10, 5>(P, s);                    \
        case 168: return FN<11, 6>(P, s);                    \
        case 176: case 180: return FN<12, 6>(P, s);          \
        case 200: return FN<13, 7>(P, s);                    \
        case 240: return FN<16, 8>(P, s);                    \
        case 256: return FN<17, 8>(P, s);                    \
    }                                                        \
    return -2;

int launch_sxc_b(const SxcBParams& p, hipStream_t s) {
    if (p.M <= 0) return 0;
    if (!sxc_supported(p.D) || p.De % 4 || !p.w_o || !p.w_p1 || !p.b_o || p.nch_p1 <= 0 || (p.nch_p1 & 1)) return -2;
    SXC_DISPATCH(launch_b, p)
}

int launch_sxc_a(const SxcAParams& p, hipStream_t s) {
    if (p.M <= 0) return 0;
    if (!sxc_supported(p.D) || (!p.tail && !p.head)) return -2;
    if (p.tail && (!p.w_p2 || !p.w_f2 || !p.b_p2 || !p.b_f2 || !p.ln_g || !p.ln_b || !p.c || !p.xres || p.nch_f2 <= 0)) return -2;
    if (p.head && (!p.w_f1 || !p.w_qkv || !p.b_f1 || !p.q || p.nch_f1 <= 0)) return -2;
    SXC_DISPATCH(launch_a, p)
}
