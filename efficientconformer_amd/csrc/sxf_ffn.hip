// Split-precision FeedForwardModule as ONE kernel (round 6): y = x + 1/2 (Swish(LN(x) W1^T + b1) W2^T + b2), optionally followed by the block-final LayerNorm
// (reference modules.py:385-395; blocks.py:122, 132-135).  Arithmetic as in split.hip: fp32 tensors, every product on the fp16 matrix pipe with operands
// split into two fp16 halves (three MFMAs per product, ~2^-21 relative) - here in the SAME-SCALE form x S = h + l (sx_common.h split2s: activations S = 2^8,
// weights S = 2^10): one accumulator per product sum, undone by one exact multiply; the (h, l / 2048) form of split.hip needs a second accumulator per tile and a
// fold, which was 40 % of this kernel's instructions.
//
// split.hip ran an FFN as LayerNorm + GEMM + GEMM: the (rows, 4 D) hidden activation went to HBM as fp32 and came back (17 GB per Small step), and at K = 120 .. 240
// a 128 x 128 tile spends as long in its split / Swish prologue and epilogue as in its products (8.8 ms per step for 1.4 ms of matrix work).  Here a wave keeps 32
// rows for the whole module, TRANSPOSED (rows = MFMA columns = lanes):
//   * LN(x) is lane-local (a lane owns half a row, its kh twin the other half): statistics by two passes over the row + one xor-32 shuffle; gamma / beta are folded into
//     W1 / b1 at pack time, so the normalised row goes straight into split B fragments (KS1 k-steps, resident in registers);
//   * per chunk of 32 hidden units: H^T = W1_c a^T (A operand = weight rows from LDS, shared by the workgroup's 4 waves = 128 rows), b1 rides in column D of the weight
//     image against a constant 1 in a^T, Swish on the accumulator registers, which ARE the B fragments of the second product (k order of the W2 image permuted to
//     the accumulator layout): Y^T += W2_c H^T.  Nothing of H leaves the registers;
//   * the weight images (fp16 h | l planes, chunk-major, packed at finalize) stream through a two-stage LDS ring: chunk c + 1 is written from registers and chunk c + 2
//     requested from L2 before chunk c's products, one LDS-only barrier per chunk;
//   * epilogue: + b2 / 2 + x (reloaded), optional LayerNorm over the row (lane-local again), 16-byte stores.
#include "kernels.h"
#include "sx_common.h"

namespace {

using namespace sx;

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

constexpr float SA = 256.0f, SW = 1024.0f, UNSCALE = 1.0f / (256.0f * 1024.0f);      // operand scales (activations, weights; encoder.hip packs the images with SW)
constexpr int ROW2 = 80;                                         // bytes per W2 image row in LDS: 32 hidden units (64 B) + 16

template <int KS1, int NT2>
struct FfnLds {
    static constexpr int DP1 = 16 * KS1, DP2 = 32 * NT2, ROW1 = DP1 * 2 + 16;
    static constexpr int W1B = 2 * 32 * ROW1, W2B = 2 * DP2 * ROW2, STAGE = W1B + W2B;
    static constexpr int PIECES = 8 * (DP1 + DP2);               // 16-byte pieces of a chunk's four planes
    static constexpr int NPC = (PIECES + 255) / 256;
};

template <int KS1, int NT2>
__global__ __launch_bounds__(256) void sxf_ffn_kernel(const SxfFfnParams p) {
    using L = FfnLds<KS1, NT2>;
    constexpr int DP1 = L::DP1, DP2 = L::DP2, ROW1 = L::ROW1, NPC = L::NPC;
    extern __shared__ __attribute__((aligned(16))) char sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, kh = lane >> 5;
    const int D = p.D;
    const int m = blockIdx.x * 128 + wave * 32 + lr;
    const int row = m < p.M ? m : p.M - 1;
    const float* xr = p.X + (size_t)row * p.ldx;
    typedef uint32_t u4v __attribute__((ext_vector_type(4)));
    // ---- the weight ring: this thread's pieces of a chunk (global: contiguous; LDS: padded rows)
    uint32_t loff[NPC];
#pragma unroll
    for (int it = 0; it < NPC; ++it) {
        const int q = tid + 256 * it;
        int o;
        if (q < 8 * DP1) { const int pl = q / (4 * DP1), rem = q - pl * 4 * DP1, r = rem / (DP1 / 8), ch = rem - r * (DP1 / 8); o = pl * 32 * ROW1 + r * ROW1 + ch * 16; }
        else { const int q2 = q - 8 * DP1, pl = q2 / (4 * DP2), rem = q2 - pl * 4 * DP2, n = rem >> 2, ch = rem & 3; o = L::W1B + pl * DP2 * ROW2 + n * ROW2 + ch * 16; }
        loff[it] = (uint32_t)o;
    }
    const char* wsrc = reinterpret_cast<const char*>(p.wimg) + (size_t)tid * 16;
    constexpr size_t CB = (size_t)L::PIECES * 16;
    u4v wreg[NPC];
    auto fetch = [&](int c) __attribute__((always_inline)) {
        c = c < p.nchunk ? c : p.nchunk - 1;                      // past the end: the last chunk again (never published)
#pragma unroll
        for (int it = 0; it < NPC; ++it)
            if (L::PIECES % 256 == 0 || tid + 256 * it < L::PIECES) wreg[it] = *reinterpret_cast<const u4v*>(wsrc + (size_t)c * CB + (size_t)it * 4096);
    };
    auto publish = [&](char* st) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < NPC; ++it)
            if (L::PIECES % 256 == 0 || tid + 256 * it < L::PIECES) *reinterpret_cast<u4v*>(st + loff[it]) = wreg[it];
    };
    fetch(0);
    // ---- this lane's half row in registers: ALL loads issued at once (unconditional 16-byte loads at clamped columns, masked by select).  The first version made
    //      three passes of ld_span4 - a branch per quad, so every load waited for the previous one: ~20 us per workgroup before the first MFMA (tools/sxf_ffn_probe.py)
    float xv[KS1][8];
#pragma unroll
    for (int s = 0; s < KS1; ++s) {
        const int x = 16 * s + 8 * kh;
        const int x0 = x < D ? x : D - 4, x1 = x + 4 < D ? x + 4 : D - 4;       // D % 4 == 0: a quad is inside the row or masked as a whole
        const float4 a = *reinterpret_cast<const float4*>(xr + x0), c = *reinterpret_cast<const float4*>(xr + x1);
        const bool ok0 = x < D, ok1 = x + 4 < D;
        xv[s][0] = ok0 ? a.x : 0.f; xv[s][1] = ok0 ? a.y : 0.f; xv[s][2] = ok0 ? a.z : 0.f; xv[s][3] = ok0 ? a.w : 0.f;
        xv[s][4] = ok1 ? c.x : 0.f; xv[s][5] = ok1 ? c.y : 0.f; xv[s][6] = ok1 ? c.z : 0.f; xv[s][7] = ok1 ? c.w : 0.f;
    }
    // ---- LayerNorm statistics (two passes over the registers, fp32; the lane holds columns 16 s + 8 kh .. + 7 of every k-step, its kh twin the rest)
    float sum = 0.f;
#pragma unroll
    for (int s = 0; s < KS1; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) sum += xv[s][e];
    sum += __shfl_xor(sum, 32);
    const float mean = sum / (float)D;
    float sq = 0.f;
#pragma unroll
    for (int s = 0; s < KS1; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float dlt = 16 * s + 8 * kh + e < D ? xv[s][e] - mean : 0.f; sq = fmaf(dlt, dlt, sq); }
    sq += __shfl_xor(sq, 32);
    const float rstd = rsqrtf(sq / (float)D + 1e-6f);
    // ---- a^T as split B fragments: (x - mean) rstd (gamma, beta live in the W1 image), column D = 1 (b1's column), 0 behind it
    f16x8 ah[KS1], al[KS1];
#pragma unroll
    for (int s = 0; s < KS1; ++s) {
        float v[8];
        uint32_t hh[4], ll[4];
#pragma unroll
        for (int e = 0; e < 8; ++e) { const int x = 16 * s + 8 * kh + e; v[e] = x < D ? (xv[s][e] - mean) * (rstd * SA) : (x == D ? SA : 0.f); }
#pragma unroll
        for (int e = 0; e < 4; ++e) split2s(v[2 * e], v[2 * e + 1], hh[e], ll[e]);
        ah[s] = as_f16x8(make_uint4(hh[0], hh[1], hh[2], hh[3])); al[s] = as_f16x8(make_uint4(ll[0], ll[1], ll[2], ll[3]));
    }
    f32x16 oacc[NT2];
#pragma unroll
    for (int t = 0; t < NT2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[t][r] = 0.f;
    publish(sm);
    fetch(1);
    lds_barrier();
    for (int c = 0; c < p.nchunk; ++c) {
        const char* st = sm + (c & 1) * L::STAGE;
#ifdef EFFCONF_DEBUG_ABI
        const int abl = p.ablate;
        if (!(abl & 8)) {
#endif
        if (c + 1 < p.nchunk) publish(sm + ((c + 1) & 1) * L::STAGE);      // stage (c + 1) & 1 was read in iteration c - 1: every wave is past the barrier that closed it
        fetch(c + 2);
#ifdef EFFCONF_DEBUG_ABI
        }
#endif
        // ---- H^T = W1_c a^T (32 hidden units x 32 rows); bias via column D
        // three accumulators, one per product kind, issued round-robin: an MFMA whose accumulator is the previous instruction's result waits ~70 cycles for it
        // (24 - 51 of them in a row on ONE accumulator were 1.7 k of a chunk's cycles at D = 120: tools/sxf_ffn_probe.py); summed once per chunk
        f32x16 h1, h2, h3;
#pragma unroll
        for (int r = 0; r < 16; ++r) { h1[r] = 0.f; h2[r] = 0.f; h3[r] = 0.f; }
        const char* w1 = st + lr * ROW1 + 16 * kh;
        // fragment reads three k-steps ahead of their MFMAs (one wave per SIMD: nothing else hides the LDS latency; left alone the compiler puts every read next to its use)
        f16x8 wh[KS1], wl[KS1];
#pragma unroll
        for (int s = 0; s < KS1; ++s) { wh[s] = *reinterpret_cast<const f16x8*>(w1 + 32 * s); wl[s] = *reinterpret_cast<const f16x8*>(w1 + 32 * ROW1 + 32 * s); }
#ifdef EFFCONF_DEBUG_ABI
        if (!(abl & 1))
#endif
#pragma unroll
        for (int s = 0; s < KS1; ++s) {
            h1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[s], ah[s], h1, 0, 0, 0);
            h2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[s], al[s], h2, 0, 0, 0);
            h3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[s], ah[s], h3, 0, 0, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x100, KS1 >= 3 ? 6 : 2 * KS1, 0);      // DS reads of the first three k-steps
#pragma unroll
        for (int s = 0; s < KS1; ++s) {
            __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);                       // the three MFMAs of k-step s
            if (s + 3 < KS1) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);     // the reads of k-step s + 3
        }
        // ---- Swish (modules.py:389) on the accumulators -> split B fragments of the second product (register 8 s + e <-> k position 8 kh + e of k-step s)
        f16x8 hbh[2], hbl[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            uint32_t hh[4], ll[4];
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float z = ((h1[8 * s + e] + h2[8 * s + e]) + h3[8 * s + e]) * UNSCALE;
#ifdef EFFCONF_DEBUG_ABI
                if (abl & 2) { v[e] = z * SA; continue; }
#endif
                v[e] = (z * SA) * sx_rcp(1.0f + sx_expf(fminf(-z, 87.0f)));
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) split2s(v[2 * e], v[2 * e + 1], hh[e], ll[e]);
            hbh[s] = as_f16x8(make_uint4(hh[0], hh[1], hh[2], hh[3])); hbl[s] = as_f16x8(make_uint4(ll[0], ll[1], ll[2], ll[3]));
        }
        // ---- Y^T += W2_c H^T (one accumulator per tile at the scale SA SW).  Units u = (tile pair, k-step): two tiles alternate so that consecutive MFMAs hit
        //      different accumulators; fragment reads run three units ahead (sched_group_barrier below: left alone the compiler reads right before every use)
        __builtin_amdgcn_sched_barrier(0);                       // its own scheduling region: one sched_group_barrier pipeline per region
        const char* w2 = st + L::W1B + lr * ROW2 + 16 * kh;
        // units = (group of up to four output tiles, k-step); inside a unit the MFMAs run kind-major over the tiles, so consecutive instructions hit different
        // accumulators (the same tile comes back after >= 4 others); the fragment reads of unit u + 1 are issued before the MFMAs of unit u
        constexpr int GS = NT2 < 4 ? NT2 : 4, NG = (NT2 + GS - 1) / GS, NU = 2 * NG;
        f16x8 vh[NU][GS], vl[NU][GS];
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
            for (int i = 0; i < GS; ++i) {
                const int t = (u >> 1) * GS + i, s2 = u & 1;
                if (t < NT2) { vh[u][i] = *reinterpret_cast<const f16x8*>(w2 + 32 * t * ROW2 + 32 * s2); vl[u][i] = *reinterpret_cast<const f16x8*>(w2 + DP2 * ROW2 + 32 * t * ROW2 + 32 * s2); }
            }
#ifdef EFFCONF_DEBUG_ABI
        if (!(abl & 4))
#endif
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int s2 = u & 1;
#pragma unroll
            for (int i = 0; i < GS; ++i) { const int t = (u >> 1) * GS + i; if (t < NT2) oacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh[u][i], hbh[s2], oacc[t], 0, 0, 0); }
#pragma unroll
            for (int i = 0; i < GS; ++i) { const int t = (u >> 1) * GS + i; if (t < NT2) oacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh[u][i], hbl[s2], oacc[t], 0, 0, 0); }
#pragma unroll
            for (int i = 0; i < GS; ++i) { const int t = (u >> 1) * GS + i; if (t < NT2) oacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl[u][i], hbh[s2], oacc[t], 0, 0, 0); }
        }
        {
            constexpr int LASTG = NT2 - (NG - 1) * GS;          // tiles of the last group
            if (NG == 1) __builtin_amdgcn_sched_group_barrier(0x100, 2 * LASTG, 0); else __builtin_amdgcn_sched_group_barrier(0x100, 2 * GS, 0);
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                if (u + 1 < NU) { if (((u + 1) >> 1) == NG - 1) __builtin_amdgcn_sched_group_barrier(0x100, 2 * LASTG, 0); else __builtin_amdgcn_sched_group_barrier(0x100, 2 * GS, 0); }
                if ((u >> 1) == NG - 1) __builtin_amdgcn_sched_group_barrier(0x008, 3 * LASTG, 0); else __builtin_amdgcn_sched_group_barrier(0x008, 3 * GS, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#ifdef EFFCONF_DEBUG_ABI
        if (!(abl & 16))
#endif
        lds_barrier();
    }
    // ---- y = x + Y + b2 / 2 (W2 and b2 are packed pre-scaled by 1/2); optional LayerNorm(y) (blocks.py:135); feature of register (t, r) = 32 t + 8 (r >> 2) + 4 kh + (r & 3)
    float ysum = 0.f;
    float4 xq[NT2][4], bq[NT2][4];
#pragma unroll
    for (int t = 0; t < NT2; ++t)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {                             // all loads first (clamped; D % 4 == 0: a quad is inside or outside as a whole)
            const int f = 32 * t + 8 * rq + 4 * kh, fc = f < D ? f : D - 4;
            xq[t][rq] = *reinterpret_cast<const float4*>(xr + fc);
            bq[t][rq] = *reinterpret_cast<const float4*>(p.b2 + fc);
        }
#pragma unroll
    for (int t = 0; t < NT2; ++t)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            const int f = 32 * t + 8 * rq + 4 * kh;
            const float4 xv4 = xq[t][rq], bz = bq[t][rq];
            oacc[t][4 * rq + 0] = fmaf(oacc[t][4 * rq + 0], UNSCALE, xv4.x + bz.x); oacc[t][4 * rq + 1] = fmaf(oacc[t][4 * rq + 1], UNSCALE, xv4.y + bz.y);
            oacc[t][4 * rq + 2] = fmaf(oacc[t][4 * rq + 2], UNSCALE, xv4.z + bz.z); oacc[t][4 * rq + 3] = fmaf(oacc[t][4 * rq + 3], UNSCALE, xv4.w + bz.w);
            if (f < D) ysum += (oacc[t][4 * rq + 0] + oacc[t][4 * rq + 1]) + (oacc[t][4 * rq + 2] + oacc[t][4 * rq + 3]);
        }
    float omean = 0.f, orstd = 1.f;
    if (p.ln_g) {
        ysum += __shfl_xor(ysum, 32);
        omean = ysum / (float)D;
        float q2 = 0.f;
#pragma unroll
        for (int t = 0; t < NT2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int f = 32 * t + 8 * (r >> 2) + 4 * kh + (r & 3);
                const float dlt = f < D ? oacc[t][r] - omean : 0.f;
                q2 = fmaf(dlt, dlt, q2);
            }
        q2 += __shfl_xor(q2, 32);
        orstd = rsqrtf(q2 / (float)D + 1e-6f);
    }
    if (m < p.M) {
        float* yr = p.Y + (size_t)m * p.ldy;
#pragma unroll
        for (int t = 0; t < NT2; ++t)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int f = 32 * t + 8 * rq + 4 * kh;
                if (f >= D) continue;
                float4 o = make_float4(oacc[t][4 * rq], oacc[t][4 * rq + 1], oacc[t][4 * rq + 2], oacc[t][4 * rq + 3]);
                if (p.ln_g) {
                    const float4 g = *reinterpret_cast<const float4*>(p.ln_g + f), bb = *reinterpret_cast<const float4*>(p.ln_b + f);
                    o.x = (o.x - omean) * orstd * g.x + bb.x; o.y = (o.y - omean) * orstd * g.y + bb.y;
                    o.z = (o.z - omean) * orstd * g.z + bb.z; o.w = (o.w - omean) * orstd * g.w + bb.w;
                }
                *reinterpret_cast<float4*>(yr + f) = o;
            }
    }
}

template <int KS1, int NT2>
int launch_ffn(const SxfFfnParams& p, hipStream_t s) {
    using L = FfnLds<KS1, NT2>;
    static_assert(2 * L::STAGE <= 160 * 1024, "weight ring of the fused split FFN");
    static LdsAttr attr;
    ensure_dynamic_lds(reinterpret_cast<const void*>(&sxf_ffn_kernel<KS1, NT2>), 2 * L::STAGE, attr);
    hipLaunchKernelGGL((sxf_ffn_kernel<KS1, NT2>), dim3((p.M + 127) / 128), dim3(256), 2 * L::STAGE, s, p);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace

// k-steps of the first product (D columns + the bias column) and 32-row output tiles; 0: width not built (the caller keeps LayerNorm + two GEMMs)
void sxf_ffn_shape(int D, int* ks1, int* nt2) { *ks1 = (D + 1 + 15) / 16; *nt2 = (D + 31) / 32; }
bool sxf_ffn_supported(int D) {
    switch (D) { case 24: case 32: case 48: case 100: case 120: case 140: case 144: case 168: case 176: case 180: case 200: case 240: case 256: return true; default: return false; }
}
size_t sxf_ffn_image_halfs(int D, int F) {                       // fp16 elements of one module's weight image
    int ks1, nt2; sxf_ffn_shape(D, &ks1, &nt2);
    return (size_t)((F + 31) / 32) * 64 * (16 * ks1 + 32 * nt2);
}

int launch_sxf_ffn(const SxfFfnParams& p, hipStream_t s) {
    if (p.M <= 0) return 0;
    if (!sxf_ffn_supported(p.D) || p.D % 4 || p.ldx % 4 || p.ldy % 4 || !p.wimg || !p.b2 || p.nchunk <= 0) return -2;
    switch (p.D) {
        case 24: return launch_ffn<2, 1>(p, s);
        case 32: return launch_ffn<3, 1>(p, s);
        case 48: return launch_ffn<4, 2>(p, s);
        case 100: return launch_ffn<7, 4>(p, s);
        case 120: return launch_ffn<8, 4>(p, s);
        case 140: return launch_ffn<9, 5>(p, s);
        case 144: return launch_ffn<10, 5>(p, s);
        case 168: return launch_ffn<11, 6>(p, s);
        case 176: case 180: return launch_ffn<12, 6>(p, s);
        case 200: return launch_ffn<13, 7>(p, s);
        case 240: return launch_ffn<16, 8>(p, s);
        case 256: return launch_ffn<17, 8>(p, s);
    }
    return -2;
}
