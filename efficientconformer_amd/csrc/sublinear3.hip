// Fused Conv2dSubsampling + Linear, third form (round 6): chunked over (output frequency, 32 channels) so that ANY channel count and model width fits.
//
// Reference: Conv2dSubsampling.forward (models/modules.py:232-249, one layer, C_in = 1: Conv2d 3x3 s2 p1 -> BatchNorm2d(eval) -> Swish -> reshape to
// (B, C*F/2, T1)) + transpose + nn.Linear(C*F/2 -> D0) of ConformerEncoder.forward (encoders.py:113-116).
//
// sublinear2.hip keeps one output frequency's whole weight slab (D0 rows x C columns) in LDS: 74 KB per slab at C = D0 = 180 (two slabs, one 4-wave workgroup
// per CU: 1.1 ms per Medium step, 3.5 x its Swish's VALU floor) and no instance at C = D0 = 360, where the front end runs as a convolution kernel + a K = 14400
// GEMM with the (frames, 14400) bf16 activation written to HBM and read back (8.4 GB per Large step).  Here - the bf16 sibling of sxf_sub.hip - a wave keeps 32
// frames (lanes = frames) and walks chunks of 32 channels at one output frequency f':
//   * first product  H^T = Wc_cb P_f'^T : the 3 x 3 convolution as ONE 16-wide MFMA k-step (A operand = 9 folded taps of 32 channels + the folded bias in tap 9
//     against a constant 1; B operand = this lane's own mel patch), fp32-accurate on bf16 MFMAs by splitting both operands (hi hi + hi lo + lo hi, as sublinear2.hip);
//   * Swish on the accumulator registers, rounded to bf16: they ARE the B fragments of the second product  Y^T += Wl_chunk H^T  (k order of the Linear's image
//     permuted to the accumulator layout at pack time); the 32 x D0 x 32 weight chunk (12 - 24 KB) streams through a two-stage LDS ring, so two or three workgroups
//     share a CU and one's Swish runs under another's MFMAs; inside a wave the second product of chunk c carries the Swish of chunk c + 1 between its MFMAs;
//   * ragged batches: workgroup = (utterance, 128 frames); the rows behind an utterance's last frame (group padding) are written as zeros.
#include "kernels.h"

namespace {

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

constexpr int ROW2 = 80;                                         // bytes per Linear image row in LDS: 32 channels (64 B) + 16

__device__ __forceinline__ uint32_t hi_bits(float x) { return __float_as_uint(x) & 0xFFFF0000u; }     // bf16 by truncation: hi + lo = x to 16 + 8 bits

// Swish of a finished first-product chunk -> bf16 B fragments of the second product, as single-instruction steps (x sigmoid(x) through v_exp_f32 / v_rcp_f32 as
// common.h swishf_): four values at a time, stage-major inside the four; value r = register r of the accumulators <-> k position 8 kh + (r & 7) of k-step r >> 3.
// Every step ends in an empty volatile asm on its result (sxf_chain.hip: pure arithmetic otherwise sinks behind the last MFMA whatever fences stand in the source).
struct SwishState { float x[16], w[16]; uint32_t hh[8]; };
constexpr int SWISH_STEPS = 16 * 5 + 8;
#define SUB3_PIN(v) asm volatile("" : "+v"(v))
__device__ __forceinline__ void swish_step(int idx, const f32x16& h, SwishState& q) {
    const int g = idx / 22, o = idx - 22 * g;
    if (o >= 20) { const int pr = 2 * g + (o - 20); q.hh[pr] = pack_bf2(q.x[2 * pr], q.x[2 * pr + 1]); SUB3_PIN(q.hh[pr]); return; }
    const int stage = o >> 2, r = 4 * g + (o & 3);
    switch (stage) {
        case 0: q.w[r] = h[r] * -1.44269504088896f; SUB3_PIN(q.w[r]); break;
        case 1: q.w[r] = __builtin_amdgcn_exp2f(q.w[r]); SUB3_PIN(q.w[r]); break;
        case 2: q.w[r] = 1.0f + q.w[r]; SUB3_PIN(q.w[r]); break;
        case 3: q.w[r] = __builtin_amdgcn_rcpf(q.w[r]); SUB3_PIN(q.w[r]); break;
        default: q.x[r] = h[r] * q.w[r]; SUB3_PIN(q.x[r]); break;
    }
}
__device__ __forceinline__ void swish_pack(const SwishState& q, bf16x8 (&nb)[2]) {
#pragma unroll
    for (int s = 0; s < 2; ++s) nb[s] = as_bf16x8(make_uint4(q.hh[4 * s], q.hh[4 * s + 1], q.hh[4 * s + 2], q.hh[4 * s + 3]));
}

template <int NT2>
struct Sub3Lds {
    static constexpr int DP2 = 32 * NT2, STAGE = DP2 * ROW2;
    static constexpr int PIECES = 4 * DP2;                       // 16-byte pieces of a chunk
    static constexpr int NPC = (PIECES + 255) / 256;
    static constexpr int CW = 2 * 32 * 32;                       // bytes of one channel block's conv taps: hi [32][16] | lo [32][16] bf16
};

template <int NT2>
__global__ __launch_bounds__(256, (NT2 <= 6 ? 2 : 1)) void sublinear3_kernel(const SubLin3Params p) {
    using L = Sub3Lds<NT2>;
    constexpr int DP2 = L::DP2, NPC = L::NPC;
    extern __shared__ __attribute__((aligned(16))) char sm[];
    char* const sCW = sm + 2 * L::STAGE;                         // conv taps of all channel blocks (resident)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, kh = lane >> 5;
    const int b = blockIdx.y, t0 = blockIdx.x * 128;
    const int nrows = p.off ? p.off[b + 1] - p.off[b] : p.To;    // rows this utterance owns in the output (ragged: group padded)
    const int nvalid = p.len ? p.len[b] : p.To;                  // frames that exist
    if (t0 >= nrows) return;                                     // the grid is sized for the longest utterance (whole workgroup leaves)
    const long long row0 = p.off ? (long long)p.off[b] : (long long)b * p.To;
    const int Tv = p.mel_len ? p.mel_len[b] : p.Tm;              // the utterance's own mel frames: the convolution's zero padding starts behind them
    const int t = t0 + 32 * wave + lr;
    const int tc = t < nvalid ? t : nvalid - 1;
    typedef uint32_t u4v __attribute__((ext_vector_type(4)));
    // ---- the Linear ring: this thread's pieces of a chunk (global: contiguous; LDS: padded rows)
    uint32_t loff[NPC];
#pragma unroll
    for (int it = 0; it < NPC; ++it) {
        const int q = tid + 256 * it, n = q >> 2, ch = q & 3;
        loff[it] = (uint32_t)(n * ROW2 + ch * 16);
    }
    const int nchunk = p.Fo * p.ncb;
    const char* wsrc = reinterpret_cast<const char*>(p.wimg) + (size_t)tid * 16;
    constexpr size_t CB = (size_t)L::PIECES * 16;
    u4v wreg[NPC];
    auto fetch = [&](int c) __attribute__((always_inline)) {
        c = c < nchunk ? c : nchunk - 1;                         // past the end: the last chunk again (never published)
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
    // ---- conv taps -> LDS (ncb blocks x 2 KB, contiguous copy)
    for (int q = tid; q < p.ncb * (L::CW / 16); q += 256)
        *reinterpret_cast<u4v*>(sCW + q * 16) = *reinterpret_cast<const u4v*>(reinterpret_cast<const char*>(p.cimg) + (size_t)q * 16);
    // ---- patch rows: mel[f][2 tc - 1 + j], j = 0 .. 2, zero outside [0, F) x [0, Tv): unconditional loads at clamped addresses, masked by select
    const float* mb = p.mel + (size_t)b * p.F * p.Tm;
    int tau[3]; bool tok[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) { const int x = 2 * tc - 1 + j; tok[j] = x >= 0 && x < Tv; tau[j] = tok[j] ? x : 0; }
    auto load_row = [&](int f, float (&dst)[3]) __attribute__((always_inline)) {
        const bool fok = f >= 0 && f < p.F;
        const float* r = mb + (size_t)(fok ? f : 0) * p.Tm;
#pragma unroll
        for (int j = 0; j < 3; ++j) { const float v = r[tau[j]]; dst[j] = fok && tok[j] ? v : 0.f; }
    };
    float pr[3][3], nx[2][3];
    load_row(-1, pr[0]); load_row(0, pr[1]); load_row(1, pr[2]);
    f32x16 oacc[NT2];
#pragma unroll
    for (int tt = 0; tt < NT2; ++tt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[tt][r] = 0.f;
    auto patch_frags = [&](bf16x8& ah, bf16x8& al) __attribute__((always_inline)) {
        // this lane's patch as split B fragments: taps 3 i + j; lane half 0 holds taps 0 .. 7, half 1 tap 8, the constant of the bias column (tap 9) and zeros
        float v[8];
        v[0] = kh ? pr[2][2] : pr[0][0]; v[1] = kh ? 1.0f : pr[0][1]; v[2] = kh ? 0.f : pr[0][2];
        v[3] = kh ? 0.f : pr[1][0]; v[4] = kh ? 0.f : pr[1][1]; v[5] = kh ? 0.f : pr[1][2];
        v[6] = kh ? 0.f : pr[2][0]; v[7] = kh ? 0.f : pr[2][1];
        uint32_t hh[4], ll[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint32_t h0 = hi_bits(v[2 * e]), h1 = hi_bits(v[2 * e + 1]);
            hh[e] = (h0 >> 16) | h1;
            ll[e] = pack_bf2(v[2 * e] - __uint_as_float(h0), v[2 * e + 1] - __uint_as_float(h1));
        }
        ah = as_bf16x8(make_uint4(hh[0], hh[1], hh[2], hh[3])); al = as_bf16x8(make_uint4(ll[0], ll[1], ll[2], ll[3]));
    };
    // H^T = Wc_cb P^T (32 channels x 32 frames): one 16-wide k-step, the three operand-half products on ONE accumulator (three accumulators - no wait between the
    // MFMAs - cost 48 v_accvgpr_write zeros, 48 v_accvgpr_read and 32 adds per chunk: 40 % of the loop's VALU instructions, profiles/r6_94_*)
    auto first_product = [&](int cb, const bf16x8& ah, const bf16x8& al, f32x16& h) __attribute__((always_inline)) {
        const char* cw = sCW + cb * L::CW + lr * 32 + 16 * kh;
        const bf16x8 wh = *reinterpret_cast<const bf16x8*>(cw), wl = *reinterpret_cast<const bf16x8*>(cw + 32 * 32);
        f32x16 z0;
#pragma unroll
        for (int r = 0; r < 16; ++r) z0[r] = 0.f;
        h = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, ah, z0, 0, 0, 0);
        h = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, al, h, 0, 0, 0);
        h = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, ah, h, 0, 0, 0);
    };
    constexpr int GS = NT2 < 4 ? NT2 : 4, NG = (NT2 + GS - 1) / GS, NU = 2 * NG, NM = 2 * NT2, Q = (SWISH_STEPS + NM - 1) / NM, LASTG = NT2 - (NG - 1) * GS;
    static_assert(NM * Q >= SWISH_STEPS, "every Swish step has its MFMA");
    // one pipeline step: Y^T += Wl_c H_c^T (stage c & 1; units = (group of up to four output tiles, k-step)) with the Swish of chunk c + 1 = (., cb1) BETWEEN its
    // MFMAs, the order fixed in the source (a fence per MFMA, its quota of Swish steps behind it); fragment reads run one unit ahead
    auto step = [&](int c, int cb1, const bf16x8& ah, const bf16x8& al, bf16x8 (&hb)[2]) __attribute__((always_inline)) {
        if (c + 1 < nchunk) publish(sm + ((c + 1) & 1) * L::STAGE);      // stage (c + 1) & 1 was read in iteration c - 1: every wave is past the barrier that closed it
        fetch(c + 2);
        __builtin_amdgcn_sched_barrier(0);
        f32x16 h;
        first_product(cb1, ah, al, h);
        const char* w2 = sm + (c & 1) * L::STAGE + lr * ROW2 + 16 * kh;
        bf16x8 vw[NU][GS];
        auto load_unit = [&](int u) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < GS; ++i) {
                const int tt = (u >> 1) * GS + i, s2 = u & 1;
                if (tt < NT2) vw[u][i] = *reinterpret_cast<const bf16x8*>(w2 + 32 * tt * ROW2 + 32 * s2);
            }
        };
        SwishState q;
        __builtin_amdgcn_sched_barrier(0);
        load_unit(0);
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int s2 = u & 1, gq = u >> 1, gsz = gq == NG - 1 ? LASTG : GS;
            const int base = 2 * GS * gq + s2 * gsz;                     // MFMAs before this unit (a function of the loop indices only: every Swish step index below is a constant)
#pragma unroll
            for (int i = 0; i < GS; ++i) {
                const int tt = gq * GS + i;
                if (tt < NT2) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (i == 0 && u + 1 < NU) load_unit(u + 1);
                    oacc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vw[u][i], hb[s2], oacc[tt], 0, 0, 0);
                    const int m0 = (base + i) * Q;
#pragma unroll
                    for (int j = 0; j < Q; ++j) if (m0 + j < SWISH_STEPS) swish_step(m0 + j, h, q);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        swish_pack(q, hb);
        lds_barrier();
    };
    publish(sm);
    fetch(1);
    lds_barrier();
    bf16x8 ah, al, hb[2];
    {   // chunk 0: first product + Swish alone
        load_row(2, nx[0]); load_row(3, nx[1]);
        patch_frags(ah, al);
        f32x16 h;
        first_product(0, ah, al, h);
        SwishState q;
#pragma unroll
        for (int j = 0; j < SWISH_STEPS; ++j) swish_step(j, h, q);
        swish_pack(q, hb);
    }
    int c = 0;
    for (int fo = 0; fo < p.Fo; ++fo) {
        for (int cb = 0; cb + 1 < p.ncb; ++cb, ++c) step(c, cb + 1, ah, al, hb);
        // the last channel block of this f': its partner is the first chunk of f' + 1 - advance the patch (rows requested one f' ahead), request the one after
#pragma unroll
        for (int j = 0; j < 3; ++j) { pr[0][j] = pr[2][j]; pr[1][j] = nx[0][j]; pr[2][j] = nx[1][j]; }
        load_row(2 * fo + 4, nx[0]); load_row(2 * fo + 5, nx[1]);       // behind the image: zeros, never used
        patch_frags(ah, al);
        step(c, 0, ah, al, hb);                                          // behind the last chunk: a product nobody consumes
        ++c;
    }
    // ---- y = Y + bias for the frames that exist, zeros for the group-padding rows; feature of register (tt, r) = 32 tt + 8 (r >> 2) + 4 kh + (r & 3)
    if (t < nrows) {
        float* yr = p.y + (size_t)(row0 + t) * p.ldy;
        const bool live = t < nvalid;
#pragma unroll
        for (int tt = 0; tt < NT2; ++tt)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int f = 32 * tt + 8 * rq + 4 * kh;
                if (f >= p.N) continue;                          // N % 4 == 0: a quad is inside or outside as a whole
                const float4 bz = *reinterpret_cast<const float4*>(p.bias + f);
                float4 o;
                o.x = live ? oacc[tt][4 * rq + 0] + bz.x : 0.f; o.y = live ? oacc[tt][4 * rq + 1] + bz.y : 0.f;
                o.z = live ? oacc[tt][4 * rq + 2] + bz.z : 0.f; o.w = live ? oacc[tt][4 * rq + 3] + bz.w : 0.f;
                *reinterpret_cast<float4*>(yr + f) = o;
            }
    }
}

template <int NT2>
int launch_sub3(const SubLin3Params& p, hipStream_t s) {
    using L = Sub3Lds<NT2>;
    const int lds = 2 * L::STAGE + p.ncb * L::CW;
    if (lds > 160 * 1024) return -2;
    static LdsAttr attr;
    ensure_dynamic_lds(reinterpret_cast<const void*>(&sublinear3_kernel<NT2>), lds, attr);
    hipLaunchKernelGGL((sublinear3_kernel<NT2>), dim3((p.rows_max + 127) / 128, p.B), dim3(256), lds, s, p);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace

// 32-column output tiles of the instance that serves width N (0: not built)
int sublinear3_tiles(int N) {
    const int nt = (N + 31) / 32;
    return N % 4 ? 0 : (nt == 1 ? 1 : (nt <= 4 ? 4 : (nt <= 6 ? 6 : (nt <= 12 ? 12 : 0))));
}

int launch_sublinear3(const SubLin3Params& p, hipStream_t s) {
    if (p.B <= 0 || p.rows_max <= 0) return 0;
    if (!p.mel || !p.cimg || !p.wimg || !p.bias || !p.y || p.N % 4 || p.ldy % 4 || p.ncb <= 0 || p.Fo <= 0 || p.B > 65535) return -2;
    switch (sublinear3_tiles(p.N)) {
        case 1: return launch_sub3<1>(p, s);
        case 4: return launch_sub3<4>(p, s);
        case 6: return launch_sub3<6>(p, s);
        case 12: return launch_sub3<12>(p, s);
    }
    return -2;
}
