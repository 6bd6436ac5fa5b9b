// Split-precision front end as ONE kernel (round 6): Conv2d(1, C, 3 x 3, stride 2, pad 1) + folded BatchNorm + Swish + transpose / flatten + Linear
// (reference modules.py:232-249 Conv2dSubsampling, encoders.py:113-116) for the one-layer subsampler of the EfficientConformer configurations.
//
// The per-module path (exact.hip ex_conv2d_flat1_kernel + split.hip sx_gemm_kernel [+ gather_rows_kernel for ragged batches]) writes the (frames, C F') fp32
// activation to HBM and reads it back - 2 x 2.8 GB per Small step at B = 256, 2.6 of the split step's 21 serial milliseconds - and runs both on the rectangle a row
// range's LONGEST utterance spans.  Here the activation never leaves the registers; the structure is sxf_ffn.hip's (a wave keeps 32 frames, TRANSPOSED: frames = MFMA
// columns = lanes), with the 3 x 3 mel patch in the place of the normalised row:
//   * hidden unit (f', c) = Swish(BN(conv)) of output frequency f' and channel c; chunk = 32 channels of one f' (k index c F' + f' of the Linear, encoders.py:114);
//   * first product  H^T = Wc_cb P_f'^T : A operand = the conv taps of 32 channels (9 taps x BN scale, the BN shift + conv bias in tap 9 against a constant 1; 16-wide
//     k-step), B operand = this lane's patch mel[2 f' - 1 .. 2 f' + 1][2 t - 1 .. 2 t + 1] (zero outside the utterance's own image: what it sees when run alone), three
//     MFMAs (h h, h l, l h); the patch rows are carried from f' to f' + 1 (row 2 f' + 1 is the next patch's first) and the next two rows are requested one f' ahead;
//   * Swish on the accumulator registers, which ARE the B fragments of the second product  Y^T += Wl_chunk H^T  (k order of the Linear's image permuted to the
//     accumulator layout at pack time, encoder.hip); Linear images stream through the two-stage LDS ring of sxf_ffn.hip (one LDS-only barrier per chunk);
//   * ragged batches: workgroup = (utterance, 128 frames); rows off[b] + t, the group-padding rows behind an utterance's last frame are written as zeros (what
//     gather_rows_kernel left there).  Arithmetic: same-scale split (sx_common.h split2s): mel at 2^6, activations at 2^8, weights at 2^10, one fp32 accumulator per sum.
#include "kernels.h"
#include "sx_common.h"

namespace {

using namespace sx;

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

constexpr float SP = 64.0f, SA = 256.0f, SW = 1024.0f;           // operand scales: mel patch, hidden activation, weights (encoder.hip packs both images with SW)
constexpr float UNS1 = 1.0f / (SP * SW), UNS2 = 1.0f / (SA * SW);
constexpr int ROW2 = 80;                                         // bytes per Linear image row in LDS: 32 hidden units (64 B) + 16

// Swish (modules.py:240) of a finished first-product chunk -> split B fragments of the second product, as 216 single-instruction steps (the scheme of sxf_chain.hip:
// exp through v_exp_f32 with the rounding of its argument corrected to first order, 1 / x by v_rcp_f32).  Four values at a time, stage-major inside the four;
// value r = register r of the accumulators <-> k position 8 kh + (r & 7) of k-step r >> 3.  Every step ends in an empty volatile asm on its result: the steps are
// pure arithmetic, which instruction selection otherwise sinks behind the last MFMA whatever fences stand in the source.
struct SwishState { float x[16], y[16], w[16]; uint32_t hh[8], ll[8]; };
constexpr int SWISH_STEPS = 16 * 11 + 8;
#define SUB_PIN(v) asm volatile("" : "+v"(v))
__device__ __forceinline__ void swish_step(int idx, const f32x16& h, SwishState& q) {
    const int g = idx / 46, o = idx - 46 * g;
    if (o >= 44) { const int pr = 2 * g + (o - 44); split2s(q.x[2 * pr], q.x[2 * pr + 1], q.hh[pr], q.ll[pr]); SUB_PIN(q.hh[pr]); SUB_PIN(q.ll[pr]); return; }
    const int stage = (o >> 2) + 2, r = 4 * g + (o & 3);
    switch (stage) {
        case 2: q.x[r] = h[r] * UNS1; SUB_PIN(q.x[r]); break;
        case 3: q.y[r] = fminf(-q.x[r], 87.0f); SUB_PIN(q.y[r]); break;
        case 4: q.w[r] = q.y[r] * 1.44269502162933349609375f; SUB_PIN(q.w[r]); break;
        case 5: q.y[r] = fmaf(q.y[r], 1.44269502162933349609375f, -q.w[r]); SUB_PIN(q.y[r]); break;
        case 6: q.w[r] = __builtin_amdgcn_exp2f(q.w[r]); SUB_PIN(q.w[r]); break;
        case 7: q.y[r] = q.y[r] * 0.693147180559945f; SUB_PIN(q.y[r]); break;
        case 8: q.w[r] = fmaf(q.w[r], q.y[r], q.w[r]); SUB_PIN(q.w[r]); break;
        case 9: q.w[r] = 1.0f + q.w[r]; SUB_PIN(q.w[r]); break;
        case 10: q.w[r] = __builtin_amdgcn_rcpf(q.w[r]); SUB_PIN(q.w[r]); break;
        case 11: q.x[r] = q.x[r] * SA; SUB_PIN(q.x[r]); break;
        default: q.x[r] = q.x[r] * q.w[r]; SUB_PIN(q.x[r]); break;
    }
}
__device__ __forceinline__ void swish_pack(const SwishState& q, f16x8 (&nh)[2], f16x8 (&nl)[2]) {
#pragma unroll
    for (int s = 0; s < 2; ++s) { nh[s] = as_f16x8(make_uint4(q.hh[4 * s], q.hh[4 * s + 1], q.hh[4 * s + 2], q.hh[4 * s + 3])); nl[s] = as_f16x8(make_uint4(q.ll[4 * s], q.ll[4 * s + 1], q.ll[4 * s + 2], q.ll[4 * s + 3])); }
}

template <int NT2>
struct SubLds {
    static constexpr int DP2 = 32 * NT2, STAGE = 2 * DP2 * ROW2;
    static constexpr int PIECES = 8 * DP2;                       // 16-byte pieces of a chunk's two planes
    static constexpr int NPC = (PIECES + 255) / 256;
    static constexpr int CW = 2 * 32 * 32;                       // bytes of one channel block's conv taps: hi [32][16] | lo [32][16] fp16
};

template <int NT2>
__global__ __launch_bounds__(256, (NT2 <= 4 ? 2 : 1)) void sxf_sublin_kernel(const SxfSubParams p) {
    using L = SubLds<NT2>;
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
        const int q = tid + 256 * it, pl = q / (4 * DP2), rem = q - pl * 4 * DP2, n = rem >> 2, ch = rem & 3;
        loff[it] = (uint32_t)(pl * DP2 * ROW2 + n * ROW2 + ch * 16);
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
    // ---- software pipeline over the chunks: the second product of chunk c (24 MFMAs at four output tiles, operands: registers + the ring) runs in ONE basic block
    //      with the first product + Swish + operand split of chunk c + 1 (3 MFMAs, ~450 VALU / transcendental instructions) - independent work that the scheduler
    //      interleaves (one matrix instruction, then a slice of the Swish); left in sequence a wave spent 1.8 k cycles on the VALU with the matrix pipe idle and
    //      0.9 k cycles queueing MFMAs with the VALU idle, per chunk
    auto patch_frags = [&](f16x8& ah, f16x8& al) __attribute__((always_inline)) {
        // this lane's patch as split B fragments: taps 3 i + j; lane half 0 holds taps 0 .. 7, half 1 tap 8, the constant of the shift column (tap 9) and zeros
        float v[8];
        v[0] = kh ? pr[2][2] : pr[0][0]; v[1] = kh ? 1.0f : pr[0][1]; v[2] = kh ? 0.f : pr[0][2];
        v[3] = kh ? 0.f : pr[1][0]; v[4] = kh ? 0.f : pr[1][1]; v[5] = kh ? 0.f : pr[1][2];
        v[6] = kh ? 0.f : pr[2][0]; v[7] = kh ? 0.f : pr[2][1];
        uint32_t hh[4], ll[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) split2s(v[2 * e] * SP, v[2 * e + 1] * SP, hh[e], ll[e]);
        ah = as_f16x8(make_uint4(hh[0], hh[1], hh[2], hh[3])); al = as_f16x8(make_uint4(ll[0], ll[1], ll[2], ll[3]));
    };
    // H^T = Wc_cb P^T (32 channels x 32 frames): one 16-wide k-step, the three operand-half products on ONE accumulator at the scale SP SW (three accumulators - no
    // wait between the MFMAs - cost 48 v_accvgpr_write zeros, 48 v_accvgpr_read and 32 adds per chunk, profiles/r6_94_*)
    auto first_product = [&](int cb, const f16x8& ah, const f16x8& al, f32x16& h) __attribute__((always_inline)) {
        const char* cw = sCW + cb * L::CW + lr * 32 + 16 * kh;
        const f16x8 wh = *reinterpret_cast<const f16x8*>(cw), wl = *reinterpret_cast<const f16x8*>(cw + 32 * 32);
        f32x16 z0;
#pragma unroll
        for (int r = 0; r < 16; ++r) z0[r] = 0.f;
        h = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, ah, z0, 0, 0, 0);
        h = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, al, h, 0, 0, 0);
        h = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, ah, h, 0, 0, 0);
    };
    constexpr int GS = NT2 < 4 ? NT2 : 4, NG = (NT2 + GS - 1) / GS, NU = 2 * NG, NM = 6 * NT2, Q = (SWISH_STEPS + NM - 1) / NM, LASTG = NT2 - (NG - 1) * GS;
    static_assert(NM * Q >= SWISH_STEPS, "every Swish step has its MFMA");
    // one pipeline step: Y^T += Wl_c H_c^T (stage c & 1; units = (group of up to four output tiles, k-step), MFMAs kind-major over the tiles: consecutive instructions
    // hit different accumulators) with the Swish of chunk c + 1 = (., cb1) BETWEEN its MFMAs, the order fixed in the source (a fence per MFMA, its quota of Swish steps
    // behind it); fragment reads run one unit ahead
    auto step = [&](int c, int cb1, const f16x8& ah, const f16x8& al, f16x8 (&hbh)[2], f16x8 (&hbl)[2]) __attribute__((always_inline)) {
        if (c + 1 < nchunk) publish(sm + ((c + 1) & 1) * L::STAGE);      // stage (c + 1) & 1 was read in iteration c - 1: every wave is past the barrier that closed it
        fetch(c + 2);
        __builtin_amdgcn_sched_barrier(0);
        f32x16 h;
        first_product(cb1, ah, al, h);
        const char* w2 = sm + (c & 1) * L::STAGE + lr * ROW2 + 16 * kh;
        f16x8 vh[NU][GS], vl[NU][GS];
        auto load_unit = [&](int u) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < GS; ++i) {
                const int tt = (u >> 1) * GS + i, s2 = u & 1;
                if (tt < NT2) { vh[u][i] = *reinterpret_cast<const f16x8*>(w2 + 32 * tt * ROW2 + 32 * s2); vl[u][i] = *reinterpret_cast<const f16x8*>(w2 + DP2 * ROW2 + 32 * tt * ROW2 + 32 * s2); }
            }
        };
        SwishState q;
        __builtin_amdgcn_sched_barrier(0);
        load_unit(0);
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int s2 = u & 1, gq = u >> 1, gsz = gq == NG - 1 ? LASTG : GS;
            const int base = 3 * (2 * GS * gq + s2 * gsz);              // MFMAs before this unit (a function of the loop indices only: every Swish step index below is a constant)
#pragma unroll
            for (int kind = 0; kind < 3; ++kind)
#pragma unroll
                for (int i = 0; i < GS; ++i) {
                    const int tt = gq * GS + i;
                    if (tt < NT2) {
                        __builtin_amdgcn_sched_barrier(0);
                        if (kind == 0 && i == 0 && u + 1 < NU) load_unit(u + 1);
                        oacc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kind == 2 ? vl[u][i] : vh[u][i], kind == 1 ? hbl[s2] : hbh[s2], oacc[tt], 0, 0, 0);
                        const int m0 = (base + kind * gsz + i) * Q;
#pragma unroll
                        for (int j = 0; j < Q; ++j) if (m0 + j < SWISH_STEPS) swish_step(m0 + j, h, q);
                    }
                }
        }
        __builtin_amdgcn_sched_barrier(0);
        swish_pack(q, hbh, hbl);
        lds_barrier();
    };
    publish(sm);
    fetch(1);
    lds_barrier();
    f16x8 ah, al, hbh[2], hbl[2];
    {   // chunk 0: first product + Swish alone
        load_row(2, nx[0]); load_row(3, nx[1]);
        patch_frags(ah, al);
        f32x16 h;
        first_product(0, ah, al, h);
        SwishState q;
#pragma unroll
        for (int j = 0; j < SWISH_STEPS; ++j) swish_step(j, h, q);
        swish_pack(q, hbh, hbl);
    }
    int c = 0;
    for (int fo = 0; fo < p.Fo; ++fo) {
        for (int cb = 0; cb + 1 < p.ncb; ++cb, ++c) step(c, cb + 1, ah, al, hbh, hbl);
        // the last channel block of this f': its partner is the first chunk of f' + 1 - advance the patch (rows requested one f' ahead), request the one after
#pragma unroll
        for (int j = 0; j < 3; ++j) { pr[0][j] = pr[2][j]; pr[1][j] = nx[0][j]; pr[2][j] = nx[1][j]; }
        load_row(2 * fo + 4, nx[0]); load_row(2 * fo + 5, nx[1]);       // behind the image: zeros, never used
        patch_frags(ah, al);
        step(c, 0, ah, al, hbh, hbl);                                    // behind the last chunk: a product nobody consumes
        ++c;
    }
    // ---- y = Y + bias for the frames that exist, zeros for the group-padding rows; feature of register (tt, r) = 32 tt + 8 (r >> 2) + 4 kh + (r & 3)
    if (t < nrows) {
        float* yr = p.y + (size_t)(row0 + t) * p.N;
        const bool live = t < nvalid;
#pragma unroll
        for (int tt = 0; tt < NT2; ++tt)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int f = 32 * tt + 8 * rq + 4 * kh;
                if (f >= p.N) continue;                          // N % 4 == 0: a quad is inside or outside as a whole
                const float4 bz = *reinterpret_cast<const float4*>(p.bias + f);
                float4 o;
                o.x = live ? fmaf(oacc[tt][4 * rq + 0], UNS2, bz.x) : 0.f; o.y = live ? fmaf(oacc[tt][4 * rq + 1], UNS2, bz.y) : 0.f;
                o.z = live ? fmaf(oacc[tt][4 * rq + 2], UNS2, bz.z) : 0.f; o.w = live ? fmaf(oacc[tt][4 * rq + 3], UNS2, bz.w) : 0.f;
                *reinterpret_cast<float4*>(yr + f) = o;
            }
    }
}

template <int NT2>
int launch_sub(const SxfSubParams& p, hipStream_t s) {
    using L = SubLds<NT2>;
    const int lds = 2 * L::STAGE + p.ncb * L::CW;
    if (lds > 160 * 1024) return -2;
    static LdsAttr attr;
    ensure_dynamic_lds(reinterpret_cast<const void*>(&sxf_sublin_kernel<NT2>), lds, attr);
    hipLaunchKernelGGL((sxf_sublin_kernel<NT2>), dim3((p.rows_max + 127) / 128, p.B), dim3(256), lds, s, p);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace

// 32-row output tiles of the built instances (0: width not built - the caller keeps the per-module kernels)
int sxf_sublin_tiles(int N) {
    const int nt = (N + 31) / 32;
    return N % 4 ? 0 : (nt == 1 ? 1 : (nt <= 4 ? 4 : (nt <= 6 ? 6 : (nt <= 12 ? 12 : 0))));
}

int launch_sxf_sublin(const SxfSubParams& p, hipStream_t s) {
    if (p.B <= 0 || p.rows_max <= 0) return 0;
    if (!p.mel || !p.cimg || !p.wimg || !p.bias || !p.y || p.N % 4 || p.ncb <= 0 || p.Fo <= 0 || p.B > 65535) return -2;
    switch (sxf_sublin_tiles(p.N)) {
        case 1: return launch_sub<1>(p, s);
        case 4: return launch_sub<4>(p, s);
        case 6: return launch_sub<6>(p, s);
        case 12: return launch_sub<12>(p, s);
    }
    return -2;
}
