// Fused row-local chains of the Conformer block (gfx950).
//
// Between the two operations of a Conformer block that mix frames (attention and the depthwise convolution) everything is
// row-local: for one frame, the attention output projection + residual, the conv-module LayerNorm, pointwise-1 + GLU
// (reference models/attentions.py:716, blocks.py:125-129, modules.py:511-514) — "chain B" — and, across the block boundary,
// pointwise-2 + residual, FFN2 with its pre-norm and half-step residual, the block's final LayerNorm, then the NEXT block's
// FFN1 (pre-norm, half-step residual), the attention pre-norm and the stacked Q/K/V projection (modules.py:519-522,
// 385-392; blocks.py:129-137, 119-126; attentions.py:651-686) — "chain A".  As separate kernels every step re-reads and
// re-writes the fp32 residual row (4D bytes each way) in an HBM burst that no compute overlaps: the s_memtime phase profile
// of the FFN kernel shows the load/store phases, not the MFMA loop, as ~2/3 of a wave's life (profiles/r1_09_*).
//
// Here a wave keeps its 32 residual rows in registers, in the MFMA C layout of the row-stationary scheme (rsgemm.hip):
//     xc[t][r]  <->  row (lane & 31), column 32t + (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
// and walks the whole chain on them:
//   * a residual GEMM accumulates straight into xc (the accumulator is initialised with xc + bias; the FFN's 1/2 is folded
//     into its second weight and bias at pack time, exact in bf16);
//   * LayerNorm needs the partner half-lane only (one xor-32 shuffle per statistic);
//   * registers [8j, 8j+8) of tile t, rounded to bf16, ARE the B fragment of k-step 2t+j of the next GEMM, provided that GEMM's
//     weight has its K index permuted within every group of 16 (position 8h+e <-> column 4h + 8(e>>2) + (e&3)) — done once at
//     pack time for every weight of a chain, so no data ever moves between lanes;
//   * all weights of the chain stream through ONE LDS-DMA ring as uniform chunks (a 64-row slab of a D-wide weight, or 32
//     hidden units' W1 rows + W2 columns of an FFN: both 2*KS KiB and 2*KS wave-DMAs), so the counted-vmcnt protocol of
//     rsgemm.hip carries over unchanged across stage boundaries;
//   * global memory is touched once per tensor, through the coalesced staging of rowstat.h: x in, A (bf16) in, x out, and the
//     bf16 products (Q+u, Q+v, K, V rows, or the GLU output) out.
// HBM bytes per row: chain A 4D (x) + 2D (A) in, 4D (x) + 8D (Q+u, Q+v, K, V) out; chain B 6D in, 4D + 2D out — against
// ~74D for the unfused block.
#include "kernels.h"
#include "rowstat.h"

#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace {

// widest 8-wave chain (k-steps of 16 columns) whose FFN stages run software-pipelined inside the wave (ffn_stage): the second accumulator and the
// refill's pointers need ~24 registers more than the D = 168 shape (12 k-steps, two waves per SIMD, 256 registers) has left - tried with smaller
// fragment batches: 64 - 97 spilled registers, some inside the chunk loop.  Shapes of <= 4 waves (one per SIMD, 512 registers) pipeline at any width.
constexpr int PIPE_MAX_KS = 8;

template <int KS>
struct Geo {
    static_assert(KS % 2 == 0, "two k-steps per 32-column tile");
    static constexpr int NT = KS / 2;               // 32-column tiles of the residual row
    static constexpr int DP = 32 * NT;              // padded width
    static constexpr int P1 = KS * 2;               // 16-byte pieces per weight row
    static constexpr int HALF = CH * P1 * 16;       // one 32-row slab (= one FFN W2 slab): KS KiB
    static constexpr int BUF = 2 * HALF;
};

struct ChainDev {
    ChainParams p;
    FastDiv32 fT, fD;
    int nf[8];            // float offsets of the LDS constant arrays (see chain_const_layout)
    int nfl_kb;           // size of the constant block in KiB (LDS-DMA pieces)
    int ldr, ld2;         // row pitch (elements) shared by every row-shaped weight of the chain / by the FFN second weights
};

// ---- C-layout helpers -----------------------------------------------------------------------------------------------------
template <int NT>
__device__ __forceinline__ void add_cvec(f32x16 (&xc)[NT], const float* sv, int half) {    // xc[t][r] += sv[column]
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v = *reinterpret_cast<const float4*>(sv + 32 * t + 8 * q + 4 * half);
            xc[t][4 * q + 0] += v.x; xc[t][4 * q + 1] += v.y; xc[t][4 * q + 2] += v.z; xc[t][4 * q + 3] += v.w;
        }
}
// two-pass statistics over the D valid columns (pad columns hold exact zeros); eps 1e-6 (reference modules.py:377, 447; blocks.py:97)
template <int NT>
__device__ __forceinline__ void ln_stats(const f32x16 (&xc)[NT], int D, float& mean, float& rstd) {
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; r += 4) sum += (xc[t][r] + xc[t][r + 1]) + (xc[t][r + 2] + xc[t][r + 3]);
    mean = (sum + __shfl_xor(sum, 32)) / (float)D;
    float var = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; r += 4) {
            const float a = xc[t][r] - mean, b = xc[t][r + 1] - mean, c = xc[t][r + 2] - mean, d = xc[t][r + 3] - mean;
            var += (a * a + b * b) + (c * c + d * d);
        }
    var += __shfl_xor(var, 32);
    var -= (float)(32 * NT - D) * mean * mean;           // the zero pad columns contributed (0 - mean)^2 each
    rstd = rsqrtf(fmaxf(var, 0.f) / (float)D + 1e-6f);
    // opaque copy: keeps the compiler from carrying all (x - mean) differences of the variance pass into the normalisation
    // (64-128 extra live registers pushed loop-invariant addresses into scratch, and scratch reloads share the vmcnt FIFO with
    // the weight DMAs: every reload in a chunk loop waited for the whole prefetch queue)
    asm volatile("" : "+v"(mean));
}
// xf = bf16((xc - mean) * rstd) as the K-permuted B fragments of the next GEMM.  The LayerNorm's gamma / beta are folded into that
// GEMM at pack time (W diag(gamma), b + W beta), so the pre-norms cost no loads; pad columns become -mean*rstd but meet zero
// weight columns
template <int KS>
__device__ __forceinline__ void norm_frags(const f32x16 (&xc)[KS / 2], float mean, float rstd, bf16x8 (&xf)[KS]) {
    const float nm = -mean * rstd;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int r = 8 * (s & 1);
        xf[s] = as_bf16x8(make_uint4(pack_bf2(fmaf(xc[s >> 1][r + 0], rstd, nm), fmaf(xc[s >> 1][r + 1], rstd, nm)),
                                     pack_bf2(fmaf(xc[s >> 1][r + 2], rstd, nm), fmaf(xc[s >> 1][r + 3], rstd, nm)),
                                     pack_bf2(fmaf(xc[s >> 1][r + 4], rstd, nm), fmaf(xc[s >> 1][r + 5], rstd, nm)),
                                     pack_bf2(fmaf(xc[s >> 1][r + 6], rstd, nm), fmaf(xc[s >> 1][r + 7], rstd, nm))));
    }
}
template <int NT>
__device__ __forceinline__ void ln_inplace(f32x16 (&xc)[NT], float mean, float rstd, const float* sg, const float* sb, int half) {
    int ofs = 4 * half;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 g = *reinterpret_cast<const float4*>(sg + ofs + 32 * t + 8 * q), b = *reinterpret_cast<const float4*>(sb + ofs + 32 * t + 8 * q);
            xc[t][4 * q + 0] = (xc[t][4 * q + 0] - mean) * rstd * g.x + b.x;
            xc[t][4 * q + 1] = (xc[t][4 * q + 1] - mean) * rstd * g.y + b.y;
            xc[t][4 * q + 2] = (xc[t][4 * q + 2] - mean) * rstd * g.z + b.z;
            xc[t][4 * q + 3] = (xc[t][4 * q + 3] - mean) * rstd * g.w + b.w;
        }
        // the next tile's gamma / beta addresses depend on this tile's last result: keeps the compiler from hoisting every read
        // of the row to the top (2 x 16 x NT live registers on top of the row itself)
        asm volatile("" : "+v"(ofs) : "v"(xc[t][15]));
    }
}

// ---- global <-> registers through the staging region ----------------------------------------------------------------------
// one staged 64-column window of residual rows -> xc tiles 2W, 2W+1 (pad columns zeroed)
template <int NT, int W, int OFF, int N>
__device__ __forceinline__ void take_x(char* stg, int lane, int D, const u32x4 (&v)[N], f32x16 (&xc)[NT]) {
    const int lr = lane & 31, half = lane >> 5;
    wave_sync();
    stage_put<OFF>(stg, lane, v);
    wave_sync();
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        const int t = 2 * W + tt;
        if (t < NT) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int col = 32 * t + 8 * q + 4 * half;
                float4 x4 = *reinterpret_cast<const float4*>(stg + lr * STG_ROW + (tt * 32 + q * 8 + half * 4) * 4);
                if (col >= D) x4 = make_float4(0.f, 0.f, 0.f, 0.f);      // D % 4 == 0: a piece is valid or not as a whole
                xc[t][4 * q + 0] = x4.x; xc[t][4 * q + 1] = x4.y; xc[t][4 * q + 2] = x4.z; xc[t][4 * q + 3] = x4.w;
            }
        }
    }
}
// residual rows -> xc, windows W0, W0+1, ... (pairs fetched together)
template <int NT, int W0>
__device__ __forceinline__ void load_x(const char* xb, size_t pitch, int D, int m_base, int M, char* stg, int lane, f32x16 (&xc)[NT]) {
    constexpr int NWIN = (NT + 1) / 2;
    if constexpr (W0 < NWIN) {
        u32x4 v[16] = {};
        stage_load<0>(xb, pitch, D * 4, m_base, M, 256 * W0, lane, v);
        if constexpr (W0 + 1 < NWIN) stage_load<8>(xb, pitch, D * 4, m_base, M, 256 * (W0 + 1), lane, v);
        take_x<NT, W0, 0>(stg, lane, D, v, xc);
        if constexpr (W0 + 1 < NWIN) take_x<NT, W0 + 1, 8>(stg, lane, D, v, xc);
        load_x<NT, W0 + 2>(xb, pitch, D, m_base, M, stg, lane, xc);
    }
}
template <int NT, int W>
__device__ __forceinline__ void store_x(char* yb, size_t pitch, int D, int m_base, int M, char* stg, int lane, const f32x16 (&xc)[NT]) {
    constexpr int NWIN = (NT + 1) / 2;
    if constexpr (W < NWIN) {
        const int lr = lane & 31, half = lane >> 5;
        wave_sync();
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int t = 2 * W + tt;
            if (t < NT) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4*>(stg + lr * STG_ROW + (tt * 32 + q * 8 + half * 4) * 4) =
                        make_float4(xc[t][4 * q + 0], xc[t][4 * q + 1], xc[t][4 * q + 2], xc[t][4 * q + 3]);
            }
        }
        wave_sync();
        stage_store(stg, yb, pitch, D * 4, m_base, M, 256 * W, lane);
        store_x<NT, W + 1>(yb, pitch, D, m_base, M, stg, lane, xc);
    }
}
// bf16 rows (natural column order) -> K-permuted B fragments: k-step s of half h = columns 16s + 4h + {0..3} and 16s + 8 + 4h + {0..3}
template <int KS, int W>
__device__ __forceinline__ void take_a(char* stg, int lane, int D, int m_base, int M, const u32x4 (&v)[8], bf16x8 (&xf)[KS]) {
    const int lr = lane & 31, half = lane >> 5;
    wave_sync();
    stage_put<0>(stg, lane, v);
    wave_sync();
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int s = 8 * W + j;
        if (s < KS) {
            const char* src = stg + lr * STG_ROW + (16 * j + 4 * half) * 2;
            uint2 lo = *reinterpret_cast<const uint2*>(src), hi = *reinterpret_cast<const uint2*>(src + 16);
            const int c0 = 16 * s + 4 * half;           // D % 4 == 0: each 4-column piece is valid or not as a whole
            if (c0 >= D || m_base + lr >= M) lo = make_uint2(0u, 0u);
            if (c0 + 8 >= D || m_base + lr >= M) hi = make_uint2(0u, 0u);
            xf[s] = as_bf16x8(make_uint4(lo.x, lo.y, hi.x, hi.y));
        }
    }
}
template <int KS, int W>
__device__ __forceinline__ void load_a(const char* ab, size_t pitch, int row_bytes, int D, int m_base, int M, char* stg, int lane, bf16x8 (&xf)[KS]) {
    constexpr int NWIN = (KS + 7) / 8;                  // 128 columns per window
    if constexpr (W < NWIN) {
        u32x4 v[8] = {};
        stage_load<0>(ab, pitch, row_bytes, m_base, M, 256 * W, lane, v);
        take_a<KS, W>(stg, lane, D, m_base, M, v, xf);
        load_a<KS, W + 1>(ab, pitch, row_bytes, D, m_base, M, stg, lane, xf);
    }
}

// PROF (tuning only, EFFCONF_CHAIN_PHASES=81 | 162 | 163: KS = 8 full chain, KS = 16 head / tail): s_memtime per phase of the FFN stages -
// 0 advance (DMA wait + barrier + refill), 1 GEMM1, 2 Swish, 3 GEMM2, 4 everything else, 5 waves
template <int KS, int NW, int NBUF, int KIND, bool PROF = false>
__global__ __launch_bounds__(NW * 64, (NW == 4 && KS <= 8) ? 2 : 1) void chain_kernel(const ChainDev cd, unsigned long long* prof = nullptr) {
    unsigned long long ph[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, t0 = 0;
    if constexpr (PROF) t0 = __builtin_readcyclecounter();
#define CH_TICK(i) do { if constexpr (PROF) { asm volatile("" ::: "memory"); const unsigned long long t1_ = __builtin_readcyclecounter(); ph[i] += t1_ - t0; t0 = t1_; } } while (0)
    using G = Geo<KS>;
    constexpr int NT = G::NT, P1 = G::P1, HALF = G::HALF, BUF = G::BUF;
    constexpr bool ISB = KIND == CHAIN_B;
    constexpr bool PRE = KIND == CHAIN_A_FULL || KIND == CHAIN_A_TAIL;     // previous block's tail: pw2, FFN2, block norm
    constexpr bool POST = KIND == CHAIN_A_FULL || KIND == CHAIN_A_HEAD;    // this block's head: FFN1, QKV
    static_assert((2 * KS) % NW == 0, "uniform DMA count per wave");
    constexpr int PER = 2 * KS / NW;
    constexpr int NTHR = NW * 64;
    const ChainParams& p = cd.p;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* stg_base = smem + NBUF * BUF;
    float* sf = reinterpret_cast<float*>(stg_base + NW * STG_BYTES);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, half = lane >> 5;
    const int m_base = (blockIdx.x * NW + wave) * 32;
    char* stg = stg_base + wave * STG_BYTES;
    const int D = p.D;

    // ---- chunk schedule: [g0] [ffn0] [ffn1] [g1]
    const int n_g0 = (ISB || PRE) ? (NT + 1) / 2 : 0;
    const int n_f0 = PRE ? p.f[0].Fp / CH : 0;
    const int n_f1 = POST ? p.f[1].Fp / CH : 0;
    const int n_g1 = (ISB || POST) ? p.g1.nchunks : 0;
    // PIPE (ffn_stage): an FFN stage of n hidden chunks is n + 1 ring chunks - ring chunk j holds [W1 rows of hidden chunk j | W2 slab of hidden
    // chunk j - 1]: what iteration j - 1 of the pipelined loop reads (the first GEMM of chunk j beside the Swish of chunk j - 1, then the second
    // GEMM of chunk j - 1), so ONE ring chunk is live per iteration and the plain ring rule (barrier k: chunk k has landed) keeps its full
    // prefetch distance.  Ring chunk 0 carries W1 of hidden chunk 0 (its W2 half is a copy nobody reads; uniform DMA count per chunk).
    constexpr bool PIPE = (NBUF >= 3) && (KS <= PIPE_MAX_KS || NW <= 4);
    const int r_f0 = n_f0 + ((PIPE && n_f0) ? 1 : 0), r_f1 = n_f1 + ((PIPE && n_f1) ? 1 : 0);
    const int e0 = n_g0, e1 = e0 + r_f0, e2 = e1 + r_f1, total = e2 + n_g1;

    // Per-lane byte offsets of this wave's PER DMA instructions, computed ONCE and kept in registers: every weight matrix of a chain
    // has the same row pitch (K = D for all of them; launch_chain checks), so one set serves the row-shaped chunks (off_r) and one
    // the FFN chunks (off_f: W1 rows, then W2 pieces).  The DMA address is (wave-uniform chunk base) + offset (saddr form), so an
    // issue costs no per-lane address arithmetic (a division, a modulo and a 64-bit multiply-add per instruction otherwise).
    uint32_t off_r[PER], off_f[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int i = wave + NW * k;
        off_r[k] = i < KS ? dma_rows32_off<P1>(cd.ldr, i, lane) : dma_rows32_off<P1>(cd.ldr, i - KS, lane) + (uint32_t)(32 * cd.ldr) * 2u;
        off_f[k] = i < KS ? off_r[k] : dma_w2_off(cd.ld2, i - KS, lane);
    }
    // Branch-free (round 4): the refill sits inside the pipelined FFN iteration now, and a branch there would cut the basic block in which the
    // compiler interleaves the next chunk's MFMAs with this chunk's Swish.  Both chunk shapes are PER wave-DMAs into consecutive KiB of the
    // buffer (HALF = KS KiB: the second half starts where wave-instruction KS lands); they differ in the two base pointers and in the lane
    // offsets of the second half only - uniform selects.
    struct Refill { const char* b_lo; const char* b_hi; char* buf; bool ffn; };
    auto issue_prep = [&](int c) __attribute__((always_inline)) -> Refill {
        const bool ffn = c >= e0 && c < e2;
        const bool second = c >= e1;
        const int cf = c - (second ? e1 : e0);
        const int nh = second ? n_f1 : n_f0;
        int c1 = PIPE ? (cf < nh ? cf : nh - 1) : cf, c2 = PIPE ? cf - 1 : cf;      // hidden chunks of the W1 / W2 halves (unused values when !ffn)
        c1 = c1 > 0 ? c1 : 0; c2 = c2 > 0 ? c2 : 0;
        const bf16_t* fw1 = second ? p.f[1].w1 : p.f[0].w1;
        const bf16_t* fw2 = second ? p.f[1].w2 : p.f[0].w2;
        const bool first_g = c < e0;
        const bf16_t* gw = first_g ? p.g0.w : p.g1.w;
        const int cg = first_g ? c : c - e2;
        const char* w1 = reinterpret_cast<const char*>(fw1 + (size_t)c1 * CH * cd.ldr);
        const char* w2 = reinterpret_cast<const char*>(fw2 + c2 * CH);
        const char* w = reinterpret_cast<const char*>(gw + (size_t)(cg > 0 ? cg : 0) * 64 * cd.ldr);
        return Refill{ffn ? w1 : w, ffn ? w2 : w, smem + (c % NBUF) * BUF, ffn};
    };
    static_assert(HALF == 64 * KS * 16, "second half of a buffer = wave-instruction KS");
    // wave-DMA k of a refill (k = 0 .. PER - 1); NC: without the compiler-level memory barrier (placed between MFMAs, see glds16_nc)
    auto issue_one = [&](const Refill& r, int k, auto nc) __attribute__((always_inline)) {
        const int i = wave + NW * k;
        if constexpr (decltype(nc)::value) glds16_nc(i < KS ? r.b_lo : r.b_hi, r.ffn ? off_f[k] : off_r[k], r.buf + 1024 * i);
        else glds16(i < KS ? r.b_lo : r.b_hi, r.ffn ? off_f[k] : off_r[k], r.buf + 1024 * i);
    };
    auto issue = [&](int c) __attribute__((always_inline)) {
        if constexpr (PIPE) {
            const Refill r = issue_prep(c);
#pragma unroll
            for (int k = 0; k < PER; ++k) issue_one(r, k, std::false_type{});
        } else {             // unpipelined shapes: round 3's branching form (the D = 168 shape is register-bound: the select form costs it 3 more spills)
            char* buf = smem + (c % NBUF) * BUF;
            if (c >= e0 && c < e2) {
                const ChainFfn& f = p.f[(c >= e1) ? 1 : 0];
                const int cc = c - ((c >= e1) ? e1 : e0);
                const char* w1 = reinterpret_cast<const char*>(f.w1 + (size_t)cc * CH * cd.ldr);
                const char* w2 = reinterpret_cast<const char*>(f.w2 + cc * CH);
#pragma unroll
                for (int k = 0; k < PER; ++k) {
                    const int i = wave + NW * k;
                    if (i < KS) glds16(w1, off_f[k], buf + 64 * i * 16);
                    else glds16(w2, off_f[k], buf + HALF + 64 * (i - KS) * 16);
                }
            } else {
                const ChainGemm& g = (c < e0) ? p.g0 : p.g1;
                const int cc = (c < e0) ? c : c - e2;
                const char* w = reinterpret_cast<const char*>(g.w + (size_t)cc * 64 * cd.ldr);
#pragma unroll
                for (int k = 0; k < PER; ++k) {
                    const int i = wave + NW * k;
                    if (i < KS) glds16(w, off_r[k], buf + 64 * i * 16);
                    else glds16(w, off_r[k], buf + HALF + 64 * (i - KS) * 16);
                }
            }
        }
    };
    // Ring protocol.  Barrier k (the k-th call of advance()) guarantees: chunks <= k+1 have landed for everybody, and everybody
    // is done with chunk k-1, whose buffer is refilled right after it.  Waiting one chunk AHEAD lets half of the waves run their
    // FFN iterations one phase late (barrier between Swish and the second GEMM instead of before the first, see ffn_stage):
    // the two waves sharing a SIMD then alternate between the MFMA pipe and the VALU instead of competing for the same one.
    // Global stores issued between barriers (st1: since the last advance, st2: the interval before) sit in the same in-order
    // vmcnt FIFO as the DMAs: they are younger than the chunk being waited for, so they are simply allowed to stay outstanding
    // (waiting for them — i.e. for HBM write acknowledgements — cost ~10k cycles per Q/K/V chunk: s_memtime profile).
    int gc = 0, st1 = 0, st2 = 0;                          // next chunk to consume; store instructions in flight
    // Unpipelined 8-wave shapes (D = 168: no registers for the pipelined loop's second accumulator) wait one chunk AHEAD (barrier k: chunk
    // k + 1 has landed too; late waves, below); everything else uses the plain rule (barrier k: chunk k has landed)
    constexpr bool AHEAD = (NW == 8) && !PIPE;
    static_assert(!AHEAD || NBUF >= 3, "wait-ahead protocol needs three ring buffers");
    // unpipelined 8-wave shapes: half of the waves run their FFN iterations one phase late (barrier between Swish and the second GEMM instead
    // of before the first), so that the two waves sharing a SIMD alternate between the matrix pipe and the VALU
    const bool late = !PIPE && (NW == 8) && wave >= NW / 2;         // waves w and w + 4 share a SIMD
    auto advance = [&]() __attribute__((always_inline)) -> const char* {
        constexpr int MAXC = AHEAD ? NBUF - 3 : NBUF - 2;
        int ahead = (AHEAD ? total - 2 : total - 1) - gc;  // chunks issued beyond the one being waited for
        const int rem = ahead;
        ahead = ahead < 0 ? 0 : (ahead > MAXC ? MAXC : ahead);
        if (st1 + st2 == 0) wait_chunks<PER, MAXC>(rem);
        // (round 6, last session) The stores are NOT added to the allowed count any more.  The count used to be [DMAs of the chunks ahead] + [stores since], on the
        // premise that one in-order FIFO retires loads and stores in issue order.  It does not: a store can retire before an older LDS-DMA, the count drops below the
        // limit early and the barrier releases readers of a chunk whose last pieces have not landed - they read the ring slot's PREVIOUS weights.  Seen as a bf16-rounding
        // size perturbation of the last utterance of a row range in ~0.5 % of forwards with several ranges in flight (tools/stream_stress.py, profiles/r6_105 .. r6_108:
        // none in 6000 iterations with the queue drained here); allowing only the DMAs ahead is safe whatever order the stores retire in (loads retire in order)
        else wait_vmcnt_dyn(PER * ahead + (p.count_stores ? st1 + ((AHEAD ? NBUF >= 4 : NBUF >= 3) ? st2 : 0) : 0));
        st2 = st1; st1 = 0;
        wg_barrier();
        const char* buf = smem + (gc % NBUF) * BUF;
        ++gc;
        return buf;
    };
    // The refill of the buffer the barrier has just released - chunk gc + NBUF - 2 after advance() moved gc on: right behind the barrier, before
    // any global store of the iteration (the counted waits assume [refill, stores] in that order).  Exactly one refill per advance(); the
    // pipelined FFN loop issues its own, spread over the iteration (ffn_stage).
    auto refill = [&]() __attribute__((always_inline)) {
        if (gc + NBUF - 2 < total) issue(gc + NBUF - 2);
    };

    // ---- constants -> LDS (zero padded so that pad columns stay exactly zero through every stage)
    float* s_b0 = sf + cd.nf[0];                           // g0 bias [DP]
    float* s_ln = sf + cd.nf[1];                           // LayerNorms: [i][gamma DP | beta DP]
    float* s_f0b1 = sf + cd.nf[2];                         // ffn0 b1 [Fp], b2 [DP]
    float* s_f0b2 = sf + cd.nf[3];
    float* s_f1b1 = sf + cd.nf[4];
    float* s_f1b2 = sf + cd.nf[5];
    float* s_g1b = sf + cd.nf[6];                          // g1 bias [64 * n_g1]
    float* s_uv = sf + cd.nf[7];                           // u [DP] | v [DP]
    constexpr int DP = G::DP;
    // one contiguous, zero padded block prepared at pack time (chain_const_layout): fire-and-forget LDS-DMA, visible after the
    // first barrier below
    for (int i = wave; i < cd.nfl_kb; i += NW) glds16(reinterpret_cast<const char*>(p.consts) + (size_t)i * 1024 + lane * 16, reinterpret_cast<char*>(sf) + i * 1024);

#pragma unroll
    for (int c = 0; c < NBUF - 1; ++c)
        if (c < total) issue(c);

    // ---- this wave's rows
    f32x16 xc[NT];
    bf16x8 xf[KS];
    if constexpr (ISB || PRE) {
        // first windows of x and A fetched together (one HBM latency instead of two), the rest by the generic loaders
        const char* xb = reinterpret_cast<const char*>(p.X);
        const char* ab = reinterpret_cast<const char*>(p.A);
        constexpr int NWX = (NT + 1) / 2;
        u32x4 vx[16] = {}, va[8] = {};
        stage_load<0>(xb, (size_t)p.ldx * 4, D * 4, m_base, p.M, 0, lane, vx);
        if constexpr (NWX > 1) stage_load<8>(xb, (size_t)p.ldx * 4, D * 4, m_base, p.M, 256, lane, vx);
        stage_load<0>(ab, (size_t)p.lda * 2, p.lda * 2, m_base, p.M, 0, lane, va);
        take_x<NT, 0, 0>(stg, lane, D, vx, xc);
        if constexpr (NWX > 1) take_x<NT, 1, 8>(stg, lane, D, vx, xc);
        take_a<KS, 0>(stg, lane, D, m_base, p.M, va, xf);
        load_x<NT, 2>(xb, (size_t)p.ldx * 4, D, m_base, p.M, stg, lane, xc);
        load_a<KS, 1>(ab, (size_t)p.lda * 2, p.lda * 2, D, m_base, p.M, stg, lane, xf);
    } else {
        load_x<NT, 0>(reinterpret_cast<const char*>(p.X), (size_t)p.ldx * 4, D, m_base, p.M, stg, lane, xc);
    }

    // chunk 0 visible to everybody before anyone's first GEMM (a late wave reads a chunk before its own barrier for it)
    if (total >= NBUF) wait_vmcnt<PER * (NBUF - 2)>(); else wait_vmcnt<0>();     // (also covers the constant block, issued first)
    wg_barrier();

    const int q0 = (half + lr) % P1;
    const int w1row = lr * (P1 * 16);
    auto wfrag = [&](const char* slab, int s) __attribute__((always_inline)) {
        int q = q0 + 2 * s;
        q -= q >= P1 ? P1 : 0;
        return *reinterpret_cast<const bf16x8*>(slab + w1row + q * 16);
    };
    const int k2 = (half + (lr >> 2)) & 3;
    const int w2off0 = lr * 64 + k2 * 16, w2off1 = lr * 64 + (k2 ^ 2) * 16;

    CH_TICK(5);
    // ---- stage: x += g0(A)   (alpha = 1: the accumulator starts at x + bias)
    if constexpr (ISB || PRE) {
        add_cvec<NT>(xc, s_b0, half);
#pragma unroll
        for (int c = 0; c < (NT + 1) / 2; ++c) {
            const char* buf = advance();
            refill();
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (2 * c + j < NT) {
                    constexpr int FB = (KS % 8 == 0) ? 8 : ((KS % 4 == 0) ? 4 : KS);
#pragma unroll
                    for (int s0 = 0; s0 < KS; s0 += FB) {
                        bf16x8 wa[FB];
#pragma unroll
                        for (int i = 0; i < FB; ++i) wa[i] = wfrag(buf + j * HALF, s0 + i);
#pragma unroll
                        for (int i = 0; i < FB; ++i) xc[2 * c + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[i], xf[s0 + i], xc[2 * c + j], 0, 0, 0);
                    }
                }
            }
        }
    }

    CH_TICK(6);
    // ---- FFN stage: x += 1/2 FFN(LN(x))  (the 1/2 lives in W2 / b2)
    // Round 4: software-pipelined inside the wave.  Rounds 1 - 3 ran a chunk as GEMM1 -> Swish -> GEMM2, strictly in that order per wave, and
    // relied on the OTHER wave of the SIMD (running one phase late) to use the matrix pipe during a wave's Swish and the VALU during its GEMMs;
    // measured, a wave alone on its SIMD needs 77 us for the D = 168 chain and two per SIMD 90 us - each wave is its own dependency chain
    // (12 accumulating MFMAs -> 16 exp / rcp chains -> 12 MFMAs), the partner fills less than a fifth of its bubbles.  Now iteration c of a wave
    // issues the first GEMM of chunk c + 1 (an independent accumulator; its W1 rows travel in the same ring chunk as chunk c's W2 slab, see the
    // chunk schedule) TOGETHER with the Swish of chunk c - MFMAs and VALU of one wave overlap - then the second GEMM of chunk c.  Same
    // operations on the same operands in the same order per accumulator: bit-identical results.
    // after(slot) is called behind every group of MFMAs (slots 0 .. G1 - 1 in gemm1, G1 .. G1 + G2 - 1 in gemm2): the pipelined loop hangs the
    // wave-DMAs of its refill there
    constexpr int FB1 = (KS % 4 == 0) ? 4 : KS, G1 = KS / FB1, TB2 = (NT % 2 == 0) ? 2 : 1, G2 = NT / TB2;
    auto no_hook = [](int) {};
    auto gemm1 = [&](const char* buf, const float* b1, auto after) __attribute__((always_inline)) -> f32x16 {
        f32x16 h;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v = *reinterpret_cast<const float4*>(b1 + 8 * q);
            h[4 * q + 0] = v.x; h[4 * q + 1] = v.y; h[4 * q + 2] = v.z; h[4 * q + 3] = v.w;
        }
        constexpr int FB = (KS % 4 == 0) ? 4 : KS;
#pragma unroll
        for (int s0 = 0; s0 < KS; s0 += FB) {
            bf16x8 wa[FB];
#pragma unroll
            for (int i = 0; i < FB; ++i) wa[i] = wfrag(buf, s0 + i);
#pragma unroll
            for (int i = 0; i < FB; ++i) h = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[i], xf[s0 + i], h, 0, 0, 0);
            after(s0 / FB);
        }
        return h;
    };
    auto gemm2 = [&](const char* w2, const bf16x8 hf0, const bf16x8 hf1, auto after) __attribute__((always_inline)) {
        constexpr int TB = (NT % 2 == 0) ? 2 : 1;
#pragma unroll
        for (int t0 = 0; t0 < NT; t0 += TB) {
            bf16x8 wb[TB][2];
#pragma unroll
            for (int i = 0; i < TB; ++i) {
                wb[i][0] = *reinterpret_cast<const bf16x8*>(w2 + (t0 + i) * 2048 + w2off0);
                wb[i][1] = *reinterpret_cast<const bf16x8*>(w2 + (t0 + i) * 2048 + w2off1);
            }
#pragma unroll
            for (int i = 0; i < TB; ++i) {
                xc[t0 + i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb[i][0], hf0, xc[t0 + i], 0, 0, 0);
                xc[t0 + i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb[i][1], hf1, xc[t0 + i], 0, 0, 0);
            }
            after(G1 + t0 / TB);
        }
    };
    auto ffn_stage = [&](const float* sb1, const float* sb2, int nchunks) __attribute__((always_inline)) {
        float mean, rstd;
        ln_stats<NT>(xc, D, mean, rstd);
        norm_frags<KS>(xc, mean, rstd, xf);
        add_cvec<NT>(xc, sb2, half);
        if constexpr (PIPE) {
            CH_TICK(4);
            // iterations whose refill certainly exists (chunk gc + NBUF - 2 < total: all of them unless this is the chain's last stage) spread
            // its PER wave-DMAs over the MFMA groups of the iteration; the others issue it behind the barrier (refill())
            int n_sure = total - gc - NBUF;
            n_sure = n_sure < 0 ? 0 : (n_sure < nchunks - 1 ? n_sure : nchunks - 1);
            f32x16 h = gemm1(advance(), sb1 + 4 * half, no_hook);     // ring chunk 0 of the stage: W1 of hidden chunk 0
            refill();
            CH_TICK(1);
            auto body = [&](int c, auto sure) __attribute__((always_inline)) {
                constexpr bool SURE = decltype(sure)::value;
                const char* buf = advance();                 // [W1 of hidden chunk c + 1 | W2 of hidden chunk c]
                if constexpr (!SURE) refill();
                CH_TICK(0);
                const Refill rf = issue_prep(SURE ? gc + NBUF - 2 : 0);
                // The L2 -> LDS path of a CU moves ~30 bytes per cycle (phase profile: 8 wave-DMAs = 8 KiB took a wave ~1080 cycles to issue while
                // the other three waves issued theirs), and an in-order wave issues nothing else meanwhile: issued in one burst behind the
                // barrier, the refill of a D = 240 chunk (32 KiB) cost as much as the chunk's 32 MFMAs.  One DMA behind every group of MFMAs
                // instead: the matrix pipe works off the group while the DMA waits for its turn.
                auto hook = [&](int slot) __attribute__((always_inline)) {
                    if constexpr (SURE) {
#pragma unroll
                        for (int k = 0; k < PER; ++k)
                            if (k * (G1 + G2) / PER == slot) issue_one(rf, k, std::true_type{});
                    }
                };
                const f32x16 hn = gemm1(buf, sb1 + (c + 1) * CH + 4 * half, hook);
                uint32_t w[8];
#pragma unroll
                for (int r = 0; r < 16; r += 2) w[r >> 1] = pack_bf2(swishf_(h[r]), swishf_(h[r + 1]));
                const bf16x8 hf0 = as_bf16x8(make_uint4(w[0], w[1], w[2], w[3])), hf1 = as_bf16x8(make_uint4(w[4], w[5], w[6], w[7]));
                CH_TICK(2);
                gemm2(buf + HALF, hf0, hf1, hook);
                h = hn;
                CH_TICK(3);
            };
            int c = 0;
            for (; c < n_sure; ++c) body(c, std::true_type{});
            for (; c + 1 < nchunks; ++c) body(c, std::false_type{});
            {   // last hidden chunk: nothing to start early
                const char* buf = advance();
                refill();
                CH_TICK(0);
                uint32_t w[8];
#pragma unroll
                for (int r = 0; r < 16; r += 2) w[r >> 1] = pack_bf2(swishf_(h[r]), swishf_(h[r + 1]));
                const bf16x8 hf0 = as_bf16x8(make_uint4(w[0], w[1], w[2], w[3])), hf1 = as_bf16x8(make_uint4(w[4], w[5], w[6], w[7]));
                CH_TICK(2);
                gemm2(buf + HALF, hf0, hf1, no_hook);
                CH_TICK(3);
            }
        } else {
            for (int c = 0; c < nchunks; ++c) {
                CH_TICK(4);
                const char* buf = smem + (gc % NBUF) * BUF;
                if (!late) { (void)advance(); refill(); }
                CH_TICK(0);
                const f32x16 h = gemm1(buf, sb1 + c * CH + 4 * half, no_hook);
                CH_TICK(1);
                uint32_t w[8];
#pragma unroll
                for (int r = 0; r < 16; r += 2) w[r >> 1] = pack_bf2(swishf_(h[r]), swishf_(h[r + 1]));
                const bf16x8 hf0 = as_bf16x8(make_uint4(w[0], w[1], w[2], w[3])), hf1 = as_bf16x8(make_uint4(w[4], w[5], w[6], w[7]));
                CH_TICK(2);
                if (late) { (void)advance(); refill(); }
                CH_TICK(0);
                gemm2(buf + HALF, hf0, hf1, no_hook);
                CH_TICK(3);
            }
        }
    };

    if constexpr (ISB) {
        // ---- conv-module pre-norm, pointwise-1 + GLU -> bf16 rows (modules.py:512-514)
        float mean, rstd;
        ln_stats<NT>(xc, D, mean, rstd);
        norm_frags<KS>(xc, mean, rstd, xf);
        for (int c = 0; c < n_g1; ++c) {
            const char* buf = advance();
            refill();
            f32x16 acc[2];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = *reinterpret_cast<const float4*>(s_g1b + 64 * c + 32 * j + 8 * q + 4 * half);
                    acc[j][4 * q + 0] = v.x; acc[j][4 * q + 1] = v.y; acc[j][4 * q + 2] = v.z; acc[j][4 * q + 3] = v.w;
                }
            constexpr int FB = 2;
#pragma unroll
            for (int s0 = 0; s0 < KS; s0 += FB) {
                bf16x8 wa[2][FB];
#pragma unroll
                for (int i = 0; i < FB; ++i) { wa[0][i] = wfrag(buf, s0 + i); wa[1][i] = wfrag(buf + HALF, s0 + i); }
#pragma unroll
                for (int i = 0; i < FB; ++i) {
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[0][i], xf[s0 + i], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[1][i], xf[s0 + i], acc[1], 0, 0, 0);
                }
            }
            // a * sigmoid(b) for channels 32c + (r&3) + 8(r>>2) + 4*half -> staging (64 B per row) -> coalesced store
            wave_sync();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float o[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = acc[0][4 * q + i] * sigmoidf_(acc[1][4 * q + i]);
                *reinterpret_cast<uint2*>(stg + lr * STG_ROW + (8 * q + 4 * half) * 2) = make_uint2(pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3]));
            }
            wave_sync();
            const int col = 32 * c + 8 * (lane & 3);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = 16 * i + (lane >> 2), m = m_base + row;
                const u32x4 v = *reinterpret_cast<const u32x4*>(stg + row * STG_ROW + 16 * (lane & 3));
                if (m < p.M && col < p.Ng) *reinterpret_cast<u32x4*>(p.glu + (size_t)m * p.ldg + col) = v;
            }
            st1 += 2;
        }
    } else {
        if constexpr (PRE) {
            ffn_stage(s_f0b1, s_f0b2, n_f0);                                        // FFN2 of the previous block
            float mean, rstd;
            ln_stats<NT>(xc, D, mean, rstd);
            ln_inplace<NT>(xc, mean, rstd, s_ln, s_ln + DP, half);                  // block output = LayerNorm(x)  (blocks.py:135)
        }
        if constexpr (POST) {
            ffn_stage(s_f1b1, s_f1b2, n_f1);                                        // FFN1 of this block
            float mean, rstd;
            ln_stats<NT>(xc, D, mean, rstd);
            norm_frags<KS>(xc, mean, rstd, xf);                                     // attention pre-norm
            CH_TICK(4);
            // x is final here (the Q/K/V projection only reads it): store it now so that its registers are free during the last stage
            store_x<NT, 0>(reinterpret_cast<char*>(p.Y), (size_t)p.ldy * 4, D, m_base, p.M, stg, lane, xc);
            st1 += 8 * ((NT + 1) / 2);
            CH_TICK(7);
            // destination row offsets of the 4 rows this lane stores per window instruction: (b, t) -> (b*Tp + t)*D
            size_t qoff[4];
            bool qok[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = m_base + 8 * i + (lane >> 3);
                const int mc = m < p.M ? m : p.M - 1;
                const int b = cd.fT.div(mc), t = mc - b * p.T;
                qoff[i] = ((size_t)b * p.Tp + t) * D;
                qok[i] = m < p.M;
            }
            const int pc8 = 8 * (lane & 7);
            for (int c = 0; c < n_g1; ++c) {
                CH_TICK(8);
                const char* buf = advance();
                CH_TICK(6);                 // PROF, Q/K/V stage: "g0 stage" = wait + barrier, "ffn gemm1" = the refill, "norms+misc" = MFMAs, "qkv stage" = write-out
                refill();
                CH_TICK(1);
                f32x16 acc[2];
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 v = *reinterpret_cast<const float4*>(s_g1b + 64 * c + 32 * j + 8 * q + 4 * half);
                        acc[j][4 * q + 0] = v.x; acc[j][4 * q + 1] = v.y; acc[j][4 * q + 2] = v.z; acc[j][4 * q + 3] = v.w;
                    }
                constexpr int FB = 2;
#pragma unroll
                for (int s0 = 0; s0 < KS; s0 += FB) {
                    bf16x8 wa[2][FB];
#pragma unroll
                    for (int i = 0; i < FB; ++i) { wa[0][i] = wfrag(buf, s0 + i); wa[1][i] = wfrag(buf + HALF, s0 + i); }
#pragma unroll
                    for (int i = 0; i < FB; ++i) {
                        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[0][i], xf[s0 + i], acc[0], 0, 0, 0);
                        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[1][i], xf[s0 + i], acc[1], 0, 0, 0);
                    }
                }
                if constexpr (PROF) { asm volatile("s_nop 0" :: "v"(acc[0][0]), "v"(acc[1][15])); }
                CH_TICK(4);
                // registers r = 8g .. 8g+7 of tile j are the columns 64c + 32j + 16g + 8*half + (0..7) of the stacked [Q | K | V]
                // (row permutation of pack_linear_chunkperm).  Q columns get + u; Q + v is derived in the attention kernel, so it is
                // never written (the Q/K/V write-out is HBM-write bound: one tensor less is a quarter of its bytes).
                {
                    const float* su = s_uv;
                    wave_sync();
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int g = 0; g < 2; ++g) {
                            const int n0 = 64 * c + 32 * j + 16 * g + 8 * half;
                            float4 ua = make_float4(0.f, 0.f, 0.f, 0.f), ub = ua;
                            if (n0 < D) { ua = *reinterpret_cast<const float4*>(su + n0); ub = *reinterpret_cast<const float4*>(su + n0 + 4); }
                            *reinterpret_cast<uint4*>(stg + lr * STG_ROW + (32 * j + 16 * g + 8 * half) * 2) =
                                make_uint4(pack_bf2(acc[j][8 * g + 0] + ua.x, acc[j][8 * g + 1] + ua.y), pack_bf2(acc[j][8 * g + 2] + ua.z, acc[j][8 * g + 3] + ua.w),
                                           pack_bf2(acc[j][8 * g + 4] + ub.x, acc[j][8 * g + 5] + ub.y), pack_bf2(acc[j][8 * g + 6] + ub.z, acc[j][8 * g + 7] + ub.w));
                        }
                    wave_sync();
                    const int n0 = 64 * c + pc8;
                    if ((D & 7) == 0) {                    // a 16-byte piece never straddles the Q | K | V boundaries
                        const int which = cd.fD.div(n0), nn0 = n0 - which * D;
                        bf16_t* dst = which == 0 ? p.qu : (which == 1 ? p.kh : p.vt);
                        const bool colok = n0 < 3 * D;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const u32x4 v = *reinterpret_cast<const u32x4*>(stg + (8 * i + (lane >> 3)) * STG_ROW + 16 * (lane & 7));
                            if (colok && qok[i]) *reinterpret_cast<u32x4*>(dst + qoff[i] + nn0) = v;
                        }
                        st1 += 4;
                    } else {                               // D % 8 == 4 (Medium's D = 180): two 8-byte halves, each inside one tensor
                        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                        const int wa = cd.fD.div(n0), wb = cd.fD.div(n0 + 4);
                        const int na = n0 - wa * D, nb = n0 + 4 - wb * D;
                        bf16_t* da = wa == 0 ? p.qu : (wa == 1 ? p.kh : p.vt);
                        bf16_t* db = wb == 0 ? p.qu : (wb == 1 ? p.kh : p.vt);
                        const bool oka = n0 < 3 * D, okb = n0 + 4 < 3 * D;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const u32x4 v = *reinterpret_cast<const u32x4*>(stg + (8 * i + (lane >> 3)) * STG_ROW + 16 * (lane & 7));
                            if (oka && qok[i]) *reinterpret_cast<u32x2*>(da + qoff[i] + na) = u32x2{v[0], v[1]};
                            if (okb && qok[i]) *reinterpret_cast<u32x2*>(db + qoff[i] + nb) = u32x2{v[2], v[3]};
                        }
                        st1 += 8;
                    }
                }
            }
        }
    }
    // ---- residual rows out
    if constexpr (!POST) store_x<NT, 0>(reinterpret_cast<char*>(p.Y), (size_t)p.ldy * 4, D, m_base, p.M, stg, lane, xc);
    if constexpr (PROF) {
        CH_TICK(8);
        if (lane == 0 && (blockIdx.x & 7) == 0) {      // a 1/8 sample of the workgroups reports (keeps the atomics out of the measurement)
            for (int i = 0; i < 9; ++i) atomicAdd(prof + i, ph[i]);
            atomicAdd(prof + 9, 1ull);
        }
    }
#undef CH_TICK
}

unsigned long long* g_chain_prof = nullptr;
void chain_prof_dump() {
    unsigned long long h[10];
    if (!g_chain_prof || hipMemcpy(h, g_chain_prof, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess || !h[9]) return;
    static const char* names[9] = {"ffn advance", "ffn gemm1", "ffn swish", "ffn gemm2", "norms+misc", "prologue", "g0 stage", "store_x", "qkv stage"};
    unsigned long long tot = 0;
    for (int i = 0; i < 9; ++i) tot += h[i];
    fprintf(stderr, "[chain phases] %s: waves %llu, cycles/wave %.0f   (pipelined shapes: 'ffn swish' = next chunk's first GEMM + Swish + part of the refill; inside the "
                    "Q/K/V stage: 'g0 stage' = wait + barrier, 'ffn gemm1' = refill, 'norms+misc' = MFMAs, 'qkv stage' = write-out)\n",
            getenv("EFFCONF_CHAIN_PHASES"), h[9], (double)tot / h[9]);
    for (int i = 0; i < 9; ++i) fprintf(stderr, "[chain phases]   %-12s %10.0f cyc/wave  %5.1f%%\n", names[i], (double)h[i] / h[9], 100.0 * h[i] / tot);
}

}  // namespace

// LDS float-region layout shared by kernel, launcher and the packer (encoder.hip builds the constant block with it)
// padded row width of the kernel instance launch_chain_kind picks for a width D (KS = 2, 4, 8, 12, 16 k-steps of 16 columns): the
// constant block is laid out in THESE units.  (Round 1 used 32 * ceil(D / 32) here: equal for every CTC config shipped, but 160
// instead of 192 for D = 140 / 144 - beta, v and every later array were read 32 floats off: EfficientConformerTransducerSmall,
// ConformerTransducerSmall.)
int chain_padded_width(int D) {
    const int ks = 2 * ((D + 31) / 32);
    return 16 * (ks <= 2 ? 2 : ks <= 4 ? 4 : ks <= 8 ? 8 : ks <= 12 ? 12 : 16);
}

int chain_const_layout(const ChainParams& p, int kind, int (&nf)[8]) {
    const int DP = chain_padded_width(p.D);
    const bool isb = kind == CHAIN_B, pre = kind == CHAIN_A_FULL || kind == CHAIN_A_TAIL, post = kind == CHAIN_A_FULL || kind == CHAIN_A_HEAD;
    int o = 0;
    nf[0] = o; o += (isb || pre) ? DP : 0;
    nf[1] = o; o += pre ? 2 * DP : 0;
    nf[2] = o; o += pre ? p.f[0].Fp : 0;
    nf[3] = o; o += pre ? DP : 0;
    nf[4] = o; o += post ? p.f[1].Fp : 0;
    nf[5] = o; o += post ? DP : 0;
    nf[6] = o; o += (isb || post) ? 64 * p.g1.nchunks : 0;
    nf[7] = o; o += post ? 2 * DP : 0;
    return (o + 255) / 256 * 256;          // whole KiB: copied by 1 KiB LDS-DMA pieces
}

namespace {

template <int KS, int NW, int NBUF, int KIND>
int launch_chain_t(const ChainParams& p, hipStream_t s) {
    using G = Geo<KS>;
    ChainDev cd;
    cd.p = p;
    cd.fT = FastDiv32(p.T > 0 ? p.T : 1);
    cd.fD = FastDiv32(p.D);
    {   // one row pitch for g0 / g1 / W1 (all have K = D), one for the W2 matrices: the kernel keeps its DMA offsets in registers
        constexpr bool isb = KIND == CHAIN_B, pre = KIND == CHAIN_A_FULL || KIND == CHAIN_A_TAIL, post = KIND == CHAIN_A_FULL || KIND == CHAIN_A_HEAD;
        int ldr = 0, ld2 = 0;
        bool ok = true;
        auto row = [&](int ld) { if (!ldr) ldr = ld; else ok = ok && ld == ldr; };
        auto w2 = [&](int ld) { if (!ld2) ld2 = ld; else ok = ok && ld == ld2; };
        if (isb || pre) row(p.g0.ldw);
        if (isb || post) row(p.g1.ldw);
        if (pre) { row(p.f[0].ldw1); w2(p.f[0].ldw2); }
        if (post) { row(p.f[1].ldw1); w2(p.f[1].ldw2); }
        if (!ok || ldr <= 0) return -6;
        cd.ldr = ldr; cd.ld2 = ld2;
    }
    const int nfl = chain_const_layout(p, KIND, cd.nf);
    cd.nfl_kb = nfl / 256;
    if (!p.consts) return -5;
    const int lds = NBUF * G::BUF + NW * STG_BYTES + nfl * 4;
    if (lds > 160 * 1024) return -4;
    static LdsAttr attr;
    ensure_dynamic_lds(reinterpret_cast<const void*>(&chain_kernel<KS, NW, NBUF, KIND, false>), lds, attr);
    const int rows_per_wg = NW * 32;
    if constexpr ((KS == 8 && KIND == CHAIN_A_FULL) || (KS == 16 && (KIND == CHAIN_A_HEAD || KIND == CHAIN_A_TAIL))) {
        // EFFCONF_CHAIN_PHASES = "<KS><kind>": 81 = the KS = 8 full chain, 162 / 163 = the KS = 16 head / tail (one instance per process)
        static const bool prof = getenv("EFFCONF_CHAIN_PHASES") != nullptr && atoi(getenv("EFFCONF_CHAIN_PHASES")) == KS * 10 + KIND;
        if (prof) {
            if (!g_chain_prof) {
                if (hipMalloc(&g_chain_prof, 128) != hipSuccess || hipMemset(g_chain_prof, 0, 128) != hipSuccess) return -1;
                atexit(chain_prof_dump);
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&chain_kernel<KS, NW, NBUF, KIND, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            }
            hipLaunchKernelGGL((chain_kernel<KS, NW, NBUF, KIND, true>), dim3((p.M + rows_per_wg - 1) / rows_per_wg), dim3(NW * 64), lds, s, cd, g_chain_prof);
            return hipGetLastError() == hipSuccess ? 0 : -1;
        }
    }
    hipLaunchKernelGGL((chain_kernel<KS, NW, NBUF, KIND, false>), dim3((p.M + rows_per_wg - 1) / rows_per_wg), dim3(NW * 64), lds, s, cd, nullptr);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// Workgroup shape.  A chain workgroup streams ALL weights of its chain through its LDS ring, one 32-unit chunk per barrier interval - a fixed
// sequence of ~40 (D = 120) to ~70 dependent chunk steps however few rows the launch has: at B = 4 x 10 s the chain launches took 52 - 90 us each,
// the same as with 100 x the rows, and were 1.7 of the forward's 2.2 ms (profiles/r4_03_small_batch_routes.txt).  The default shapes (8 waves = 256
// rows at D <= 192: two waves per SIMD alternating between the matrix pipe and the VALU, one barrier for eight waves) are the throughput shapes.
// Launches of at most `small_m` rows (option chain_small_m, default 4096: B <= 8 utterances of 10 s at the first stage) run as 2-WAVE workgroups
// (64 rows): four times the workgroups, one wave per SIMD (a chunk step is the wave's own dependency chain, not two waves' issue slots), a
// two-wave barrier.  Every wave computes its 32 rows exactly as in the wide shapes (same instruction sequence per wave, weights in the same
// order): the rows are bit-identical whichever shape ran them (tests/test_gpu_round4.py).  Measured: 90 -> 77 us (D = 168), 55 -> 48 us (D = 120)
// per chain-A launch, forward at B = 4: 2.18 -> 2.08 ms.  The rest IS the dependency chain of one wave (12 accumulating MFMAs, Swish on 16
// values, 12 MFMAs, per chunk, 42 FFN chunks + 8 Q/K/V chunks): going below needs the hidden units of a chunk split across waves, i.e. a
// different summation order than the throughput shapes (not bit-identical) - not built.
template <int KIND>
int launch_chain_kind(const ChainParams& p, hipStream_t s) {
    const int ks = 2 * ((p.D + 31) / 32);
    const bool small = p.small_m > 0 && p.M <= p.small_m;
    if (ks <= 2) return launch_chain_t<2, 4, 4, KIND>(p, s);
    if (ks <= 4) return small ? launch_chain_t<4, 2, 4, KIND>(p, s) : launch_chain_t<4, 8, 4, KIND>(p, s);
    if (ks <= 8) return small ? launch_chain_t<8, 2, 4, KIND>(p, s) : (p.variant == 1 ? launch_chain_t<8, 4, 2, KIND>(p, s) : launch_chain_t<8, 8, 4, KIND>(p, s));
    if (ks <= 12) return small ? launch_chain_t<12, 2, 3, KIND>(p, s) : launch_chain_t<12, 8, 3, KIND>(p, s);
    return launch_chain_t<16, 4, 3, KIND>(p, s);          // D = 240 / 256 already runs one wave per SIMD (4 waves); its 2-wave shape measured slower (84 against 79 us)
}

}  // namespace

// D <= 256: the residual row fits the register file.  The Q/K/V-emitting half stores 16-byte pieces when D % 8 == 0 (they never
// straddle the Q | K | V boundaries) and 8-byte halves otherwise (Medium's D = 180).
bool chain_supported(int D) { return D % 4 == 0 && D >= 16 && D <= 256; }
// At KS = 16 (D = 240 / 256) the chains run at one wave per SIMD; the tail and the head each fit the register file (AGPRs as the
// overflow, no scratch), the combined tail + head kernel does not (see chain_full_supported).
bool chain_head_supported(int D) { return chain_supported(D); }      // D % 4 == 0: the Q/K/V rows leave as 16- or 8-byte pieces
bool chain_tail_supported(int D) { return chain_supported(D); }
// tail + next head in ONE kernel: up to D = 192 the whole state fits the register file; at D = 240 (KS = 16) the combined kernel spills
// (199 us per block against 181 us for the five per-GEMM kernels) while the tail and the head as TWO chain launches do not
bool chain_full_supported(int D, int dmax) { return chain_head_supported(D) && D <= dmax; }

int launch_chain(const ChainParams& p, int kind, hipStream_t s) {
    if (p.M <= 0) return 0;
    if (!chain_supported(p.D)) return -2;
    if (p.pair >= 5 && kind != CHAIN_B && chain3_supported(p.D) && p.D >= p.pair_min_d && !(p.pair_small_max >= 0 && p.M <= p.pair_small_max) &&
        (kind == CHAIN_A_HEAD || p.f[0].w2cm) && (kind == CHAIN_A_TAIL || p.f[1].w2cm)) return launch_chain3(p, kind, s);
    if (p.pair && chain2_supported(p.D) && p.D >= p.pair_min_d && !(p.pair_small_max >= 0 && p.M <= p.pair_small_max)) return launch_chain2(p, kind, s);
    switch (kind) {
        case CHAIN_B: return launch_chain_kind<CHAIN_B>(p, s);
        case CHAIN_A_FULL: return launch_chain_kind<CHAIN_A_FULL>(p, s);
        case CHAIN_A_HEAD: return launch_chain_kind<CHAIN_A_HEAD>(p, s);
        case CHAIN_A_TAIL: return launch_chain_kind<CHAIN_A_TAIL>(p, s);
    }
    return -3;
}
