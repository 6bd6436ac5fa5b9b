// Row-stationary MFMA kernels for the Conformer block (gfx950): activations stay in registers,
// weights stream through an LDS-DMA ring.
//
// The block's GEMMs are tall and skinny (M = B*T rows in the 10^4..10^5 range, K = D in 120..384,
// N in D..4D), so the classic C-tile GEMM re-reads A once per N tile and spends most of its time in
// prologue/epilogue.  Here the MFMA operands are swapped (C^T = W X^T):
//
//   * a wave owns RT*32 activation rows; lane l holds row (l & 31) of each 32-row tile, split in two
//     halves across lanes l and l+32 -> X is the MFMA *B* operand and lives in registers for the whole kernel
//     (v_mfma_f32_32x32x16_bf16: B[k][n]: n = lane & 31, k = 8*(lane>>5) + e);
//   * weight rows are the MFMA *A* operand, streamed in 32-row chunks by LDS-DMA
//     (global_load_lds_dwordx4: no staging registers) into a 3/4-deep ring shared by all waves of the
//     workgroup, with counted s_waitcnt vmcnt(N) and one raw s_barrier per chunk, so two chunks are always in
//     flight while one is computed (a one-deep register-staged prefetch was L2-latency bound:
//     profiles/r1_01_*).  The LDS image is lane-linear, so the bank-conflict swizzle is applied to the
//     SOURCE address (piece rotation by row) and undone on the read;
//   * the result tile C^T[n][m] leaves every lane with 16 output columns of ITS OWN row m, so
//     bias/activation/residual are lane-local, and a second GEMM can consume the first one's
//     accumulators directly as its B operand: accumulator registers [8s, 8s+8) of a 32-row result tile are,
//     after bf16 packing, exactly the B fragment of k-step s — provided the second weight's K index is
//     permuted within each group of 16 at pack time (position 8h+e <-> 4h + 8(e>>2) + (e&3)).
//
// ffn_fused_kernel: y = x + alpha * (Swish(a W1^T + b1) W2^T + b2)   (a = LayerNorm(x) as bf16)
//   replaces FeedForwardModule.forward + the half-step residual (reference models/modules.py:385-392,
//   blocks.py:122, 132); the 4D-wide hidden activation never exists in memory.
// rs_gemm_kernel: one GEMM with the QKV-scatter / GLU / residual epilogues (see below).
#include "kernels.h"
#include "rowstat.h"
#include <cstdio>
#include <cstdlib>

namespace {

// ---- LayerNorm in the prologue: lane (row m, half) owns columns 16s + 8*half + e of its row; row statistics need the
// partner half-lane only (one xor-32 shuffle).  eps 1e-6, two-pass fp32 (reference modules.py:377, 447, 500).
template <int KS>
__device__ __forceinline__ void layernorm_regs(float4 (&ra)[KS], float4 (&rb)[KS], int D, bool live, int half, const float* sg, const float* sb,
                                               bf16x8 (&xf)[KS]);

template <int KS>
__device__ __forceinline__ void load_row_layernorm(const float* xrow, int D, bool live, int half, const float* sg, const float* sb,
                                                   bf16x8 (&xf)[KS]) {
    float4 ra[KS], rb[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {                      // unconditional clamped loads
        const int c0 = s * 16 + half * 8;
        ra[s] = *reinterpret_cast<const float4*>(xrow + (c0 < D - 4 ? c0 : D - 4));
        rb[s] = *reinterpret_cast<const float4*>(xrow + (c0 + 4 < D - 4 ? c0 + 4 : D - 4));
    }
    layernorm_regs<KS>(ra, rb, D, live, half, sg, sb, xf);
}

template <int KS>
__device__ __forceinline__ void layernorm_regs(float4 (&ra)[KS], float4 (&rb)[KS], int D, bool live, int half, const float* sg, const float* sb,
                                               bf16x8 (&xf)[KS]) {
    float sum = 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int c0 = s * 16 + half * 8;
        if (c0 >= D) ra[s] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c0 + 4 >= D) rb[s] = make_float4(0.f, 0.f, 0.f, 0.f);
        sum += (ra[s].x + ra[s].y) + (ra[s].z + ra[s].w) + (rb[s].x + rb[s].y) + (rb[s].z + rb[s].w);
    }
    const float mean = (sum + __shfl_xor(sum, 32)) / (float)D;
    float var = 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int c0 = s * 16 + half * 8;
        // branch-free: a padded chunk holds zeros and subtracts 0 instead of the mean (one select per chunk, no exec-masked blocks)
        const float ma = c0 < D ? mean : 0.f, mb = c0 + 4 < D ? mean : 0.f;
        { const float a = ra[s].x - ma, b = ra[s].y - ma, c = ra[s].z - ma, d = ra[s].w - ma; var += (a * a + b * b) + (c * c + d * d); }
        { const float a = rb[s].x - mb, b = rb[s].y - mb, c = rb[s].z - mb, d = rb[s].w - mb; var += (a * a + b * b) + (c * c + d * d); }
    }
    const float rstd = rsqrtf((var + __shfl_xor(var, 32)) / (float)D + 1e-6f);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int c0 = s * 16 + half * 8;               // gamma / beta live in LDS, zero padded to KS*16 columns
        const float4 g0 = *reinterpret_cast<const float4*>(sg + c0), g1 = *reinterpret_cast<const float4*>(sg + c0 + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(sb + c0), b1 = *reinterpret_cast<const float4*>(sb + c0 + 4);
        // No column / row guards: columns >= D have x = 0 (zeroed above) and gamma = beta = 0, so they come out as 0; rows >= M
        // were fetched from a clamped (valid) row and are never stored.  With per-element selects the compiler sinks the LDS
        // reads of gamma / beta under 2*KS exec-masked branches, each waiting for its own LDS round trip.
        uint4 w;
        w.x = pack_bf2((ra[s].x - mean) * rstd * g0.x + b0.x, (ra[s].y - mean) * rstd * g0.y + b0.y);
        w.y = pack_bf2((ra[s].z - mean) * rstd * g0.z + b0.z, (ra[s].w - mean) * rstd * g0.w + b0.w);
        w.z = pack_bf2((rb[s].x - mean) * rstd * g1.x + b1.x, (rb[s].y - mean) * rstd * g1.y + b1.y);
        w.w = pack_bf2((rb[s].z - mean) * rstd * g1.z + b1.z, (rb[s].w - mean) * rstd * g1.w + b1.w);
        xf[s] = as_bf16x8(w);
        (void)live;
    }
}


// y = x + alpha * (acc + b2) on a pair of 64-column windows: x in through the staging region, updated in place, y out
template <int KS, int NT2, int W>
__device__ __forceinline__ void ffn_epilogue_window(char* stg, int lr, int half, const float* sb2, float alpha, const f32x16 (&acc)[NT2]) {
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        if (2 * W + tt < NT2) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4* cell = reinterpret_cast<float4*>(stg + lr * STG_ROW + (tt * 32 + q * 8 + half * 4) * 4);
                const float4 bv = *reinterpret_cast<const float4*>(sb2 + (2 * W + tt) * 32 + q * 8 + half * 4);
                const float4 x4 = *cell;
                float4 o;
                o.x = x4.x + alpha * (acc[2 * W + tt][q * 4 + 0] + bv.x);
                o.y = x4.y + alpha * (acc[2 * W + tt][q * 4 + 1] + bv.y);
                o.z = x4.z + alpha * (acc[2 * W + tt][q * 4 + 2] + bv.z);
                o.w = x4.w + alpha * (acc[2 * W + tt][q * 4 + 3] + bv.w);
                *cell = o;
            }
        }
    }
}
template <int KS, int NT2, int W0>
__device__ __forceinline__ void ffn_epilogue_pair(const char* xb, char* yb, size_t xpitch, size_t ypitch, int row_bytes, int m_base, int M,
                                                  char* stg, int lane, const float* sb2, float alpha, const f32x16 (&acc)[NT2]) {
    constexpr int NWIN = (NT2 + 1) / 2;
    if constexpr (W0 < NWIN) {
        u32x4 v[16] = {};
        stage_load<0>(xb, xpitch, row_bytes, m_base, M, 256 * W0, lane, v);
        if constexpr (W0 + 1 < NWIN) stage_load<8>(xb, xpitch, row_bytes, m_base, M, 256 * (W0 + 1), lane, v);
        wave_sync();
        stage_put<0>(stg, lane, v);
        wave_sync();
        ffn_epilogue_window<KS, NT2, W0>(stg, lane & 31, lane >> 5, sb2, alpha, acc);
        wave_sync();
        stage_store(stg, yb, ypitch, row_bytes, m_base, M, 256 * W0, lane);
        if constexpr (W0 + 1 < NWIN) {
            wave_sync();
            stage_put<8>(stg, lane, v);
            wave_sync();
            ffn_epilogue_window<KS, NT2, W0 + 1>(stg, lane & 31, lane >> 5, sb2, alpha, acc);
            wave_sync();
            stage_store(stg, yb, ypitch, row_bytes, m_base, M, 256 * (W0 + 1), lane);
        }
        ffn_epilogue_pair<KS, NT2, W0 + 2>(xb, yb, xpitch, ypitch, row_bytes, m_base, M, stg, lane, sb2, alpha, acc);
    }
}

template <int KS, int NT2, int NBUF>
struct FfnSmem {
    static constexpr int P1 = KS * 2;                      // 16-byte pieces per W1 row
    static constexpr int W1_BYTES = CH * P1 * 16;
    static constexpr int W2_BYTES = NT2 * 32 * 64;
    static constexpr int BUF = W1_BYTES + W2_BYTES;
    static constexpr int RING = NBUF * BUF;
};
// Staging region of the coalesced prologue / epilogue: its own LDS when that fits next to the ring, otherwise it aliases the last
// ring buffer (free until the first in-loop DMA issue, which follows a workgroup barrier; all ring buffers are free after the loop)
template <int KS, int NT2, int NW, int NBUF>
struct FfnStage {
    using SM = FfnSmem<KS, NT2, NBUF>;
    static constexpr int BYTES = NW * STG_BYTES;
    static constexpr int MISC = (KS * 16 * 4 /*Fp*/ + NT2 * 32 + KS * 32) * 4;
    static constexpr bool ALIAS = SM::RING + BYTES + MISC > 160 * 1024;
    static_assert(!ALIAS || SM::BUF >= BYTES, "staging region must fit one ring buffer");
};

// KS: k16-steps over D (D <= 16*KS), NT2 = KS/2: 32-col tiles over D, RT: 32-row tiles per wave, NW: waves, NBUF: ring depth
// PROF: per-phase s_memtime accounting (tuning only, EFFCONF_FFN_PHASES=1): 0 prologue, 1 chunk wait + barrier, 2 DMA issue,
// 3 GEMM1, 4 bias + Swish + pack, 5 GEMM2, 6 epilogue, 7 waves
template <int KS, int NT2, int RT, int NW, int NBUF, bool PROF = false>
__global__ __launch_bounds__(NW * 64) void ffn_fused_kernel(const FfnParams p, unsigned long long* prof = nullptr) {
    unsigned long long ph[7] = {0, 0, 0, 0, 0, 0, 0}, t0 = 0;
    if constexpr (PROF) t0 = __builtin_readcyclecounter();
#define FFN_TICK(i) do { if constexpr (PROF) { asm volatile("" ::: "memory"); const unsigned long long t1_ = __builtin_readcyclecounter(); ph[i] += t1_ - t0; t0 = t1_; } } while (0)
    using SM = FfnSmem<KS, NT2, NBUF>;
    static_assert(NT2 * 2 == KS && (2 * KS) % NW == 0, "uniform DMA count per wave");
    constexpr int PER = 2 * KS / NW;                       // DMA instructions per wave per chunk
    constexpr int P1 = SM::P1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sb1 = reinterpret_cast<float*>(smem + SM::RING);   // whole first bias in LDS
    float* sb2 = sb1 + p.Fp;                                  // second bias, NT2*32 entries
    float* sgam = sb2 + NT2 * 32;                             // LayerNorm gamma | beta (KS*16 each, zero padded)
    float* sbet = sgam + KS * 16;
    constexpr int NTHR = NW * 64;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, half = lane >> 5;
    const int m_base = (blockIdx.x * NW + wave) * (RT * 32);
    const int nchunks = p.Fp / CH;
    using ST = FfnStage<KS, NT2, NW, NBUF>;
    static_assert(RT == 1, "one 32-row tile per wave");
    char* stg = (ST::ALIAS ? smem + (NBUF - 1) * SM::BUF : reinterpret_cast<char*>(sbet + KS * 16)) + wave * STG_BYTES;

    // per-lane byte offsets of this wave's PER DMA instructions, computed once (the address arithmetic - a division, a modulo and
    // a 64-bit multiply-add per instruction - was ~80 cycles per DMA: the "dma_issue" phase of EFFCONF_FFN_PHASES); per chunk only
    // the wave-uniform base moves
    uint32_t doff[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int i = wave + NW * k;
        doff[k] = i < KS ? dma_rows32_off<P1>(p.ldw1, i, lane) : dma_w2_off(p.ldw2, i - KS, lane);
    }
    auto issue = [&](int c) {
        char* buf = smem + (c % NBUF) * SM::BUF;
        const char* w1 = reinterpret_cast<const char*>(p.W1 + (size_t)c * CH * p.ldw1);
        const char* w2 = reinterpret_cast<const char*>(p.W2 + c * CH);
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = wave + NW * k;
            if (i < KS) glds16(w1, doff[k], buf + 64 * i * 16);
            else glds16(w2, doff[k], buf + SM::W1_BYTES + 64 * (i - KS) * 16);
        }
    };

    // ---- bias -> LDS, this lane's activation row fragments (B operand of GEMM1), then start the weight stream
    for (int i = tid; i < p.Fp; i += NTHR) sb1[i] = p.b1[i];
    for (int i = tid; i < NT2 * 32; i += NTHR) sb2[i] = i < p.D ? p.b2[i] : 0.f;
    const bool fuse_ln = p.ln_g != nullptr;
    if (fuse_ln) {
        for (int i = tid; i < KS * 16; i += NTHR) { sgam[i] = i < p.D ? p.ln_g[i] : 0.f; sbet[i] = i < p.D ? p.ln_b[i] : 0.f; }
        __syncthreads();
    }
#pragma unroll
    for (int c = 0; c < NBUF - 1; ++c)                     // start the weight stream, then fetch this wave's rows
        if (c < nchunks) issue(c);
    bf16x8 xf[RT][KS];
    {
        const bool live = m_base + lr < p.M;
        if (fuse_ln) {
            float4 ra[KS], rb[KS];
            fetch_row_f32<KS>(p.X, p.ldx, p.D, m_base, p.M, stg, lane, ra, rb);
            layernorm_regs<KS>(ra, rb, p.D, live, half, sgam, sbet, xf[0]);
        } else {
            uint4 raw[KS];
            fetch_row_bf16<KS>(p.A, p.lda, m_base, p.M, stg, lane, raw);
#pragma unroll
            for (int s = 0; s < KS; ++s) xf[0][s] = as_bf16x8(mask_chunk(raw[s], live ? p.D - (s * 16 + half * 8) : 0));
        }
    }

    f32x16 acc[RT][NT2];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int t = 0; t < NT2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[rt][t][r] = 0.f;
    FFN_TICK(0);

    // per-lane read offsets: W1 piece (2s + half) of row lr; W2 pieces of rows 32t + lr
    const int q0 = (half + lr) % P1;
    const int k2 = (half + (lr >> 2)) & 3;
    const int w2off0 = lr * 64 + k2 * 16, w2off1 = lr * 64 + (k2 ^ 2) * 16;

    for (int c = 0; c < nchunks; ++c) {
        wait_chunks<PER, NBUF - 2>(nchunks - 1 - c);       // chunk c has landed (this wave's pieces)
        wg_barrier();                                      // ... and everybody's; everybody is done with chunk c-1
        FFN_TICK(1);
        if (c + NBUF - 1 < nchunks) issue(c + NBUF - 1);   // refill the buffer chunk c-1 used.  (Spreading these DMA instructions over
                                                           // GEMM1 was slower: a wave's MFMAs queue behind a DMA that waits for the
                                                           // CU's 64 B/clk address path, which the four waves' 32 KB per chunk keep busy
                                                           // for ~512 cycles either way.)
        FFN_TICK(2);
        const char* buf = smem + (c % NBUF) * SM::BUF;
        const char* w1 = buf + lr * (P1 * 16);
        const char* w2 = buf + SM::W1_BYTES;
        const float* b1 = sb1 + c * CH;
        // ---- GEMM1: H^T[j][m] = sum_k W1[j][k] a[m][k]
        f32x16 h[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 16; ++r) h[rt][r] = 0.f;
        // fragments are fetched in batches of FB before the MFMAs that consume them, so the LDS latency is paid once per
        // batch instead of once per MFMA pair (matters most at one wave per SIMD)
        constexpr int FB = (KS % 8 == 0) ? 8 : ((KS % 4 == 0) ? 4 : KS);     // must divide KS
#pragma unroll
        for (int s0 = 0; s0 < KS; s0 += FB) {
            bf16x8 wa[FB];
#pragma unroll
            for (int i = 0; i < FB; ++i) {
                int q = q0 + 2 * (s0 + i);
                q -= q >= P1 ? P1 : 0;
                wa[i] = *reinterpret_cast<const bf16x8*>(w1 + q * 16);
            }
#pragma unroll
            for (int i = 0; i < FB; ++i)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) h[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[i], xf[rt][s0 + i], h[rt], 0, 0, 0);
        }
        if constexpr (PROF) { asm volatile("s_nop 0" :: "v"(h[0][0]), "v"(h[0][15])); }   // GEMM1 results retired before the tick
        FFN_TICK(3);
        // ---- bias + Swish, pack to bf16: registers [8s, 8s+8) are the B fragment of k-step s of GEMM2
        bf16x8 hf[RT][2];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            uint32_t w[8];
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const int j0 = (r & 3) + 8 * (r >> 2) + 4 * half;
                w[r >> 1] = pack_bf2(swishf_(h[rt][r] + b1[j0]), swishf_(h[rt][r + 1] + b1[j0 + 1]));
            }
            hf[rt][0] = as_bf16x8(make_uint4(w[0], w[1], w[2], w[3]));
            hf[rt][1] = as_bf16x8(make_uint4(w[4], w[5], w[6], w[7]));
        }
        if constexpr (PROF) { asm volatile("s_nop 0" :: "v"(hf[0][0]), "v"(hf[0][1])); }
        FFN_TICK(4);
        // ---- GEMM2: Y^T[n][m] += sum_j W2p[n][j] H^T[j][m]
        constexpr int TB = (NT2 % 4 == 0) ? 4 : ((NT2 % 2 == 0) ? 2 : 1);    // must divide NT2
#pragma unroll
        for (int t0 = 0; t0 < NT2; t0 += TB) {
            bf16x8 wb[TB][2];
#pragma unroll
            for (int i = 0; i < TB; ++i) {
                wb[i][0] = *reinterpret_cast<const bf16x8*>(w2 + (t0 + i) * 2048 + w2off0);
                wb[i][1] = *reinterpret_cast<const bf16x8*>(w2 + (t0 + i) * 2048 + w2off1);
            }
#pragma unroll
            for (int i = 0; i < TB; ++i)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    acc[rt][t0 + i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb[i][0], hf[rt][0], acc[rt][t0 + i], 0, 0, 0);
                    acc[rt][t0 + i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb[i][1], hf[rt][1], acc[rt][t0 + i], 0, 0, 0);
                }
        }
        if constexpr (PROF) { asm volatile("s_nop 0" :: "v"(acc[0][NT2 - 1][0]), "v"(acc[0][0][0])); }
        FFN_TICK(5);
    }

    // ---- epilogue: y[m][n] = x[m][n] + alpha * (acc + b2[n]);  lane owns row m, columns 32t + 8q + 4*half + (0..3).
    // Per 64-column window: the residual tile comes in coalesced through the staging region, every lane updates its own
    // elements in place, and the tile leaves coalesced.
    if constexpr (ST::ALIAS) wg_barrier();                 // every wave is done with the ring before it becomes staging memory
    {
        constexpr int NWIN = (NT2 + 1) / 2;
        const char* xb = reinterpret_cast<const char*>(p.X);
        char* yb = reinterpret_cast<char*>(p.Y);
        ffn_epilogue_pair<KS, NT2, 0>(xb, yb, (size_t)p.ldx * 4, (size_t)p.ldy * 4, p.D * 4, m_base, p.M, stg, lane, sb2, p.alpha, acc[0]);
    }
    if constexpr (PROF) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        FFN_TICK(6);
        if (lane == 0 && (blockIdx.x & 7) == 0) {      // a 1/8 sample of the workgroups reports (keeps the atomics out of the measurement)
            for (int i = 0; i < 7; ++i) atomicAdd(prof + i, ph[i]);
            atomicAdd(prof + 7, 1ull);
        }
    }
#undef FFN_TICK
}

unsigned long long* g_ffn_prof = nullptr;     // EFFCONF_FFN_PHASES=1: device counters, dumped at exit
void ffn_prof_dump() {
    unsigned long long all[8 * 8];
    if (!g_ffn_prof || hipMemcpy(all, g_ffn_prof, sizeof(all), hipMemcpyDeviceToHost) != hipSuccess) return;
    static const char* names[7] = {"prologue", "wait+barrier", "dma_issue", "gemm1", "swish", "gemm2", "epilogue"};
    for (int c = 0; c < 8; ++c) {
        const unsigned long long* h = all + 8 * c;
        if (!h[7]) continue;
        unsigned long long tot = 0;
        for (int i = 0; i < 7; ++i) tot += h[i];
        fprintf(stderr, "[ffn phases] KS=%d: waves %llu, cycles/wave %.0f\n", 4 * c, h[7], (double)tot / h[7]);
        for (int i = 0; i < 7; ++i) fprintf(stderr, "[ffn phases]   %-13s %10.0f cyc/wave  %5.1f%%\n", names[i], (double)h[i] / h[7], 100.0 * h[i] / tot);
    }
}

template <int KS, int NT2, int RT, int NW, int NBUF>
int launch_ffn_t(const FfnParams& p, hipStream_t s) {
    using SM = FfnSmem<KS, NT2, NBUF>;
    using ST = FfnStage<KS, NT2, NW, NBUF>;
    const int lds = SM::RING + p.Fp * 4 + NT2 * 32 * 4 + KS * 32 * 4 + (ST::ALIAS ? 0 : ST::BYTES);
    if (lds > 160 * 1024) return -4;
    static LdsAttr attr, attr_prof;
    ensure_dynamic_lds(reinterpret_cast<const void*>(&ffn_fused_kernel<KS, NT2, RT, NW, NBUF, false>), lds, attr);
    ensure_dynamic_lds(reinterpret_cast<const void*>(&ffn_fused_kernel<KS, NT2, RT, NW, NBUF, true>), lds, attr_prof);
    const int rows_per_wg = NW * RT * 32;
    static const bool prof = getenv("EFFCONF_FFN_PHASES") != nullptr;
    if (prof) {
        if (!g_ffn_prof) {
            if (hipMalloc(&g_ffn_prof, 512) != hipSuccess || hipMemset(g_ffn_prof, 0, 512) != hipSuccess) return -1;
            atexit(ffn_prof_dump);
        }
        hipLaunchKernelGGL((ffn_fused_kernel<KS, NT2, RT, NW, NBUF, true>), dim3((p.M + rows_per_wg - 1) / rows_per_wg),
                           dim3(NW * 64), lds, s, p, g_ffn_prof + 8 * ((KS / 4) & 7));
        return hipGetLastError() == hipSuccess ? 0 : -1;
    }
    hipLaunchKernelGGL((ffn_fused_kernel<KS, NT2, RT, NW, NBUF, false>), dim3((p.M + rows_per_wg - 1) / rows_per_wg), dim3(NW * 64),
                       lds, s, p, nullptr);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ---------------------------------------------------------------------------------------------
// rs_gemm_kernel: one GEMM, activations stationary:  C^T[n][m] = sum_k W[n][k] a[m][k]  (K = D <= 384)
//   RS_RESID : y[m][n] = R[m][n] + alpha * (acc + bias[n])      fp32   (attention output projection +
//              residual, attentions.py:716 + blocks.py:126; pointwise-2 + residual, modules.py:519 + blocks.py:129)
//   RS_F32   : y[m][n] = acc + bias[n]                          fp32   (conv_res 1x1 strided conv, blocks.py:106-110)
//   RS_GLU   : g[m][j] = (acc_a + b_a) * sigmoid(acc_b + b_b)   bf16   (pointwise-1 + GLU, modules.py:513-514;
//              weight rows interleaved per 32 channels: chunk 2i = a, chunk 2i+1 = b)
//   RS_QKV   : Q+u, Q+v, K, V scattered to head-major [B][H][Tg][dpad] (attentions.py:651-686)   (odd d fallback)
//   RS_QKV_NAT: Q+u, Q+v, K, V as plain rows [B*Tp][D] with 8-byte row-contiguous stores; the weight rows of every
//              32-row chunk are permuted at pack time (row j <-> column 16(j>>4) + 8((j>>2)&1) + 4((j>>3)&1) + (j&3)) so that
//              a lane's accumulators r = 0..7 / 8..15 are the columns 32c + 16(r>>3) + 8*half + (r&7): each 16-byte store of a
//              lane pair fills one 32-byte sector per row (half-filled sectors doubled the HBM write traffic: profiles/r1_06)
enum { RS_RESID = 0, RS_F32 = 1, RS_GLU = 2, RS_QKV = 3, RS_QKV_NAT = 4 };

struct RsDev { GemmParams p; int nchunks; FastDiv32 fD, fd; };

// G = output tiles (32 columns each) accumulated in registers before they are flushed.  Flushes of the QKV / GLU
// variants contain only stores and sit at the FRONT of an iteration (before the next DMA issue), so the counted vmcnt
// never under-waits (memory ops retire in order; extra stores in the FIFO can only make the wait stricter).  The
// residual variants read R, so they keep the whole row (G >= N/32) and flush once after the loop.
// second launch-bounds argument = minimum waves per SIMD: small workgroups want several co-resident per CU so that one
// workgroup's prologue / flush memory phases overlap another's MFMA phase
template <int KS, int G, int RT, int NW, int NBUF, int EPI>
__global__ __launch_bounds__(NW * 64, (NW == 4 && RT == 1 && EPI != RS_RESID) ? (KS <= 8 ? 3 : (KS <= 12 ? 2 : 1)) : (NW == 8 ? 2 : 1))
void rs_gemm_kernel(const RsDev gd) {
    const GemmParams& p = gd.p;
    static_assert(KS % NW == 0, "uniform DMA count per wave");
    static_assert(EPI != RS_GLU || G % 2 == 0, "GLU flushes (a, b) pairs");
    constexpr int PER = KS / NW;
    constexpr int P1 = KS * 2;
    constexpr int BUF = CH * P1 * 16;
    constexpr int NTHR = NW * 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sbias = reinterpret_cast<float*>(smem + NBUF * BUF);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, half = lane >> 5;
    const int m_base = (blockIdx.x * NW + wave) * (RT * 32);
    const int nchunks = gd.nchunks;

    uint32_t doff[PER];                                        // per-lane DMA byte offsets, computed once (see ffn_fused_kernel)
#pragma unroll
    for (int k = 0; k < PER; ++k) doff[k] = dma_rows32_off<P1>(p.ldw, wave + NW * k, lane);
    auto issue = [&](int c) {
        char* buf = smem + (c % NBUF) * BUF;
        const char* w = reinterpret_cast<const char*>(p.W + (size_t)c * CH * p.ldw);
#pragma unroll
        for (int k = 0; k < PER; ++k) glds16(w, doff[k], buf + 64 * (wave + NW * k) * 16);
    };

    for (int i = tid; i < nchunks * CH; i += NTHR) sbias[i] = p.bias[i];
    float* sgam = sbias + nchunks * CH;                       // LayerNorm gamma | beta (KS*16 each, zero padded)
    float* sbet = sgam + KS * 16;
    // QKV epilogues: the rel-pos bias u (D floats; Q + v is derived from Q + u in the attention kernel) also lives in LDS.  Read from global inside the flush (under the
    // `which == 0` branch) every Q tile exposed one L2 round trip before its stores: ~16 of them per workgroup.
    float* su = sbet + KS * 16;
    if constexpr (EPI == RS_QKV || EPI == RS_QKV_NAT)
        for (int i = tid; i < p.D; i += NTHR) su[i] = p.u[i];                            // visible after the first wg_barrier
    const bool fuse_ln = p.X != nullptr;
    if (fuse_ln) {
        for (int i = tid; i < KS * 16; i += NTHR) { sgam[i] = i < p.K ? p.ln_g[i] : 0.f; sbet[i] = i < p.K ? p.ln_b[i] : 0.f; }
        __syncthreads();
    }
    bf16x8 xf[RT][KS];
    int rowm[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const int m = m_base + rt * 32 + lr;
        rowm[rt] = m;
        const int mc = m < p.M ? m : p.M - 1;
        if (fuse_ln) {
            load_row_layernorm<KS>(p.X + (size_t)mc * p.ldx, p.K, m < p.M, half, sgam, sbet, xf[rt]);
            continue;
        }
        size_t src = mc;
        if (p.a_rows > 0) { const int b = mc / p.a_rows, r = mc - b * p.a_rows; src = (size_t)b * p.a_pitch + (size_t)r * p.a_stride; }
        const bf16_t* arow = p.A + src * p.lda;
        uint4 raw[KS];                                   // unconditional clamped loads, masked afterwards (see ffn_fused_kernel)
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int c0 = s * 16 + half * 8;
            raw[s] = *reinterpret_cast<const uint4*>(arow + (c0 < p.lda - 8 ? c0 : p.lda - 8));
        }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int c0 = s * 16 + half * 8;
            xf[rt][s] = as_bf16x8(mask_chunk(raw[s], (m < p.M) ? p.K - c0 : 0));
        }
    }
    // per-row constants of the QKV scatter
    int qb[RT], qtq[RT], qtoff[RT];
    size_t qrow[RT];
    if constexpr (EPI == RS_QKV_NAT) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int m = rowm[rt] < p.M ? rowm[rt] : p.M - 1;
            const int b = m / p.T, t = m - b * p.T;
            qrow[rt] = ((size_t)b * (p.Tg * p.G) + t) * p.D;
        }
    }
    if constexpr (EPI == RS_QKV) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int m = rowm[rt] < p.M ? rowm[rt] : p.M - 1;
            const int b = m / p.T, t = m - b * p.T;
            qb[rt] = b; qtq[rt] = t / p.G; qtoff[rt] = t - qtq[rt] * p.G;
        }
    }
#pragma unroll
    for (int c = 0; c < NBUF - 1; ++c)
        if (c < nchunks) issue(c);

    f32x16 acc[RT][G];

    // flush tiles [0, ntiles) of the group whose first chunk is c0: lane owns row m, columns 32(c0+g) + 8q + 4*half + i
    auto flush = [&](int c0, int ntiles) __attribute__((always_inline)) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int m = rowm[rt];
            const int mc = m < p.M ? m : p.M - 1;
            if constexpr (EPI == RS_RESID || EPI == RS_F32) {
                // whole-row variants: batches of GB tiles — one burst of unconditional clamped residual loads, then consume
                constexpr int GB = G < 4 ? G : 4;
                float* yr = reinterpret_cast<float*>(p.C) + (size_t)mc * p.ldc;
#pragma unroll
                for (int g0 = 0; g0 < G; g0 += GB) {
                    float4 rv[GB * 4];
                    if constexpr (EPI == RS_RESID) {
#pragma unroll
                        for (int g = 0; g < GB; ++g)
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const int n = (c0 + g0 + g) * CH + q * 8 + half * 4;
                                rv[g * 4 + q] = *reinterpret_cast<const float4*>(p.R + (size_t)mc * p.ldr + (n < p.N - 4 ? n : p.N - 4));
                            }
                    }
#pragma unroll
                    for (int g = 0; g < GB; ++g) {
                        const float* bias = sbias + (c0 + g0 + g) * CH;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int nl = q * 8 + half * 4, n = (c0 + g0 + g) * CH + nl;
                            float4 o;
                            o.x = acc[rt][g0 + g][q * 4 + 0] + bias[nl + 0]; o.y = acc[rt][g0 + g][q * 4 + 1] + bias[nl + 1];
                            o.z = acc[rt][g0 + g][q * 4 + 2] + bias[nl + 2]; o.w = acc[rt][g0 + g][q * 4 + 3] + bias[nl + 3];
                            if constexpr (EPI == RS_RESID) {
                                const float4 xv = rv[g * 4 + q];
                                o.x = xv.x + p.alpha * o.x; o.y = xv.y + p.alpha * o.y; o.z = xv.z + p.alpha * o.z; o.w = xv.w + p.alpha * o.w;
                            }
                            if (g0 + g < ntiles && n < p.N && m < p.M) *reinterpret_cast<float4*>(yr + n) = o;
                        }
                    }
                }
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                if (g < ntiles && m < p.M) {
                const float* bias = sbias + (c0 + g) * CH;
                if constexpr (EPI == RS_RESID || EPI == RS_F32) {
                } else if constexpr (EPI == RS_GLU) {
                    if ((g & 1) == 0 && g + 1 < G) {                       // tiles (g, g+1) = (a, b) of channel block (c0+g)/2
                    constexpr int gb = (G > 1) ? 1 : 0;
                    const float* bias_b = bias + CH;
                    bf16_t* gr = reinterpret_cast<bf16_t*>(p.C) + (size_t)m * p.ldc;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int nl = q * 8 + half * 4, j = ((c0 + g) >> 1) * CH + nl;
                        if (j >= p.ldc) continue;
                        float o[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            o[i] = (acc[rt][g][q * 4 + i] + bias[nl + i]) * sigmoidf_(acc[rt][(g + gb) % G][q * 4 + i] + bias_b[nl + i]);
                        *reinterpret_cast<uint2*>(gr + j) = make_uint2(pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3]));
                    }
                    }
                } else if constexpr (EPI == RS_QKV_NAT) {
                    // columns 32(c0+g) + 16*half + r: two runs of 8 consecutive columns -> 16-byte stores when D % 8 == 0
                    // (a run never straddles the Q|K|V boundary), else four runs of 4 (8-byte stores)
                    if ((p.D & 7) == 0) {
#pragma unroll
                        for (int r0 = 0; r0 < 16; r0 += 8) {
                            const int n0 = (c0 + g) * CH + 16 * (r0 >> 3) + 8 * half;   // lane pair -> one full 32-byte sector per row
                            if (n0 < p.N) {
                                const int which = gd.fD.div(n0), nn0 = n0 - which * p.D;
                                float v[8];
#pragma unroll
                                for (int i = 0; i < 8; ++i) v[i] = acc[rt][g][r0 + i] + bias[(i & 3) + 8 * ((r0 + i) >> 2) + 4 * half];
                                const size_t idx = qrow[rt] + nn0;
                                if (which == 0) {
                                    float uu[8];
                                    *reinterpret_cast<float4*>(uu) = *reinterpret_cast<const float4*>(su + nn0);
                                    *reinterpret_cast<float4*>(uu + 4) = *reinterpret_cast<const float4*>(su + nn0 + 4);
                                    *reinterpret_cast<uint4*>(p.qu + idx) = make_uint4(pack_bf2(v[0] + uu[0], v[1] + uu[1]), pack_bf2(v[2] + uu[2], v[3] + uu[3]),
                                                                                     pack_bf2(v[4] + uu[4], v[5] + uu[5]), pack_bf2(v[6] + uu[6], v[7] + uu[7]));
                                } else {
                                    bf16_t* dst = which == 1 ? p.kh : p.vt;
                                    *reinterpret_cast<uint4*>(dst + idx) = make_uint4(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7]));
                                }
                            }
                        }
                    } else {
#pragma unroll
                        for (int r0 = 0; r0 < 16; r0 += 4) {
                            const int n0 = (c0 + g) * CH + 16 * (r0 >> 3) + 8 * half + (r0 & 7);   // 4 consecutive columns, one `which`
                            if (n0 < p.N) {
                                const int which = gd.fD.div(n0), nn0 = n0 - which * p.D;
                                float v[4];
#pragma unroll
                                for (int i = 0; i < 4; ++i) v[i] = acc[rt][g][r0 + i] + bias[(i) + 8 * (r0 >> 2) + 4 * half];
                                const size_t idx = qrow[rt] + nn0;
                                if (which == 0) {
                                    const float4 u4 = *reinterpret_cast<const float4*>(su + nn0);
                                    *reinterpret_cast<uint2*>(p.qu + idx) = make_uint2(pack_bf2(v[0] + u4.x, v[1] + u4.y), pack_bf2(v[2] + u4.z, v[3] + u4.w));
                                } else {
                                    bf16_t* dst = which == 1 ? p.kh : p.vt;
                                    *reinterpret_cast<uint2*>(dst + idx) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
                                }
                            }
                        }
                    }
                } else {   // RS_QKV
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int nl = q * 8 + half * 4, n0 = (c0 + g) * CH + nl;
                        if (n0 >= p.N) continue;
                        // D % 4 == 0: the 4 columns share `which`; heads may split inside the group only between pairs when d is even
                        const int which = gd.fD.div(n0), nn0 = n0 - which * p.D;
                        bf16_t* dst = which == 1 ? p.kh : (which == 2 ? p.vt : p.qu);
#pragma unroll
                        for (int i2 = 0; i2 < 4; i2 += 2) {
                            const float v0 = acc[rt][g][q * 4 + i2] + bias[nl + i2], v1 = acc[rt][g][q * 4 + i2 + 1] + bias[nl + i2 + 1];
                            const int flat = qtoff[rt] * p.D + nn0 + i2;
                            const int h = gd.fd.div(flat), x = flat - h * p.d;
                            const size_t idx = ((size_t)(qb[rt] * p.H + h) * p.Tg + qtq[rt]) * p.dpad + x;
                            if ((p.d & 1) == 0) {
                                if (which == 0) {
                                    const float u0 = su[nn0 + i2], u1 = su[nn0 + i2 + 1];
                                    *reinterpret_cast<uint32_t*>(p.qu + idx) = pack_bf2(v0 + u0, v1 + u1);
                                } else {
                                    *reinterpret_cast<uint32_t*>(dst + idx) = pack_bf2(v0, v1);
                                }
                            } else {
                                const int flat1 = flat + 1;
                                const int h1 = gd.fd.div(flat1), x1 = flat1 - h1 * p.d;
                                const size_t idx1 = ((size_t)(qb[rt] * p.H + h1) * p.Tg + qtq[rt]) * p.dpad + x1;
                                if (which == 0) {
                                    p.qu[idx] = f2bf(v0 + su[nn0 + i2]);
                                    p.qu[idx1] = f2bf(v1 + su[nn0 + i2 + 1]);
                                } else { dst[idx] = f2bf(v0); dst[idx1] = f2bf(v1); }
                            }
                        }
                    }
                }
                }
            }
        }
    };

    const int q0 = (half + lr) % P1;
    for (int cg = 0; cg < nchunks; cg += G) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int c = cg + g;
            if (c < nchunks) {
            wait_chunks<PER, NBUF - 2>(nchunks - 1 - c);
            wg_barrier();
            if (g == 0 && cg > 0) flush(cg - G, G);            // stores of the previous group, BEFORE the next DMA issue
            if (c + NBUF - 1 < nchunks) issue(c + NBUF - 1);
            const char* w = smem + (c % NBUF) * BUF + lr * (P1 * 16);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[rt][g][r] = 0.f;
            constexpr int FB = (KS % 8 == 0) ? 8 : ((KS % 4 == 0) ? 4 : KS);   // fragment batches (divisor of KS), see ffn_fused_kernel
#pragma unroll
            for (int s0 = 0; s0 < KS; s0 += FB) {
                bf16x8 wa[FB];
#pragma unroll
                for (int i = 0; i < FB; ++i) {
                    int q = q0 + 2 * (s0 + i);
                    q -= q >= P1 ? P1 : 0;
                    wa[i] = *reinterpret_cast<const bf16x8*>(w + q * 16);
                }
#pragma unroll
                for (int i = 0; i < FB; ++i)
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) acc[rt][g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[i], xf[rt][s0 + i], acc[rt][g], 0, 0, 0);
            }
            }
        }
    }
    {   // last (possibly partial) group
        const int c0 = ((nchunks - 1) / G) * G;
        flush(c0, nchunks - c0);
    }
}

template <int KS, int G, int RT, int NW, int NBUF, int EPI>
int launch_rs_t(const RsDev& gd, hipStream_t s) {
    const int lds = NBUF * CH * KS * 32 + gd.nchunks * CH * 4 + KS * 32 * 4 + ((EPI == RS_QKV || EPI == RS_QKV_NAT) ? ((gd.p.D + 3) & ~3) * 4 : 0);
    if (lds > 160 * 1024) return -4;
    if (gd.p.ldw < KS * 16) return -6;      // every weight row is read KS * 16 columns wide: the packing must be at least that wide
    if ((EPI == RS_RESID || EPI == RS_F32) && gd.nchunks > G) return -5;
    static LdsAttr attr;
    ensure_dynamic_lds(reinterpret_cast<const void*>(&rs_gemm_kernel<KS, G, RT, NW, NBUF, EPI>), lds, attr);
    const int rows_per_wg = NW * RT * 32;
    hipLaunchKernelGGL((rs_gemm_kernel<KS, G, RT, NW, NBUF, EPI>), dim3((gd.p.M + rows_per_wg - 1) / rows_per_wg), dim3(NW * 64), lds, s, gd);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// configuration class from max(K, N_resident): KS k-steps; residual variants keep KS/2 output tiles, the others 4
// Shape heuristics (tuned on MI355X, see profiles/): small-N GEMMs are dominated by per-workgroup fixed latency, so
// they use 4-wave workgroups (128 rows) at several workgroups per CU; variant 1 = 8 waves (256 rows).
template <int EPI>
int launch_rs_ks(const RsDev& gd, hipStream_t s) {
    constexpr bool whole = (EPI == RS_RESID || EPI == RS_F32);
    const int width = whole ? (gd.p.K > gd.p.N ? gd.p.K : gd.p.N) : gd.p.K;
    const int ks = (width + 15) / 16;
    const bool big = gd.p.rs_variant == 1;
    if constexpr (whole) {      // KS/2 resident output tiles per row
        if (ks <= 2) return launch_rs_t<2, 1, 2, 2, 4, EPI>(gd, s);
        if (ks <= 4) return launch_rs_t<4, 2, 2, 4, 4, EPI>(gd, s);
        if (ks <= 8) return big ? launch_rs_t<8, 4, 1, 8, 4, EPI>(gd, s) : launch_rs_t<8, 4, 1, 4, 4, EPI>(gd, s);
        if (ks <= 12) return launch_rs_t<12, 6, 1, 4, 4, EPI>(gd, s);
        if (ks <= 16) return big ? launch_rs_t<16, 8, 1, 8, 4, EPI>(gd, s) : launch_rs_t<16, 8, 1, 4, 4, EPI>(gd, s);
        if (ks <= 20) return launch_rs_t<20, 10, 1, 4, 4, EPI>(gd, s);
        return launch_rs_t<24, 12, 1, 4, 4, EPI>(gd, s);
    } else {                    // QKV / GLU: groups of 4 tiles, stores only
        if (ks <= 2) return launch_rs_t<2, 2, 1, 2, 4, EPI>(gd, s);
        if (ks <= 4) return launch_rs_t<4, 4, 1, 4, 4, EPI>(gd, s);
        if (ks <= 8) return big ? launch_rs_t<8, 4, 1, 8, 4, EPI>(gd, s) : launch_rs_t<8, 4, 1, 4, 4, EPI>(gd, s);
        if (ks <= 12) return launch_rs_t<12, 4, 1, 4, 4, EPI>(gd, s);
        if (ks <= 16) return big ? launch_rs_t<16, 4, 1, 8, 4, EPI>(gd, s) : launch_rs_t<16, 4, 1, 4, 4, EPI>(gd, s);
        if (ks <= 20) return launch_rs_t<20, 4, 1, 4, 4, EPI>(gd, s);
        return launch_rs_t<24, 4, 1, 8, 4, EPI>(gd, s);
    }
}

}  // namespace

bool ffn_fused_supported(int D) { return D % 4 == 0 && D <= 384; }

int launch_ffn_fused(const FfnParams& p, hipStream_t s) {
    if (p.M <= 0) return 0;
    if (!ffn_fused_supported(p.D) || p.Fp % CH || p.lda % 8 || p.ldw1 % 8 || p.ldw2 % 8) return -2;
    const int ks = (p.D + 15) / 16;
    // rows per workgroup / waves per SIMD trade-offs, selected per width class (option "ffn_variant" overrides for tuning)
    const int var = p.variant;
    if (ks <= 2) return launch_ffn_t<2, 1, 1, 4, 4>(p, s);
    if (ks <= 4) return launch_ffn_t<4, 2, 1, 8, 4>(p, s);
    if (ks <= 8) {
        if (var == 1) return launch_ffn_t<8, 4, 1, 4, 3>(p, s);      // 128 rows, 3 workgroups per CU
        if (var == 2) return launch_ffn_t<8, 4, 1, 4, 4>(p, s);
        return launch_ffn_t<8, 4, 1, 8, 4>(p, s);
    }
    if (ks <= 12) {
        if (var == 1) return launch_ffn_t<12, 6, 1, 4, 4>(p, s);
        return launch_ffn_t<12, 6, 1, 8, 3>(p, s);                   // 2 waves per SIMD, 256 rows; ring of 3 leaves room for the staging region
    }
    if (ks <= 16) return launch_ffn_t<16, 8, 1, 4, 3>(p, s);
    if (ks <= 20) return launch_ffn_t<20, 10, 1, 4, 3>(p, s);
    return launch_ffn_t<24, 12, 1, 4, 3>(p, s);
}

bool rs_gemm_supported(int K) { return K % 4 == 0 && K <= 384; }
bool rs_gemm_resident_supported(int K, int N) { return rs_gemm_supported(K) && N <= 384; }

// epi: RS_* (0 resid, 1 f32, 2 glu, 3 qkv).  W packed [>= nchunks*32][ldw >= round_up(K,64)], bias padded likewise.
int launch_rs_gemm(const GemmParams& p, int epi, hipStream_t s) {
    if (p.M <= 0 || p.N <= 0) return 0;
    if (!rs_gemm_supported(p.K) || p.lda % 8 || p.ldw % 8) return -2;
    RsDev gd;
    gd.p = p;
    gd.nchunks = (p.N + CH - 1) / CH;
    if (epi == RS_QKV || epi == RS_QKV_NAT) { gd.fD = FastDiv32(p.D); gd.fd = FastDiv32(p.d); }
    switch (epi) {
        case RS_RESID: return launch_rs_ks<RS_RESID>(gd, s);
        case RS_F32: return launch_rs_ks<RS_F32>(gd, s);
        case RS_GLU: return launch_rs_ks<RS_GLU>(gd, s);
        case RS_QKV: return launch_rs_ks<RS_QKV>(gd, s);
        case RS_QKV_NAT: return launch_rs_ks<RS_QKV_NAT>(gd, s);
    }
    return -3;
}
