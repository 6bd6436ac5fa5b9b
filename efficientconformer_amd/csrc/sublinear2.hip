// Fused Conv2dSubsampling + Linear, second generation: row-stationary, the 3x3 convolution on the MFMA pipe (gfx950).
//
// Reference: Conv2dSubsampling.forward (models/modules.py:232-249, one layer, C_in = 1: Conv2d 3x3 s2 p1 -> BatchNorm2d(eval) ->
// Swish -> reshape to (B, C*F/2, T1)) + transpose + nn.Linear(C*F/2 -> D0) of ConformerEncoder.forward (encoders.py:113-116).
//
// sublinear.hip computes the convolution on the VALU (9 fp32 FMAs + Swish per A element, ~20 lane-ops per element against 2*D0/16
// MFMA lane-ops): its A-tile producer, not the GEMM, bounds it (11 % of the bf16 MFMA peak, 12 % of the Small step).  Here:
//   * a wave owns 32 output frames, lane (lane & 31) owns ONE frame (both half-waves), as in the row-stationary kernels (rsgemm.hip);
//   * per output frequency f the convolution of all C channels is an MFMA: conv^T[c][t] = sum_tap Wc[c][tap] * P[tap][t] with the
//     folded conv weights as A operand (32 channels x 16 tap slots: 9 taps, slot 9 = folded bias against a constant 1, rest zero; held
//     in registers for the whole kernel) and the frame's 3x3 mel patch as B operand - each lane supplies its OWN frame's taps, no data
//     moves between lanes.  fp32 accuracy on bf16 MFMAs by splitting both operands: W_hi P_hi + W_hi P_lo + W_lo P_hi (error 2^-16
//     relative, below the bf16 rounding of the result that the reference path's consumers see anyway);
//   * the MFMA result leaves every lane with 16 channels OF ITS FRAME per 32-channel group in C layout; after Swish, registers [8j, 8j+8)
//     rounded to bf16 ARE the B fragment of k-step 2g + j of the Linear GEMM (its weight is packed with the K index permuted per 16, as
//     every chain weight: chain.hip) - the (B*T1, C*40) activation never exists, not even in LDS;
//   * the Linear weight streams through an LDS-DMA ring, one slab (all D0 rows x the C_pad columns of one f) per step, counted vmcnt +
//     one barrier per f (rowstat.h); K order (f, c): packed at finalize from linear.weight[n][c*F/2 + f];
//   * output rows leave through the coalesced staging of rowstat.h (256-byte row segments).
#include "kernels.h"
#include "rowstat.h"

namespace {

template <int NT>
__device__ __forceinline__ void store_rows(char* yb, size_t pitch, int N, int m_base, int M, char* stg, int lane, const f32x16 (&xc)[NT]) {
    const int lr = lane & 31, half = lane >> 5;
#pragma unroll
    for (int w = 0; w < (NT + 1) / 2; ++w) {
        wave_sync();
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int t = 2 * w + tt;
            if (t < NT) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4*>(stg + lr * STG_ROW + (tt * 32 + q * 8 + half * 4) * 4) =
                        make_float4(xc[t][4 * q + 0], xc[t][4 * q + 1], xc[t][4 * q + 2], xc[t][4 * q + 3]);
            }
        }
        wave_sync();
        stage_store(stg, yb, pitch, N * 4, m_base, M, 256 * w, lane);
    }
}

// one wave-wide 4-byte LDS-DMA: lane i's dword at g lands at lds_wave_base + 4 i (rowstat.h's glds16 with the dword opcode)
__device__ __forceinline__ void glds4(const void* g, char* lds_wave_base) {
    const uint32_t l = (uint32_t)(uintptr_t)(lds_void_t*)lds_wave_base;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off" ::"s"(l), "v"(g) : "memory", "m0");
}

__device__ __forceinline__ uint32_t split_hi(float x) { return __float_as_uint(x) & 0xFFFF0000u; }   // bf16 by truncation: hi + lo = x exactly in 16 + 8 bits

// CG = 32-channel groups (C <= 32 CG), NT = 32-column output tiles (D0 <= 32 NT), NBUF ring slabs
template <int CG, int NT, int NBUF, int NW>
__global__ __launch_bounds__(NW * 64, NW == 8 ? 2 : 1) void sublinear2_kernel(const float* __restrict__ mel, int F, int Tm, int T1, int M,
                                                            const float* __restrict__ ctab /*[CG*32][16]*/, const bf16_t* __restrict__ Wp,
                                                            const float* __restrict__ bias, int N, float* __restrict__ out, int ldc,
                                                            const int* __restrict__ rag_off, const int* __restrict__ rag_t1, const int* __restrict__ rag_tm, int rag_n) {
    constexpr int KSF = 2 * CG, P1 = 2 * KSF;                    // k-steps and 16-byte pieces per weight row of one f
    constexpr int SLAB = NT * CH * P1 * 16;                      // bytes of one f's weight slab
    constexpr int NDMA = NT * CH * P1 / 64, PER = NDMA / NW;     // wave-DMAs per slab
    static_assert(NDMA % NW == 0, "uniform DMA count per wave");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, half = lane >> 5;
    const int m_base = (blockIdx.x * NW + wave) * 32;
    const int F2 = F / 2;
    const int ldw = P1 * 8;                                      // elements per packed weight row (one f)

    uint32_t doff[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {                              // instruction i covers 64 slots of the slab: 32-row sub-chunk i / (P1/2), its piece range
        const int i = wave + NW * k;
        const int sub = i / (P1 / 2), ii = i - sub * (P1 / 2);
        doff[k] = (uint32_t)(sub * CH * ldw) * 2u + dma_rows32_off<P1>(ldw, ii, lane);
    }
    auto issue = [&](int f) __attribute__((always_inline)) {
        char* buf = smem + (f % NBUF) * SLAB;
        const char* w = reinterpret_cast<const char*>(Wp + (size_t)f * NT * CH * ldw);
#pragma unroll
        for (int k = 0; k < PER; ++k) glds16(w, doff[k], buf + 64 * (wave + NW * k) * 16);
    };

    // ---- conv weights as MFMA A fragments (row = channel, 8 tap slots per half-wave), split into bf16 hi / lo
    bf16x8 whi[CG], wlo[CG];
#pragma unroll
    for (int g = 0; g < CG; ++g) {
        const float* src = ctab + (size_t)(g * 32 + lr) * 16 + half * 8;
        const float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
        const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        uint32_t h[4], l[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint32_t h0 = split_hi(v[2 * e]), h1 = split_hi(v[2 * e + 1]);
            h[e] = (h0 >> 16) | h1;
            l[e] = pack_bf2(v[2 * e] - __uint_as_float(h0), v[2 * e + 1] - __uint_as_float(h1));
        }
        whi[g] = as_bf16x8(make_uint4(h[0], h[1], h[2], h[3]));
        wlo[g] = as_bf16x8(make_uint4(l[0], l[1], l[2], l[3]));
    }
#pragma unroll
    for (int g = 0; g < CG; ++g) asm volatile("" :: "v"(whi[g]), "v"(wlo[g]));      // the compiler's own waits for these loads end HERE, not inside the DMA loop

    // ---- this lane's frame.  The mel patch is a rolling 3-row window.  Its look-ahead rows arrive by LDS-DMA as well (6 dword pieces per
    //      wave and frequency into a private two-slot stage): no register is ever written behind the compiler's back - hand-issued loads
    //      into VGPRs were tried first and broke as soon as the allocator moved one of those registers (AGPR / scratch copies taken before
    //      the data had landed: NaNs on Medium) - and the queue stays countable: per iteration 6 row pieces, then the slab's PER pieces.
    const int m = m_base + lr, mc = m < M ? m : M - 1;
    // rectangular batch: row m = (b, t) of B x T1.  Ragged batch (rag_off != null): utterance b owns rows [rag_off[b], rag_off[b + 1]) of
    // the concatenated row space, has rag_t1[b] frames after the subsampling (rows past them pad the utterance to a multiple of the
    // attention group size: written as zeros) and rag_tm[b] mel frames (the conv's zero padding starts THERE); Tm stays the row pitch.
    int b, t, tmb = Tm;
    bool pad_row = false;
    if (rag_off) {
        b = ragged_find(rag_off, rag_n, mc); t = mc - rag_off[b]; tmb = rag_tm[b];
        pad_row = t >= rag_t1[b];
        t = pad_row ? 0 : t;
    } else { b = mc / T1; t = mc - b * T1; }
    const float* melb = mel + (size_t)b * F * Tm;
    int col[3]; bool cok[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) { const int tc = 2 * t - 1 + j; cok[j] = tc >= 0 && tc < tmb; col[j] = tc < 0 ? 0 : (tc < tmb ? tc : tmb - 1); }
    auto row_ptr = [&](int fr) { return melb + (size_t)(fr < 0 ? 0 : (fr < F ? fr : F - 1)) * Tm; };
    auto mask_row = [&](int fr, float (&r)[3]) __attribute__((always_inline)) {     // zero outside the image (conv padding 1)
        const bool ok = fr >= 0 && fr < F;
#pragma unroll
        for (int j = 0; j < 3; ++j) r[j] = (ok && cok[j]) ? r[j] : 0.f;
    };
    char* mstage = smem + NBUF * SLAB + wave * (2 * 6 * 256);     // [2 slots][6 pieces][64 lanes] floats
    auto issue_rows = [&](int fr, int slot) __attribute__((always_inline)) {         // rows fr, fr + 1 -> stage slot
        const float* p0 = row_ptr(fr);
        const float* p1 = row_ptr(fr + 1);
#pragma unroll
        for (int j = 0; j < 3; ++j) glds4(p0 + col[j], mstage + (slot * 6 + j) * 256);
#pragma unroll
        for (int j = 0; j < 3; ++j) glds4(p1 + col[j], mstage + (slot * 6 + 3 + j) * 256);
    };
    float r0[3], r1[3], r2[3];
    {   // the first window by ordinary loads, retired before the loop (the empty asm makes the compiler place its wait here)
        const float* pa = row_ptr(-1); const float* pb = row_ptr(0); const float* pc = row_ptr(1);
#pragma unroll
        for (int j = 0; j < 3; ++j) { r0[j] = pa[col[j]]; r1[j] = pb[col[j]]; r2[j] = pc[col[j]]; }
#pragma unroll
        for (int j = 0; j < 3; ++j) asm volatile("" :: "v"(r0[j]), "v"(r1[j]), "v"(r2[j]));
        mask_row(-1, r0); mask_row(0, r1); mask_row(1, r2);
    }
    issue_rows(2, 0);                                             // rows 2, 3: consumed in iteration 0
#pragma unroll
    for (int c = 0; c < NBUF - 1; ++c)
        if (c < F2) issue(c);

    f32x16 xc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) xc[nt][r] = 0.f;

    const int q0 = (half + lr) % P1;
    for (int f = 0; f < F2; ++f) {
        // Counted wait: the queue holds (oldest first) ... slab f, the row pieces issued in iteration f - 1, then - only with a three-slab
        // ring - slab f + 1.  Everything but that youngest slab must have landed.
        if (NBUF >= 3 && f + 1 < F2) wait_vmcnt<PER>(); else wait_vmcnt<0>();
        wg_barrier();                                             // everybody's pieces of slab f are in; everybody is done with slab f - 1
        float n1[3], n2[3];                                       // rows 2f + 2, 2f + 3 from the stage (this wave's own pieces: its wait above covers them)
        {
            const float* st = reinterpret_cast<const float*>(mstage + (f & 1) * 6 * 256) + lane;
#pragma unroll
            for (int j = 0; j < 3; ++j) { n1[j] = st[j * 64]; n2[j] = st[(3 + j) * 64]; }
        }
        mask_row(2 * f + 2, n1); mask_row(2 * f + 3, n2);
        // ---- patch fragments of this frame at frequency f: tap = 3 i + j; half 0 holds taps 0..7, half 1 tap 8, the constant 1 (bias), zeros
        float tp[8];
        if (half == 0) { tp[0] = r0[0]; tp[1] = r0[1]; tp[2] = r0[2]; tp[3] = r1[0]; tp[4] = r1[1]; tp[5] = r1[2]; tp[6] = r2[0]; tp[7] = r2[1]; }
        else { tp[0] = r2[2]; tp[1] = 1.0f; tp[2] = 0.f; tp[3] = 0.f; tp[4] = 0.f; tp[5] = 0.f; tp[6] = 0.f; tp[7] = 0.f; }
        uint32_t ph[4], pl[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint32_t h0 = split_hi(tp[2 * e]), h1 = split_hi(tp[2 * e + 1]);
            ph[e] = (h0 >> 16) | h1;
            pl[e] = pack_bf2(tp[2 * e] - __uint_as_float(h0), tp[2 * e + 1] - __uint_as_float(h1));
        }
        const bf16x8 phi = as_bf16x8(make_uint4(ph[0], ph[1], ph[2], ph[3])), plo = as_bf16x8(make_uint4(pl[0], pl[1], pl[2], pl[3]));
        // roll the window to rows 2f+1 .. 2f+3 (frequency f + 1), then fetch rows 2f+4, 2f+5 (frequency f + 2) and the next slab - in THIS order
#pragma unroll
        for (int j = 0; j < 3; ++j) { r0[j] = r2[j]; r1[j] = n1[j]; r2[j] = n2[j]; }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the stage reads above are done before their slot's next refill is issued... (the other slot is refilled now; this orders slot f & 1 for iteration f + 1's issue)
        issue_rows(2 * f + 4, (f + 1) & 1);
        if (f + NBUF - 1 < F2) issue(f + NBUF - 1);

        // ---- convolution of all channels on the MFMA pipe, Swish, round: the Linear GEMM's B fragments of this f
        bf16x8 xf[KSF];
#pragma unroll
        for (int g = 0; g < CG; ++g) {
            f32x16 cv;
#pragma unroll
            for (int r = 0; r < 16; ++r) cv[r] = 0.f;
            cv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(whi[g], phi, cv, 0, 0, 0);
            cv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(whi[g], plo, cv, 0, 0, 0);
            cv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wlo[g], phi, cv, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float y[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) y[e] = swishf_(cv[8 * j + e]);
                xf[2 * g + j] = as_bf16x8(make_uint4(pack_bf2(y[0], y[1]), pack_bf2(y[2], y[3]), pack_bf2(y[4], y[5]), pack_bf2(y[6], y[7])));
            }
        }
        // ---- Linear: out^T[n][t] += W[n][f, :] . X^T[:, t]
        const char* buf = smem + (f % NBUF) * SLAB + lr * (P1 * 16);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const char* w = buf + nt * CH * P1 * 16;
            constexpr int FB = 4;                               // fragment batches: KSF = 8 / 12 registers' worth of weight fragments at a time
#pragma unroll
            for (int s0 = 0; s0 < KSF; s0 += FB) {
                bf16x8 wa[FB];
#pragma unroll
                for (int i = 0; i < FB; ++i) {
                    int q = q0 + 2 * (s0 + i);
                    q -= q >= P1 ? P1 : 0;
                    wa[i] = *reinterpret_cast<const bf16x8*>(w + q * 16);
                }
#pragma unroll
                for (int i = 0; i < FB; ++i) xc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[i], xf[s0 + i], xc[nt], 0, 0, 0);
            }
        }
    }
    wait_vmcnt<0>();                                              // the last iterations' look-ahead rows (never used)
    // ---- epilogue: + bias, rows out through the staging region (the ring is free: barrier first)
    __syncthreads();
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int n = nt * 32 + q * 8 + half * 4;
            const float4 bz = *reinterpret_cast<const float4*>(bias + n);          // bias is padded to a multiple of 128
            xc[nt][4 * q + 0] += bz.x; xc[nt][4 * q + 1] += bz.y; xc[nt][4 * q + 2] += bz.z; xc[nt][4 * q + 3] += bz.w;
            if (pad_row) { xc[nt][4 * q + 0] = 0.f; xc[nt][4 * q + 1] = 0.f; xc[nt][4 * q + 2] = 0.f; xc[nt][4 * q + 3] = 0.f; }
        }
    store_rows<NT>(reinterpret_cast<char*>(out), (size_t)ldc * 4, N, m_base, M, smem + wave * STG_BYTES, lane, xc);
}

template <int CG, int NT, int NBUF, int NW>
int launch2(const float* mel, int B, int F, int Tm, int T1, const float* ctab, const bf16_t* Wp, const float* bias, int N, float* out, int ldc,
            hipStream_t s, const RaggedRows* rg, const int* rag_tm) {
    constexpr int P1 = 4 * CG, SLAB = NT * CH * P1 * 16;
    const int M = rg ? rg->rows : B * T1;
    const int lds = NBUF * SLAB + NW * 2 * 6 * 256 > NW * STG_BYTES ? NBUF * SLAB + NW * 2 * 6 * 256 : NW * STG_BYTES;
    if (lds > 160 * 1024) return -4;
    static LdsAttr attr;
    ensure_dynamic_lds(reinterpret_cast<const void*>(&sublinear2_kernel<CG, NT, NBUF, NW>), lds, attr);
    hipLaunchKernelGGL((sublinear2_kernel<CG, NT, NBUF, NW>), dim3((M + NW * 32 - 1) / (NW * 32)), dim3(NW * 64), lds, s, mel, F, Tm, T1, M, ctab, Wp, bias, N, out, ldc,
                       rg ? rg->off : nullptr, rg ? rg->len : nullptr, rag_tm, rg ? rg->n : 0);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace

// supported: F = 80, C <= 128 with D0 <= 128 (Small, Transducer-Small: 8 waves, two per SIMD) and C, D0 <= 192 (Medium, Transducer-Medium:
// 4 waves, one per SIMD - at two per SIMD the 6-group instance spills, and a spill of a hand-loaded window register breaks the counted
// waits: its first run ended in a memory fault).  Wider front ends (Large: D0 = 360) use sublinear.hip / the separate kernels.
int sublinear2_groups(int F, int C, int N) {
    if (F != 80 || N % 4) return 0;
    if (C <= 128 && N <= 128) return 4;
    if (C <= 192 && N <= 192) return 6;
    return 0;
}

// ctab: [32 * groups][16] fp32 (9 folded taps, folded bias, zeros); Wp: [F/2][32 * NT rows][32 * groups] bf16, K-permuted per 16 (encoder.hip)
int launch_sublinear2(const float* mel, int B, int F, int Tm, int T1, const float* ctab, const bf16_t* Wp, const float* bias, int C, int N,
                      float* out, int ldc, hipStream_t s, const RaggedRows* rg, const int* rag_tm) {
    if (B <= 0 || T1 <= 0) return 0;
    if (rg && (!rag_tm || rg->rows <= 0)) return rg->rows <= 0 ? 0 : -2;
    switch (sublinear2_groups(F, C, N)) {
        case 4: return launch2<4, 4, 3, 8>(mel, B, F, Tm, T1, ctab, Wp, bias, N, out, ldc, s, rg, rag_tm);
        case 6: return launch2<6, 6, 2, 4>(mel, B, F, Tm, T1, ctab, Wp, bias, N, out, ldc, s, rg, rag_tm);
    }
    return -2;
}
