// Column-pair form of the fused row-local chains (gfx950): the chains of chain.hip for the WIDE stages (D = 129 .. 256), two waves per SIMD.
//
// chain.hip keeps a wave's 32 residual rows in registers over the full padded width: 32 x 256 fp32 + their bf16 copy are 192 registers
// before any temporary, so the D = 240 chains run ONE wave per SIMD (4-wave, 128-row workgroups, 467 of 512 registers) and the D = 168
// chains two per SIMD at 256 registers with spills and no room for the software-pipelined FFN loop.  A lone wave is its own dependency
// chain - 16 accumulating MFMAs -> Swish -> 16 MFMAs per 32 hidden units, its refill DMAs blocking everything behind them - and only four
// waves issue the L2 -> LDS weight stream (5.6 B / cycle per issuing wave, profiles/r4_03_lds_fill_rate.txt): 0.08 of the MFMA peak.
//
// Here a workgroup is 8 waves on the same 128 rows: a PAIR of waves (w, w + 4: the two waves of one SIMD) per 32 rows.
//   * wave A (cw = 0) owns the residual columns [0, DP / 2), wave B (cw = 1) [DP / 2, DP): NT / 2 accumulator tiles each (64 registers
//     at D = 240 instead of 128); both hold the full normalised row as bf16 B fragments (xf: they exchange their halves through LDS
//     after every LayerNorm / load, 8 KiB per wave);
//   * FFN: hidden chunks ALTERNATE between the two - the owner of chunk j runs the first GEMM (all K) + Swish and publishes the 32 x 32
//     bf16 tile hf(j) through LDS (2 KiB), and in the NEXT iteration BOTH waves run the second GEMM of chunk j on their own column tiles.
//     Per iteration one wave issues 16 + 8 MFMAs and the Swish, its partner 8 MFMAs and (having the slack) its share of the refill: the
//     SIMD's matrix pipe sees 32 MFMAs per 32 hidden units as before, from two instruction streams that cover each other's stalls;
//   * Q/K/V and GLU chunks alternate as well, and the owner of chunk c writes its tile out DURING chunk c + 1 (the partner's MFMAs): the
//     HBM-bound write-out (2100 of a chunk's 6090 cycles in chain.hip's phase profile) runs beside the other wave's matrix work;
//   * eight waves issue the weight ring's LDS-DMAs (PER = KS / 4 per wave and chunk).
// Every accumulator sees the same operations on the same operands in the same order as in chain.hip (k order of the first GEMM, chunk
// order of the second, LayerNorm sums continued from wave A's partial in wave B), so the rows are BIT-IDENTICAL to chain.hip's - which
// is how this file is tested (tests/test_gpu_round5.py: option chain_pair on / off).  Weights, constant blocks and parameters are
// chain.hip's (same packing, same ring chunk layout with the FFN stage software-pipelined: ring chunk j = [W1 rows of hidden chunk j |
// W2 slab of hidden chunk j - 1]).
// Registers: ~200 per wave -> two waves per SIMD; the tail and the head of chain A fit ONE kernel at D = 240 / 256 (chain.hip: two
// launches there).  Reference: models/modules.py:385-392, 511-522; blocks.py:119-137; attentions.py:651-686, 716.
#include "kernels.h"
#include "rowstat.h"

#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace {

constexpr int NW2 = 8, NBUF2 = 3;
constexpr int S2_ROW = 144;                  // staging row pitch: 128-byte windows + 16 (36 dwords: conflict-free 16-byte accesses, rows x pieces either way)
constexpr int S2_WIN = 32 * S2_ROW;          // 4608: the window area; its bytes [0, 4096) also carry the xf / hf exchanges between the waves of a pair
constexpr int S2_BYTES = S2_WIN + 512;       // + the LayerNorm hand-off slots (never touched by the window / exchange traffic: no barrier between their last read and the next private use)

template <int KS>
struct Geo2 {
    static_assert(KS % 4 == 0, "an even number of 32-column tiles per wave");
    static constexpr int NT = KS / 2, NTH = NT / 2, KSH = KS / 2;
    static constexpr int DP = 16 * KS, P1 = KS * 2;
    static constexpr int HALF = CH * P1 * 16, BUF = 2 * HALF;
    static constexpr int PER = 2 * KS / NW2;             // wave-DMAs per wave and ring chunk
};

struct ChainDev2 {
    ChainParams p;
    FastDiv32 fT, fD;
    int nf[8];
    int nfl_kb;
    int ldr, ld2;
};

template <int V> using ic = std::integral_constant<int, V>;
// compile-time loop: f(ic<0>{}), f(ic<1>{}), ... - array indices derived from the counter are constants BEFORE any optimisation pass (a runtime offset
// into a register array that only becomes constant after unrolling can leave the array in scratch)
template <int I, int N, class F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(ic<I>{}); static_for<I + 1, N>(f); }
}

// ---- 128-byte-window staging (private to a wave): lane = (row 8i + lane / 8, piece lane % 8)
template <int OFF, int N>
__device__ __forceinline__ void s2_load(const char* base, size_t pitch, int row_bytes, int m_base, int M, int wbyte, int lane, u32x4 (&v)[N], bool nt = false) {
    int cb = wbyte + 16 * (lane & 7);
    cb = cb < row_bytes - 16 ? cb : row_bytes - 16;
    if (nt) {               // streamed once: keep the rows from displacing the weights in L2 (wave-uniform branch)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m_base + 8 * i + (lane >> 3);
            v[OFF + i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(base + (size_t)(m < M ? m : M - 1) * pitch + cb));
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m_base + 8 * i + (lane >> 3);
            v[OFF + i] = *reinterpret_cast<const u32x4*>(base + (size_t)(m < M ? m : M - 1) * pitch + cb);
        }
    }
}
template <int OFF, int N>
__device__ __forceinline__ void s2_put(char* stg, int lane, const u32x4 (&v)[N]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(stg + (8 * i + (lane >> 3)) * S2_ROW + 16 * (lane & 7)) = v[OFF + i];
}
__device__ __forceinline__ void s2_store(const char* stg, char* base, size_t pitch, int row_bytes, int m_base, int M, int wbyte, int lane, bool nt = false) {
    const int cb = wbyte + 16 * (lane & 7);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m_base + 8 * i + (lane >> 3);
        const u32x4 v = *reinterpret_cast<const u32x4*>(stg + (8 * i + (lane >> 3)) * S2_ROW + 16 * (lane & 7));
        if (m < M && cb < row_bytes) {
            if (nt) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(base + (size_t)m * pitch + cb));
            else *reinterpret_cast<u32x4*>(base + (size_t)m * pitch + cb) = v;
        }
    }
}

// DMA: 0 = every wave issues its refill in one burst right behind the chunk barrier; 1 = the owner of an FFN chunk hangs its wave-DMAs behind the MFMA groups
// of its first GEMM (its partner, which has the slack, still bursts)
// PROF (tuning only, EFFCONF_CHAIN2_PHASES=<10 KS + kind>): s_memtime per phase, wave A and wave B of every 8th workgroup's first pair
template <int KS, int KIND, int DMA, bool PROF = false>
__global__ __launch_bounds__(NW2 * 64, 1) void chain2_kernel(const ChainDev2 cd, unsigned long long* prof = nullptr) {
    unsigned long long ph[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, t0 = 0;
    if constexpr (PROF) t0 = __builtin_readcyclecounter();
#define C2_TICK(i) do { if constexpr (PROF) { asm volatile("" ::: "memory"); const unsigned long long t1_ = __builtin_readcyclecounter(); ph[i] += t1_ - t0; t0 = t1_; } } while (0)
    using G = Geo2<KS>;
    constexpr int NT = G::NT, NTH = G::NTH, KSH = G::KSH, P1 = G::P1, HALF = G::HALF, BUF = G::BUF, PER = G::PER, DP = G::DP;
    constexpr bool ISB = KIND == CHAIN_B;
    constexpr bool PRE = KIND == CHAIN_A_FULL || KIND == CHAIN_A_TAIL;
    constexpr bool POST = KIND == CHAIN_A_FULL || KIND == CHAIN_A_HEAD;
    const ChainParams& p = cd.p;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* stg_base = smem + NBUF2 * BUF;
    float* sf = reinterpret_cast<float*>(stg_base + NW2 * S2_BYTES);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pr = wave & 3, cw = wave >> 2;                  // row tile of the workgroup; column half (waves w and w + 4 share a SIMD)
    const int lr = lane & 31, half = lane >> 5;
    const int m_base = (blockIdx.x * 4 + pr) * 32;
    char* stg = stg_base + wave * S2_BYTES;                   // this wave's staging region
    const char* stgp = stg_base + (wave ^ 4) * S2_BYTES;      // the partner's (read only)
    const int D = p.D;

    // ---- chunk schedule: [g0] [ffn0] [ffn1] [g1]; an FFN stage of n hidden chunks = n + 1 ring chunks (see chain.hip, PIPE)
    const int n_g0 = (ISB || PRE) ? (NT + 1) / 2 : 0;
    const int n_f0 = PRE ? p.f[0].Fp / CH : 0;
    const int n_f1 = POST ? p.f[1].Fp / CH : 0;
    const int n_g1 = (ISB || POST) ? p.g1.nchunks : 0;
    const int r_f0 = n_f0 + (n_f0 ? 1 : 0), r_f1 = n_f1 + (n_f1 ? 1 : 0);
    const int e0 = n_g0, e1 = e0 + r_f0, e2 = e1 + r_f1, total = e2 + n_g1;

    // DUTY (DMA >= 2): the two waves of a pair take the refills in turns - ring chunk k is issued (2 PER wave-DMAs, wave-instructions pr + 4 k') by the
    // waves with cw == (k & 1), in iteration k - 2; the FFN stages make the OTHER wave of that iteration the owner of the hidden chunk (first GEMM +
    // Swish: the long half), the Q/K/V / GLU stages the same wave (its partner writes the previous chunk out)
    constexpr bool DUTY = DMA >= 2;
    constexpr int NOFF = DUTY ? 2 * PER : PER, ISTR = DUTY ? 4 : NW2;
    const int i0 = DUTY ? pr : wave;
    const bool cm = p.w2cm && (!PRE || p.f[0].w2cm) && (!POST || p.f[1].w2cm);      // second FFN weights from their chunk-major images
    uint32_t off_r[NOFF], off_f[NOFF];
#pragma unroll
    for (int k = 0; k < NOFF; ++k) {
        const int i = i0 + ISTR * k;
        off_r[k] = i < KS ? dma_rows32_off<P1>(cd.ldr, i, lane) : dma_rows32_off<P1>(cd.ldr, i - KS, lane) + (uint32_t)(32 * cd.ldr) * 2u;
        off_f[k] = i < KS ? off_r[k] : (cm ? (uint32_t)((i - KS) * 1024 + lane * 16) : dma_w2_off(cd.ld2, i - KS, lane));
    }
    struct Refill { const char* b_lo; const char* b_hi; char* buf; bool ffn; };
    auto issue_prep = [&](int c) __attribute__((always_inline)) -> Refill {
        const bool ffn = c >= e0 && c < e2;
        const bool second = c >= e1;
        const int cf = c - (second ? e1 : e0);
        const int nh = second ? n_f1 : n_f0;
        int c1 = cf < nh ? cf : nh - 1, c2 = cf - 1;
        c1 = c1 > 0 ? c1 : 0; c2 = c2 > 0 ? c2 : 0;
        const bf16_t* fw1 = second ? p.f[1].w1 : p.f[0].w1;
        const bf16_t* fw2 = cm ? (second ? p.f[1].w2cm : p.f[0].w2cm) : (second ? p.f[1].w2 : p.f[0].w2);
        const bool first_g = c < e0;
        const bf16_t* gw = first_g ? p.g0.w : p.g1.w;
        const int cg = first_g ? c : c - e2;
        const char* w1 = reinterpret_cast<const char*>(fw1 + (size_t)c1 * CH * cd.ldr);
        const char* w2 = reinterpret_cast<const char*>(fw2 + (cm ? (size_t)c2 * (DP * 32) : (size_t)c2 * CH));
        const char* w = reinterpret_cast<const char*>(gw + (size_t)(cg > 0 ? cg : 0) * 64 * cd.ldr);
        return Refill{ffn ? w1 : w, ffn ? w2 : w, smem + (c % NBUF2) * BUF, ffn};
    };
    static_assert(HALF == 64 * KS * 16, "second half of a buffer = wave-instruction KS");
    auto issue_one = [&](const Refill& r, int k, auto nc) __attribute__((always_inline)) {
        const int i = i0 + ISTR * k;
        if constexpr (decltype(nc)::value) glds16_nc(i < KS ? r.b_lo : r.b_hi, r.ffn ? off_f[k] : off_r[k], r.buf + 1024 * i);
        else glds16(i < KS ? r.b_lo : r.b_hi, r.ffn ? off_f[k] : off_r[k], r.buf + 1024 * i);
    };
    auto issue = [&](int c) __attribute__((always_inline)) {
        const Refill r = issue_prep(c);
#pragma unroll
        for (int k = 0; k < NOFF; ++k) issue_one(r, k, std::false_type{});
    };
    // Ring protocol of chain.hip (plain rule): barrier k = chunk k has landed for everybody and everybody is done with chunk k - 1, whose
    // buffer takes chunk k + 2 - issued anywhere in iteration k, but BEFORE the iteration's global stores (counted waits: [refill, stores])
    int gc = 0, st1 = 0, st2 = 0;
    auto advance = [&]() __attribute__((always_inline)) -> const char* {
        if constexpr (DUTY) {
            // the issuer of chunk gc has nothing younger in its queue than the stores it issued since (chunk gc + 2 comes in iteration gc)
            if (cw == (gc & 1)) wait_vmcnt_dyn(p.count_stores ? st1 + st2 : 0);      // stores may retire before an older DMA (chain.hip, advance()): drain, do not count them
        } else {
            constexpr int MAXC = NBUF2 - 2;
            int ahead = total - 1 - gc;
            const int rem = ahead;
            ahead = ahead < 0 ? 0 : (ahead > MAXC ? MAXC : ahead);
            if (st1 + st2 == 0) wait_chunks<PER, MAXC>(rem);
            else wait_vmcnt_dyn(PER * ahead + (p.count_stores ? st1 + st2 : 0));           // without the stores since: they may retire before an older DMA (chain.hip, advance())
        }
        C2_TICK(1);
        st2 = st1; st1 = 0;
        wg_barrier();
        C2_TICK(0);
        const char* buf = smem + (gc % NBUF2) * BUF;
        ++gc;
        return buf;
    };
    // the refill of iteration gc - 1 (chunk gc + 1); DUTY: only the waves whose turn it is (cw == parity of the iteration = of the chunk)
    auto refill = [&]() __attribute__((always_inline)) {
        if (DUTY && cw != ((gc - 1) & 1)) return;
        if (gc + NBUF2 - 2 < total) issue(gc + NBUF2 - 2);
        C2_TICK(11);
    };

    float* s_b0 = sf + cd.nf[0];
    float* s_ln = sf + cd.nf[1];
    float* s_f0b1 = sf + cd.nf[2];
    float* s_f0b2 = sf + cd.nf[3];
    float* s_f1b1 = sf + cd.nf[4];
    float* s_f1b2 = sf + cd.nf[5];
    float* s_g1b = sf + cd.nf[6];
    float* s_uv = sf + cd.nf[7];
    for (int i = wave; i < cd.nfl_kb; i += NW2) glds16(reinterpret_cast<const char*>(p.consts) + (size_t)i * 1024 + lane * 16, reinterpret_cast<char*>(sf) + i * 1024);
#pragma unroll
    for (int c = 0; c < NBUF2 - 1; ++c)
        if (c < total && (!DUTY || cw == (c & 1))) issue(c);

    // ---- this wave's state: NTH residual tiles (its column half), the whole normalised row as B fragments
    f32x16 xc[NTH];
    bf16x8 xf[KS];
    const int ct0 = cw * NTH;                                // first tile / (x 2) first k-step of the wave's column half

    // own fragments -> both waves of the pair hold xf[0 .. KS): through the staging regions, 4 fragments (4 KiB) per round.  Every wave reads BOTH
    // halves back from LDS (its own from its own region): a branch on cw around the register array would be if-converted into a dynamically
    // indexed store and put xf into scratch
    const char* stgA = stg_base + pr * S2_BYTES;             // staging of the pair's wave A / wave B
    const char* stgB = stg_base + (pr + 4) * S2_BYTES;
    auto publish_xf = [&](const bf16x8 (&own)[KSH]) __attribute__((always_inline)) {
#pragma unroll
        for (int r0 = 0; r0 < KSH; r0 += 4) {
            if (r0 > 0) wg_barrier();                        // the partner has read the previous round
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (r0 + i < KSH) *reinterpret_cast<bf16x8*>(stg + i * 1024 + lane * 16) = own[r0 + i];
            wg_barrier();
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (r0 + i < KSH) {
                    xf[r0 + i] = *reinterpret_cast<const bf16x8*>(stgA + i * 1024 + lane * 16);
                    xf[KSH + r0 + i] = *reinterpret_cast<const bf16x8*>(stgB + i * 1024 + lane * 16);
                }
        }
        wg_barrier();                                        // the staging regions are private again
    };

    // ---- rows in: x (the wave's column half, fp32) and, for the g0 kinds, the bf16 operand rows (each wave its half of the k-steps, then exchanged)
    {
        const char* xb = reinterpret_cast<const char*>(p.X);
        u32x4 vx[4 * NTH] = {};
        static_for<0, NTH>([&](auto I) { constexpr int tt = decltype(I)::value; s2_load<4 * tt>(xb, (size_t)p.ldx * 4, D * 4, m_base, p.M, 128 * (ct0 + tt), lane, vx, (p.nt & 1) != 0); });
        constexpr int NWA = (KSH + 3) / 4;                   // 128-byte windows (4 k-steps) of the wave's half of the operand row
        u32x4 va[4 * NWA] = {};
        if constexpr (ISB || PRE) {
            const char* ab = reinterpret_cast<const char*>(p.A);
            static_for<0, NWA>([&](auto I) { constexpr int w = decltype(I)::value; s2_load<4 * w>(ab, (size_t)p.lda * 2, p.lda * 2, m_base, p.M, cw * KSH * 32 + 128 * w, lane, va, (p.nt & 1) != 0); });
        }
        static_for<0, NTH>([&](auto I) {
            constexpr int tt = decltype(I)::value;
            wave_sync();
            s2_put<4 * tt>(stg, lane, vx);
            wave_sync();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int col = 32 * (ct0 + tt) + 8 * q + 4 * half;
                float4 x4 = *reinterpret_cast<const float4*>(stg + lr * S2_ROW + (q * 8 + half * 4) * 4);
                if (col >= D) x4 = make_float4(0.f, 0.f, 0.f, 0.f);
                xc[tt][4 * q + 0] = x4.x; xc[tt][4 * q + 1] = x4.y; xc[tt][4 * q + 2] = x4.z; xc[tt][4 * q + 3] = x4.w;
            }
        });
        if constexpr (ISB || PRE) {
            bf16x8 own[KSH];
            static_for<0, NWA>([&](auto I) {
                constexpr int w = decltype(I)::value;
                wave_sync();
                s2_put<4 * w>(stg, lane, va);
                wave_sync();
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if constexpr (true) if (4 * w + j < KSH) {
                        const int s = cw * KSH + 4 * w + j;
                        const char* src = stg + lr * S2_ROW + (16 * j + 4 * half) * 2;
                        uint2 lo = *reinterpret_cast<const uint2*>(src), hi = *reinterpret_cast<const uint2*>(src + 16);
                        const int c0 = 16 * s + 4 * half;
                        if (c0 >= D || m_base + lr >= p.M) lo = make_uint2(0u, 0u);
                        if (c0 + 8 >= D || m_base + lr >= p.M) hi = make_uint2(0u, 0u);
                        own[4 * w + j] = as_bf16x8(make_uint4(lo.x, lo.y, hi.x, hi.y));
                    }
                }
            });
            wave_sync();
            publish_xf(own);
        }
    }
    // constants and the first chunk visible to everybody
    if (!DUTY && total >= NBUF2) wait_vmcnt<PER * (NBUF2 - 2)>(); else wait_vmcnt<0>();
    wg_barrier();
    C2_TICK(6);

    const int q0 = (half + lr) % P1;
    const int w1row = lr * (P1 * 16);
    auto wfrag = [&](const char* slab, int s) __attribute__((always_inline)) {
        int q = q0 + 2 * s;
        q -= q >= P1 ? P1 : 0;
        return *reinterpret_cast<const bf16x8*>(slab + w1row + q * 16);
    };
    const int k2 = (half + (lr >> 2)) & 3;
    const int w2off0 = lr * 64 + k2 * 16 + ct0 * 2048, w2off1 = lr * 64 + (k2 ^ 2) * 16 + ct0 * 2048;      // the wave's column tiles of a W2 slab

    auto add_cvec2 = [&](const float* sv) __attribute__((always_inline)) {       // xc[tt][r] += sv[column]
#pragma unroll
        for (int tt = 0; tt < NTH; ++tt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = *reinterpret_cast<const float4*>(sv + 32 * (ct0 + tt) + 8 * q + 4 * half);
                xc[tt][4 * q + 0] += v.x; xc[tt][4 * q + 1] += v.y; xc[tt][4 * q + 2] += v.z; xc[tt][4 * q + 3] += v.w;
            }
    };
    // LayerNorm statistics in chain.hip's summation order: wave A sums its tiles, wave B continues from A's partial and finishes (the xor-32
    // shuffle, the division, the rsqrt), and hands the result back - four barriers per norm.  eps 1e-6 (modules.py:377, 447; blocks.py:97)
    auto ln_stats2 = [&](float& mean, float& rstd) __attribute__((always_inline)) {
        float* my = reinterpret_cast<float*>(stg + S2_WIN) + lane;
        const float* pa = reinterpret_cast<const float*>(stgp + S2_WIN) + lane;
        auto psum = [&](float sum) __attribute__((always_inline)) {
#pragma unroll
            for (int t = 0; t < NTH; ++t)
#pragma unroll
                for (int r = 0; r < 16; r += 4) sum += (xc[t][r] + xc[t][r + 1]) + (xc[t][r + 2] + xc[t][r + 3]);
            return sum;
        };
        auto pvar = [&](float var, float mu) __attribute__((always_inline)) {
#pragma unroll
            for (int t = 0; t < NTH; ++t)
#pragma unroll
                for (int r = 0; r < 16; r += 4) {
                    const float a = xc[t][r] - mu, b = xc[t][r + 1] - mu, c = xc[t][r + 2] - mu, d = xc[t][r + 3] - mu;
                    var += (a * a + b * b) + (c * c + d * d);
                }
            return var;
        };
        mean = 0.f; rstd = 0.f;
        if (cw == 0) *my = psum(0.f);
        wg_barrier();
        if (cw == 1) { const float sum = psum(*pa); mean = (sum + __shfl_xor(sum, 32)) / (float)D; *my = mean; }
        wg_barrier();
        if (cw == 0) { mean = *pa; *my = pvar(0.f, mean); }
        wg_barrier();
        if (cw == 1) {
            float var = pvar(*pa, mean);
            var += __shfl_xor(var, 32);
            var -= (float)(32 * NT - D) * mean * mean;
            rstd = rsqrtf(fmaxf(var, 0.f) / (float)D + 1e-6f);
            *my = rstd;
        }
        wg_barrier();
        if (cw == 0) rstd = *pa;
        asm volatile("" : "+v"(mean));
        C2_TICK(5);
    };
    // bf16((x - mean) * rstd) of the wave's columns as K-permuted B fragments, then both halves to both waves
    auto norm_xf = [&](float mean, float rstd) __attribute__((always_inline)) {
        const float nm = -mean * rstd;
        bf16x8 own[KSH];
#pragma unroll
        for (int s = 0; s < KSH; ++s) {
            const int r = 8 * (s & 1);
            own[s] = as_bf16x8(make_uint4(pack_bf2(fmaf(xc[s >> 1][r + 0], rstd, nm), fmaf(xc[s >> 1][r + 1], rstd, nm)),
                                          pack_bf2(fmaf(xc[s >> 1][r + 2], rstd, nm), fmaf(xc[s >> 1][r + 3], rstd, nm)),
                                          pack_bf2(fmaf(xc[s >> 1][r + 4], rstd, nm), fmaf(xc[s >> 1][r + 5], rstd, nm)),
                                          pack_bf2(fmaf(xc[s >> 1][r + 6], rstd, nm), fmaf(xc[s >> 1][r + 7], rstd, nm))));
        }
        publish_xf(own);
        C2_TICK(5);
    };
    auto store_x2 = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int tt = 0; tt < NTH; ++tt) {
            wave_sync();
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<float4*>(stg + lr * S2_ROW + (q * 8 + half * 4) * 4) = make_float4(xc[tt][4 * q + 0], xc[tt][4 * q + 1], xc[tt][4 * q + 2], xc[tt][4 * q + 3]);
            wave_sync();
            s2_store(stg, reinterpret_cast<char*>(p.Y), (size_t)p.ldy * 4, D * 4, m_base, p.M, 128 * (ct0 + tt), lane, (p.nt & 2) != 0);
        }
        st1 += 4 * NTH;
        C2_TICK(10);
    };

    // ---- stage: x += g0(A).  Ring chunk c carries the weight rows of tiles 2c and 2c + 1: wave A's tiles come first, then wave B's
    if constexpr (ISB || PRE) {
        add_cvec2(s_b0);
#pragma unroll
        for (int c = 0; c < (NT + 1) / 2; ++c) {
            const char* buf = advance();
            refill();
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                constexpr int FB = (KS % 8 == 0) ? 8 : 4;
                const int t = 2 * c + j;                     // compile-time after unrolling
                if (t < NT && cw == t / NTH) {
                    const int tt = t % NTH;
#pragma unroll
                    for (int s0 = 0; s0 < KS; s0 += FB) {
                        bf16x8 wa[FB];
#pragma unroll
                        for (int i = 0; i < FB; ++i) wa[i] = wfrag(buf + j * HALF, s0 + i);
#pragma unroll
                        for (int i = 0; i < FB; ++i) xc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[i], xf[s0 + i], xc[tt], 0, 0, 0);
                    }
                }
            }
            if constexpr (PROF) asm volatile("s_nop 0" :: "v"(xc[0][0]), "v"(xc[NTH - 1][15]));
            C2_TICK(7);
        }
    }

    // ---- FFN stage: x += 1/2 FFN(LN(x)).  Iteration j (ring chunk j): the owner of hidden chunk j (cw == j & 1) runs its first GEMM and Swish and
    //      publishes hf(j); both waves run the second GEMM of chunk j - 1 on their own column tiles (hf(j - 1): the owner's registers / the
    //      partner's staging region, visible since this iteration's barrier)
    constexpr int FB1 = 4, G1 = KS / FB1;
    static_assert(G1 == PER, "one wave-DMA behind every MFMA group of the first GEMM");
    auto gemm1 = [&](const char* buf, const float* b1, auto after) __attribute__((always_inline)) -> f32x16 {
        f32x16 h;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v = *reinterpret_cast<const float4*>(b1 + 8 * q);
            h[4 * q + 0] = v.x; h[4 * q + 1] = v.y; h[4 * q + 2] = v.z; h[4 * q + 3] = v.w;
        }
#pragma unroll
        for (int s0 = 0; s0 < KS; s0 += FB1) {
            bf16x8 wa[FB1];
#pragma unroll
            for (int i = 0; i < FB1; ++i) wa[i] = wfrag(buf, s0 + i);
#pragma unroll
            for (int i = 0; i < FB1; ++i) h = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[i], xf[s0 + i], h, 0, 0, 0);
            after(s0 / FB1);
        }
        return h;
    };
    auto gemm2 = [&](const char* w2, const bf16x8 hf0, const bf16x8 hf1) __attribute__((always_inline)) {
#pragma unroll
        for (int tt = 0; tt < NTH; ++tt) {
            const bf16x8 wb0 = *reinterpret_cast<const bf16x8*>(w2 + tt * 2048 + w2off0);
            const bf16x8 wb1 = *reinterpret_cast<const bf16x8*>(w2 + tt * 2048 + w2off1);
            xc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb0, hf0, xc[tt], 0, 0, 0);
            xc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb1, hf1, xc[tt], 0, 0, 0);
        }
    };
    auto no_hook = [](int) {};
    auto ffn_stage = [&](const float* sb1, const float* sb2, int n) __attribute__((always_inline)) {
        float mean, rstd;
        ln_stats2(mean, rstd);
        norm_xf(mean, rstd);
        add_cvec2(sb2);
        bf16x8 ho0 = {}, ho1 = {};                           // hf of the wave's latest own chunk
        // own half of an iteration: first GEMM of chunk j (the refill's wave-DMAs behind its MFMA groups, or in a burst before it), Swish, publish -
        // and, in the SAME basic block (the compiler interleaves its MFMAs with the Swish's VALU), the second GEMM of chunk j - 1
        auto own_iter = [&](const char* buf, int j, auto first) __attribute__((always_inline)) {
            constexpr bool FIRST = decltype(first)::value;
            bf16x8 u0 = {}, u1 = {};
            if constexpr (!FIRST) { u0 = *reinterpret_cast<const bf16x8*>(stgp + lane * 16); u1 = *reinterpret_cast<const bf16x8*>(stgp + 1024 + lane * 16); }
            f32x16 h;
            if constexpr (DMA == 3 && !FIRST) {
                // the second GEMM of chunk j - 1 (independent accumulators) between the MFMAs of the first GEMM's dependent chain: one column tile
                // (two MFMAs) behind every group of FB1
                static_assert(NTH <= G1, "a column tile per MFMA group");
                h = gemm1(buf, sb1 + j * CH + 4 * half, [&](int g) __attribute__((always_inline)) {
                    if (g < NTH) {
                        const bf16x8 wb0 = *reinterpret_cast<const bf16x8*>(buf + HALF + g * 2048 + w2off0);
                        const bf16x8 wb1 = *reinterpret_cast<const bf16x8*>(buf + HALF + g * 2048 + w2off1);
                        xc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb0, u0, xc[g], 0, 0, 0);
                        xc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb1, u1, xc[g], 0, 0, 0);
                    }
                });
            } else if (DMA == 1 && gc + NBUF2 - 2 < total) {
                const Refill rf = issue_prep(gc + NBUF2 - 2);
                h = gemm1(buf, sb1 + j * CH + 4 * half, [&](int g) __attribute__((always_inline)) { issue_one(rf, g, std::true_type{}); });
            } else {
                refill();
                h = gemm1(buf, sb1 + j * CH + 4 * half, no_hook);
            }
            if constexpr (PROF) asm volatile("s_nop 0" :: "v"(h[0]), "v"(h[15]));
            C2_TICK(2);
            uint32_t w[8];
#pragma unroll
            for (int r = 0; r < 16; r += 2) w[r >> 1] = pack_bf2(swishf_(h[r]), swishf_(h[r + 1]));
            ho0 = as_bf16x8(make_uint4(w[0], w[1], w[2], w[3])); ho1 = as_bf16x8(make_uint4(w[4], w[5], w[6], w[7]));
            if constexpr (!FIRST && DMA != 3) gemm2(buf + HALF, u0, u1);
            *reinterpret_cast<bf16x8*>(stg + lane * 16) = ho0;
            *reinterpret_cast<bf16x8*>(stg + 1024 + lane * 16) = ho1;
            if constexpr (PROF) asm volatile("s_nop 0" :: "v"(xc[0][0]), "v"(xc[NTH - 1][15]));
            C2_TICK(3);
        };
        // owner of hidden chunk j: cw == (j & 1) - DUTY: the wave whose turn it is NOT to refill in iteration j (global ring index gc + j)
        const int op = DUTY ? ((gc + 1) & 1) : 0;            // owner(j) = the waves with cw == ((j + op) & 1)
        {   // j = 0
            const char* buf = advance();
            if (cw == op) own_iter(buf, 0, std::true_type{}); else refill();
        }
        for (int j = 1; j < n; ++j) {
            const char* buf = advance();
            if (((j + op) & 1) == cw) own_iter(buf, j, std::false_type{});
            else { refill(); gemm2(buf + HALF, ho0, ho1); if constexpr (PROF) asm volatile("s_nop 0" :: "v"(xc[0][0]), "v"(xc[NTH - 1][15])); C2_TICK(4); }
        }
        {   // j = n: the second GEMM of the last hidden chunk
            const char* buf = advance();
            refill();
            if (((n - 1 + op) & 1) == cw) gemm2(buf + HALF, ho0, ho1);
            else gemm2(buf + HALF, *reinterpret_cast<const bf16x8*>(stgp + lane * 16), *reinterpret_cast<const bf16x8*>(stgp + 1024 + lane * 16));
            if constexpr (PROF) asm volatile("s_nop 0" :: "v"(xc[0][0]), "v"(xc[NTH - 1][15]));
            C2_TICK(4);
        }
    };

    // bias of the two 32-row slabs of ring chunk c -> accumulators
    auto acc_bias = [&](f32x16 (&acc)[2], int c) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = *reinterpret_cast<const float4*>(s_g1b + 64 * c + 32 * j + 8 * q + 4 * half);
                acc[j][4 * q + 0] = v.x; acc[j][4 * q + 1] = v.y; acc[j][4 * q + 2] = v.z; acc[j][4 * q + 3] = v.w;
            }
    };
    auto g1_mfma = [&](f32x16 (&acc)[2], const char* buf) __attribute__((always_inline)) {
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
    };

    if constexpr (ISB) {
        // ---- conv-module pre-norm, pointwise-1 + GLU -> bf16 rows (modules.py:512-514).  GLU chunks alternate between the pair; the owner of chunk c
        //      writes it out during chunk c + 1
        float mean, rstd;
        ln_stats2(mean, rstd);
        norm_xf(mean, rstd);
        f32x16 acc[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        auto glu_out = [&](int c) __attribute__((always_inline)) {
            wave_sync();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float o[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = acc[0][4 * q + i] * sigmoidf_(acc[1][4 * q + i]);
                *reinterpret_cast<uint2*>(stg + lr * S2_ROW + (8 * q + 4 * half) * 2) = make_uint2(pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3]));
            }
            wave_sync();
            const int col = 32 * c + 8 * (lane & 3);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = 16 * i + (lane >> 2), m = m_base + row;
                const u32x4 v = *reinterpret_cast<const u32x4*>(stg + row * S2_ROW + 16 * (lane & 3));
                if (m < p.M && col < p.Ng) *reinterpret_cast<u32x4*>(p.glu + (size_t)m * p.ldg + col) = v;
            }
            st1 += 2;
        };
        const int op = DUTY ? (gc & 1) : 0;                  // owner(c) = the waves with cw == ((c + op) & 1): DUTY - the wave that refills in that iteration
        for (int c = 0; c < n_g1; ++c) {
            const char* buf = advance();
            refill();
            if (((c + op) & 1) == cw) { acc_bias(acc, c); g1_mfma(acc, buf); if constexpr (PROF) asm volatile("s_nop 0" :: "v"(acc[0][0]), "v"(acc[1][15])); C2_TICK(8); }
            else if (c >= 1) { glu_out(c - 1); C2_TICK(9); }
        }
        if (n_g1 >= 1 && ((n_g1 - 1 + op) & 1) == cw) glu_out(n_g1 - 1);
    } else {
        if constexpr (PRE) {
            ffn_stage(s_f0b1, s_f0b2, n_f0);                                        // FFN2 of the previous block
            float mean, rstd;
            ln_stats2(mean, rstd);
            // block output = LayerNorm(x)  (blocks.py:135)
            const float* sg = s_ln + 32 * ct0 + 4 * half;
            const float* sb = s_ln + DP + 32 * ct0 + 4 * half;
            int ofs = 0;
#pragma unroll
            for (int t = 0; t < NTH; ++t) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 g = *reinterpret_cast<const float4*>(sg + ofs + 32 * t + 8 * q), b = *reinterpret_cast<const float4*>(sb + ofs + 32 * t + 8 * q);
                    xc[t][4 * q + 0] = (xc[t][4 * q + 0] - mean) * rstd * g.x + b.x;
                    xc[t][4 * q + 1] = (xc[t][4 * q + 1] - mean) * rstd * g.y + b.y;
                    xc[t][4 * q + 2] = (xc[t][4 * q + 2] - mean) * rstd * g.z + b.z;
                    xc[t][4 * q + 3] = (xc[t][4 * q + 3] - mean) * rstd * g.w + b.w;
                }
                asm volatile("" : "+v"(ofs) : "v"(xc[t][15]));
            }
        }
        if constexpr (POST) {
            ffn_stage(s_f1b1, s_f1b2, n_f1);                                        // FFN1 of this block
            float mean, rstd;
            ln_stats2(mean, rstd);
            norm_xf(mean, rstd);                                                    // attention pre-norm
            store_x2();                                                             // x is final: its registers are free during the Q/K/V stage
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
            f32x16 acc[2];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
            // registers r = 8g .. 8g+7 of tile j are the columns 64c + 32j + 16g + 8*half + (0..7) of the stacked [Q | K | V] (row permutation of
            // pack_linear_chunkperm); Q columns get + u (Q + v is derived in the attention kernel)
            auto qkv_out = [&](int c) __attribute__((always_inline)) {
                const float* su = s_uv;
                wave_sync();
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                        const int n0 = 64 * c + 32 * j + 16 * g + 8 * half;
                        float4 ua = make_float4(0.f, 0.f, 0.f, 0.f), ub = ua;
                        if (n0 < D) { ua = *reinterpret_cast<const float4*>(su + n0); ub = *reinterpret_cast<const float4*>(su + n0 + 4); }
                        *reinterpret_cast<uint4*>(stg + lr * S2_ROW + (32 * j + 16 * g + 8 * half) * 2) =
                            make_uint4(pack_bf2(acc[j][8 * g + 0] + ua.x, acc[j][8 * g + 1] + ua.y), pack_bf2(acc[j][8 * g + 2] + ua.z, acc[j][8 * g + 3] + ua.w),
                                       pack_bf2(acc[j][8 * g + 4] + ub.x, acc[j][8 * g + 5] + ub.y), pack_bf2(acc[j][8 * g + 6] + ub.z, acc[j][8 * g + 7] + ub.w));
                    }
                wave_sync();
                const int n0 = 64 * c + pc8;
                if ((D & 7) == 0) {
                    const int which = cd.fD.div(n0), nn0 = n0 - which * D;
                    bf16_t* dst = which == 0 ? p.qu : (which == 1 ? p.kh : p.vt);
                    const bool colok = n0 < 3 * D;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const u32x4 v = *reinterpret_cast<const u32x4*>(stg + (8 * i + (lane >> 3)) * S2_ROW + 16 * (lane & 7));
                        if (colok && qok[i]) { if (p.nt & 2) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(dst + qoff[i] + nn0)); else *reinterpret_cast<u32x4*>(dst + qoff[i] + nn0) = v; }
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
                        const u32x4 v = *reinterpret_cast<const u32x4*>(stg + (8 * i + (lane >> 3)) * S2_ROW + 16 * (lane & 7));
                        if (oka && qok[i]) *reinterpret_cast<u32x2*>(da + qoff[i] + na) = u32x2{v[0], v[1]};
                        if (okb && qok[i]) *reinterpret_cast<u32x2*>(db + qoff[i] + nb) = u32x2{v[2], v[3]};
                    }
                    st1 += 8;
                }
            };
            const int op = DUTY ? (gc & 1) : 0;              // owner(c) = the waves with cw == ((c + op) & 1): DUTY - the wave that refills in that iteration
            for (int c = 0; c < n_g1; ++c) {
                const char* buf = advance();
                refill();
                if (((c + op) & 1) == cw) { acc_bias(acc, c); g1_mfma(acc, buf); if constexpr (PROF) asm volatile("s_nop 0" :: "v"(acc[0][0]), "v"(acc[1][15])); C2_TICK(8); }
                else if (c >= 1) { qkv_out(c - 1); C2_TICK(9); }
            }
            if (n_g1 >= 1 && ((n_g1 - 1 + op) & 1) == cw) qkv_out(n_g1 - 1);
        }
    }
    // ---- residual rows out
    if constexpr (!POST) store_x2();
    if constexpr (PROF) {
        C2_TICK(12);
        if (lane == 0 && pr == 0 && (blockIdx.x & 7) == 0) {
            for (int i = 0; i < 13; ++i) atomicAdd(prof + 16 * cw + i, ph[i]);
            atomicAdd(prof + 16 * cw + 15, 1ull);
        }
    }
#undef C2_TICK
}

unsigned long long* g_chain2_prof = nullptr;
void chain2_prof_dump() {
    unsigned long long h[32];
    if (!g_chain2_prof || hipMemcpy(h, g_chain2_prof, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess || !h[15]) return;
    static const char* names[13] = {"barrier", "vmcnt wait", "ffn gemm1", "ffn swish+gemm2", "ffn gemm2 only", "LN + exchange", "prologue", "g0 mfma", "g1 mfma", "g1 write-out", "store_x", "refill issue", "tail"};
    for (int w = 0; w < 2; ++w) {
        const unsigned long long* q = h + 16 * w;
        unsigned long long tot = 0;
        for (int i = 0; i < 13; ++i) tot += q[i];
        fprintf(stderr, "[chain2 phases] %s wave %c: waves %llu, cycles/wave %.0f\n", getenv("EFFCONF_CHAIN2_PHASES"), w ? 'B' : 'A', q[15], (double)tot / q[15]);
        for (int i = 0; i < 13; ++i) fprintf(stderr, "[chain2 phases]   %-16s %10.0f cyc/wave  %5.1f%%\n", names[i], (double)q[i] / q[15], 100.0 * q[i] / tot);
    }
}

template <int KS, int KIND, int DMA>
int launch_chain2_t(const ChainParams& p, hipStream_t s) {
    using G = Geo2<KS>;
    ChainDev2 cd;
    cd.p = p;
    cd.fT = FastDiv32(p.T > 0 ? p.T : 1);
    cd.fD = FastDiv32(p.D);
    {
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
    const int lds = NBUF2 * G::BUF + NW2 * S2_BYTES + nfl * 4;
    if (lds > 160 * 1024) return -4;
    static LdsAttr attr;
    ensure_dynamic_lds(reinterpret_cast<const void*>(&chain2_kernel<KS, KIND, DMA, false>), lds, attr);
    if constexpr (DMA >= 1) {
#ifdef EFFCONF_PHASE_PROF    // in-kernel phase profiles: a tuning build (tools/build_ablate.py); the product has neither the getenv nor the profiling instantiation
        static const bool prof = getenv("EFFCONF_CHAIN2_PHASES") != nullptr && atoi(getenv("EFFCONF_CHAIN2_PHASES")) == DMA * 1000 + KS * 10 + KIND;
#else
        constexpr bool prof = false;
#endif
        if (prof) {
            if (!g_chain2_prof) {
                if (hipMalloc(&g_chain2_prof, 256) != hipSuccess || hipMemset(g_chain2_prof, 0, 256) != hipSuccess) return -1;
                atexit(chain2_prof_dump);
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&chain2_kernel<KS, KIND, DMA, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            }
            hipLaunchKernelGGL((chain2_kernel<KS, KIND, DMA, true>), dim3((p.M + 127) / 128), dim3(NW2 * 64), lds, s, cd, g_chain2_prof);
            return hipGetLastError() == hipSuccess ? 0 : -1;
        }
    }
    hipLaunchKernelGGL((chain2_kernel<KS, KIND, DMA, false>), dim3((p.M + 127) / 128), dim3(NW2 * 64), lds, s, cd, nullptr);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

template <int KIND>
int launch_chain2_kind(const ChainParams& p, hipStream_t s) {
    const int ks = chain_padded_width(p.D) / 16;
    // option chain_pair: 1 = burst refills by every wave, 2 = the FFN owner's refill behind its MFMA groups, 3 = refills in turns (the wave with the
    // short half of the iteration), 4 = 3 + the second GEMM between the MFMAs of the first
#define C2_MODES(KSV) switch (p.pair) { case 1: return launch_chain2_t<KSV, KIND, 0>(p, s); case 2: return launch_chain2_t<KSV, KIND, 1>(p, s); \
                                        case 3: return launch_chain2_t<KSV, KIND, 2>(p, s); default: return launch_chain2_t<KSV, KIND, 3>(p, s); }   /* 5: chain3.hip where it exists, mode 4 here */
    if (ks == 12) C2_MODES(12)
    if (ks == 16) C2_MODES(16)
#undef C2_MODES
    return -2;
}

}  // namespace

bool chain2_supported(int D) { const int ks = chain_padded_width(D) / 16; return chain_supported(D) && (ks == 12 || ks == 16); }

// LDS: ring + staging + constant block (the combined tail + head block of D = 256 with 4 D hidden units: 150 KiB)
int launch_chain2(const ChainParams& p, int kind, hipStream_t s) {
    if (p.M <= 0) return 0;
    if (!chain2_supported(p.D)) return -2;
    switch (kind) {
        case CHAIN_B: return launch_chain2_kind<CHAIN_B>(p, s);
        case CHAIN_A_FULL: return launch_chain2_kind<CHAIN_A_FULL>(p, s);
        case CHAIN_A_HEAD: return launch_chain2_kind<CHAIN_A_HEAD>(p, s);
        case CHAIN_A_TAIL: return launch_chain2_kind<CHAIN_A_TAIL>(p, s);
    }
    return -3;
}
