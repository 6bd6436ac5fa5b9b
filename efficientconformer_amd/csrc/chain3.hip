// Role-split form of chain A for the widest stages (padded width 256: D = 193 .. 256), THREE waves per SIMD (gfx950).
//
// What the phase profiles of chain.hip / chain2.hip say about the D = 240 stage (profiles/r5_02_*, r5_04_*): a 1-KiB LDS-DMA blocks the
// issuing wave for 130 - 185 cycles (a wave streams 5.6 B / cycle, profiles/r4_03_lds_fill_rate.txt), a hidden chunk is 32 of them, and in
// a kernel whose waves all hold 256 registers the waves that issue the weight stream are the waves that issue the MFMAs: whatever the
// split of the work (one wave per SIMD; a column pair per SIMD, hidden chunks in turns, refills in turns), an iteration costs the sum
// - ~2000 cycles for 1024 cycles of matrix work.  The stream has to be issued by waves that have nothing else to do, i.e. the row tile has
// to be spread over MORE waves with FEWER registers each.  Here 32 rows belong to three waves (w, w + 4, w + 8: one SIMD), <= 168 registers:
//   * wave A holds the normalised row as B fragments (xf, 64 registers) and runs the FIRST GEMM of every hidden chunk - 16 MFMAs of one
//     dependency chain - with the Swish of the PREVIOUS chunk interleaved (a second accumulator), and publishes the 32 x 32 bf16 tile hf
//     through LDS (2 KiB).  It never touches the weight stream;
//   * waves B0 / B1 hold the fp32 residual / accumulator tiles of one column half each (64 registers), run the SECOND GEMM of the chunk two
//     iterations back on them (8 independent MFMAs each) and issue the ring's LDS-DMAs (4 per wave and chunk): their blocked time is the
//     time they would otherwise spend waiting for A;
//   * LayerNorm statistics: B0 -> B1 -> back, in chain.hip's summation order; the normalised fragments go to A (and, for the Q/K/V stage,
//     to all three) through LDS; Q/K/V chunks rotate over the three waves - MFMAs, then the two column tiles written out in the next two
//     iterations, one each, beside the next owners' MFMAs.
// Ring chunk i of an FFN stage = [W1 rows of hidden chunk i | W2 slab (chunk-major image) of hidden chunk i - 2]: n + 2 ring chunks for n
// hidden chunks.  Every accumulator sees chain.hip's operations in chain.hip's order: the rows are bit-identical (tests/test_gpu_round5.py).
// The two roles are separate code paths from the top (a common path would keep A's fragments and B's accumulators live in each other's
// code: 190 registers); they execute the same sequence of workgroup barriers - kept in one table below (PROTOCOL).
// Reference: models/modules.py:385-392, 519-522; blocks.py:119-137; attentions.py:651-686.
#include "kernels.h"
#include "rowstat.h"

#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace {

constexpr int NW3 = 12, NBUF3 = 3, NB3 = 8;                  // waves; ring buffers; waves that issue the ring (the B waves)
constexpr int S3_ROW = 144;                                  // staging window: 32 rows x (128 + 16) bytes
constexpr int S3_WIN = 32 * S3_ROW;                          // 4608: one window per B wave; [0, 4096) also carries hf slots / fragment exchanges
constexpr int S3_TILE = 2 * S3_WIN + 512 + 256;              // per row tile: window 0, window 1, LayerNorm hand-off slots, the sink of wave A's L2 prefetches

template <int KS>
struct Geo3 {
    static_assert(KS % 4 == 0, "an even number of 32-column tiles per B wave");
    static constexpr int NT = KS / 2, NTH = NT / 2, KSH = KS / 2;
    static constexpr int DP = 16 * KS, P1 = KS * 2;
    static constexpr int HALF = CH * P1 * 16, BUF = 2 * HALF;
    static constexpr int PER = 2 * KS / NB3;
    static constexpr int XR = (KSH + 3) / 4;                 // rounds of a fragment exchange (4 fragments per B wave and round)
};

struct ChainDev3 {
    ChainParams p;
    FastDiv32 fT, fD;
    int nf[8];
    int nfl_kb;
    int ldr;
};

// `ofs` (feeding the next addresses) made to depend on a whole accumulator tile.  Device pass only: on the host side of the compilation a 64-byte "v" operand is
// not a valid x86 constraint, and the failed instantiation silently leaves the kernel's host stub undefined
#if defined(__HIP_DEVICE_COMPILE__)
#define C3_PIN_TILE(ofs, tile) asm volatile("" : "+v"(ofs) : "v"(tile))
#else
#define C3_PIN_TILE(ofs, tile) ((void)0)
#endif

template <int V> using ic3 = std::integral_constant<int, V>;
template <int I, int N, class F> __device__ __forceinline__ void static_for3(F&& f) {
    if constexpr (I < N) { f(ic3<I>{}); static_for3<I + 1, N>(f); }
}

template <int OFF, int N>
__device__ __forceinline__ void s3_load(const char* base, size_t pitch, int row_bytes, int m_base, int M, int wbyte, int lane, u32x4 (&v)[N]) {
    int cb = wbyte + 16 * (lane & 7);
    cb = cb < row_bytes - 16 ? cb : row_bytes - 16;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m_base + 8 * i + (lane >> 3);
        v[OFF + i] = *reinterpret_cast<const u32x4*>(base + (size_t)(m < M ? m : M - 1) * pitch + cb);
    }
}
template <int OFF, int N>
__device__ __forceinline__ void s3_put(char* stg, int lane, const u32x4 (&v)[N]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(stg + (8 * i + (lane >> 3)) * S3_ROW + 16 * (lane & 7)) = v[OFF + i];
}
__device__ __forceinline__ void s3_store(const char* stg, char* base, size_t pitch, int row_bytes, int m_base, int M, int wbyte, int lane) {
    const int cb = wbyte + 16 * (lane & 7);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m_base + 8 * i + (lane >> 3);
        const u32x4 v = *reinterpret_cast<const u32x4*>(stg + (8 * i + (lane >> 3)) * S3_ROW + 16 * (lane & 7));
        if (m < M && cb < row_bytes) *reinterpret_cast<u32x4*>(base + (size_t)m * pitch + cb) = v;
    }
}

// PROTOCOL - the workgroup barriers both roles execute, in order (XB = 2 XR barriers of a fragment exchange):
//   prologue:  [XB if PRE]  1 (constants + first chunk)
//   g0 stage:  n_g0 advances                                       (PRE)
//   FFN stage: 4 (statistics)  XB  n + 2 advances                  (PRE: ffn0, POST: ffn1)
//   block norm: 4                                                  (PRE)
//   Q/K/V:     4  XB  n_g1 advances  2 (drain)                     (POST)
// PROF (tuning only, EFFCONF_CHAIN3_PHASES=<kind>): s_memtime per phase, the three waves of every 8th workgroup's first row tile
template <int KS, int KIND, bool PROF = false>
__global__ __launch_bounds__(NW3 * 64, 1) void chain3_kernel(const ChainDev3 cd, unsigned long long* prof = nullptr) {
    unsigned long long ph[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, t0 = 0;
    if constexpr (PROF) t0 = __builtin_readcyclecounter();
#define C3_TICK(i) do { if constexpr (PROF) { asm volatile("" ::: "memory"); const unsigned long long t1_ = __builtin_readcyclecounter(); ph[i] += t1_ - t0; t0 = t1_; } } while (0)
#define C3_DUMP() do { if constexpr (PROF) { C3_TICK(9); if (lane == 0 && pr == 0 && (blockIdx.x & 7) == 0) { for (int i = 0; i < 10; ++i) atomicAdd(prof + 16 * role + i, ph[i]); atomicAdd(prof + 16 * role + 15, 1ull); } } } while (0)
    if constexpr (PROF) {          // which SIMD does wave w of a 12-wave workgroup run on?  (HW_REG_HW_ID bits 5:4; workgroup 0 reports)
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) prof[48 + (threadIdx.x >> 6)] = ((hwid >> 4) & 3) | (((hwid >> 8) & 15) << 4) | (1u << 16);
    }
    using G = Geo3<KS>;
    constexpr int NT = G::NT, NTH = G::NTH, KSH = G::KSH, P1 = G::P1, HALF = G::HALF, BUF = G::BUF, PER = G::PER, DP = G::DP, XR = G::XR;
    constexpr bool PRE = KIND == CHAIN_A_FULL || KIND == CHAIN_A_TAIL;
    constexpr bool POST = KIND == CHAIN_A_FULL || KIND == CHAIN_A_HEAD;
    static_assert(KIND != CHAIN_B, "chain A only");
    const ChainParams& p = cd.p;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* tile_base = smem + NBUF3 * BUF;
    float* sf = reinterpret_cast<float*>(tile_base + 4 * S3_TILE);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pr = wave & 3, role = wave >> 2;               // row tile; 0 = A, 1 / 2 = B0 / B1 (waves w, w + 4, w + 8 share a SIMD)
    const int lr = lane & 31, half = lane >> 5;
    const int m_base = (blockIdx.x * 4 + pr) * 32;
    char* win0 = tile_base + pr * S3_TILE;                   // the row tile's two windows and its hand-off slots
    char* win1 = win0 + S3_WIN;
    float* slot0 = reinterpret_cast<float*>(win0 + 2 * S3_WIN) + lane;       // B0's hand-off slot, B1's
    float* slot1 = slot0 + 64;
    const int D = p.D;

    const int n_g0 = PRE ? (NT + 1) / 2 : 0;
    const int n_f0 = PRE ? p.f[0].Fp / CH : 0;
    const int n_f1 = POST ? p.f[1].Fp / CH : 0;
    const int n_g1 = POST ? p.g1.nchunks : 0;
    const int r_f0 = n_f0 ? n_f0 + 2 : 0, r_f1 = n_f1 ? n_f1 + 2 : 0;
    const int e0 = n_g0, e1 = e0 + r_f0, e2 = e1 + r_f1, total = e2 + n_g1;

    const float* s_b0 = sf + cd.nf[0];
    const float* s_ln = sf + cd.nf[1];
    const float* s_f0b1 = sf + cd.nf[2];
    const float* s_f0b2 = sf + cd.nf[3];
    const float* s_f1b1 = sf + cd.nf[4];
    const float* s_f1b2 = sf + cd.nf[5];
    const float* s_g1b = sf + cd.nf[6];
    const float* s_uv = sf + cd.nf[7];

    const int q0 = (half + lr) % P1;
    const int w1row = lr * (P1 * 16);
    auto wfrag = [&](const char* slab, int s) __attribute__((always_inline)) {
        int q = q0 + 2 * s;
        q -= q >= P1 ? P1 : 0;
        return *reinterpret_cast<const bf16x8*>(slab + w1row + q * 16);
    };
    // every wave reads the published fragments of a round: fragments r0 .. r0 + 3 from window 0, KSH + r0 .. from window 1
    auto read_round = [&](bf16x8 (&xf)[KS], auto r0c) __attribute__((always_inline)) {
        constexpr int r0 = decltype(r0c)::value;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (r0 + i < KSH) {
                xf[r0 + i] = *reinterpret_cast<const bf16x8*>(win0 + i * 1024 + lane * 16);
                xf[KSH + r0 + i] = *reinterpret_cast<const bf16x8*>(win1 + i * 1024 + lane * 16);
            }
    };
    // destination rows of the Q/K/V write-out: (b, t) -> (b * Tp + t) * D for the 2 rows a lane stores per half-tile instruction pair
    // bias of the two 32-row slabs of ring chunk c -> accumulators; the MFMAs of a Q/K/V chunk
    auto acc_bias = [&](f32x16 (&acc)[2], int c) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = *reinterpret_cast<const float4*>(s_g1b + 64 * c + 32 * j + 8 * q + 4 * half);
                acc[j][4 * q + 0] = v.x; acc[j][4 * q + 1] = v.y; acc[j][4 * q + 2] = v.z; acc[j][4 * q + 3] = v.w;
            }
    };
    auto g1_mfma = [&](f32x16 (&acc)[2], const bf16x8 (&xf)[KS], const char* buf) __attribute__((always_inline)) {
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
    // tile j (32 columns) of Q/K/V chunk c out through window `stg`: registers r = 8g .. 8g + 7 of the tile are the columns 64c + 32j + 16g + 8 half + (0..7)
    // of the stacked [Q | K | V] (row permutation of pack_linear_chunkperm); Q columns get + u.  Returns the number of store instructions
    auto qkv_out_tile = [&](const f32x16& a, int c, int j, char* stg) __attribute__((always_inline)) -> int {
        wave_sync();
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int n0 = 64 * c + 32 * j + 16 * g + 8 * half;
            float4 ua = make_float4(0.f, 0.f, 0.f, 0.f), ub = ua;
            if (n0 < D) { ua = *reinterpret_cast<const float4*>(s_uv + n0); ub = *reinterpret_cast<const float4*>(s_uv + n0 + 4); }
            *reinterpret_cast<uint4*>(stg + lr * S3_ROW + (16 * g + 8 * half) * 2) =
                make_uint4(pack_bf2(a[8 * g + 0] + ua.x, a[8 * g + 1] + ua.y), pack_bf2(a[8 * g + 2] + ua.z, a[8 * g + 3] + ua.w),
                           pack_bf2(a[8 * g + 4] + ub.x, a[8 * g + 5] + ub.y), pack_bf2(a[8 * g + 6] + ub.z, a[8 * g + 7] + ub.w));
        }
        wave_sync();
        const int n0 = 64 * c + 32 * j + 8 * (lane & 3);
        int nst = 0;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = 16 * i + (lane >> 2), m = m_base + row;
            const int mc = m < p.M ? m : p.M - 1;
            const int b = cd.fT.div(mc), t = mc - b * p.T;
            const size_t qoff = ((size_t)b * p.Tp + t) * D;
            const u32x4 v = *reinterpret_cast<const u32x4*>(stg + row * S3_ROW + 16 * (lane & 3));
            if ((D & 7) == 0) {
                const int which = cd.fD.div(n0), nn0 = n0 - which * D;
                bf16_t* dst = which == 0 ? p.qu : (which == 1 ? p.kh : p.vt);
                if (n0 < 3 * D && m < p.M) *reinterpret_cast<u32x4*>(dst + qoff + nn0) = v;
                nst += 1;
            } else {                               // D % 8 == 4: two 8-byte halves, each inside one tensor
                typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                const int wa = cd.fD.div(n0), wb = cd.fD.div(n0 + 4);
                const int na = n0 - wa * D, nb = n0 + 4 - wb * D;
                bf16_t* da = wa == 0 ? p.qu : (wa == 1 ? p.kh : p.vt);
                bf16_t* db = wb == 0 ? p.qu : (wb == 1 ? p.kh : p.vt);
                if (n0 < 3 * D && m < p.M) *reinterpret_cast<u32x2*>(da + qoff + na) = u32x2{v[0], v[1]};
                if (n0 + 4 < 3 * D && m < p.M) *reinterpret_cast<u32x2*>(db + qoff + nb) = u32x2{v[2], v[3]};
                nst += 2;
            }
        }
        return nst;
    };
    // The Q/K/V stage, common to the three roles (each holds xf): chunk c belongs to the wave with role == (c + op) % 3, which runs its MFMAs in
    // iteration c and writes its two column tiles out in iterations c + 1 (window 0) and c + 2 (window 1).  adv() = the role's ring advance,
    // rf() its refill (a no-op for A), st = its store counter
    auto qkv_stage = [&](const bf16x8 (&xf)[KS], auto adv, auto rf, int& st, int op) __attribute__((always_inline)) {
        f32x16 acc[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        int mine = (role + 3 - op) % 3;                      // first chunk of this wave
#pragma unroll 1
        for (int c = 0; c < n_g1 + 2; ++c) {
            const char* buf = nullptr;
            if (c < n_g1) buf = adv(); else wg_barrier();
            const int ph = c - mine;                         // 0: MFMAs of chunk `mine`, 1 / 2: its column tiles out
            if (ph == 0 && c < n_g1) { acc_bias(acc, c); g1_mfma(acc, xf, buf); rf(); }
            else {
                rf();
                if (ph == 1 && mine < n_g1) st += qkv_out_tile(acc[0], mine, 0, win0);
                else if (ph == 2 && mine < n_g1) { st += qkv_out_tile(acc[1], mine, 1, win1); mine += 3; }
            }
        }
    };

    if (role == 0) {
        // =================================================================== role A: first GEMM + Swish of every hidden chunk
        int gc = 0;
        // L2 prefetch.  Every weight chunk is a FIRST touch for the XCD's L2 (the workgroups of a launch walk the weight stream in step, nothing is reused
        // later, and the activation traffic of the launch evicts it before the next one): with two chunks of LDS prefetch the ring fill waited for memory-side
        // latency in every iteration (vmcnt wait = 1/3 of a B wave's life, profiles/r5_10_*).  Wave A, which idles at the barriers more than half of its
        // life, touches every 128-byte line of chunk gc + PFD once per iteration: one 4-byte LDS-DMA per lane (4 A waves x 64 lanes = the chunk's 256 lines)
        // into a sink nobody reads - no destination register, nothing to wait for until the kernel ends
        const int PFD = (p.nt >> 4) & 15;                    // tuning (option chain_nt, bits 4..7): prefetch distance in chunks, 0 = off
        char* sink = win0 + 2 * S3_WIN + 512;
        const uint32_t pf_off = (uint32_t)(((pr * 64 + lane) & 127) * 128);
        auto prefetch = [&](int c) __attribute__((always_inline)) {
            if (!(PFD > 0 && c < total)) return;
            const bool ffn = c >= e0 && c < e2;
            const bool second = c >= e1;
            const int cf = c - (second ? e1 : e0);
            const int nh = second ? n_f1 : n_f0;
            int c1 = cf < nh ? cf : nh - 1, c2 = cf - 2;
            c1 = c1 > 0 ? c1 : 0; c2 = c2 > 0 ? c2 : 0;
            const bf16_t* fw1 = second ? p.f[1].w1 : p.f[0].w1;
            const bf16_t* fw2 = second ? p.f[1].w2cm : p.f[0].w2cm;
            const bool first_g = c < e0;
            const bf16_t* gw = first_g ? p.g0.w : p.g1.w;
            const int cg = first_g ? c : c - e2;
            const char* w1 = reinterpret_cast<const char*>(fw1 + (size_t)c1 * CH * cd.ldr);
            const char* w2 = reinterpret_cast<const char*>(fw2 + (size_t)c2 * (DP * 32));
            const char* w = reinterpret_cast<const char*>(gw + (size_t)(cg > 0 ? cg : 0) * 64 * cd.ldr);
            const char* lo = ffn ? w1 : w;
            const char* hi = ffn ? w2 : w + (size_t)32 * cd.ldr * 2;
            const char* base = (pr & 2) ? hi : lo;            // A waves 0, 1: the chunk's first 16 KiB (128 lines), 2, 3: its second
            const uint32_t l = (uint32_t)(uintptr_t)(lds_void_t*)sink;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2" ::"s"(l), "v"(pf_off), "s"(base) : "memory", "m0");
        };
        auto advanceA = [&]() __attribute__((always_inline)) -> const char* {
            wg_barrier();
            C3_TICK(0);
            const char* buf = smem + (gc % NBUF3) * BUF;
            ++gc;
            prefetch(gc + NBUF3 - 2 + PFD);
            return buf;
        };
        auto no_rf = []() {};
        for (int c = NBUF3 - 1; c < NBUF3 - 1 + PFD; ++c) prefetch(c);
        bf16x8 xf[KS];
        if constexpr (PRE) {
#pragma unroll
            for (int r = 0; r < 2 * XR; ++r) wg_barrier();     // the B waves exchange the operand fragments
        }
        wg_barrier();
        if constexpr (PRE)
#pragma unroll 1
            for (int c = 0; c < n_g0; ++c) (void)advanceA();
        auto gemm1 = [&](const char* buf, const float* b1, auto between) __attribute__((always_inline)) -> f32x16 {
            f32x16 h;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = *reinterpret_cast<const float4*>(b1 + 8 * q);
                h[4 * q + 0] = v.x; h[4 * q + 1] = v.y; h[4 * q + 2] = v.z; h[4 * q + 3] = v.w;
            }
            constexpr int FB = 4;
#pragma unroll
            for (int s0 = 0; s0 < KS; s0 += FB) {
                bf16x8 wa[FB];
#pragma unroll
                for (int i = 0; i < FB; ++i) wa[i] = wfrag(buf, s0 + i);
#pragma unroll
                for (int i = 0; i < FB; ++i) h = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[i], xf[s0 + i], h, 0, 0, 0);
                between(s0 / FB);
            }
            return h;
        };
        auto swish_out = [&](const f32x16& h, char* slot) __attribute__((always_inline)) {
            uint32_t w[8];
#pragma unroll
            for (int r = 0; r < 16; r += 2) w[r >> 1] = pack_bf2(swishf_(h[r]), swishf_(h[r + 1]));
            *reinterpret_cast<bf16x8*>(slot + lane * 16) = as_bf16x8(make_uint4(w[0], w[1], w[2], w[3]));
            *reinterpret_cast<bf16x8*>(slot + 1024 + lane * 16) = as_bf16x8(make_uint4(w[4], w[5], w[6], w[7]));
        };
        auto ffn_stage_a = [&](const float* sb1, int n) __attribute__((always_inline)) {
#pragma unroll
            for (int r = 0; r < 4; ++r) wg_barrier();         // statistics (B0 -> B1 -> B0)
            static_for3<0, XR>([&](auto I) {                  // normalised fragments from the B waves
                constexpr int r = decltype(I)::value;
                if (r > 0) wg_barrier();
                wg_barrier();
                read_round(xf, ic3<4 * r>{});
            });
            wg_barrier();
            // iteration i: first GEMM of hidden chunk i (i < n) with the Swish of chunk i - 1 in the same basic block; hf(i - 1) -> window (i - 1) & 1
            f32x16 hp;
            {   // i = 0
                const char* buf = advanceA();
                hp = gemm1(buf, sb1 + 4 * half, [](int) {});
            }
#pragma unroll 1
            for (int i = 1; i < n; ++i) {
                const char* buf = advanceA();
                if (p.nt & 256) continue;                    // timing-only ablation (option chain_nt bit 8): no first GEMM / Swish
                const f32x16 hn = gemm1(buf, sb1 + i * CH + 4 * half, [](int) {});
                swish_out(hp, ((i - 1) & 1) ? win1 : win0);
                hp = hn;
                if constexpr (PROF) asm volatile("s_nop 0" :: "v"(hp[0]), "v"(hp[15]));
                C3_TICK(2);
            }
            {   // i = n
                (void)advanceA();
                swish_out(hp, ((n - 1) & 1) ? win1 : win0);
            }
            (void)advanceA();                                 // i = n + 1: the B waves' last second GEMM
        };
        if constexpr (PRE) {
            ffn_stage_a(s_f0b1, n_f0);
#pragma unroll
            for (int r = 0; r < 4; ++r) wg_barrier();         // block norm statistics
        }
        if constexpr (POST) {
            ffn_stage_a(s_f1b1, n_f1);
#pragma unroll
            for (int r = 0; r < 4; ++r) wg_barrier();         // attention pre-norm statistics
            static_for3<0, XR>([&](auto I) {
                constexpr int r = decltype(I)::value;
                if (r > 0) wg_barrier();
                wg_barrier();
                read_round(xf, ic3<4 * r>{});
            });
            wg_barrier();
            int st = 0;
            C3_TICK(5);
            qkv_stage(xf, advanceA, no_rf, st, gc % 3);
            C3_TICK(8);
        }
        wait_vmcnt<0>();                                      // the prefetches write LDS: nothing may be in flight when the workgroup's LDS is handed on
        C3_DUMP();
        return;
    }

    // ======================================================================= role B: residual / accumulator tiles of one column half, the weight ring
    const int cw = role - 1, bidx = wave - 4;
    const int ct0 = cw * NTH;
    char* stg = cw ? win1 : win0;                            // this wave's window
    float* my = cw ? slot1 : slot0;
    const float* pa = cw ? slot0 : slot1;
    uint32_t off_r[PER], off_f[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int i = bidx + NB3 * k;
        off_r[k] = i < KS ? dma_rows32_off<P1>(cd.ldr, i, lane) : dma_rows32_off<P1>(cd.ldr, i - KS, lane) + (uint32_t)(32 * cd.ldr) * 2u;
        off_f[k] = i < KS ? off_r[k] : (uint32_t)((i - KS) * 1024 + lane * 16);        // second FFN weights: chunk-major images (contiguous slabs)
    }
    auto issue = [&](int c) __attribute__((always_inline)) {
        const bool ffn = c >= e0 && c < e2;
        const bool second = c >= e1;
        const int cf = c - (second ? e1 : e0);
        const int nh = second ? n_f1 : n_f0;
        int c1 = cf < nh ? cf : nh - 1, c2 = cf - 2;
        c1 = c1 > 0 ? c1 : 0; c2 = c2 > 0 ? c2 : 0;
        const bf16_t* fw1 = second ? p.f[1].w1 : p.f[0].w1;
        const bf16_t* fw2 = second ? p.f[1].w2cm : p.f[0].w2cm;
        const bool first_g = c < e0;
        const bf16_t* gw = first_g ? p.g0.w : p.g1.w;
        const int cg = first_g ? c : c - e2;
        const char* w1 = reinterpret_cast<const char*>(fw1 + (size_t)c1 * CH * cd.ldr);
        const char* w2 = reinterpret_cast<const char*>(fw2 + (size_t)c2 * (DP * 32));
        const char* w = reinterpret_cast<const char*>(gw + (size_t)(cg > 0 ? cg : 0) * 64 * cd.ldr);
        const char* b_lo = ffn ? w1 : w;
        const char* b_hi = ffn ? w2 : w;
        char* buf = smem + (c % NBUF3) * BUF;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = bidx + NB3 * k;
            glds16(i < KS ? b_lo : b_hi, ffn ? off_f[k] : off_r[k], buf + 1024 * i);
        }
    };
    int gc = 0, st1 = 0, st2 = 0;
    auto advance = [&]() __attribute__((always_inline)) -> const char* {
        constexpr int MAXC = NBUF3 - 2;
        int ahead = total - 1 - gc;
        const int rem = ahead;
        ahead = ahead < 0 ? 0 : (ahead > MAXC ? MAXC : ahead);
        if (st1 + st2 == 0) wait_chunks<PER, MAXC>(rem);
        else wait_vmcnt_dyn(PER * ahead + (p.count_stores ? st1 + st2 : 0));               // without the stores since: they may retire before an older DMA (chain.hip, advance())
        C3_TICK(1);
        st2 = st1; st1 = 0;
        wg_barrier();
        C3_TICK(0);
        const char* buf = smem + (gc % NBUF3) * BUF;
        ++gc;
        return buf;
    };
    auto refill = [&]() __attribute__((always_inline)) {
        if (gc + NBUF3 - 2 < total) issue(gc + NBUF3 - 2);
        C3_TICK(3);
    };
    for (int i = bidx; i < cd.nfl_kb; i += NB3) glds16(reinterpret_cast<const char*>(p.consts) + (size_t)i * 1024 + lane * 16, reinterpret_cast<char*>(sf) + i * 1024);
#pragma unroll
    for (int c = 0; c < NBUF3 - 1; ++c)
        if (c < total) issue(c);

    f32x16 xc[NTH];
    // own fragments -> window (4 per round); `reader` waves read both windows
    auto publish = [&](const bf16x8 (&own)[KSH], bf16x8 (&xf)[KS], bool reader) __attribute__((always_inline)) {
        static_for3<0, XR>([&](auto I) {
            constexpr int r = decltype(I)::value;
            if (r > 0) wg_barrier();
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (4 * r + i < KSH) *reinterpret_cast<bf16x8*>(stg + i * 1024 + lane * 16) = own[4 * r + i];
            wg_barrier();
            if (reader) read_round(xf, ic3<4 * r>{});
        });
        wg_barrier();
    };
    auto ln_stats = [&](float& mean, float& rstd) __attribute__((always_inline)) {
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
    };
    auto norm_own = [&](float mean, float rstd, bf16x8 (&own)[KSH]) __attribute__((always_inline)) {
        const float nm = -mean * rstd;
#pragma unroll
        for (int s = 0; s < KSH; ++s) {
            const int r = 8 * (s & 1);
            own[s] = as_bf16x8(make_uint4(pack_bf2(fmaf(xc[s >> 1][r + 0], rstd, nm), fmaf(xc[s >> 1][r + 1], rstd, nm)),
                                          pack_bf2(fmaf(xc[s >> 1][r + 2], rstd, nm), fmaf(xc[s >> 1][r + 3], rstd, nm)),
                                          pack_bf2(fmaf(xc[s >> 1][r + 4], rstd, nm), fmaf(xc[s >> 1][r + 5], rstd, nm)),
                                          pack_bf2(fmaf(xc[s >> 1][r + 6], rstd, nm), fmaf(xc[s >> 1][r + 7], rstd, nm))));
        }
    };
    auto add_cvec = [&](const float* sv) __attribute__((always_inline)) {
        int ofs = 32 * ct0 + 4 * half;
#pragma unroll
        for (int tt = 0; tt < NTH; ++tt) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = *reinterpret_cast<const float4*>(sv + ofs + 32 * tt + 8 * q);
                xc[tt][4 * q + 0] += v.x; xc[tt][4 * q + 1] += v.y; xc[tt][4 * q + 2] += v.z; xc[tt][4 * q + 3] += v.w;
            }
            // the next tile's addresses depend on this tile's result: keeps the compiler from reading the whole vector first (64 registers on top of
            // the rows and the operand fragments)
            C3_PIN_TILE(ofs, xc[tt]);
        }
    };
    auto store_x = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int tt = 0; tt < NTH; ++tt) {
            wave_sync();
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<float4*>(stg + lr * S3_ROW + (q * 8 + half * 4) * 4) = make_float4(xc[tt][4 * q + 0], xc[tt][4 * q + 1], xc[tt][4 * q + 2], xc[tt][4 * q + 3]);
            wave_sync();
            s3_store(stg, reinterpret_cast<char*>(p.Y), (size_t)p.ldy * 4, D * 4, m_base, p.M, 128 * (ct0 + tt), lane);
        }
        st1 += 4 * NTH;
    };

    // ---- rows in
    {
        const char* xb = reinterpret_cast<const char*>(p.X);
        u32x4 vx[4 * NTH] = {};
        static_for3<0, NTH>([&](auto I) { constexpr int tt = decltype(I)::value; s3_load<4 * tt>(xb, (size_t)p.ldx * 4, D * 4, m_base, p.M, 128 * (ct0 + tt), lane, vx); });
        static_for3<0, NTH>([&](auto I) {
            constexpr int tt = decltype(I)::value;
            wave_sync();
            s3_put<4 * tt>(stg, lane, vx);
            wave_sync();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int col = 32 * (ct0 + tt) + 8 * q + 4 * half;
                float4 x4 = *reinterpret_cast<const float4*>(stg + lr * S3_ROW + (q * 8 + half * 4) * 4);
                if (col >= D) x4 = make_float4(0.f, 0.f, 0.f, 0.f);
                xc[tt][4 * q + 0] = x4.x; xc[tt][4 * q + 1] = x4.y; xc[tt][4 * q + 2] = x4.z; xc[tt][4 * q + 3] = x4.w;
            }
        });
    }
    if constexpr (PRE) {
        // ---- x += g0(A): the bf16 operand rows (each B wave loads half of the k-steps, both end up with all of them), then the wave's own column tiles
        bf16x8 xa[KS];
        {
            // one exchange round per 128-byte window (4 k-steps) of the wave's half of the operand row: load, fragments, window, barrier, both halves back.
            // Window by window (not all loads first): with xc and the growing xa live, a second set of staging registers does not fit 168
            static_assert(XR == (KSH + 3) / 4, "one round per window");
            const char* ab = reinterpret_cast<const char*>(p.A);
            static_for3<0, XR>([&](auto I) {
                constexpr int w = decltype(I)::value;
                u32x4 va[4] = {};
                s3_load<0>(ab, (size_t)p.lda * 2, p.lda * 2, m_base, p.M, cw * KSH * 32 + 128 * w, lane, va);
                if (w > 0) wg_barrier();                     // the previous round has been read
                wave_sync();
                s3_put<0>(stg, lane, va);
                wave_sync();
                bf16x8 own4[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int s = cw * KSH + 4 * w + j;
                    const char* src = stg + lr * S3_ROW + (16 * j + 4 * half) * 2;
                    uint2 lo = *reinterpret_cast<const uint2*>(src), hi = *reinterpret_cast<const uint2*>(src + 16);
                    const int c0 = 16 * s + 4 * half;
                    if (c0 >= D || m_base + lr >= p.M) lo = make_uint2(0u, 0u);
                    if (c0 + 8 >= D || m_base + lr >= p.M) hi = make_uint2(0u, 0u);
                    own4[j] = as_bf16x8(make_uint4(lo.x, lo.y, hi.x, hi.y));
                }
                wave_sync();
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (4 * w + j < KSH) *reinterpret_cast<bf16x8*>(stg + j * 1024 + lane * 16) = own4[j];
                wg_barrier();
                read_round(xa, ic3<4 * w>{});
            });
            wg_barrier();
        }
        if (total >= NBUF3) wait_vmcnt<PER * (NBUF3 - 2)>(); else wait_vmcnt<0>();
        wg_barrier();
        add_cvec(s_b0);
#pragma unroll
        for (int c = 0; c < (NT + 1) / 2; ++c) {
            const char* buf = advance();
            refill();
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                constexpr int FB = 2;                        // xc + the operand fragments are 128 registers here: small batches keep the stage inside 168
                const int t = 2 * c + j;
                if (t < NT && cw == t / NTH) {
                    const int tt = t % NTH;
#pragma unroll
                    for (int s0 = 0; s0 < KS; s0 += FB) {
                        bf16x8 wa[FB];
#pragma unroll
                        for (int i = 0; i < FB; ++i) wa[i] = wfrag(buf + j * HALF, s0 + i);
#pragma unroll
                        for (int i = 0; i < FB; ++i) xc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[i], xa[s0 + i], xc[tt], 0, 0, 0);
                    }
                }
            }
        }
    } else {
        if (total >= NBUF3) wait_vmcnt<PER * (NBUF3 - 2)>(); else wait_vmcnt<0>();
        wg_barrier();
    }

    const int k2 = (half + (lr >> 2)) & 3;
    const int w2off0 = lr * 64 + k2 * 16 + ct0 * 2048, w2off1 = lr * 64 + (k2 ^ 2) * 16 + ct0 * 2048;
    auto gemm2 = [&](const char* w2, const bf16x8 hf0, const bf16x8 hf1) __attribute__((always_inline)) {
#pragma unroll
        for (int tt = 0; tt < NTH; ++tt) {
            const bf16x8 wb0 = *reinterpret_cast<const bf16x8*>(w2 + tt * 2048 + w2off0);
            const bf16x8 wb1 = *reinterpret_cast<const bf16x8*>(w2 + tt * 2048 + w2off1);
            xc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb0, hf0, xc[tt], 0, 0, 0);
            xc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb1, hf1, xc[tt], 0, 0, 0);
        }
    };
    bf16x8 xnone[KS];                                        // publish() target of a wave that does not read
    auto ffn_stage_b = [&](const float* sb2, int n) __attribute__((always_inline)) {
        float mean, rstd;
        ln_stats(mean, rstd);
        {
            bf16x8 own[KSH];
            norm_own(mean, rstd, own);
            publish(own, xnone, false);
        }
        add_cvec(sb2);
        // iteration i: second GEMM of hidden chunk i - 2 (hf from window (i - 2) & 1, published by A in iteration i - 1), and the refill
        const bool rf_first = (p.nt & 4) == 0;               // tuning (option chain_nt, bit 4): the refill behind the second GEMM instead of in front of it
#pragma unroll 1
        for (int i = 0; i < n + 2; ++i) {
            const char* buf = advance();
            if (rf_first && !(p.nt & 1024)) refill();          // bit 10 (timing-only ablation): no weight stream
            if (i >= 2 && !(p.nt & 512)) {                   // bit 9 (timing-only ablation): no second GEMM
                const char* slot = ((i - 2) & 1) ? win1 : win0;
                const bf16x8 h0 = *reinterpret_cast<const bf16x8*>(slot + lane * 16), h1 = *reinterpret_cast<const bf16x8*>(slot + 1024 + lane * 16);
                gemm2(buf + HALF, h0, h1);
                if constexpr (PROF) asm volatile("s_nop 0" :: "v"(xc[0][0]), "v"(xc[NTH - 1][15]));
            }
            C3_TICK(2);
            if (!rf_first) refill();
        }
    };
    if constexpr (PRE) {
        ffn_stage_b(s_f0b2, n_f0);
        float mean, rstd;
        ln_stats(mean, rstd);
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
            C3_PIN_TILE(ofs, xc[t]);
        }
    }
    if constexpr (POST) {
        ffn_stage_b(s_f1b2, n_f1);
        float mean, rstd;
        ln_stats(mean, rstd);
        bf16x8 xf[KS];
        {
            bf16x8 own[KSH];
            norm_own(mean, rstd, own);
            store_x();                                       // x is final: its registers are free before the full fragment set arrives
            publish(own, xf, true);
        }
        C3_TICK(5);
        qkv_stage(xf, advance, refill, st1, gc % 3);
        C3_TICK(8);
    } else {
        store_x();
    }
    C3_DUMP();
#undef C3_TICK
#undef C3_DUMP
}

unsigned long long* g_chain3_prof = nullptr;
void chain3_prof_dump() {
    unsigned long long h[64];
    if (!g_chain3_prof || hipMemcpy(h, g_chain3_prof, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess || !h[15]) return;
    fprintf(stderr, "[chain3 phases] wave -> SIMD (CU) of workgroup 0:");
    for (int w = 0; w < 12; ++w) fprintf(stderr, " %d:%llu(%llu)", w, h[48 + w] & 3, (h[48 + w] >> 4) & 15);
    fprintf(stderr, "\n");
    static const char* names[10] = {"barrier", "vmcnt wait (B)", "ffn compute", "refill issue (B)", "-", "LN / exchange / g0 / rows", "-", "-", "qkv stage", "rest"};
    for (int w = 0; w < 3; ++w) {
        const unsigned long long* q = h + 16 * w;
        if (!q[15]) continue;
        unsigned long long tot = 0;
        for (int i = 0; i < 10; ++i) tot += q[i];
        fprintf(stderr, "[chain3 phases] kind %s role %d: waves %llu, cycles/wave %.0f\n", getenv("EFFCONF_CHAIN3_PHASES"), w, q[15], (double)tot / q[15]);
        for (int i = 0; i < 10; ++i) if (q[i]) fprintf(stderr, "[chain3 phases]   %-26s %10.0f cyc/wave  %5.1f%%\n", names[i], (double)q[i] / q[15], 100.0 * q[i] / tot);
    }
}

template <int KS, int KIND>
int launch_chain3_t(const ChainParams& p, hipStream_t s) {
    using G = Geo3<KS>;
    ChainDev3 cd;
    cd.p = p;
    cd.fT = FastDiv32(p.T > 0 ? p.T : 1);
    cd.fD = FastDiv32(p.D);
    {
        constexpr bool pre = KIND == CHAIN_A_FULL || KIND == CHAIN_A_TAIL, post = KIND == CHAIN_A_FULL || KIND == CHAIN_A_HEAD;
        int ldr = 0;
        bool ok = true;
        auto row = [&](int ld) { if (!ldr) ldr = ld; else ok = ok && ld == ldr; };
        if (pre) { row(p.g0.ldw); row(p.f[0].ldw1); ok = ok && p.f[0].w2cm; }
        if (post) { row(p.g1.ldw); row(p.f[1].ldw1); ok = ok && p.f[1].w2cm; }
        if (!ok || ldr <= 0) return -6;
        cd.ldr = ldr;
    }
    const int nfl = chain_const_layout(p, KIND, cd.nf);
    cd.nfl_kb = nfl / 256;
    if (!p.consts) return -5;
    const int lds = NBUF3 * G::BUF + 4 * S3_TILE + nfl * 4;
    if (lds > 160 * 1024) return -4;
    static LdsAttr attr;
    ensure_dynamic_lds(reinterpret_cast<const void*>(&chain3_kernel<KS, KIND, false>), lds, attr);
#ifdef EFFCONF_PHASE_PROF    // in-kernel phase profiles: a tuning build (tools/build_ablate.py); the product has neither the getenv nor the profiling instantiation
    static const bool prof = getenv("EFFCONF_CHAIN3_PHASES") != nullptr && atoi(getenv("EFFCONF_CHAIN3_PHASES")) == KIND;
#else
    constexpr bool prof = false;
#endif
    if (prof) {
        if (!g_chain3_prof) {
            if (hipMalloc(&g_chain3_prof, 512) != hipSuccess || hipMemset(g_chain3_prof, 0, 512) != hipSuccess) return -1;
            atexit(chain3_prof_dump);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&chain3_kernel<KS, KIND, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        }
        hipLaunchKernelGGL((chain3_kernel<KS, KIND, true>), dim3((p.M + 127) / 128), dim3(NW3 * 64), lds, s, cd, g_chain3_prof);
        return hipGetLastError() == hipSuccess ? 0 : -1;
    }
    hipLaunchKernelGGL((chain3_kernel<KS, KIND, false>), dim3((p.M + 127) / 128), dim3(NW3 * 64), lds, s, cd, nullptr);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace

bool chain3_supported(int D) { return chain_supported(D) && chain_padded_width(D) == 256; }

int launch_chain3(const ChainParams& p, int kind, hipStream_t s) {
    if (p.M <= 0) return 0;
    if (!chain3_supported(p.D)) return -2;
    switch (kind) {
        case CHAIN_A_FULL: return launch_chain3_t<16, CHAIN_A_FULL>(p, s);
        case CHAIN_A_HEAD: return launch_chain3_t<16, CHAIN_A_HEAD>(p, s);
        case CHAIN_A_TAIL: return launch_chain3_t<16, CHAIN_A_TAIL>(p, s);
    }
    return -3;
}
