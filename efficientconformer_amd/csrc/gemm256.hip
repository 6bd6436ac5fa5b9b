// Wide-model bf16 MFMA GEMM (gfx950): 256 x BN x 64 block tiles with BOTH operands streamed by LDS-DMA.
//
//   C[m, n] = sum_k A[m, k] * W[n, k] + bias[n]      (same operands and epilogues as gemm.hip; launch_gemm picks this kernel)
//
// Serves the Linear / pointwise-conv layers of the widths the row-stationary kernels do not hold (D = 512, 720: EfficientConformer
// Large stages 2-3, Conformer Large; reference models/layers.py:57-67, call sites modules.py:378-382, 502-508, attentions.py:57-60).
//
//   * 8 waves as 4 (M) x 2 (N); a wave owns 64 x BN/2 of the tile: 2 x BN/64 accumulators of v_mfma_f32_32x32x16_bf16.
//   * Two LDS buffers of [256 + BN rows][128 bytes]; k-tile t+1 is written by global_load_lds_dwordx4 (no staging registers, no
//     ds_write pass) while k-tile t feeds the MFMAs: one s_waitcnt vmcnt(0) + s_barrier per k-tile.
//   * The DMA destination is lane-linear, so the bank swizzle lives on the SOURCE address: the 16-byte slot s of tile row r holds
//     the row's k-chunk s ^ ((r >> 1) & 7).  A fragment read (lane = row lr, k-chunk 2 kk + half) is conflict-free for the
//     ds_read_b128 lane groups of this chip (16 lanes: 8 even + 8 odd rows, all (row & 1, slot) pairs distinct), and every DMA
//     instruction still fetches 8 whole 128-byte row segments.
//   * K tail (K % 64 != 0, K % 8 == 0): chunks at or beyond column K are fetched from a 16-byte zero block instead of the row, so
//     neither stale workspace bytes nor the next row's data reach the MFMA (the packed weights are zero there as well).
//   * Epilogues leave through a per-wave LDS staging region as 16-byte pieces (bf16 rows, fp32 residual rows, GLU, natural-layout
//     Q+u | K | V), 32 rows at a time.
#include "rowstat.h"

namespace {

constexpr int G_BM = 256, G_BK = 64, G_NW = 8;
constexpr int G_ROWB = G_BK * 2;   // bytes per tile row

__device__ __attribute__((aligned(16))) uint32_t g_zero_chunk[4] = {0u, 0u, 0u, 0u};

struct G256Dev {
    GemmParams p;
    FastDiv32 fD;
    int n_tiles, m_tiles;
};

template <int BN, int EPI>
__global__ __launch_bounds__(G_NW * 64, 2) void gemm256_kernel(const G256Dev gd) {
    const GemmParams& p = gd.p;
    constexpr int NF = BN / 64, WCOLS = BN / 2;
    constexpr int A_BYTES = G_BM * G_ROWB, W_BYTES = BN * G_ROWB, BUF = A_BYTES + W_BYTES;
    constexpr int NA = G_BM / 8 / G_NW, NWD = BN / 8 / G_NW;     // DMA instructions (8 tile rows each) per wave and k-tile
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, half = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int id = xcd_remap(blockIdx.x, gd.m_tiles * gd.n_tiles);
    const int tm = id / gd.n_tiles, tn = id - tm * gd.n_tiles;
    const int m0 = tm * G_BM, n0 = tn * BN;

    // ---- per-lane DMA sources: instruction j of a tile covers tile rows 8j .. 8j+7, lane -> (row 8j + lane/8, slot lane%8)
    const char* pa[NA];
    const char* pw[NWD];
    int ca[NA];                                                  // first k column (inside a k-tile) of this lane's A chunk
#pragma unroll
    for (int k = 0; k < NA; ++k) {
        const int r = 8 * (wave * NA + k) + (lane >> 3);
        const int cs = (lane & 7) ^ ((r >> 1) & 7);
        int m = m0 + r;
        m = m < p.M ? m : p.M - 1;
        size_t src = m;
        if (p.a_rows > 0) {
            const int b = m / p.a_rows, rr = m - b * p.a_rows;
            src = (size_t)b * p.a_pitch + (size_t)rr * p.a_stride;
        }
        pa[k] = reinterpret_cast<const char*>(p.A + src * p.lda + cs * 8);
        ca[k] = cs * 8;
    }
#pragma unroll
    for (int k = 0; k < NWD; ++k) {
        const int r = 8 * (wave * NWD + k) + (lane >> 3);
        const int cs = (lane & 7) ^ ((r >> 1) & 7);
        int n = n0 + r;
        n = n < p.N ? n : p.N - 1;                               // rows past N: any in-bounds row (their columns are never stored)
        pw[k] = reinterpret_cast<const char*>(p.W + (size_t)n * p.ldw + cs * 8);
    }
    const int nk = (p.K + G_BK - 1) / G_BK;
    const bool ktail = (p.K & (G_BK - 1)) != 0;
    auto issue = [&](int t, int buf) __attribute__((always_inline)) {
        char* base = smem + buf * BUF;
        const size_t kb = (size_t)t * G_ROWB;
        if (ktail && t == nk - 1) {
            const int k0 = t * G_BK;
#pragma unroll
            for (int k = 0; k < NA; ++k) {
                const char* src = (k0 + ca[k] < p.K) ? pa[k] + kb : reinterpret_cast<const char*>(g_zero_chunk);
                glds16(src, base + (wave * NA + k) * 1024);
            }
        } else {
#pragma unroll
            for (int k = 0; k < NA; ++k) glds16(pa[k] + kb, base + (wave * NA + k) * 1024);
        }
#pragma unroll
        for (int k = 0; k < NWD; ++k) glds16(pw[k] + kb, base + A_BYTES + (wave * NWD + k) * 1024);
    };

    f32x16 acc[2][NF];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < NF; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    // fragment byte offsets inside a 32-row group: row lr, k-chunk 2 kk + half at slot (2 kk + half) ^ ((lr >> 1) & 7)
    const int s0 = (half ^ ((lr >> 1) & 7)) * 16;
    int foff[G_BK / 16];
#pragma unroll
    for (int kk = 0; kk < G_BK / 16; ++kk) foff[kk] = lr * G_ROWB + ((kk * 32) ^ s0);

    auto compute = [&](int buf) __attribute__((always_inline)) {
        const char* a = smem + buf * BUF + (wm * 64) * G_ROWB;
        const char* b = smem + buf * BUF + A_BYTES + (wn * WCOLS) * G_ROWB;
#pragma unroll
        for (int kk = 0; kk < G_BK / 16; ++kk) {
            bf16x8 af[2], bf[NF];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) af[mi] = *reinterpret_cast<const bf16x8*>(a + mi * 32 * G_ROWB + foff[kk]);
#pragma unroll
            for (int ni = 0; ni < NF; ++ni) bf[ni] = *reinterpret_cast<const bf16x8*>(b + ni * 32 * G_ROWB + foff[kk]);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < NF; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mi], bf[ni], acc[mi][ni], 0, 0, 0);
        }
    };

    issue(0, 0);
    for (int t = 0; t < nk; ++t) {
        wait_vmcnt<0>();                  // this wave's pieces of k-tile t have landed ...
        wg_barrier();                     // ... and everybody's; everybody is done reading the other buffer (k-tile t-1)
        if (t + 1 < nk) issue(t + 1, (t + 1) & 1);
        compute(t & 1);
    }
    wg_barrier();                         // the operand buffers are free: they become the staging regions below

    // ---- epilogue.  C layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    const int lcol = lr, lrow = 4 * half;
    const int nw0 = n0 + wn * WCOLS;       // first column of this wave
    if constexpr (EPI == EPI_BF16 || EPI == EPI_SWISH_BF16 || EPI == EPI_QKV_NAT) {
        constexpr int PITCH = WCOLS * 2 + 16, PPR = WCOLS / 8, RPI = 64 / PPR, NIT = 32 / RPI;
        char* wbuf = smem + wave * 32 * PITCH;
        float add[NF];
#pragma unroll
        for (int ni = 0; ni < NF; ++ni) {
            const int n = nw0 + ni * 32 + lcol, nc = n < p.N ? n : p.N - 1;
            add[ni] = p.bias[nc];
            if constexpr (EPI == EPI_QKV_NAT) { if (gd.fD.div(nc) == 0) add[ni] += p.u[nc]; }
        }
        const int piece = lane % PPR, prow = lane / PPR;
        const int nb = nw0 + piece * 8;
        const int Tp = p.Tg * p.G;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
            for (int ni = 0; ni < NF; ++ni) {
                const bool live = nw0 + ni * 32 + lcol < p.N;    // pad columns of the row buffers stay zero
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float val = acc[mi][ni][r] + add[ni];
                    if constexpr (EPI == EPI_SWISH_BF16) val = swishf_(val);
                    val = live ? val : 0.f;
                    const int row = (r & 3) + 8 * (r >> 2) + lrow;
                    *reinterpret_cast<bf16_t*>(wbuf + row * PITCH + (ni * 32 + lcol) * 2) = f2bf(val);
                }
            }
            wave_sync();
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int row = it * RPI + prow, m = m0 + wm * 64 + mi * 32 + row;
                const uint4 v = *reinterpret_cast<const uint4*>(wbuf + row * PITCH + piece * 16);
                if (m >= p.M) continue;
                if constexpr (EPI == EPI_QKV_NAT) {
                    if (nb >= p.N) continue;                     // D % 8 == 0 (launcher): a piece never straddles Q | K | V
                    const int which = gd.fD.div(nb), nn = nb - which * p.D;
                    const int b = m / p.T, t = m - b * p.T;
                    bf16_t* dst = which == 1 ? p.kh : (which == 2 ? p.vt : p.qu);
                    *reinterpret_cast<uint4*>(dst + ((size_t)b * Tp + t) * p.D + nn) = v;
                } else {
                    if (nb < p.ldc) *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.C) + (size_t)m * p.ldc + nb) = v;
                }
            }
            wave_sync();
        }
    } else if constexpr (EPI == EPI_GLU_BF16) {
        // W rows interleaved per 32 channels (a | b): fragment pairs (2q, 2q+1) of a wave are (a, b) of 32 output channels
        constexpr int NP = NF / 2, PITCH = NP * 64 + 16, PPR = NP * 4, RPI = 64 / PPR, NIT = 32 / RPI;
        static_assert(NF % 2 == 0, "GLU needs both halves in one wave");
        char* wbuf = smem + wave * 32 * PITCH;
        float ba[NP], bb[NP];
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const int na = nw0 + q * 64 + lcol, nbb = na + 32;
            ba[q] = p.bias[na < p.N ? na : p.N - 1];
            bb[q] = p.bias[nbb < p.N ? nbb : p.N - 1];
        }
        const int piece = lane % PPR, prow = lane / PPR;
        const int jb = nw0 / 2 + piece * 8;                      // first output channel of this lane's piece
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                const bool live = nw0 + q * 64 + lcol < p.N;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + lrow;
                    const float val = (acc[mi][2 * q][r] + ba[q]) * sigmoidf_(acc[mi][2 * q + 1][r] + bb[q]);
                    *reinterpret_cast<bf16_t*>(wbuf + row * PITCH + (q * 32 + lcol) * 2) = f2bf(live ? val : 0.f);
                }
            }
            wave_sync();
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int row = it * RPI + prow, m = m0 + wm * 64 + mi * 32 + row;
                const uint4 v = *reinterpret_cast<const uint4*>(wbuf + row * PITCH + piece * 16);
                if (m < p.M && jb < p.ldc) *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.C) + (size_t)m * p.ldc + jb) = v;
            }
            wave_sync();
        }
    } else {   // EPI_RESID_F32 / EPI_F32: fp32 rows, residual in and sum out per 16-byte piece
        constexpr int PITCH = WCOLS * 4 + 16, PPR = WCOLS / 4, RPI = 64 / PPR, NIT = 32 / RPI;
        char* wbuf = smem + wave * 32 * PITCH;
        float bias[NF];
#pragma unroll
        for (int ni = 0; ni < NF; ++ni) { const int n = nw0 + ni * 32 + lcol; bias[ni] = p.bias[n < p.N ? n : p.N - 1]; }
        const int piece = lane % PPR, prow = lane / PPR, nb = nw0 + piece * 4;
        const int nbc = nb < p.N - 4 ? nb : p.N - 4;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            float4 rv[NIT];
            if constexpr (EPI == EPI_RESID_F32) {
#pragma unroll
                for (int it = 0; it < NIT; ++it) {               // all residual pieces first (unconditional, clamped)
                    const int m = m0 + wm * 64 + mi * 32 + it * RPI + prow;
                    rv[it] = *reinterpret_cast<const float4*>(p.R + (size_t)(m < p.M ? m : p.M - 1) * p.ldr + nbc);
                }
            }
#pragma unroll
            for (int ni = 0; ni < NF; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + lrow;
                    float val = acc[mi][ni][r] + bias[ni];
                    if constexpr (EPI == EPI_RESID_F32) val *= p.alpha;
                    *reinterpret_cast<float*>(wbuf + row * PITCH + (ni * 32 + lcol) * 4) = val;
                }
            wave_sync();
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int row = it * RPI + prow, m = m0 + wm * 64 + mi * 32 + row;
                float4 a = *reinterpret_cast<const float4*>(wbuf + row * PITCH + piece * 16);
                if constexpr (EPI == EPI_RESID_F32) a = make_float4(rv[it].x + a.x, rv[it].y + a.y, rv[it].z + a.z, rv[it].w + a.w);
                if (m < p.M && nb < p.N) *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.C) + (size_t)m * p.ldc + nb) = a;
            }
            wave_sync();
        }
    }
}

template <int BN, int EPI>
int launch256_t(G256Dev& gd, hipStream_t s) {
    const GemmParams& p = gd.p;
    gd.n_tiles = (p.N + BN - 1) / BN;
    gd.m_tiles = (p.M + G_BM - 1) / G_BM;
    constexpr int WCOLS = BN / 2;
    constexpr int stage = G_NW * 32 * (WCOLS * 4 + 16);          // the widest (fp32) staging region
    constexpr int ring = 2 * (G_BM + BN) * G_ROWB;
    const int lds = ring > stage ? ring : stage;
    static LdsAttr attr;
    ensure_dynamic_lds(reinterpret_cast<const void*>(&gemm256_kernel<BN, EPI>), lds, attr);
    hipLaunchKernelGGL((gemm256_kernel<BN, EPI>), dim3(gd.m_tiles * gd.n_tiles), dim3(G_NW * 64), lds, s, gd);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

template <int EPI>
int launch256_bn(G256Dev& gd, int bn, hipStream_t s) {
    return bn == 128 ? launch256_t<128, EPI>(gd, s) : launch256_t<256, EPI>(gd, s);
}

}  // namespace

// what the kernel needs beyond launch_gemm's own checks: whole 16-byte k-chunks and 16-byte-aligned row / piece addresses
bool gemm256_supported(const GemmParams& p, int epi) {
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    if (p.K % 8 || p.lda % 8 || p.ldw % 64 || p.ldw < ec_round_up(p.K, 64) || !al16(p.A) || !al16(p.W)) return false;
    switch (epi) {
        case EPI_BF16: case EPI_SWISH_BF16: case EPI_GLU_BF16: return p.ldc % 8 == 0 && al16(p.C);
        case EPI_QKV_NAT: return p.D % 8 == 0 && al16(p.qu) && al16(p.kh) && al16(p.vt);
        case EPI_RESID_F32: return p.N % 4 == 0 && p.N >= 4 && p.ldc % 4 == 0 && p.ldr % 4 == 0 && al16(p.C) && al16(p.R);
        case EPI_F32: return p.N % 4 == 0 && p.N >= 4 && p.ldc % 4 == 0 && al16(p.C);
    }
    return false;
}

// bn: 256 or 128 columns per workgroup tile
int launch_gemm256(const GemmParams& p, int epi, int bn, hipStream_t s) {
    if (p.M <= 0 || p.N <= 0 || p.K <= 0) return 0;
    if (!gemm256_supported(p, epi) || (bn != 128 && bn != 256)) return -2;
    G256Dev gd;
    gd.p = p;
    gd.fD = FastDiv32(epi == EPI_QKV_NAT ? p.D : 1);
    switch (epi) {
        case EPI_F32: return launch256_bn<EPI_F32>(gd, bn, s);
        case EPI_BF16: return launch256_bn<EPI_BF16>(gd, bn, s);
        case EPI_SWISH_BF16: return launch256_bn<EPI_SWISH_BF16>(gd, bn, s);
        case EPI_RESID_F32: return launch256_bn<EPI_RESID_F32>(gd, bn, s);
        case EPI_GLU_BF16: return launch256_bn<EPI_GLU_BF16>(gd, bn, s);
        case EPI_QKV_NAT: return launch256_bn<EPI_QKV_NAT>(gd, bn, s);
    }
    return -3;
}
