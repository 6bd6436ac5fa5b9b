// Row-stationary MFMA kernels for the Conformer block (gfx950): activations stay in registers,
// weights stream through LDS.
//
// The block's GEMMs are tall and skinny (M = B*T rows in the 10^4..10^5 range, K = D in 120..384,
// N in D..4D), so the classic C-tile GEMM re-reads A once per N tile and spends most of its time in
// prologue/epilogue.  Here the MFMA operands are swapped (C^T = W X^T):
//
//   * a wave owns RT*32 activation rows; lane l holds row (l & 31) of each 32-row tile, split in two
//     halves across lanes l and l+32 -> X is the MFMA *B* operand and lives in registers for the whole kernel
//     (v_mfma_f32_32x32x16_bf16: B[k][n]: n = lane & 31, k = 8*(lane>>5) + e);
//   * weight rows are the MFMA *A* operand, staged in 32-row chunks through a double-buffered LDS ring
//     shared by all waves of the workgroup (every wave needs every weight row exactly once);
//   * the result tile C^T[n][m] leaves every lane with 16 output columns of ITS OWN row m, so
//     bias/activation/residual are lane-local, and a second GEMM can consume the first one's
//     accumulators directly as its B operand: accumulator registers [8s, 8s+8) of a 32-row result tile are,
//     after bf16 packing, exactly the B fragment of k-step s — provided the second weight's K index is
//     permuted within each group of 16 at pack time (position 8h+e <-> 4h + 8(e>>2) + (e&3)).
//
// ffn_fused_kernel: y = x + alpha * (Swish(a W1^T + b1) W2^T + b2)   (a = LayerNorm(x) as bf16)
//   replaces FeedForwardModule.forward + the half-step residual (reference models/modules.py:385-392,
//   blocks.py:122, 132); the 4D-wide hidden activation never exists in memory.
#include "kernels.h"

namespace {

constexpr int CH = 32;   // weight rows per LDS chunk == hidden units per step

template <int KS, int NT2>
struct FfnSmem {
    static constexpr int W1ROW = KS * 32 + 16;            // bytes per W1 row in LDS (KS*16 bf16 + 16 B pad)
    static constexpr int W2ROW = CH * 2 + 16;             // 80 B
    static constexpr int W1_BYTES = CH * W1ROW;
    static constexpr int W2_BYTES = NT2 * 32 * W2ROW;
    static constexpr int BUF = W1_BYTES + W2_BYTES + 128;  // + b1 chunk (32 floats)
    static constexpr int TOTAL = 2 * BUF;
};

// KS: k16-steps over D (D <= 16*KS), NT2: 32-col tiles over D (D <= 32*NT2), RT: 32-row tiles per wave, NW: waves
template <int KS, int NT2, int RT, int NW>
__global__ __launch_bounds__(NW * 64) void ffn_fused_kernel(const FfnParams p) {
    using SM = FfnSmem<KS, NT2>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NTHR = NW * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 31, half = lane >> 5;
    const int m_base = (blockIdx.x * NW + wave) * (RT * 32);

    // ---- this lane's activation row fragments (B operand of GEMM1), loaded once
    bf16x8 xf[RT][KS];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const int m = m_base + rt * 32 + lr;
        const bf16_t* arow = p.A + (size_t)(m < p.M ? m : p.M - 1) * p.lda;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int c0 = s * 16 + half * 8;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (c0 < p.D && m < p.M) v = mask_chunk(*reinterpret_cast<const uint4*>(arow + c0), p.D - c0);
            xf[rt][s] = as_bf16x8(v);
        }
    }
    f32x16 acc[RT][NT2];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int t = 0; t < NT2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[rt][t][r] = 0.f;

    // ---- weight staging: chunk c = hidden units [32c, 32c+32)
    constexpr int W1_CHUNKS = CH * KS * 2;                 // 16-byte pieces of a W1 chunk (KS*16 bf16 per row)
    constexpr int W2_CHUNKS = NT2 * 32 * 4;                // 4 pieces (32 bf16) per W2 row
    constexpr int N1 = (W1_CHUNKS + NTHR - 1) / NTHR, N2 = (W2_CHUNKS + NTHR - 1) / NTHR;
    uint4 r1[N1], r2[N2];
#pragma unroll
    for (int i = 0; i < N1; ++i) r1[i] = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < N2; ++i) r2[i] = make_uint4(0, 0, 0, 0);
    float rb = 0.f;
    auto load_chunk = [&](int c) {
#pragma unroll
        for (int i = 0; i < N1; ++i) {
            const int q = tid + i * NTHR;
            if (W1_CHUNKS % NTHR == 0 || q < W1_CHUNKS) {
                const int row = q / (KS * 2), pc = q - row * (KS * 2);
                r1[i] = *reinterpret_cast<const uint4*>(p.W1 + (size_t)(c * CH + row) * p.ldw1 + pc * 8);
            }
        }
#pragma unroll
        for (int i = 0; i < N2; ++i) {
            const int q = tid + i * NTHR;
            if (W2_CHUNKS % NTHR == 0 || q < W2_CHUNKS) {
                const int row = q >> 2, pc = q & 3;
                r2[i] = *reinterpret_cast<const uint4*>(p.W2 + (size_t)row * p.ldw2 + c * CH + pc * 8);
            }
        }
        if (tid < CH) rb = p.b1[c * CH + tid];
    };
    auto store_chunk = [&](int buf) {
        char* w1 = smem + buf * SM::BUF;
        char* w2 = w1 + SM::W1_BYTES;
#pragma unroll
        for (int i = 0; i < N1; ++i) {
            const int q = tid + i * NTHR;
            if (W1_CHUNKS % NTHR == 0 || q < W1_CHUNKS) {
                const int row = q / (KS * 2), pc = q - row * (KS * 2);
                *reinterpret_cast<uint4*>(w1 + row * SM::W1ROW + pc * 16) = r1[i];
            }
        }
#pragma unroll
        for (int i = 0; i < N2; ++i) {
            const int q = tid + i * NTHR;
            if (W2_CHUNKS % NTHR == 0 || q < W2_CHUNKS) {
                const int row = q >> 2, pc = q & 3;
                *reinterpret_cast<uint4*>(w2 + row * SM::W2ROW + pc * 16) = r2[i];
            }
        }
        if (tid < CH) reinterpret_cast<float*>(w2 + SM::W2_BYTES)[tid] = rb;
    };

    const int nchunks = p.Fp / CH;
    load_chunk(0);
    store_chunk(0);
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        const int buf = c & 1;
        if (c + 1 < nchunks) load_chunk(c + 1);
        const char* w1 = smem + buf * SM::BUF + lr * SM::W1ROW + half * 16;
        const char* w2 = smem + buf * SM::BUF + SM::W1_BYTES + lr * SM::W2ROW + half * 16;
        const float* b1 = reinterpret_cast<const float*>(smem + buf * SM::BUF + SM::W1_BYTES + SM::W2_BYTES);
        // ---- GEMM1: H^T[j][m] = sum_k W1[j][k] a[m][k]
        f32x16 h[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 16; ++r) h[rt][r] = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(w1 + s * 32);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) h[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, xf[rt][s], h[rt], 0, 0, 0);
        }
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
        // ---- GEMM2: Y^T[n][m] += sum_j W2p[n][j] H^T[j][m]
#pragma unroll
        for (int t = 0; t < NT2; ++t)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(w2 + t * 32 * SM::W2ROW + s2 * 32);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
                    acc[rt][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, hf[rt][s2], acc[rt][t], 0, 0, 0);
            }
        if (c + 1 < nchunks) store_chunk(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: y[m][n] = x[m][n] + alpha * (acc + b2[n]);  lane owns row m, columns 32t + 8q + 4*half + (0..3)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const int m = m_base + rt * 32 + lr;
        if (m >= p.M) continue;
        const float* xr = p.X + (size_t)m * p.ldx;
        float* yr = p.Y + (size_t)m * p.ldy;
#pragma unroll
        for (int t = 0; t < NT2; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = t * 32 + q * 8 + half * 4;
                if (n >= p.D) continue;
                const float4 xv = *reinterpret_cast<const float4*>(xr + n);
                const float4 bv = *reinterpret_cast<const float4*>(p.b2 + n);
                float4 o;
                o.x = xv.x + p.alpha * (acc[rt][t][q * 4 + 0] + bv.x);
                o.y = xv.y + p.alpha * (acc[rt][t][q * 4 + 1] + bv.y);
                o.z = xv.z + p.alpha * (acc[rt][t][q * 4 + 2] + bv.z);
                o.w = xv.w + p.alpha * (acc[rt][t][q * 4 + 3] + bv.w);
                *reinterpret_cast<float4*>(yr + n) = o;
            }
    }
}

// ---------------------------------------------------------------------------------------------
// rs_gemm_kernel: one GEMM, activations stationary:  C^T[n][m] = sum_k W[n][k] a[m][k]  (K = D <= 384)
//   RS_RESID : y[m][n] = R[m][n] + alpha * (acc + bias[n])      fp32   (attention output projection +
//              residual, attentions.py:716 + blocks.py:126; pointwise-2 + residual, modules.py:519 + blocks.py:129)
//   RS_F32   : y[m][n] = acc + bias[n]                          fp32   (conv_res 1x1 strided conv, blocks.py:106-110)
//   RS_GLU   : g[m][j] = (acc_a + b_a) * sigmoid(acc_b + b_b)   bf16   (pointwise-1 + GLU, modules.py:513-514;
//              weight rows interleaved per 32 channels: chunk 2i = a, chunk 2i+1 = b)
//   RS_QKV   : Q+u, Q+v, K, V scattered to head-major [B][H][Tg][dpad] (attentions.py:651-686)
enum { RS_RESID = 0, RS_F32 = 1, RS_GLU = 2, RS_QKV = 3 };

struct FastDiv32 {   // exact for n * d < 2^32
    uint32_t mul, d;
    __host__ __device__ FastDiv32() : mul(0), d(1) {}
    __host__ explicit FastDiv32(uint32_t dd) : mul(dd > 1 ? (uint32_t)((1ull << 32) / dd + 1) : 0), d(dd) {}
    __device__ __forceinline__ uint32_t div(uint32_t n) const { return d == 1 ? n : __umulhi(n, mul); }
};
struct RsDev { GemmParams p; int nchunks; FastDiv32 fD, fd; };

template <int KS, int RT, int NW, int EPI>
__global__ __launch_bounds__(NW * 64) void rs_gemm_kernel(const RsDev gd) {
    const GemmParams& p = gd.p;
    constexpr int WROW = KS * 32 + 16;
    constexpr int BUF = CH * WROW + 128;
    constexpr int NTHR = NW * 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 31, half = lane >> 5;
    const int m_base = (blockIdx.x * NW + wave) * (RT * 32);

    bf16x8 xf[RT][KS];
    int rowm[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const int m = m_base + rt * 32 + lr;
        rowm[rt] = m;
        const int mc = m < p.M ? m : p.M - 1;
        size_t src = mc;
        if (p.a_rows > 0) { const int b = mc / p.a_rows, r = mc - b * p.a_rows; src = (size_t)b * p.a_pitch + (size_t)r * p.a_stride; }
        const bf16_t* arow = p.A + src * p.lda;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int c0 = s * 16 + half * 8;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (c0 < p.K && m < p.M) v = mask_chunk(*reinterpret_cast<const uint4*>(arow + c0), p.K - c0);
            xf[rt][s] = as_bf16x8(v);
        }
    }
    // per-row constants of the QKV scatter
    int qb[RT], qtq[RT], qtoff[RT];
    if constexpr (EPI == RS_QKV) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int m = rowm[rt] < p.M ? rowm[rt] : p.M - 1;
            const int b = m / p.T, t = m - b * p.T;
            qb[rt] = b; qtq[rt] = t / p.G; qtoff[rt] = t - qtq[rt] * p.G;
        }
    }

    constexpr int W_CHUNKS = CH * KS * 2;
    constexpr int N1 = (W_CHUNKS + NTHR - 1) / NTHR;
    uint4 r1[N1];
#pragma unroll
    for (int i = 0; i < N1; ++i) r1[i] = make_uint4(0, 0, 0, 0);
    float rb = 0.f;
    auto load_chunk = [&](int c) {
#pragma unroll
        for (int i = 0; i < N1; ++i) {
            const int q = tid + i * NTHR;
            if (W_CHUNKS % NTHR == 0 || q < W_CHUNKS) {
                const int row = q / (KS * 2), pc = q - row * (KS * 2);
                r1[i] = *reinterpret_cast<const uint4*>(p.W + (size_t)(c * CH + row) * p.ldw + pc * 8);
            }
        }
        if (tid < CH) rb = p.bias[c * CH + tid];
    };
    auto store_chunk = [&](int buf) {
        char* w = smem + buf * BUF;
#pragma unroll
        for (int i = 0; i < N1; ++i) {
            const int q = tid + i * NTHR;
            if (W_CHUNKS % NTHR == 0 || q < W_CHUNKS) {
                const int row = q / (KS * 2), pc = q - row * (KS * 2);
                *reinterpret_cast<uint4*>(w + row * WROW + pc * 16) = r1[i];
            }
        }
        if (tid < CH) reinterpret_cast<float*>(w + CH * WROW)[tid] = rb;
    };

    f32x16 prev[RT];   // GLU: the 'a' half of the current channel block
    load_chunk(0);
    store_chunk(0);
    __syncthreads();
    for (int c = 0; c < gd.nchunks; ++c) {
        const int buf = c & 1;
        if (c + 1 < gd.nchunks) load_chunk(c + 1);
        const char* w = smem + buf * BUF + lr * WROW + half * 16;
        const float* bias = reinterpret_cast<const float*>(smem + buf * BUF + CH * WROW);
        f32x16 acc[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[rt][r] = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(w + s * 32);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, xf[rt][s], acc[rt], 0, 0, 0);
        }
        // ---- epilogue of this 32-column chunk: lane owns row m, columns n = 32c + 8q + 4*half + i
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int m = rowm[rt];
            if constexpr (EPI == RS_RESID || EPI == RS_F32) {
                if (m < p.M) {
                    float* yr = reinterpret_cast<float*>(p.C) + (size_t)m * p.ldc;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int nl = q * 8 + half * 4, n = c * CH + nl;
                        if (n >= p.N) continue;
                        float4 o;
                        o.x = acc[rt][q * 4 + 0] + bias[nl + 0]; o.y = acc[rt][q * 4 + 1] + bias[nl + 1];
                        o.z = acc[rt][q * 4 + 2] + bias[nl + 2]; o.w = acc[rt][q * 4 + 3] + bias[nl + 3];
                        if constexpr (EPI == RS_RESID) {
                            const float4 xv = *reinterpret_cast<const float4*>(p.R + (size_t)m * p.ldr + n);
                            o.x = xv.x + p.alpha * o.x; o.y = xv.y + p.alpha * o.y; o.z = xv.z + p.alpha * o.z; o.w = xv.w + p.alpha * o.w;
                        }
                        *reinterpret_cast<float4*>(yr + n) = o;
                    }
                }
            } else if constexpr (EPI == RS_GLU) {
                if ((c & 1) == 0) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) prev[rt][r] = acc[rt][r] + bias[(r & 3) + 8 * (r >> 2) + 4 * half];
                } else if (m < p.M) {
                    bf16_t* gr = reinterpret_cast<bf16_t*>(p.C) + (size_t)m * p.ldc;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int nl = q * 8 + half * 4, j = (c >> 1) * CH + nl;
                        if (j >= p.ldc) continue;
                        float g[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) g[i] = prev[rt][q * 4 + i] * sigmoidf_(acc[rt][q * 4 + i] + bias[nl + i]);
                        *reinterpret_cast<uint2*>(gr + j) = make_uint2(pack_bf2(g[0], g[1]), pack_bf2(g[2], g[3]));
                    }
                }
            } else {   // RS_QKV
                if (m < p.M) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int nl = q * 8 + half * 4, n0 = c * CH + nl;
                        if (n0 >= p.N) continue;
                        // D % 4 == 0: the 4 columns share `which`; heads may split inside the group only between pairs when d is even
                        const int which = gd.fD.div(n0), nn0 = n0 - which * p.D;
                        bf16_t* dst = which == 1 ? p.kh : (which == 2 ? p.vt : p.qu);
#pragma unroll
                        for (int i2 = 0; i2 < 4; i2 += 2) {
                            float v0 = acc[rt][q * 4 + i2] + bias[nl + i2], v1 = acc[rt][q * 4 + i2 + 1] + bias[nl + i2 + 1];
                            const int flat = qtoff[rt] * p.D + nn0 + i2;
                            const int h = gd.fd.div(flat), x = flat - h * p.d;
                            const size_t idx = ((size_t)(qb[rt] * p.H + h) * p.Tg + qtq[rt]) * p.dpad + x;
                            if ((p.d & 1) == 0) {
                                if (which == 0) {
                                    const float u0 = p.u[nn0 + i2], u1 = p.u[nn0 + i2 + 1], w0 = p.v[nn0 + i2], w1 = p.v[nn0 + i2 + 1];
                                    *reinterpret_cast<uint32_t*>(p.qu + idx) = pack_bf2(v0 + u0, v1 + u1);
                                    *reinterpret_cast<uint32_t*>(p.qv + idx) = pack_bf2(v0 + w0, v1 + w1);
                                } else {
                                    *reinterpret_cast<uint32_t*>(dst + idx) = pack_bf2(v0, v1);
                                }
                            } else {
                                const int flat1 = flat + 1;
                                const int h1 = gd.fd.div(flat1), x1 = flat1 - h1 * p.d;
                                const size_t idx1 = ((size_t)(qb[rt] * p.H + h1) * p.Tg + qtq[rt]) * p.dpad + x1;
                                if (which == 0) {
                                    p.qu[idx] = f2bf(v0 + p.u[nn0 + i2]); p.qv[idx] = f2bf(v0 + p.v[nn0 + i2]);
                                    p.qu[idx1] = f2bf(v1 + p.u[nn0 + i2 + 1]); p.qv[idx1] = f2bf(v1 + p.v[nn0 + i2 + 1]);
                                } else { dst[idx] = f2bf(v0); dst[idx1] = f2bf(v1); }
                            }
                        }
                    }
                }
            }
        }
        if (c + 1 < gd.nchunks) store_chunk(buf ^ 1);
        __syncthreads();
    }
}

template <int KS, int RT, int NW, int EPI>
int launch_rs_t(const RsDev& gd, hipStream_t s) {
    constexpr int LDS = 2 * (CH * (KS * 32 + 16) + 128);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&rs_gemm_kernel<KS, RT, NW, EPI>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr_set = true;
    }
    const int rows_per_wg = NW * RT * 32;
    hipLaunchKernelGGL((rs_gemm_kernel<KS, RT, NW, EPI>), dim3((gd.p.M + rows_per_wg - 1) / rows_per_wg), dim3(NW * 64), LDS, s, gd);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

template <int EPI>
int launch_rs_ks(const RsDev& gd, hipStream_t s) {
    const int ks = (gd.p.K + 15) / 16;
    if (ks <= 2) return launch_rs_t<2, 2, 8, EPI>(gd, s);
    if (ks <= 4) return launch_rs_t<4, 2, 8, EPI>(gd, s);
    if (ks <= 8) return launch_rs_t<8, 2, 8, EPI>(gd, s);
    if (ks <= 12) return launch_rs_t<12, 2, 8, EPI>(gd, s);
    if (ks <= 16) return launch_rs_t<16, 2, 8, EPI>(gd, s);
    if (ks <= 20) return launch_rs_t<20, 1, 8, EPI>(gd, s);
    return launch_rs_t<24, 1, 8, EPI>(gd, s);
}

template <int KS, int NT2, int RT, int NW>
int launch_ffn_t(const FfnParams& p, hipStream_t s) {
    using SM = FfnSmem<KS, NT2>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ffn_fused_kernel<KS, NT2, RT, NW>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, SM::TOTAL);
        attr_set = true;
    }
    const int rows_per_wg = NW * RT * 32;
    hipLaunchKernelGGL((ffn_fused_kernel<KS, NT2, RT, NW>), dim3((p.M + rows_per_wg - 1) / rows_per_wg), dim3(NW * 64),
                       SM::TOTAL, s, p);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace

bool ffn_fused_supported(int D) { return D % 4 == 0 && D <= 384; }

int launch_ffn_fused(const FfnParams& p, hipStream_t s) {
    if (p.M <= 0) return 0;
    if (!ffn_fused_supported(p.D) || p.Fp % CH || p.lda % 8 || p.ldw1 % 8 || p.ldw2 % 8) return -2;
    const int ks = (p.D + 15) / 16;
    // D <= 128: 8 waves x 32 rows; D <= 256: 4 waves x 64 rows (one wave per SIMD, 512 registers); else 4 x 32
    if (ks <= 2) return launch_ffn_t<2, 1, 1, 8>(p, s);
    if (ks <= 4) return launch_ffn_t<4, 2, 1, 8>(p, s);
    if (ks <= 8) return launch_ffn_t<8, 4, 1, 8>(p, s);
    if (ks <= 12) return launch_ffn_t<12, 6, 2, 4>(p, s);
    if (ks <= 16) return launch_ffn_t<16, 8, 2, 4>(p, s);
    if (ks <= 20) return launch_ffn_t<20, 10, 1, 4>(p, s);
    return launch_ffn_t<24, 12, 1, 4>(p, s);
}

bool rs_gemm_supported(int K) { return K % 4 == 0 && K <= 384; }

// epi: RS_* (0 resid, 1 f32, 2 glu, 3 qkv).  W packed [>= nchunks*32][ldw >= round_up(K,64)], bias padded likewise.
int launch_rs_gemm(const GemmParams& p, int epi, hipStream_t s) {
    if (p.M <= 0 || p.N <= 0) return 0;
    if (!rs_gemm_supported(p.K) || p.lda % 8 || p.ldw % 8) return -2;
    RsDev gd;
    gd.p = p;
    gd.nchunks = (p.N + CH - 1) / CH;
    if (epi == RS_QKV) { gd.fD = FastDiv32(p.D); gd.fd = FastDiv32(p.d); }
    switch (epi) {
        case RS_RESID: return launch_rs_ks<RS_RESID>(gd, s);
        case RS_F32: return launch_rs_ks<RS_F32>(gd, s);
        case RS_GLU: return launch_rs_ks<RS_GLU>(gd, s);
        case RS_QKV: return launch_rs_ks<RS_QKV>(gd, s);
    }
    return -3;
}
