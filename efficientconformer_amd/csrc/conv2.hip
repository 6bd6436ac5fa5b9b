// Two-layer Conv2dSubsampling (plain Conformer configs: ConformerCTC* / ConformerTransducer*, reference
// models/modules.py:218-249 with subsampling_layers = 2, configs/ConformerCTCLarge.json).
//
//  layer 1 (C_in = 1)  : subsample_conv_cl_kernel — 3x3 s2 p1 conv + folded BatchNorm2d + Swish, written CHANNEL-LAST
//                        act1[b][f1][t1][Cp] (bf16, Cp = round_up(C, 64), pad channels zero) so that layer 2 reads
//                        contiguous channel vectors;
//  layer 2 (C -> C)    : conv2_igemm_kernel — the 3x3 s2 conv as an implicit GEMM on the MFMA pipe
//                        (this layer is 26 % of ConformerCTCLarge's FLOPs):
//                          rows m = (b, f2, t2), K = 9 taps x Cp channels (K order (tap, c_in), weights packed likewise with
//                          BatchNorm folded), N = C; the A tile of tap (i, j) is a gather of act1 rows (f1, t1) = (2f2+i-1,
//                          2t2+j-1) (zero rows outside), i.e. no im2col buffer exists;
//                        epilogue: folded bias + Swish, written as out2[(b, t2)][f2*C + n] — the A operand of the following
//                        Linear with its K axis in (f2, c) order (the reference's feature order is c*F2 + f2,
//                        modules.py:247; the Linear weight is permuted accordingly at pack time).
// Tiling as gemm.hip: 128 x 128 x 64, 4 waves, v_mfma_f32_32x32x16_bf16, register-staged double buffer.
#include "kernels.h"

namespace {

constexpr int CL_TT = 8;

__global__ __launch_bounds__(256) void subsample_conv_cl_kernel(const float* __restrict__ mel, int F, int Tm, int T1,
                                                                const float* __restrict__ w9, const float* __restrict__ bias,
                                                                int C, int Cp, bf16_t* __restrict__ out, const int* __restrict__ rag_tm) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TW = 2 * CL_TT + 1;
    float* sm = reinterpret_cast<float*>(smem);        // [F + 1][TW + 1]  row 0 = frequency -1 (zero pad)
    float* sw = sm + (F + 1) * (TW + 1);               // [Cp][10]
    const int tiles = (T1 + CL_TT - 1) / CL_TT;
    const int b = blockIdx.x / tiles, t0 = (blockIdx.x % tiles) * CL_TT;
    const int tid = threadIdx.x, F1 = F / 2;
    // ragged batch: utterance b has rag_tm[b] mel frames in rows of pitch Tm; the zero padding starts there, and its layer-1 image is zero
    // behind its own (rag_tm[b] - 1) / 2 + 1 frames - what layer 2 sees when the utterance runs alone
    const int Tmb = rag_tm ? rag_tm[b] : Tm;
    const int T1b = rag_tm ? (Tmb - 1) / 2 + 1 : T1;
    for (int i = tid; i < (F + 1) * TW; i += 256) {
        const int fr = i / TW, tc = i - fr * TW;
        const int f = fr - 1, t = 2 * t0 - 1 + tc;
        const float v = mel[((size_t)b * F + (f < 0 ? 0 : f)) * Tm + (t < 0 ? 0 : (t < Tm ? t : Tm - 1))];
        sm[fr * (TW + 1) + tc] = (f >= 0 && t >= 0 && t < Tmb) ? v : 0.f;
    }
    for (int i = tid; i < Cp * 10; i += 256) {
        const int c = i / 10, j = i - c * 10;
        sw[i] = c < C ? (j < 9 ? w9[c * 9 + j] : bias[c]) : 0.f;
    }
    __syncthreads();
    // thread = one channel pair with its 2 x 10 folded taps in registers; it walks the tile's (frame, frequency) positions, reading each
    // 3 x 3 mel patch as wave-wide LDS broadcasts (9 dwords per position) - the first version fetched taps AND patch from LDS for
    // every output pair (29 ds_read_b32 + two integer divisions per two outputs) and ran at a quarter of its 1 GB write bound.
    typedef float clf2 __attribute__((ext_vector_type(2)));
    const int cpairs = Cp / 2;
    for (int cp = tid; cp < cpairs; cp += 256) {
        clf2 w[10];
#pragma unroll
        for (int j = 0; j < 10; ++j) w[j] = clf2{sw[(2 * cp) * 10 + j], sw[(2 * cp + 1) * 10 + j]};
        const bool ok0 = 2 * cp < C, ok1 = 2 * cp + 1 < C;
        for (int tl = 0; tl < CL_TT; ++tl) {
            const int t = t0 + tl;
            if (t >= T1) break;
            bf16_t* orow = out + (((size_t)b * F1) * T1 + t) * Cp + 2 * cp;
#pragma unroll 4
            for (int f = 0; f < F1; ++f) {
                const float* m = sm + (2 * f) * (TW + 1) + 2 * tl;
                clf2 a = w[9];
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) { const float x = m[i * (TW + 1) + j]; a = __builtin_elementwise_fma(w[i * 3 + j], clf2{x, x}, a); }
                const float r0 = ok0 && t < T1b ? swishf_(a.x) : 0.f, r1 = ok1 && t < T1b ? swishf_(a.y) : 0.f;
                *reinterpret_cast<uint32_t*>(orow + (size_t)f * T1 * Cp) = pack_bf2(r0, r1);
            }
        }
    }
}

constexpr int BM = 128, BN = 128, BK = 64, LROW = BK * 2 + 16;

__global__ __launch_bounds__(256) void conv2_igemm_kernel(const bf16_t* __restrict__ act1, int F1, int T1, int Cp,
                                                          const bf16_t* __restrict__ W, int ldw, const float* __restrict__ bias,
                                                          int F2, int T2, int M, int N, bf16_t* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sA = smem;
    char* sB = smem + 2 * BM * LROW;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int n_tiles = (N + BN - 1) / BN, m_tiles = (M + BM - 1) / BM;
    const int id = xcd_remap(blockIdx.x, m_tiles * n_tiles);
    const int tm = id / n_tiles, tn = id - tm * n_tiles;
    const int m0 = tm * BM, n0 = tn * BN;
    // ---- staging rows: thread covers rows srow + 32 i (i < 4), chunk kc of the k-tile
    const int srow = tid >> 3, kc = tid & 7;
    int rb_[4], rf2[4], rt2[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int m = m0 + srow + 32 * i;
        m = m < M ? m : M - 1;
        const int t2 = m % T2, rest = m / T2;
        rt2[i] = t2; rf2[i] = rest % F2; rb_[i] = rest / F2;
    }
    const bf16_t* b_ptr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) b_ptr[i] = W + (size_t)(n0 + srow + 32 * i) * ldw + kc * 8;

    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    const int kt_per_tap = Cp / BK, nk = 9 * kt_per_tap;
    const int frag_off = (lane & 31) * LROW + (lane >> 5) * 16;
    for (int kt = -1; kt < nk; ++kt) {
        const int buf = kt & 1;
        const bool more = kt + 1 < nk;
        uint4 ra[4], rb[4];
        bool ok[4] = {false, false, false, false};
#pragma unroll
        for (int i = 0; i < 4; ++i) { ra[i] = make_uint4(0, 0, 0, 0); rb[i] = ra[i]; }
        if (more) {
            const int tap = (kt + 1) / kt_per_tap, kin = ((kt + 1) - tap * kt_per_tap) * BK + kc * 8;
            const int ti = tap / 3, tj = tap - ti * 3;
#pragma unroll
            for (int i = 0; i < 4; ++i) {                     // unconditional loads at clamped positions, zeroed if outside
                const int f1 = 2 * rf2[i] + ti - 1, t1 = 2 * rt2[i] + tj - 1;
                const int f1c = f1 < 0 ? 0 : (f1 < F1 ? f1 : F1 - 1), t1c = t1 < 0 ? 0 : (t1 < T1 ? t1 : T1 - 1);
                ra[i] = *reinterpret_cast<const uint4*>(act1 + (((size_t)rb_[i] * F1 + f1c) * T1 + t1c) * Cp + kin);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) rb[i] = *reinterpret_cast<const uint4*>(b_ptr[i] + (kt + 1) * BK);
#pragma unroll
            for (int i = 0; i < 4; ++i) {                     // validity now, the zeroing at publish time: masking here made the
                const int f1 = 2 * rf2[i] + ti - 1, t1 = 2 * rt2[i] + tj - 1;     // wave wait for the loads it had just issued
                ok[i] = f1 >= 0 && f1 < F1 && t1 >= 0 && t1 < T1;
            }
        }
        if (kt >= 0) {
            const char* a = sA + buf * BM * LROW + (wm * 64) * LROW + frag_off;
            const char* b = sB + buf * BN * LROW + (wn * 64) * LROW + frag_off;
#pragma unroll
            for (int kk = 0; kk < BK / 16; ++kk) {
                bf16x8 af[2], bf[2];
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) af[mi] = *reinterpret_cast<const bf16x8*>(a + mi * 32 * LROW + kk * 32);
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) bf[ni] = *reinterpret_cast<const bf16x8*>(b + ni * 32 * LROW + kk * 32);
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mi], bf[ni], acc[mi][ni], 0, 0, 0);
            }
        }
        if (more) {
            char* a = sA + (buf ^ 1) * BM * LROW;
            char* b = sB + (buf ^ 1) * BN * LROW;
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<uint4*>(a + (srow + 32 * i) * LROW + kc * 16) = ok[i] ? ra[i] : make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<uint4*>(b + (srow + 32 * i) * LROW + kc * 16) = rb[i];
        }
        __syncthreads();
    }
    // ---- epilogue: out2[(b, t2)][f2*N + n] = swish(acc + bias).  The wave's 64 x 64 tile is staged through LDS (the operand buffers
    // are free after the loop's last barrier) and leaves as 16-byte pieces, 128 contiguous bytes per row - not one 2-byte store and
    // two integer divisions per element.
    const int lcol = lane & 31, lrow = 4 * (lane >> 5);
    constexpr int PITCH = 64 * 2 + 16;
    char* wbuf = smem + wave * 64 * PITCH;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int n = n0 + wn * 64 + ni * 32 + lcol;
        const float bz = bias[n < N ? n : N - 1];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = mi * 32 + (r & 3) + 8 * (r >> 2) + lrow;
                *reinterpret_cast<bf16_t*>(wbuf + row * PITCH + (ni * 32 + lcol) * 2) = f2bf(swishf_(acc[mi][ni][r] + bz));
            }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int piece = lane & 7, prow = lane >> 3, nb = n0 + wn * 64 + piece * 8;
    const bool piece_ok = (N % 8) == 0 && nb < N;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int row = it * 8 + prow, m = m0 + wm * 64 + row;
        if (m >= M) continue;
        const int t2 = m % T2, rest = m / T2;
        const int f2 = rest % F2, b = rest / F2;
        bf16_t* dst = out + ((size_t)b * T2 + t2) * ((size_t)F2 * N) + (size_t)f2 * N + nb;
        if (piece_ok) {
            *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(wbuf + row * PITCH + piece * 16);
        } else {
            for (int e = 0; e < 8; ++e)
                if (nb + e < N) dst[e] = *reinterpret_cast<const bf16_t*>(wbuf + row * PITCH + piece * 16 + e * 2);
        }
    }
}

}  // namespace

int launch_subsample_conv_cl(const float* mel, int B, int F, int Tm, int T1, const float* w9, const float* bias, int C, int Cp,
                             bf16_t* out, hipStream_t s, const int* rag_tm) {
    if (B <= 0 || T1 <= 0) return 0;
    if (F % 4 || Cp % 64 || Cp < C) return -2;
    const int tiles = (T1 + CL_TT - 1) / CL_TT;
    const size_t lds = ((size_t)(F + 1) * (2 * CL_TT + 2) + (size_t)Cp * 10) * sizeof(float);
    hipLaunchKernelGGL(subsample_conv_cl_kernel, dim3(B * tiles), dim3(256), lds, s, mel, F, Tm, T1, w9, bias, C, Cp, out, rag_tm);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// act1 [B][F1][T1][Cp] bf16 -> out2 [B*T2][F2*N] bf16; W packed [round_up(N,128)][9*Cp] in (tap, c_in) K order, BN folded
int launch_conv2_igemm(const bf16_t* act1, int B, int F1, int T1, int Cp, const bf16_t* W, int ldw, const float* bias,
                       int N, int F2, int T2, bf16_t* out, hipStream_t s) {
    if (B <= 0 || T2 <= 0) return 0;
    if (Cp % 64 || ldw != 9 * Cp) return -2;
    const int M = B * F2 * T2;
    const size_t lds = 2 * (BM + BN) * LROW;
    static LdsAttr attr;
    ensure_dynamic_lds(reinterpret_cast<const void*>(&conv2_igemm_kernel), (int)lds, attr);
    const int n_tiles = (N + BN - 1) / BN, m_tiles = (M + BM - 1) / BM;
    hipLaunchKernelGGL(conv2_igemm_kernel, dim3(m_tiles * n_tiles), dim3(256), lds, s, act1, F1, T1, Cp, W, ldw, bias, F2, T2, M, N, out);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
