// Fused Conv2dSubsampling + Linear (gfx950).
//
// Reference: Conv2dSubsampling.forward (models/modules.py:232-249, one layer, C_in = 1: Conv2d 3x3 s2 p1 -> BatchNorm2d(eval)
// -> Swish -> reshape to (B, C*F/2, T1)) followed by the transpose and nn.Linear(C*F/2 -> D0) of
// ConformerEncoder.forward (models/encoders.py:113-116).
//
// Unfused, the (B*T1, C*40) bf16 activation is written and read back once (2 x 984 MB for B = 128 LibriSpeech-shaped
// utterances: profiles/r1_03_*: 0.62 + 0.51 ms).  Here it never exists: the kernel is the tiled MFMA GEMM
// (128 x 128 x 64 tiles, v_mfma_f32_32x32x16_bf16) whose A tile is PRODUCED instead of loaded — every thread computes
// 4 x 8 conv outputs per k-tile from a 17 x 3 mel patch it keeps in registers.  To make the patch reusable the GEMM's K
// axis is re-ordered at pack time to (f-chunk, channel, f-within-chunk): k' = (fc*Cp + c)*8 + e  <->  reference feature
// c*(F/2) + 8*fc + e, Cp = round_up(C, 8); a thread sweeps the channels c for a fixed block of 8 output frequencies and
// reloads its patch only F/16 = 5 times.  The conv is fp32 (9 FMA + folded BN + Swish per element), rounded to bf16 exactly
// where the unfused path rounded it, so results differ from it only by the summation order of the fp32 accumulators.
// VALU-bound by construction (~20 lane-ops per A element vs 2*D0/16 MFMA-lane-ops); HBM traffic = mel in + D0 fp32 out.
#include "kernels.h"

namespace {

constexpr int BM = 128, BK = 64, LROW = BK * 2 + 16;
typedef float slf2 __attribute__((ext_vector_type(2)));

template <int NT>   // NT = number of 128-wide output tiles (D0 <= 128*NT)
__global__ __launch_bounds__(256, NT == 1 ? 2 : 1) void sublinear_kernel(const float* __restrict__ mel, int F, int Tm, int T1, int M,
                                                        const float* __restrict__ w9, const float* __restrict__ cbias, int C, int Cp,
                                                        const bf16_t* __restrict__ W, int ldw, const float* __restrict__ bias, int N,
                                                        float* __restrict__ out, int ldc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sA = smem;                                   // [2][BM][LROW]
    char* sB = sA + 2 * BM * LROW;                     // [2][NT*128][LROW]
    float* sw = reinterpret_cast<float*>(sB + 2 * NT * 128 * LROW);   // [Cp][12]: 9 folded taps, bias, pad
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.x * BM;
    for (int i = tid; i < Cp * 12; i += 256) {
        const int c = i / 12, j = i - c * 12;
        sw[i] = (c < C && j < 10) ? (j < 9 ? w9[c * 9 + j] : cbias[c]) : 0.f;
    }
    // ---- A producer: thread = (row, 4 consecutive chunks of 8 k')
    const int prow = tid >> 1, kc0 = (tid & 1) * 4;
    const int pm = m0 + prow < M ? m0 + prow : M - 1;
    const int pb = pm / T1, pt = pm - pb * T1;
    const float* melb = mel + (size_t)pb * F * Tm;
    float patch[17][3];
    int cur_fc = -1;
    auto load_patch = [&](int fc) __attribute__((always_inline)) {
        // clamped, unconditional loads, ALL issued before the first select: the empty asm pins each loaded value, otherwise the
        // compiler sinks every load under its select's condition (exec-masked branches) and waits for each row's three loads
        // before the next row's - 17 serialized memory latencies per patch instead of one
#pragma unroll
        for (int i = 0; i < 17; ++i) {
            const int fr = 16 * fc - 1 + i;
            const int frc = fr < 0 ? 0 : (fr < F ? fr : F - 1);
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int tc = 2 * pt - 1 + j;
                patch[i][j] = melb[(unsigned)(frc * Tm + (tc < 0 ? 0 : (tc < Tm ? tc : Tm - 1)))];
            }
        }
#pragma unroll
        for (int i = 0; i < 17; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) asm volatile("" : "+v"(patch[i][j]));
#pragma unroll
        for (int i = 0; i < 17; ++i) {
            const int fr = 16 * fc - 1 + i;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int tc = 2 * pt - 1 + j;
                patch[i][j] = (fr >= 0 && fr < F && tc >= 0 && tc < Tm) ? patch[i][j] : 0.f;
            }
        }
    };
    const int srow = tid >> 3, kc = tid & 7;         // B staging (packed weight rows, K' order), as in gemm.hip

    f32x16 acc[NT][2][2];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[nt][mi][ni][r] = 0.f;

    const int nk = 5 * Cp * 8 / BK;
    __syncthreads();                                   // conv weights visible
    const int frag_off = (lane & 31) * LROW + (lane >> 5) * 16;
    // one staging site (iteration -1 is the prologue): produce A / load B of tile kt+1, MFMA on tile kt, publish tile kt+1
    for (int kt = -1; kt < nk; ++kt) {
        const int buf = kt & 1;
        const bool more = kt + 1 < nk;
        uint4 pa[4], rb[NT * 4];
#pragma unroll
        for (int i = 0; i < 4; ++i) pa[i] = make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NT * 4; ++i) rb[i] = make_uint4(0, 0, 0, 0);
        if (more) {
#pragma unroll
            for (int i = 0; i < NT * 4; ++i) rb[i] = *reinterpret_cast<const uint4*>(W + (size_t)(srow + 32 * i) * ldw + (kt + 1) * BK + kc * 8);
            const int q0 = (kt + 1) * 8 + kc0;         // chunk index = fc*Cp + c
            const int fc = q0 / Cp, c0 = q0 - fc * Cp;
            if (fc != cur_fc) { load_patch(fc); cur_fc = fc; }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float* w = sw + (c0 + i) * 12;
                const float4 w0 = *reinterpret_cast<const float4*>(w), w1 = *reinterpret_cast<const float4*>(w + 4), w2 = *reinterpret_cast<const float4*>(w + 8);
                const float wt[9] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, w2.x};
                float r[8];
#pragma unroll
                for (int e = 0; e < 8; e += 2) {        // two outputs at a time: the Swish's plain multiplies / add go out as packed f32 ops
                    slf2 a = slf2{w2.y, w2.y};          // folded bias
#pragma unroll
                    for (int ii = 0; ii < 3; ++ii)
#pragma unroll
                        for (int jj = 0; jj < 3; ++jj)
                            a = __builtin_elementwise_fma(slf2{wt[ii * 3 + jj], wt[ii * 3 + jj]}, slf2{patch[2 * e + ii][jj], patch[2 * e + 2 + ii][jj]}, a);
                    const slf2 t = a * slf2{-1.44269504088896f, -1.44269504088896f};
                    const slf2 d = slf2{__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)} + slf2{1.0f, 1.0f};
                    const slf2 y = a * slf2{__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
                    r[e] = y.x; r[e + 1] = y.y;
                }
                pa[i] = make_uint4(pack_bf2(r[0], r[1]), pack_bf2(r[2], r[3]), pack_bf2(r[4], r[5]), pack_bf2(r[6], r[7]));
            }
        }
        if (kt >= 0) {
            const char* a = sA + buf * BM * LROW + (wm * 64) * LROW + frag_off;
#pragma unroll
            for (int kk = 0; kk < BK / 16; ++kk) {
                bf16x8 af[2];
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) af[mi] = *reinterpret_cast<const bf16x8*>(a + mi * 32 * LROW + kk * 32);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const char* b = sB + buf * NT * 128 * LROW + (nt * 128 + wn * 64) * LROW + frag_off;
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) {
                        const bf16x8 bf = *reinterpret_cast<const bf16x8*>(b + ni * 32 * LROW + kk * 32);
#pragma unroll
                        for (int mi = 0; mi < 2; ++mi)
                            acc[nt][mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mi], bf, acc[nt][mi][ni], 0, 0, 0);
                    }
                }
            }
        }
        if (more) {
            const int nb = buf ^ 1;
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<uint4*>(sA + nb * BM * LROW + prow * LROW + (kc0 + i) * 16) = pa[i];
#pragma unroll
            for (int i = 0; i < NT * 4; ++i) *reinterpret_cast<uint4*>(sB + nb * NT * 128 * LROW + (srow + 32 * i) * LROW + kc * 16) = rb[i];
        }
        __syncthreads();
    }
    // ---- epilogue: out[m][n] = acc + bias  (fp32); C layout: col = lane & 31, row = (r & 3) + 8*(r >> 2) + 4*(lane >> 5)
    const int lcol = lane & 31, lrow = 4 * (lane >> 5);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int n = nt * 128 + wn * 64 + ni * 32 + lcol;
            if (n >= N) continue;
            const float bz = bias[n];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + lrow;
                    if (m < M) out[(size_t)m * ldc + n] = acc[nt][mi][ni][r] + bz;
                }
        }
}

template <int NT>
int launch_t(const float* mel, int B, int F, int Tm, int T1, const float* w9, const float* cbias, int C, int Cp,
             const bf16_t* W, int ldw, const float* bias, int N, float* out, int ldc, hipStream_t s) {
    const int M = B * T1;
    const size_t lds = 2 * (BM + NT * 128) * LROW + (size_t)Cp * 12 * 4;
    static LdsAttr attr;
    ensure_dynamic_lds(reinterpret_cast<const void*>(&sublinear_kernel<NT>), (int)lds, attr);
    hipLaunchKernelGGL((sublinear_kernel<NT>), dim3((M + BM - 1) / BM), dim3(256), lds, s, mel, F, Tm, T1, M, w9, cbias, C, Cp,
                       W, ldw, bias, N, out, ldc);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace

// N <= 256: two 128-wide output tiles keep the double-buffered A/B tiles + conv weights inside the 160 KB LDS
bool sublinear_fused_supported(int F, int N) { return F == 80 && N <= 256; }

// W: packed [round_up(N,128)][ldw] bf16 in K' order (see file header), ldw = 5*Cp*8
int launch_sublinear_fused(const float* mel, int B, int F, int Tm, int T1, const float* w9, const float* cbias, int C,
                           const bf16_t* W, int ldw, const float* bias, int N, float* out, int ldc, hipStream_t s) {
    if (B <= 0 || T1 <= 0) return 0;
    if (!sublinear_fused_supported(F, N)) return -2;
    const int Cp = ec_round_up(C, 8);
    if (ldw != 5 * Cp * 8 || ldw % 64) return -2;
    if (N <= 128) return launch_t<1>(mel, B, F, Tm, T1, w9, cbias, C, Cp, W, ldw, bias, N, out, ldc, s);
    return launch_t<2>(mel, B, F, Tm, T1, w9, cbias, C, Cp, W, ldw, bias, N, out, ldc, s);
}
