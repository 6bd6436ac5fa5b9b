// Convolution kernels of the encoder (gfx950), both HBM-bound / VALU-light:
//
//  * subsample_conv_kernel — first Conv2dSubsampling layer (reference modules.py:232-249 with
//    C_in = 1): 3x3 stride-2 pad-1 conv + folded BatchNorm2d(eval) + Swish, LDS-staged mel tile,
//    writes the (B*T1, C*F/2) bf16 A-operand of the following Linear directly (feature = c*F/2 + f),
//    i.e. the reshape (modules.py:247) and the transpose (encoders.py:113) cost nothing.
//  * dwconv_kernel — depthwise Conv1d k taps, stride s, "same" zero pre-padding (layers.py:100, 131-136)
//    + folded BatchNorm1d(eval) + Swish (modules.py:516-518) on channel-last bf16 rows; LDS time tile
//    with (k-1) halo, channels across lanes.
#include "kernels.h"

namespace {

constexpr int SUB_TT = 8;     // output frames per workgroup

__global__ __launch_bounds__(256) void subsample_conv_kernel(const float* __restrict__ mel, int F, int Tm, int T1,
                                                             const float* __restrict__ w9, const float* __restrict__ bias,
                                                             int C, bf16_t* out, int ldo) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TW = 2 * SUB_TT + 1;                 // input frames needed: 2t-1 .. 2t+1
    float* sm = reinterpret_cast<float*>(smem);        // [F + 1][TW + 1]  row 0 = frequency -1 (zero pad)
    float* sw = sm + (F + 1) * (TW + 1);               // [C][10]: 9 taps + bias
    const int tiles = (T1 + SUB_TT - 1) / SUB_TT;
    const int b = blockIdx.x / tiles, t0 = (blockIdx.x % tiles) * SUB_TT;
    const int tid = threadIdx.x;
    const int F2 = F / 2;
    for (int i = tid; i < (F + 1) * TW; i += 256) {
        const int fr = i / TW, tc = i - fr * TW;
        const int f = fr - 1, t = 2 * t0 - 1 + tc;
        const bool ok = f >= 0 && t >= 0 && t < Tm;
        const float v = mel[((size_t)b * F + (f < 0 ? 0 : f)) * Tm + (t < 0 ? 0 : (t < Tm ? t : Tm - 1))];   // clamped, unconditional
        sm[fr * (TW + 1) + tc] = ok ? v : 0.f;
    }
    for (int i = tid; i < C * 10; i += 256) {
        const int c = i / 10, j = i - c * 10;
        sw[i] = j < 9 ? w9[c * 9 + j] : bias[c];
    }
    __syncthreads();
    const int pairs = C * F2 / 2;                      // two adjacent f per thread (F2 is even for F = 80)
    for (int tl = 0; tl < SUB_TT; ++tl) {
        const int t = t0 + tl;
        if (t >= T1) break;
        bf16_t* orow = out + ((size_t)b * T1 + t) * ldo;
        for (int q = tid; q < pairs; q += 256) {
            const int c = q / (F2 / 2), f = 2 * (q - c * (F2 / 2));
            const float* w = sw + c * 10;
            float r[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                // output (f+e, t): input rows 2(f+e)-1 .. +1 -> LDS rows 2(f+e) .. +2 ; cols 2tl .. 2tl+2
                const float* m = sm + (2 * (f + e)) * (TW + 1) + 2 * tl;
                float a = w[9];
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) a = fmaf(w[i * 3 + j], m[i * (TW + 1) + j], a);
                r[e] = swishf_(a);
            }
            *reinterpret_cast<uint32_t*>(orow + c * F2 + f) = pack_bf2(r[0], r[1]);
        }
    }
}

constexpr int DW_TT = 64;   // output frames per workgroup
constexpr int DW_CC = 64;   // channels per workgroup (one per lane)

__global__ __launch_bounds__(256) void dwconv_kernel(const bf16_t* __restrict__ g, int T, int To, int C, int ld,
                                                     const float* __restrict__ w_kc, const float* __restrict__ bias,
                                                     int ksize, int stride, bf16_t* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* sg = reinterpret_cast<bf16_t*>(smem);       // [rows][DW_CC]
    const int ctiles = (C + DW_CC - 1) / DW_CC;
    const int ttiles = (To + DW_TT - 1) / DW_TT;
    int id = blockIdx.x;
    const int ct = id % ctiles; id /= ctiles;
    const int tt = id % ttiles; const int b = id / ttiles;
    const int c0 = ct * DW_CC, to0 = tt * DW_TT;
    const int half = (ksize - 1) / 2;
    const int rows = (DW_TT - 1) * stride + ksize;
    const int tin0 = to0 * stride - half;
    const int tid = threadIdx.x;
    // stage rows tin0 .. tin0+rows-1, channels c0..c0+63 (16-byte chunks of 8 channels)
    for (int i = tid; i < rows * (DW_CC / 8); i += 256) {
        const int r = i / (DW_CC / 8), ch = (i - r * (DW_CC / 8)) * 8;
        const int t = tin0 + r, c = c0 + ch;
        const int tc = t < 0 ? 0 : (t < T ? t : T - 1), cc = c < ld - 8 ? c : ld - 8;      // clamped, unconditional load
        const uint4 v = *reinterpret_cast<const uint4*>(g + ((size_t)b * T + tc) * ld + cc);
        *reinterpret_cast<uint4*>(sg + r * DW_CC + ch) = mask_chunk(v, (t >= 0 && t < T && c < ld) ? C - c : 0);
    }
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    const int c = c0 + lane;
    if (c >= ld) return;
    const bool live = c < C;
    float wreg[31];
#pragma unroll
    for (int j = 0; j < 31; ++j) wreg[j] = (live && j < ksize) ? w_kc[j * C + c] : 0.f;
    const float bz = live ? bias[c] : 0.f;
    for (int tl = wave; tl < DW_TT; tl += 4) {
        const int to = to0 + tl;
        if (to >= To) break;
        float a = bz;
        const bf16_t* p = sg + (tl * stride) * DW_CC + lane;
#pragma unroll
        for (int j = 0; j < 31; ++j)
            if (j < ksize) a = fmaf(wreg[j], bf2f(p[j * DW_CC]), a);
        out[((size_t)b * To + to) * ld + c] = live ? f2bf(swishf_(a)) : (bf16_t)0;
    }
}

}  // namespace

int launch_subsample_conv(const float* mel, int B, int F, int Tm, int T1, const float* w9, const float* bias, int C,
                          bf16_t* out, int ldo, hipStream_t s) {
    if (B <= 0 || T1 <= 0) return 0;
    if (F % 4 || (C * (F / 2)) % 2) return -2;
    const int tiles = (T1 + SUB_TT - 1) / SUB_TT;
    const size_t lds = ((size_t)(F + 1) * (2 * SUB_TT + 2) + (size_t)C * 10) * sizeof(float);
    hipLaunchKernelGGL(subsample_conv_kernel, dim3(B * tiles), dim3(256), lds, s, mel, F, Tm, T1, w9, bias, C, out, ldo);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_dwconv(const bf16_t* g, int B, int T, int To, int C, int ld, const float* w_kc, const float* bias,
                  int ksize, int stride, bf16_t* out, hipStream_t s) {
    if (B <= 0 || To <= 0) return 0;
    if (ksize > 31 || ksize < 1 || (ksize & 1) == 0 || ld % 8) return -2;
    const int ctiles = (C + DW_CC - 1) / DW_CC, ttiles = (To + DW_TT - 1) / DW_TT;
    const int rows = (DW_TT - 1) * stride + ksize;
    const size_t lds = (size_t)rows * DW_CC * sizeof(bf16_t);
    hipLaunchKernelGGL(dwconv_kernel, dim3(B * ttiles * ctiles), dim3(256), lds, s, g, T, To, C, ld, w_kc, bias,
                       ksize, stride, out);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
