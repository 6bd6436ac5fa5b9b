// fp32-operand ("exact") kernels: the opt-in precision mode of libeffconf (effconf_encoder_set_option "exact_fp32").
//
// The product path rounds MFMA operands to bf16 (HISTORY.md numerics policy); on random-weight models a few per-frame top-2 logit
// margins are smaller than that rounding, so greedy labels can flip.  This mode runs the same forward (reference
// models/encoders.py:97-142, blocks.py:119-137, attentions.py:549-718, modules.py:232-249, 385-395, 511-525) with fp32 operands
// end to end - fp32 MFMA (v_mfma_f32_32x32x2_f32: exact products, fp32 accumulation) for every GEMM, fp32 attention, fp32
// convolutions, accurate expf / IEEE division for the sigmoids - so that label sequences are identical to the reference's CPU
// fp32 path wherever its margins exceed fp32 summation-order noise (~1e-5).  It is a correctness mode, built simply: ~10x
// slower than the bf16 path and not tuned.
#include "kernels.h"

namespace {

__device__ __forceinline__ float ex_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

constexpr int XBM = 64, XBN = 64, XBK = 16, XPAD = 68;

__global__ __launch_bounds__(256) void ex_gemm_kernel(ExGemmParams p) {
    __shared__ float sA[XBK][XPAD], sB[XBK][XPAD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.x * XBM, n0 = blockIdx.y * XBN;
    const int lrow = tid >> 2, kq = (tid & 3) * 4;
    int am = m0 + lrow; am = am < p.M ? am : p.M - 1;
    const long long arow = p.a_rows ? (long long)(am / p.a_rows) * p.a_pitch + (long long)(am % p.a_rows) * p.a_stride : am;
    int bn = n0 + lrow; bn = bn < p.N ? bn : p.N - 1;
    const float* ap = p.A + arow * p.lda;
    const float* bp = p.W + (long long)bn * p.ldw;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int k0 = 0; k0 < p.K; k0 += XBK) {
        const int k = k0 + kq;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
        if (k < p.K) { a = *reinterpret_cast<const float4*>(ap + k); b = *reinterpret_cast<const float4*>(bp + k); }
        sA[kq + 0][lrow] = a.x; sA[kq + 1][lrow] = a.y; sA[kq + 2][lrow] = a.z; sA[kq + 3][lrow] = a.w;
        sB[kq + 0][lrow] = b.x; sB[kq + 1][lrow] = b.y; sB[kq + 2][lrow] = b.z; sB[kq + 3][lrow] = b.w;
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < XBK; kk += 2) {
            const float fa = sA[kk + (lane >> 5)][wm * 32 + (lane & 31)];
            const float fb = sB[kk + (lane >> 5)][wn * 32 + (lane & 31)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    const int n = n0 + wn * 32 + (lane & 31);
    if (n >= p.N) return;
    const float bz = p.bias ? p.bias[n] : 0.f;
    float* cbase = p.C;
    int ncol = n;
    if (p.split_cols > 0) { cbase += (size_t)(n / p.split_cols) * p.split_stride; ncol = n % p.split_cols; }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m >= p.M) continue;
        float v = acc[r] + bz;
        if (p.epi == 1) v = v * ex_sigmoid(v);
        else if (p.epi == 2) v = p.R[(size_t)m * p.ldr + n] + p.alpha * v;
        const long long crow = p.c_rows ? (long long)(m / p.c_rows) * p.c_pitch + m % p.c_rows : m;
        cbase[crow * p.ldc + ncol] = v;
    }
}

// 3x3 stride-2 pad-1 conv + BatchNorm(eval, as per-channel scale / shift with the conv bias folded in) + Swish  (modules.py:232-249)
// in (B, Cin, F, T) -> out (B, Co, Fo, To), or flat = 1: (B, To, Co*Fo) with feature index co*Fo + fo (modules.py:247 + encoders.py:113)
__global__ __launch_bounds__(256) void ex_conv2d_kernel(const float* __restrict__ in, int B, int Cin, int F, int T, const float* __restrict__ w,
                                                        const float* __restrict__ scale, const float* __restrict__ shift, int Co, int Fo, int To,
                                                        float* __restrict__ out, int flat, const int* __restrict__ tlen) {
    const long long total = (long long)B * Co * Fo * To;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        int to, fo, co, b;
        if (flat) {                     // thread order = the flat output's memory order (b, to, co, fo): coalesced 4-byte stores.  (Until round 4 the
            fo = (int)(idx % Fo);       // threads ran over (b, co, fo, to) in both layouts and the flat one was written with a stride of Co Fo floats:
            long long q = idx / Fo;     // 4.8 ms per 85-utterance range of the Small model, a quarter of the label-exact step.)  Same arithmetic per element.
            co = (int)(q % Co); q /= Co;
            to = (int)(q % To);
            b = (int)(q / To);
        } else {
            to = (int)(idx % To);
            long long q = idx / To;
            fo = (int)(q % Fo); q /= Fo;
            co = (int)(q % Co);
            b = (int)(q / Co);
        }
        const int Tv = tlen ? tlen[b] : T;                       // ragged batches: the utterance's own frames (zero behind them)
        if (tlen && !flat && to >= (Tv - 1) / 2 + 1) { out[(((size_t)b * Co + co) * Fo + fo) * To + to] = 0.f; continue; }
        float acc = 0.f;
        for (int ci = 0; ci < Cin; ++ci) {
            const float* ip = in + ((size_t)b * Cin + ci) * F * T;
            const float* wp = w + ((size_t)co * Cin + ci) * 9;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int f = 2 * fo - 1 + i;
                if (f < 0 || f >= F) continue;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int t = 2 * to - 1 + j;
                    if (t < 0 || t >= Tv) continue;
                    acc = fmaf(ip[(size_t)f * T + t], wp[i * 3 + j], acc);
                }
            }
        }
        float y = acc * scale[co] + shift[co];
        y = y * ex_sigmoid(y);
        if (flat) out[((size_t)b * To + to) * ((size_t)Co * Fo) + (size_t)co * Fo + fo] = y;
        else out[(((size_t)b * Co + co) * Fo + fo) * To + to] = y;
    }
}

// The one-layer subsampler (Cin = 1) in the flat layout, one thread per (utterance, output frame, output frequency): the 3 x 3 mel patch
// is loaded ONCE (zero outside the image) and all Co channels are computed from it - 9 loads per Co outputs instead of 9 per output, the
// weights / scale / shift are wave-uniform (scalar loads), and for a fixed channel consecutive lanes store consecutive floats.  The sum runs
// over the taps in ex_conv2d_kernel's order with fmaf(0, w, acc) = acc for the taps that kernel skips: bit-identical to it (13.7 -> ~1.5 ms
// per Small step of the label-exact modes).
__global__ __launch_bounds__(256) void ex_conv2d_flat1_kernel(const float* __restrict__ in, int B, int F, int T, const float* __restrict__ w,
                                                              const float* __restrict__ scale, const float* __restrict__ shift, int Co, int Fo, int To,
                                                              float* __restrict__ out, const int* __restrict__ tlen) {
    const long long total = (long long)B * To * Fo;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int fo = (int)(idx % Fo);
        const long long q = idx / Fo;
        const int to = (int)(q % To), b = (int)(q / To);
        const float* ip = in + (size_t)b * F * T;
        const int Tv = tlen ? tlen[b] : T;
        float pt[9];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int f = 2 * fo - 1 + i;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int t = 2 * to - 1 + j;
                const bool ok = f >= 0 && f < F && t >= 0 && t < Tv;
                pt[i * 3 + j] = ok ? ip[(size_t)(ok ? f : 0) * T + (ok ? t : 0)] : 0.f;
            }
        }
        float* op = out + ((size_t)b * To + to) * ((size_t)Co * Fo) + fo;
        for (int co = 0; co < Co; ++co) {
            const float* wp = w + (size_t)co * 9;
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < 9; ++k) acc = fmaf(pt[k], wp[k], acc);
            float y = acc * scale[co] + shift[co];
            y = y * ex_sigmoid(y);
            op[(size_t)co * Fo] = y;
        }
    }
}

__global__ __launch_bounds__(256) void ex_glu_kernel(const float* __restrict__ in, long long M, int N, float* __restrict__ out) {
    const long long total = M * N;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const long long m = idx / N; const int n = (int)(idx % N);
        const float a = in[m * 2 * N + n], g = in[m * 2 * N + N + n];
        out[idx] = a * ex_sigmoid(g);                               // GLU over channels (activations.py:37-39)
    }
}

// depthwise conv, k taps, "same" zero padding, stride s, BatchNorm folded into w_kc / bias, Swish (modules.py:516-518; layers.py:100, 122-136)
__global__ __launch_bounds__(256) void ex_dwconv_kernel(const float* __restrict__ g, int B, int T, int To, int C, const float* __restrict__ w_kc,
                                                        const float* __restrict__ bias, int ks, int stride, float* __restrict__ out) {
    const long long total = (long long)B * To * C;
    const int half = (ks - 1) / 2;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int c = (int)(idx % C);
        const long long q = idx / C;
        const int to = (int)(q % To), b = (int)(q / To);
        float acc = 0.f;
        for (int j = 0; j < ks; ++j) {
            const int t = stride * to + j - half;
            if (t >= 0 && t < T) acc = fmaf(w_kc[(size_t)j * C + c], g[((size_t)b * T + t) * C + c], acc);
        }
        const float y = acc + bias[c];
        out[idx] = y * ex_sigmoid(y);
    }
}

// The same depthwise convolution with the input window of TO consecutive output frames of one channel held in registers: S (TO - 1) + KS
// loads for TO outputs instead of KS per output (2.9 ms -> per Small step of the label-exact modes for the kernel above: 15 dependent L2 round
// trips per output).  Consecutive lanes <-> consecutive channels (coalesced); taps in registers; every output sums its taps in ascending order
// with fmaf(w, 0, acc) = acc for the taps the kernel above skips: bit-identical to it.
template <int KS, int S>
__global__ __launch_bounds__(256) void ex_dwconv_tiled_kernel(const float* __restrict__ g, int B, int T, int To, int C, const float* __restrict__ w_kc,
                                                              const float* __restrict__ bias, float* __restrict__ out) {
    constexpr int TO = 8, W = S * (TO - 1) + KS, HALF = (KS - 1) / 2;
    const int ntile = (To + TO - 1) / TO;
    const long long total = (long long)B * ntile * C;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int c = (int)(idx % C);
        const long long q = idx / C;
        const int tile = (int)(q % ntile), b = (int)(q / ntile);
        const int to0 = tile * TO, t0 = S * to0 - HALF;
        const float* gp = g + (size_t)b * T * C + c;
        float x[W], w[KS];
#pragma unroll
        for (int i = 0; i < W; ++i) {
            const int t = t0 + i;
            const bool ok = t >= 0 && t < T;
            x[i] = ok ? gp[(size_t)(ok ? t : 0) * C] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < KS; ++j) w[j] = w_kc[(size_t)j * C + c];
        const float bz = bias[c];
#pragma unroll
        for (int o = 0; o < TO; ++o) {
            if (to0 + o >= To) break;
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < KS; ++j) acc = fmaf(w[j], x[S * o + j], acc);
            const float y = acc + bz;
            out[((size_t)b * To + to0 + o) * C + c] = y * ex_sigmoid(y);
        }
    }
}

// one wave per (utterance, head, grouped query row): S = ((Q+u) K^T + (Q+v) E[Tg-1+j-i]^T) / sqrt(d), additive -1e9 key mask,
// softmax, P V  (attentions.py:549-718; closed form SURVEY.md 8a-6)
__global__ __launch_bounds__(64) void ex_attention_kernel(ExAttnParams p) {
    extern __shared__ float sm[];
    float* qu = sm;
    float* qv = sm + p.d;
    float* sc = sm + 2 * p.d;
    const int i = blockIdx.x, h = blockIdx.y, b = blockIdx.z, lane = threadIdx.x;
    const size_t row0 = ((size_t)b * p.Tp + (size_t)p.G * i) * p.D + (size_t)h * p.d;
    for (int x = lane; x < p.d; x += 64) {
        const int n = (h * p.d + x) % p.D;
        const float q = p.q[row0 + x];
        qu[x] = q + p.u[n];
        qv[x] = q + p.vb[n];
    }
    __syncthreads();
    const int len = p.lens[b];
    const float rs = sqrtf((float)p.d);
    float mx = -INFINITY;
    for (int j = lane; j < p.Tg; j += 64) {
        const float* kr = p.k + ((size_t)b * p.Tp + (size_t)p.G * j) * p.D + (size_t)h * p.d;
        const float* er = p.e + (size_t)p.G * (p.Tg - 1 + j - i) * p.D + (size_t)h * p.d;
        float s1 = 0.f, s2 = 0.f;
        for (int x = 0; x < p.d; ++x) { s1 = fmaf(qu[x], kr[x], s1); s2 = fmaf(qv[x], er[x], s2); }
        float s = (s1 + s2) / rs;
        if (p.G * j >= len || j - i > p.band_r || i - j > p.band_l) s += -1e9f;      // attentions.py:698-701, 1377-1403: ONE additive mask = max(padding, streaming)
        sc[j] = s;
        mx = fmaxf(mx, s);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float sum = 0.f;
    for (int j = lane; j < p.Tg; j += 64) { const float e = expf(sc[j] - mx); sc[j] = e; sum += e; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if (p.att) for (int j = lane; j < p.Tg; j += 64) p.att[(((size_t)b * p.H + h) * p.Tg + i) * p.Tg + j] = sc[j] / sum;
    __syncthreads();
    for (int x = lane; x < p.d; x += 64) {
        float acc = 0.f;
        for (int j = 0; j < p.Tg; ++j) acc = fmaf(sc[j] / sum, p.v[((size_t)b * p.Tp + (size_t)p.G * j) * p.D + (size_t)h * p.d + x], acc);
        p.out[row0 + x] = acc;
    }
}

// The same arithmetic, tiled: a workgroup (XQT / 4 waves) owns XQT = 32 (16 when LDS is short) grouped query rows of one (utterance, head) and walks the keys in blocks of
// 64.  The one-wave-per-row kernel above re-reads every K / E / V row from L2 for every query row (3.6 TFLOP/s: 100 of the 131 ms of an exact
// step); here a key block's K rows, the XQT + 63 rows of E its (query, key) pairs touch and later its V rows are staged once in LDS and shared
// by its rows.  Lane <-> key in the score phase (each lane 4 query rows: K fragment read once, E fragment per row at row offset
// lane - r, Q + u / Q + v rows as broadcast reads), lane <-> output column in the P V phase.  Every sum runs in the order of the kernel above
// (x ascending for the dot products, j ascending for P V, the same per-lane partial maxima / sums and butterflies for the softmax), and the
// zero pad columns only add fmaf(0, 0, s) = s: the two kernels agree BIT FOR BIT (tests/test_gpu_round3.py compares them).
constexpr int XKB = 64;

// global rows -> LDS rows of pitch P (zero pad columns): one thread per 4-column group, rows strided over the workgroup; 8-byte loads
// when the head spans are 8-byte aligned (even head width)
template <class F>
__device__ __forceinline__ void ex_stage_rows(float* dst, int nrows, int P, int d, int r0, int c, int rstep, bool vec2, F rowptr) {
    if (r0 >= rstep) return;                          // tail threads that do not fill a whole row group
    for (int r = r0; r < nrows; r += rstep) {
        const float* src = rowptr(r) + c;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (vec2) {
            if (c < d) { const float2 a = *reinterpret_cast<const float2*>(src); v.x = a.x; v.y = a.y; }
            if (c + 2 < d) { const float2 b = *reinterpret_cast<const float2*>(src + 2); v.z = b.x; v.w = b.y; }
        } else {
            if (c < d) v.x = src[0];
            if (c + 1 < d) v.y = src[1];
            if (c + 2 < d) v.z = src[2];
            if (c + 3 < d) v.w = src[3];
        }
        *reinterpret_cast<float4*>(dst + r * P + c) = v;
    }
}

template <int XQT>
__global__ __launch_bounds__(XQT * 16) void ex_attention2_kernel(ExAttnParams p, int P, int TgP) {
    constexpr int NTH = XQT * 16;                     // 4 query rows per wave
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* sq = sm;                                   // [2][XQT][P]: Q + u | Q + v, pad columns zero
    float* sk = sq + 2 * XQT * P;                     // [XKB][P]: K block (score phase), V block (P V phase)
    float* se = sk + XKB * P;                         // [XKB + XQT - 1][P]: E rows Tg - 1 + (jb - i0) - (XQT - 1) + w
    float* sc = se + (XKB + XQT - 1) * P;             // [XQT][TgP]: scores, then probabilities
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i0 = blockIdx.x * XQT, h = blockIdx.y, b = blockIdx.z;
    const int d = p.d, Tg = p.Tg;
    const size_t hb = (size_t)h * d;
    for (int idx = tid; idx < XQT * P; idx += NTH) {
        const int r = idx / P, x = idx - r * P;
        const int i = i0 + r < Tg ? i0 + r : Tg - 1;
        float a = 0.f, c = 0.f;
        if (x < d) {
            const int n = (int)((hb + x) % p.D);
            const float q = p.q[((size_t)b * p.Tp + (size_t)p.G * i) * p.D + hb + x];
            a = q + p.u[n]; c = q + p.vb[n];
        }
        sq[idx] = a; sq[XQT * P + idx] = c;
    }
    const int len = p.lens[b];
    const float rs = sqrtf((float)d);
    const int tpr = P >> 2, sr0 = tid / tpr, scol = (tid - sr0 * tpr) * 4, srstep = NTH / tpr;      // staging: this thread's row phase / columns
    const bool vec2 = (d & 1) == 0;
    const size_t gd = (size_t)p.G * p.D;
    const float* kbase = p.k + (size_t)b * p.Tp * p.D + hb;
    const float* vbase = p.v + (size_t)b * p.Tp * p.D + hb;
    const float* ebase = p.e + hb;
    for (int jb = 0; jb < Tg; jb += XKB) {
        __syncthreads();                              // the previous block's readers are done (first pass: sq complete)
        ex_stage_rows(sk, XKB, P, d, sr0, scol, srstep, vec2, [&](int r) { const int j = jb + r < Tg ? jb + r : Tg - 1; return kbase + gd * j; });
        ex_stage_rows(se, XKB + XQT - 1, P, d, sr0, scol, srstep, vec2, [&](int w) {
            int rel = Tg - 1 + jb - i0 - (XQT - 1) + w;
            rel = rel < 0 ? 0 : (rel > 2 * Tg - 2 ? 2 * Tg - 2 : rel);       // rows no valid (i, j) pair touches
            return ebase + gd * rel; });
        __syncthreads();
        float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
        const float* kr = sk + lane * P;
        const float* er = se + (lane + XQT - 1 - 4 * wave) * P;             // row of (i0 + 4 wave, jb + lane); next query row: one row up
        const float* qa = sq + 4 * wave * P;
        for (int x = 0; x < P; x += 4) {
            const float4 kq = *reinterpret_cast<const float4*>(kr + x);
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const float4 a = *reinterpret_cast<const float4*>(qa + rr * P + x), c = *reinterpret_cast<const float4*>(qa + (XQT + rr) * P + x);
                const float4 eq = *reinterpret_cast<const float4*>(er - rr * P + x);
                s1[rr] = fmaf(a.x, kq.x, s1[rr]); s1[rr] = fmaf(a.y, kq.y, s1[rr]); s1[rr] = fmaf(a.z, kq.z, s1[rr]); s1[rr] = fmaf(a.w, kq.w, s1[rr]);
                s2[rr] = fmaf(c.x, eq.x, s2[rr]); s2[rr] = fmaf(c.y, eq.y, s2[rr]); s2[rr] = fmaf(c.z, eq.z, s2[rr]); s2[rr] = fmaf(c.w, eq.w, s2[rr]);
            }
        }
        const int j = jb + lane;
        if (j < Tg) {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                float s = (s1[rr] + s2[rr]) / rs;
                const int qi = i0 + 4 * wave + rr;
                if (p.G * j >= len || j - qi > p.band_r || qi - j > p.band_l) s += -1e9f;      // attentions.py:698-701, 1377-1403: ONE additive mask = max(padding, streaming)
                sc[(4 * wave + rr) * TgP + j] = s;
            }
        }
    }
    __syncthreads();
    // softmax of this wave's 4 rows (the partial maxima / sums of the kernel above: lane l owns keys l, l + 64, ...)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        float* row = sc + (4 * wave + rr) * TgP;
        float mx = -INFINITY;
        for (int j = lane; j < Tg; j += 64) mx = fmaxf(mx, row[j]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        float sum = 0.f;
        for (int j = lane; j < Tg; j += 64) { const float e = expf(row[j] - mx); row[j] = e; sum += e; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        for (int j = lane; j < TgP; j += 64) row[j] = j < Tg ? row[j] / sum : 0.f;
        const int i = i0 + 4 * wave + rr;
        if (p.att && i < Tg) for (int j = lane; j < Tg; j += 64) p.att[(((size_t)b * p.H + h) * Tg + i) * Tg + j] = row[j];
    }
    // P V: lane <-> output columns lane, lane + 64, lane + 128
    float acc[4][3];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) { acc[rr][0] = 0.f; acc[rr][1] = 0.f; acc[rr][2] = 0.f; }
    for (int jb = 0; jb < Tg; jb += XKB) {
        __syncthreads();
        ex_stage_rows(sk, XKB, P, d, sr0, scol, srstep, vec2, [&](int r) { const int j = jb + r < Tg ? jb + r : Tg - 1; return vbase + gd * j; });
        __syncthreads();
        int nj = Tg - jb < XKB ? Tg - jb : XKB;
        nj = (nj + 3) & ~3;                           // probabilities of the pad keys are zero, their V rows finite: fmaf(0, v, acc) = acc
        const bool c1 = lane + 64 < P, c2 = lane + 128 < P;
        for (int jj = 0; jj < nj; jj += 4) {
            float v0[4], v1[4], v2[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float* vr = sk + (jj + t) * P;
                v0[t] = vr[lane]; v1[t] = c1 ? vr[lane + 64] : 0.f; v2[t] = c2 ? vr[lane + 128] : 0.f;
            }
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const float4 pr = *reinterpret_cast<const float4*>(sc + (4 * wave + rr) * TgP + jb + jj);
                acc[rr][0] = fmaf(pr.x, v0[0], acc[rr][0]); acc[rr][0] = fmaf(pr.y, v0[1], acc[rr][0]); acc[rr][0] = fmaf(pr.z, v0[2], acc[rr][0]); acc[rr][0] = fmaf(pr.w, v0[3], acc[rr][0]);
                acc[rr][1] = fmaf(pr.x, v1[0], acc[rr][1]); acc[rr][1] = fmaf(pr.y, v1[1], acc[rr][1]); acc[rr][1] = fmaf(pr.z, v1[2], acc[rr][1]); acc[rr][1] = fmaf(pr.w, v1[3], acc[rr][1]);
                acc[rr][2] = fmaf(pr.x, v2[0], acc[rr][2]); acc[rr][2] = fmaf(pr.y, v2[1], acc[rr][2]); acc[rr][2] = fmaf(pr.z, v2[2], acc[rr][2]); acc[rr][2] = fmaf(pr.w, v2[3], acc[rr][2]);
            }
        }
    }
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int i = i0 + 4 * wave + rr;
        if (i >= Tg) continue;
        float* orow = p.out + ((size_t)b * p.Tp + (size_t)p.G * i) * p.D + hb;
        if (lane < d) orow[lane] = acc[rr][0];
        if (lane + 64 < d) orow[lane + 64] = acc[rr][1];
        if (lane + 128 < d) orow[lane + 128] = acc[rr][2];
    }
}

inline int grid_for(long long total) { long long g = (total + 255) / 256; return (int)(g < 1 ? 1 : (g > 65535 ? 65535 : g)); }

}  // namespace

int launch_ex_gemm(const ExGemmParams& p, hipStream_t s) {
    if (p.M <= 0 || p.N <= 0) return 0;
    if (p.K % 4 || p.lda % 4 || p.ldw % 4) return -2;
    hipLaunchKernelGGL(ex_gemm_kernel, dim3((p.M + XBM - 1) / XBM, (p.N + XBN - 1) / XBN), dim3(256), 0, s, p);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_ex_conv2d(const float* in, int B, int Cin, int F, int T, const float* w, const float* scale, const float* shift, int Co,
                     float* out, int flat, hipStream_t s, const int* tlen) {
    const int Fo = (F - 1) / 2 + 1, To = (T - 1) / 2 + 1;
    if (Cin == 1 && flat) {
        hipLaunchKernelGGL(ex_conv2d_flat1_kernel, dim3(grid_for((long long)B * To * Fo)), dim3(256), 0, s, in, B, F, T, w, scale, shift, Co, Fo, To, out, tlen);
        return hipGetLastError() == hipSuccess ? 0 : -1;
    }
    hipLaunchKernelGGL(ex_conv2d_kernel, dim3(grid_for((long long)B * Co * Fo * To)), dim3(256), 0, s, in, B, Cin, F, T, w, scale, shift, Co, Fo, To, out, flat, tlen);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_ex_glu(const float* in, long long M, int N, float* out, hipStream_t s) {
    if (M <= 0) return 0;
    hipLaunchKernelGGL(ex_glu_kernel, dim3(grid_for(M * N)), dim3(256), 0, s, in, M, N, out);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_ex_dwconv(const float* g, int B, int T, int To, int C, const float* w_kc, const float* bias, int ks, int stride, float* out, hipStream_t s) {
    {
        const int grid = grid_for((long long)B * ((To + 7) / 8) * C);
#define EX_DW_CASE(K, S) if (ks == K && stride == S) { hipLaunchKernelGGL((ex_dwconv_tiled_kernel<K, S>), dim3(grid), dim3(256), 0, s, g, B, T, To, C, w_kc, bias, out); \
                                                       return hipGetLastError() == hipSuccess ? 0 : -1; }
        EX_DW_CASE(15, 1) EX_DW_CASE(15, 2) EX_DW_CASE(31, 1) EX_DW_CASE(31, 2)
#undef EX_DW_CASE
    }
    hipLaunchKernelGGL(ex_dwconv_kernel, dim3(grid_for((long long)B * To * C)), dim3(256), 0, s, g, B, T, To, C, w_kc, bias, ks, stride, out);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_ex_attention(const ExAttnParams& p, hipStream_t s) {
    if (p.variant != 1 && p.d <= 192) {                // tiled kernel unless its LDS image does not fit (very long utterances at wide heads)
        int P = (p.d + 3) / 4 * 4;
        if (((P / 4) & 1) == 0) P += 4;                // P / 4 odd: the 16-byte reads of 16 consecutive rows fall into 16 different bank groups
        const int TgP = (p.Tg + 3) / 4 * 4;
        auto lds_for = [&](int qt) { return ((size_t)(2 * qt + XKB + XKB + qt - 1) * P + (size_t)qt * TgP) * 4; };
        static LdsAttr attr32, attr16;
        if (p.variant != 2 && lds_for(32) <= 160 * 1024) {               // 8 waves: two per SIMD (variant 2: the 16-row shape, for tests)
            ensure_dynamic_lds(reinterpret_cast<const void*>(&ex_attention2_kernel<32>), (int)lds_for(32), attr32);
            hipLaunchKernelGGL(ex_attention2_kernel<32>, dim3((p.Tg + 31) / 32, p.H, p.B), dim3(512), lds_for(32), s, p, P, TgP);
            return hipGetLastError() == hipSuccess ? 0 : -1;
        }
        if (lds_for(16) <= 160 * 1024) {
            ensure_dynamic_lds(reinterpret_cast<const void*>(&ex_attention2_kernel<16>), (int)lds_for(16), attr16);
            hipLaunchKernelGGL(ex_attention2_kernel<16>, dim3((p.Tg + 15) / 16, p.H, p.B), dim3(256), lds_for(16), s, p, P, TgP);
            return hipGetLastError() == hipSuccess ? 0 : -1;
        }
    }
    const size_t lds = (size_t)(2 * p.d + p.Tg) * 4;
    if (lds > 64 * 1024) return -2;
    hipLaunchKernelGGL(ex_attention_kernel, dim3(p.Tg, p.H, p.B), dim3(64), lds, s, p);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
