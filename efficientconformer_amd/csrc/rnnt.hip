// RNN-T greedy decode (gfx950): prediction network (Embedding + 1-layer LSTM), joint network and the greedy loop.
//
// Reference: Transducer.gready_search_decoding (models/transducer.py:139-186), RnnDecoder.forward
// (models/decoders.py:41-70: Embedding -> nn.LSTM, batch_first, 1 layer), JointNetwork.forward in decoding mode
// (models/joint_networks.py:80-104: linear_encoder(f) + linear_decoder(g) -> tanh -> linear_joint), and
// max_consec_dec_step (transducer.py:83, 173).
//
// The reference decodes one utterance at a time with one joint_network call (three nn.Linear, a softmax and an
// .argmax() host sync) per (frame, token) decision.  Here:
//   * everything input-independent is hoisted into tables at finalize():  Gin[y] = W_ih emb[y] + b_ih + b_hh for every
//     token id y (the LSTM only ever sees embedding rows), so a decoder step is one 4H x H mat-vec;
//   * linear_encoder(f) is one fp32 GEMM over all (b, t) per batch (the reference recomputes it per decision,
//     joint_networks.py:82);
//   * the sequential part runs as ONE persistent workgroup per utterance: state (h, c, g_dec, z) lives in LDS, weights
//     stream from L2 in a k-major float4 layout (W4[k/4][n] = W[n][4k/4 .. +3]) so that a wave's loads are contiguous
//     1 KB rows, and RNNT_KF consecutive encoder frames are evaluated speculatively per pass over the joint weight
//     (between two emitted tokens the decoder output does not change, so the extra frames are exactly the frames the
//     reference would evaluate next as long as it keeps reading blanks).
// All arithmetic is fp32 (the decisions feed back into the recurrence: one flipped argmax changes every later token,
// so the head is kept at the reference's precision; softmax().log().argmax() of transducer.py:164 is argmax(logits)).
// Bound: L2 -> CU bandwidth on the weights (4H*H*4 + J*H*4 bytes per token, V*J*4 per joint pass).
#include "kernels.h"
#include "../../include/effconf.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

int ec_fail(const char* msg);

namespace {

constexpr int NT = 512;        // threads per utterance
constexpr int KF = 4;          // encoder frames per joint pass
constexpr int JMC = 2;         // joint columns per thread and pass

// ------------------------------------------------------------------------------------------------------------------
// fp32 GEMM  C[m][n] = sum_k A[m][k] * B[n][k] + bias[n]   (nn.Linear semantics), K % 4 == 0.  64x64x16 tiles.
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sgemm_nt_kernel(const float* __restrict__ A, int lda, const float* __restrict__ Bw, int ldb,
                                                       const float* __restrict__ bias, float* __restrict__ C, int ldc, int M, int N, int K) {
    __shared__ float As[16][68], Bs[16][68];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int lr = tid >> 2, lk = (tid & 3) * 4;
    const int am = m0 + lr < M ? m0 + lr : M - 1, bn = n0 + lr < N ? n0 + lr : N - 1;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    for (int k0 = 0; k0 < K; k0 += 16) {
        const int kk = k0 + lk < K ? k0 + lk : K - 4;                       // clamped, masked below
        float4 a = *reinterpret_cast<const float4*>(A + (size_t)am * lda + kk);
        float4 b = *reinterpret_cast<const float4*>(Bw + (size_t)bn * ldb + kk);
        if (k0 + lk >= K) { a = make_float4(0.f, 0.f, 0.f, 0.f); b = a; }
        __syncthreads();
        As[lk + 0][lr] = a.x; As[lk + 1][lr] = a.y; As[lk + 2][lr] = a.z; As[lk + 3][lr] = a.w;
        Bs[lk + 0][lr] = b.x; Bs[lk + 1][lr] = b.y; Bs[lk + 2][lr] = b.z; Bs[lk + 3][lr] = b.w;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const float4 av = *reinterpret_cast<const float4*>(&As[k][ty * 4]);
            const float4 bv = *reinterpret_cast<const float4*>(&Bs[k][tx * 4]);
            const float ar[4] = {av.x, av.y, av.z, av.w}, br[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(ar[i], br[j], acc[i][j]);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + ty * 4 + i;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + tx * 4 + j;
            if (n < N) C[(size_t)m * ldc + n] = acc[i][j] + (bias ? bias[n] : 0.f);
        }
    }
}

int launch_sgemm_nt(const float* A, int lda, const float* Bw, int ldb, const float* bias, float* C, int ldc, int M, int N, int K, hipStream_t s) {
    if (M <= 0 || N <= 0) return 0;
    if (K % 4 || K < 4) return -2;
    hipLaunchKernelGGL(sgemm_nt_kernel, dim3((N + 63) / 64, (M + 63) / 64), dim3(256), 0, s, A, lda, Bw, ldb, bias, C, ldc, M, N, K);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ------------------------------------------------------------------------------------------------------------------
// mat-vec against a k-major float4 weight image: y[n] = sum_k W[n][k] x[k], W4[(k/4)*N + n] = W[n][k..k+3].
// Thread owns columns n = n0 + tid + NT*m (m < MC); x is NX vectors in LDS (pitch xp floats), read as broadcasts.
// ------------------------------------------------------------------------------------------------------------------
template <int MC, int NX>
__device__ __forceinline__ void matvec(const float4* __restrict__ W4, int N, int K4, int n0, const float* xs, int xp, float (&acc)[MC][NX]) {
    int col[MC];
#pragma unroll
    for (int m = 0; m < MC; ++m) {
        const int n = n0 + (int)threadIdx.x + NT * m;
        col[m] = n < N ? n : N - 1;                                          // clamped: garbage columns are never used
#pragma unroll
        for (int x = 0; x < NX; ++x) acc[m][x] = 0.f;
    }
#pragma unroll 4
    for (int k4 = 0; k4 < K4; ++k4) {
        float4 w[MC];
#pragma unroll
        for (int m = 0; m < MC; ++m) w[m] = W4[(size_t)k4 * N + col[m]];
#pragma unroll
        for (int x = 0; x < NX; ++x) {
            const float4 xv = *reinterpret_cast<const float4*>(xs + x * xp + 4 * k4);
#pragma unroll
            for (int m = 0; m < MC; ++m)
                acc[m][x] = fmaf(w[m].w, xv.w, fmaf(w[m].z, xv.z, fmaf(w[m].y, xv.y, fmaf(w[m].x, xv.x, acc[m][x]))));
        }
    }
}

struct RnntDev {
    const float* gin;        // [V][4H]   W_ih emb[y] + b_ih + b_hh
    const float4* whh4;      // [H/4][4H]
    const float4* wd4;       // [H/4][J]
    const float* bd;         // [J]
    const float4* wj4;       // [J/4][V]
    const float* bj;         // [V]
    int H, J, V, max_consec;
};

__device__ __forceinline__ float sigmoid_precise(float x) { return 1.0f / (1.0f + expf(-x)); }

// One workgroup = one utterance.  fe: [B][T][J] = linear_encoder(f) incl. bias.
__global__ __launch_bounds__(NT) void rnnt_greedy_kernel(RnntDev w, const float* __restrict__ fe, const int64_t* __restrict__ lens,
                                                         int T, int* __restrict__ tokens, int* __restrict__ counts, int max_tok) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int H = w.H, J = w.J, V = w.V;
    float* sh = lds;                       // [H]   h
    float* sc = sh + H;                    // [H]   c
    float* sg = sc + H;                    // [4H]  gate pre-activations
    float* sgd = sg + 4 * H;               // [J]   linear_decoder(h)
    float* sz = sgd + J;                   // [KF][J] tanh(fe + gd)
    float* sbv = sz + KF * J;              // [KF][NT/64] wave-best values
    int* sbi = reinterpret_cast<int*>(sbv + KF * (NT / 64));   // [KF][NT/64] wave-best indices
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x;
    int Tb = (int)lens[b];
    Tb = Tb < 0 ? 0 : (Tb > T ? T : Tb);
    for (int i = tid; i < 2 * H; i += NT) sh[i] = 0.f;   // hidden = None -> zeros (decoders.py:59-60)
    __syncthreads();
    int y = 0, enc_step = 0, consec = 0, ntok = 0;       // y = x.new_zeros(1,1) (transducer.py:151)
    const float* feb = fe + (size_t)b * T * J;

    while (enc_step < Tb) {
        // ---- decoder step: gates = Gin[y] + W_hh h   (torch LSTM gate order i, f, g, o)
        for (int n0 = 0; n0 < 4 * H; n0 += NT * 5) {
            float acc[5][1];
            matvec<5, 1>(w.whh4, 4 * H, H / 4, n0, sh, H, acc);
#pragma unroll
            for (int m = 0; m < 5; ++m) {
                const int n = n0 + tid + NT * m;
                if (n < 4 * H) sg[n] = acc[m][0] + w.gin[(size_t)y * 4 * H + n];
            }
        }
        __syncthreads();
        for (int j = tid; j < H; j += NT) {
            const float ig = sigmoid_precise(sg[j]), fg = sigmoid_precise(sg[H + j]), gg = tanhf(sg[2 * H + j]), og = sigmoid_precise(sg[3 * H + j]);
            const float c = fg * sc[j] + ig * gg;
            sc[j] = c;
            sh[j] = og * tanhf(c);
        }
        __syncthreads();
        // ---- gd = linear_decoder(h)
        for (int n0 = 0; n0 < J; n0 += NT * 2) {
            float acc[2][1];
            matvec<2, 1>(w.wd4, J, H / 4, n0, sh, H, acc);
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int n = n0 + tid + NT * m;
                if (n < J) sgd[n] = acc[m][0] + w.bd[n];
            }
        }
        __syncthreads();
        // ---- joint loop: KF frames per pass, until a token is emitted or the utterance ends
        bool emitted = false;
        while (!emitted && enc_step < Tb) {
            for (int i = tid; i < KF * J; i += NT) {
                const int kf = i / J, j = i - kf * J;
                const int t = enc_step + kf < Tb ? enc_step + kf : Tb - 1;    // clamped: frames past the end are never consulted
                sz[i] = tanhf(feb[(size_t)t * J + j] + sgd[j]);
            }
            __syncthreads();
            float bv[KF]; int bi[KF];
#pragma unroll
            for (int kf = 0; kf < KF; ++kf) { bv[kf] = -INFINITY; bi[kf] = 0x7fffffff; }
            for (int n0 = 0; n0 < V; n0 += NT * JMC) {
                float acc[JMC][KF];
                matvec<JMC, KF>(w.wj4, V, J / 4, n0, sz, J, acc);
#pragma unroll
                for (int m = 0; m < JMC; ++m) {
                    const int n = n0 + tid + NT * m;
                    const bool ok = n < V;
                    const float bz = w.bj[ok ? n : V - 1];
#pragma unroll
                    for (int kf = 0; kf < KF; ++kf) {
                        const float v = acc[m][kf] + bz;
                        if (ok && (v > bv[kf] || (v == bv[kf] && n < bi[kf]))) { bv[kf] = v; bi[kf] = n; }
                    }
                }
            }
#pragma unroll
            for (int kf = 0; kf < KF; ++kf) {
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) {
                    const float ov = __shfl_xor(bv[kf], o);
                    const int oi = __shfl_xor(bi[kf], o);
                    if (ov > bv[kf] || (ov == bv[kf] && oi < bi[kf])) { bv[kf] = ov; bi[kf] = oi; }
                }
                if (lane == 0) { sbv[kf * (NT / 64) + wave] = bv[kf]; sbi[kf * (NT / 64) + wave] = bi[kf]; }
            }
            __syncthreads();
            // ---- every thread walks the KF decisions identically (transducer.py:158-176)
#pragma unroll
            for (int kf = 0; kf < KF; ++kf) {
                if (emitted || enc_step >= Tb) break;
                float v = sbv[kf * (NT / 64)]; int pred = sbi[kf * (NT / 64)];
#pragma unroll
                for (int q = 1; q < NT / 64; ++q) {
                    const float ov = sbv[kf * (NT / 64) + q]; const int oi = sbi[kf * (NT / 64) + q];
                    if (ov > v || (ov == v && oi < pred)) { v = ov; pred = oi; }
                }
                if (pred == 0 || consec == w.max_consec) { consec = 0; ++enc_step; }
                else {
                    ++consec;
                    if (tid == 0 && ntok < max_tok) tokens[(size_t)b * max_tok + ntok] = pred;
                    ++ntok;
                    y = pred;
                    emitted = true;
                }
            }
            __syncthreads();     // sz / sbv are rewritten by the next pass
        }
    }
    if (tid == 0) counts[b] = ntok < max_tok ? ntok : max_tok;
    for (int i = ntok + tid; i < max_tok; i += NT) tokens[(size_t)b * max_tok + i] = 0;
}


// ------------------------------------------------------------------------------------------------------------------
// Cluster decode: CW workgroups share CU utterances and run them in lockstep rounds.  Every workgroup owns 1/CW of the
// LSTM's hidden units (their 4 gate columns and cell state), of the decoder projection's columns and of the vocabulary, so the
// weights a round streams from L2 are read ONCE per cluster instead of once per utterance, and the per-step latency of an
// utterance drops from "10.7 MB through one CU" to "1.3 MB through each of CW CUs" plus three cluster barriers
// (counter in global memory: agent-scope release / acquire, bounded spin).  State machines (token, frame, counters) are
// replicated: every workgroup takes the same decisions from the same exchanged argmax candidates.  The arithmetic (k order
// of every dot product, libm tanhf / expf) is that of rnnt_greedy_kernel, so both produce identical tokens.
// ------------------------------------------------------------------------------------------------------------------
constexpr int CW = 8;          // workgroups per cluster
constexpr int CU = 8;          // utterances per cluster
constexpr int CKF = 2;         // encoder frames per joint pass
constexpr int CNT = 512;
constexpr int CLB = 8;         // weight loads kept in flight per thread (H/4 and J/4 must be multiples)

struct ClusterArgs {
    RnntDev w;
    const float* fe; const int64_t* lens; int T, B;
    int* tokens; int* counts; int max_tok;
    float* xh; float* xgd; float* xav; int* xai; unsigned* cnt; int* status;     // exchange buffers (per cluster), barrier counters
    int ncl;
};

// Exchange data and the barrier counter are accessed with relaxed agent-scope atomics (sc1 loads / stores served at the coherent
// level) instead of plain accesses bracketed by agent-scope release / acquire fences: the fences cost a full L2 write-back +
// invalidate per workgroup per barrier (~40 us each here; three barriers per round made the cluster no faster than one workgroup per
// utterance).  Ordering: every wave drains its stores (s_waitcnt vmcnt(0) in __syncthreads) before thread 0 bumps the counter; the
// consumers read only after thread 0 has seen the counter and a second __syncthreads.
__device__ __forceinline__ void xstore(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void xstorei(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float xload(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int xloadi(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ bool cluster_sync(unsigned* cnt, unsigned& epoch, int* status) {
    __syncthreads();
    epoch += CW;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        long spins = 0;
        while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1l << 26)) { __hip_atomic_store(status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }   // never hang the GPU
        }
    }
    __syncthreads();
    return __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0;
}

__global__ __launch_bounds__(CNT) void rnnt_cluster_kernel(const ClusterArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const RnntDev& w = a.w;
    const int H = w.H, J = w.J, V = w.V;
    const int HU = H / CW, JU = J / CW, VU = (V + CW - 1) / CW;
    float* sh = lds;                         // [CU][H]   h of every utterance
    float* sgd = sh + CU * H;                // [CU][J]
    float* sz = sgd + CU * J;                // [CU*CKF][J]
    float* sc = sz + CU * CKF * J;           // [CU][HU]  cell state of this workgroup's units
    float* sg = sc + CU * HU;                // [CU][4*HU] gate pre-activations
    float* sred = sg + CU * 4 * HU;          // [CU*CKF][8] partial argmax values, then indices
    int* sredi = reinterpret_cast<int*>(sred + CU * CKF * 8);
    __shared__ int s_y[CU], s_step[CU], s_consec[CU], s_ntok[CU], s_need[CU], s_T[CU], s_pred[CU * CKF];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cl = blockIdx.x % a.ncl, wg = blockIdx.x / a.ncl;      // cluster members are a.ncl apart: same XCD when ncl % 8 == 0
    const int u0 = cl * CU;
    float* xh = a.xh + (size_t)cl * CU * H;
    float* xgd = a.xgd + (size_t)cl * CU * J;
    float* xav = a.xav + (size_t)cl * CW * CU * CKF;
    int* xai = a.xai + (size_t)cl * CW * CU * CKF;
    unsigned* cnt = a.cnt + cl * 64;          // one counter per 256-byte line
    unsigned epoch = 0;

    for (int i = tid; i < CU * H; i += CNT) sh[i] = 0.f;             // hidden = None -> zeros
    for (int i = tid; i < CU * J; i += CNT) sgd[i] = 0.f;
    for (int i = tid; i < CU * HU; i += CNT) sc[i] = 0.f;
    if (tid < CU) {
        const int b = u0 + tid;
        int Tb = b < a.B ? (int)a.lens[b] : 0;
        Tb = Tb < 0 ? 0 : (Tb > a.T ? a.T : Tb);
        s_T[tid] = Tb; s_y[tid] = 0; s_step[tid] = 0; s_consec[tid] = 0; s_ntok[tid] = 0; s_need[tid] = Tb > 0;
    }
    __syncthreads();

    bool ok = true;
    while (ok) {
        bool active = false, dec = false;
#pragma unroll
        for (int u = 0; u < CU; ++u) { const bool act = s_step[u] < s_T[u]; active |= act; dec |= act && s_need[u]; }
        if (!active) break;
        if (dec) {
            // ---- phase 1: gates of this workgroup's units, cell update, h slice out
            if (tid < 4 * HU) {
                const int gate = tid / HU, unit = tid - gate * HU;
                const int n = gate * H + wg * HU + unit;
                float acc[CU];
#pragma unroll
                for (int u = 0; u < CU; ++u) acc[u] = 0.f;
                const float4* __restrict__ wp = w.whh4 + n;
                float4 wn[CLB];
#pragma unroll
                for (int i = 0; i < CLB; ++i) wn[i] = wp[(size_t)i * 4 * H];
                for (int k0 = 0; k0 < H / 4; k0 += CLB) {                  // next batch of weight loads in flight under this batch's FMAs
                    float4 wv[CLB];
#pragma unroll
                    for (int i = 0; i < CLB; ++i) wv[i] = wn[i];
                    const int kn = k0 + CLB < H / 4 ? k0 + CLB : k0;       // last iteration: harmless re-load
#pragma unroll
                    for (int i = 0; i < CLB; ++i) wn[i] = wp[(size_t)(kn + i) * 4 * H];
#pragma unroll
                    for (int i = 0; i < CLB; ++i)
#pragma unroll
                        for (int u = 0; u < CU; ++u) {
                            const float4 xv = *reinterpret_cast<const float4*>(sh + u * H + 4 * (k0 + i));
                            acc[u] = fmaf(wv[i].w, xv.w, fmaf(wv[i].z, xv.z, fmaf(wv[i].y, xv.y, fmaf(wv[i].x, xv.x, acc[u]))));
                        }
                }
#pragma unroll
                for (int u = 0; u < CU; ++u) sg[u * 4 * HU + tid] = acc[u] + w.gin[(size_t)s_y[u] * 4 * H + n];
            }
            __syncthreads();
            for (int i = tid; i < CU * HU; i += CNT) {
                const int u = i / HU, unit = i - u * HU;
                float hv = sh[u * H + wg * HU + unit];
                if (s_need[u] && s_step[u] < s_T[u]) {
                    const float* g = sg + u * 4 * HU + unit;
                    const float ig = sigmoid_precise(g[0]), fg = sigmoid_precise(g[HU]), gg = tanhf(g[2 * HU]), og = sigmoid_precise(g[3 * HU]);
                    const float c = fg * sc[i] + ig * gg;
                    sc[i] = c;
                    hv = og * tanhf(c);
                }
                xstore(xh + u * H + wg * HU + unit, hv);
            }
            ok = cluster_sync(cnt, epoch, a.status);
            for (int i = tid; i < CU * H; i += CNT) sh[i] = xload(xh + i);
            __syncthreads();
            // ---- phase 2: this workgroup's columns of linear_decoder(h)
            if (tid < JU * 4) {
                const int lc = tid % JU, ug = tid / JU;                   // 4 groups of CU/4 utterances
                const int n = wg * JU + lc;
                float acc[CU / 4];
#pragma unroll
                for (int q = 0; q < CU / 4; ++q) acc[q] = 0.f;
                const float4* __restrict__ wp = w.wd4 + n;
                float4 wn[CLB];
#pragma unroll
                for (int i = 0; i < CLB; ++i) wn[i] = wp[(size_t)i * J];
                for (int k0 = 0; k0 < H / 4; k0 += CLB) {
                    float4 wv[CLB];
#pragma unroll
                    for (int i = 0; i < CLB; ++i) wv[i] = wn[i];
                    const int kn = k0 + CLB < H / 4 ? k0 + CLB : k0;
#pragma unroll
                    for (int i = 0; i < CLB; ++i) wn[i] = wp[(size_t)(kn + i) * J];
#pragma unroll
                    for (int i = 0; i < CLB; ++i)
#pragma unroll
                        for (int q = 0; q < CU / 4; ++q) {
                            const float4 xv = *reinterpret_cast<const float4*>(sh + (ug * (CU / 4) + q) * H + 4 * (k0 + i));
                            acc[q] = fmaf(wv[i].w, xv.w, fmaf(wv[i].z, xv.z, fmaf(wv[i].y, xv.y, fmaf(wv[i].x, xv.x, acc[q]))));
                        }
                }
#pragma unroll
                for (int q = 0; q < CU / 4; ++q) {
                    const int u = ug * (CU / 4) + q;
                    const bool upd = s_need[u] && s_step[u] < s_T[u];
                    xstore(xgd + u * J + n, upd ? acc[q] + w.bd[n] : sgd[u * J + n]);
                }
            }
            ok = cluster_sync(cnt, epoch, a.status) && ok;
            for (int i = tid; i < CU * J; i += CNT) sgd[i] = xload(xgd + i);
            __syncthreads();
        }
        // ---- phase 3: joint on CKF frames per utterance, this workgroup's slice of the vocabulary
        for (int i = tid; i < CU * CKF * J; i += CNT) {
            const int row = i / J, j = i - row * J, u = row / CKF, kf = row - u * CKF;
            const int b = u0 + u < a.B ? u0 + u : a.B - 1;
            int t = s_step[u] + kf;
            t = t < s_T[u] ? t : (s_T[u] > 0 ? s_T[u] - 1 : 0);
            sz[i] = tanhf(a.fe[((size_t)b * a.T + t) * J + j] + sgd[u * J + j]);
        }
        __syncthreads();
        {
            constexpr int RG = CNT / 128, RPG = CU * CKF / RG;          // 4 row groups of 4 rows
            const int lc = tid & 127, rg = tid >> 7;
            const int n = wg * VU + lc;
            const bool colok = lc < VU && n < V;
            const int nc = colok ? n : V - 1;
            float acc[RPG];
#pragma unroll
            for (int q = 0; q < RPG; ++q) acc[q] = 0.f;
            const float4* __restrict__ wp = w.wj4 + nc;
            float4 wn[CLB];
#pragma unroll
            for (int i = 0; i < CLB; ++i) wn[i] = wp[(size_t)i * V];
            for (int k0 = 0; k0 < J / 4; k0 += CLB) {
                float4 wv[CLB];
#pragma unroll
                for (int i = 0; i < CLB; ++i) wv[i] = wn[i];
                const int kn = k0 + CLB < J / 4 ? k0 + CLB : k0;
#pragma unroll
                for (int i = 0; i < CLB; ++i) wn[i] = wp[(size_t)(kn + i) * V];
#pragma unroll
                for (int i = 0; i < CLB; ++i)
#pragma unroll
                    for (int q = 0; q < RPG; ++q) {
                        const float4 xv = *reinterpret_cast<const float4*>(sz + (rg * RPG + q) * J + 4 * (k0 + i));
                        acc[q] = fmaf(wv[i].w, xv.w, fmaf(wv[i].z, xv.z, fmaf(wv[i].y, xv.y, fmaf(wv[i].x, xv.x, acc[q]))));
                    }
            }
            const float bz = w.bj[nc];
#pragma unroll
            for (int q = 0; q < RPG; ++q) {
                float bv = colok ? acc[q] + bz : -INFINITY; int bi = colok ? n : 0x7fffffff;
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) {
                    const float ov = __shfl_xor(bv, o); const int oi = __shfl_xor(bi, o);
                    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
                }
                if (lane == 0) { sred[(rg * RPG + q) * 8 + (wave & 1)] = bv; sredi[(rg * RPG + q) * 8 + (wave & 1)] = bi; }
            }
        }
        __syncthreads();
        if (tid < CU * CKF) {
            float bv = sred[tid * 8]; int bi = sredi[tid * 8];
            const float ov = sred[tid * 8 + 1]; const int oi = sredi[tid * 8 + 1];
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
            xstore(xav + wg * CU * CKF + tid, bv); xstorei(xai + wg * CU * CKF + tid, bi);
        }
        ok = cluster_sync(cnt, epoch, a.status) && ok;
        if (tid < CU * CKF) {
            float bv = xload(xav + tid); int bi = xloadi(xai + tid);
            for (int q = 1; q < CW; ++q) {
                const float ov = xload(xav + q * CU * CKF + tid); const int oi = xloadi(xai + q * CU * CKF + tid);
                if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
            }
            s_pred[tid] = bi;
        }
        __syncthreads();
        // ---- decisions (transducer.py:158-176), identical in every workgroup
        if (tid < CU) {
            const int u = tid, b = u0 + u;
            int step = s_step[u], consec = s_consec[u], ntok = s_ntok[u], need = 0;
            for (int kf = 0; kf < CKF && step < s_T[u]; ++kf) {
                const int pred = s_pred[u * CKF + kf];
                if (pred == 0 || consec == w.max_consec) { consec = 0; ++step; }
                else {
                    ++consec;
                    if (wg == 0 && b < a.B && ntok < a.max_tok) a.tokens[(size_t)b * a.max_tok + ntok] = pred;
                    ++ntok;
                    s_y[u] = pred;
                    need = 1;
                    break;
                }
            }
            s_step[u] = step; s_consec[u] = consec; s_ntok[u] = ntok; s_need[u] = need;
        }
        __syncthreads();
    }
    if (wg == 0) {
        for (int u = 0; u < CU; ++u) {
            const int b = u0 + u;
            if (b >= a.B) continue;
            const int ntok = s_ntok[u] < a.max_tok ? s_ntok[u] : a.max_tok;
            if (tid == 0) a.counts[b] = ntok;
            for (int i = ntok + tid; i < a.max_tok; i += CNT) a.tokens[(size_t)b * a.max_tok + i] = 0;
        }
    }
}

inline size_t cluster_exchange_bytes(int ncl, int H, int J) {
    return (size_t)ncl * ((size_t)CU * H * 4 + (size_t)CU * J * 4 + (size_t)CW * CU * CKF * 8 + 256) + 256;
}
inline bool cluster_supported(const EcRnntConfig& c) {
    return c.dim_decoder % (4 * CW) == 0 && c.dim_joint % (4 * CW) == 0 && (c.dim_decoder / 4) % CLB == 0 && (c.dim_joint / 4) % CLB == 0 && 4 * (c.dim_decoder / CW) <= CNT && 4 * (c.dim_joint / CW) <= CNT &&
           (c.vocab_size + CW - 1) / CW <= 128;
}

struct HostT { std::vector<int64_t> shape; std::vector<float> data; };

}  // namespace

struct EcRnnt {
    EcRnntConfig cfg;
    std::map<std::string, HostT> host;
    std::vector<void*> allocs;
    RnntDev dev{};
    float* we = nullptr;     // linear_encoder.weight [J][De]
    float* be = nullptr;
    bool finalized = false;
    int cluster_mode = -1;   // -1 auto (cluster decode for batches >= 2*CU), 0 per-utterance kernel, 1 force cluster
};

namespace {

void* upload(EcRnnt* r, const void* src, size_t bytes) {
    void* d = nullptr;
    if (hipMalloc(&d, bytes) != hipSuccess) return nullptr;
    r->allocs.push_back(d);
    if (src && hipMemcpy(d, src, bytes, hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    return d;
}

// W [N][K] row-major -> float4 image [K/4][N]
std::vector<float> kmajor4(const std::vector<float>& W, int N, int K) {
    std::vector<float> o((size_t)N * K);
    for (int k4 = 0; k4 < K / 4; ++k4)
        for (int n = 0; n < N; ++n)
            for (int e = 0; e < 4; ++e) o[((size_t)k4 * N + n) * 4 + e] = W[(size_t)n * K + 4 * k4 + e];
    return o;
}

const HostT* need(EcRnnt* r, const char* key, std::initializer_list<int64_t> shape, std::string& err) {
    auto it = r->host.find(key);
    if (it == r->host.end()) { err = std::string("missing tensor ") + key; return nullptr; }
    if (it->second.shape != std::vector<int64_t>(shape)) { err = std::string("bad shape for ") + key; return nullptr; }
    return &it->second;
}

}  // namespace

extern "C" {

EcRnnt* effconf_rnnt_create(const EcRnntConfig* c) {
    if (!c) { ec_fail("null config"); return nullptr; }
    if (c->num_layers != 1) { ec_fail("RnnDecoder with num_layers == 1 is native (every shipped Transducer config)"); return nullptr; }
    if (c->joint_mode != 0 || c->joint_act != 0) { ec_fail("joint network: 'sum' + 'tanh' is native (every shipped Transducer config)"); return nullptr; }
    if (c->dim_encoder % 4 || c->dim_decoder % 4 || c->dim_joint % 4 || c->dim_encoder < 4 || c->dim_decoder < 4 || c->dim_joint < 4 ||
        c->vocab_size < 2 || c->max_consec_dec_step < 0) { ec_fail("dims must be positive multiples of 4"); return nullptr; }
    EcRnnt* r = new EcRnnt();
    r->cfg = *c;
    return r;
}

void effconf_rnnt_destroy(EcRnnt* r) {
    if (!r) return;
    for (void* p : r->allocs) (void)hipFree(p);
    delete r;
}

int effconf_rnnt_load_tensor(EcRnnt* r, const char* key, const float* host, const int64_t* shape, int32_t ndim) {
    if (!r || !key || !host) return ec_fail("null argument");
    HostT t;
    int64_t n = 1;
    for (int i = 0; i < ndim; ++i) { t.shape.push_back(shape[i]); n *= shape[i]; }
    t.data.assign(host, host + n);
    r->host[key] = std::move(t);
    r->finalized = false;
    return 0;
}

int effconf_rnnt_finalize(EcRnnt* r) {
    if (!r) return ec_fail("null handle");
    for (void* p : r->allocs) (void)hipFree(p);
    r->allocs.clear();
    const int H = r->cfg.dim_decoder, J = r->cfg.dim_joint, V = r->cfg.vocab_size, De = r->cfg.dim_encoder;
    std::string err;
    const HostT* emb = need(r, "decoder.embedding.weight", {V, H}, err);
    const HostT* wih = need(r, "decoder.rnn.weight_ih_l0", {4 * H, H}, err);
    const HostT* whh = need(r, "decoder.rnn.weight_hh_l0", {4 * H, H}, err);
    const HostT* bih = need(r, "decoder.rnn.bias_ih_l0", {4 * H}, err);
    const HostT* bhh = need(r, "decoder.rnn.bias_hh_l0", {4 * H}, err);
    const HostT* we = need(r, "joint_network.linear_encoder.weight", {J, De}, err);
    const HostT* be = need(r, "joint_network.linear_encoder.bias", {J}, err);
    const HostT* wd = need(r, "joint_network.linear_decoder.weight", {J, H}, err);
    const HostT* bd = need(r, "joint_network.linear_decoder.bias", {J}, err);
    const HostT* wj = need(r, "joint_network.linear_joint.weight", {V, J}, err);
    const HostT* bj = need(r, "joint_network.linear_joint.bias", {V}, err);
    if (!emb || !wih || !whh || !bih || !bhh || !we || !be || !wd || !bd || !wj || !bj) return ec_fail(err.c_str());
    std::vector<float> bsum(4 * H);
    for (int i = 0; i < 4 * H; ++i) bsum[i] = bih->data[i] + bhh->data[i];
    const std::vector<float> whh4 = kmajor4(whh->data, 4 * H, H), wd4 = kmajor4(wd->data, J, H), wj4 = kmajor4(wj->data, V, J);
    float* d_emb = (float*)upload(r, emb->data.data(), emb->data.size() * 4);
    float* d_wih = (float*)upload(r, wih->data.data(), wih->data.size() * 4);
    float* d_bsum = (float*)upload(r, bsum.data(), bsum.size() * 4);
    float* d_gin = (float*)upload(r, nullptr, (size_t)V * 4 * H * 4);
    r->dev.whh4 = (const float4*)upload(r, whh4.data(), whh4.size() * 4);
    r->dev.wd4 = (const float4*)upload(r, wd4.data(), wd4.size() * 4);
    r->dev.bd = (const float*)upload(r, bd->data.data(), bd->data.size() * 4);
    r->dev.wj4 = (const float4*)upload(r, wj4.data(), wj4.size() * 4);
    r->dev.bj = (const float*)upload(r, bj->data.data(), bj->data.size() * 4);
    r->we = (float*)upload(r, we->data.data(), we->data.size() * 4);
    r->be = (float*)upload(r, be->data.data(), be->data.size() * 4);
    if (!d_emb || !d_wih || !d_bsum || !d_gin || !r->dev.whh4 || !r->dev.wd4 || !r->dev.bd || !r->dev.wj4 || !r->dev.bj || !r->we || !r->be)
        return ec_fail("device allocation / upload failed");
    // Gin[y] = W_ih emb[y] + (b_ih + b_hh)   for every token id (Embedding rows are the only LSTM inputs, decoders.py:55)
    if (launch_sgemm_nt(d_emb, H, d_wih, H, d_bsum, d_gin, 4 * H, V, 4 * H, H, nullptr) != 0) return ec_fail("Gin GEMM launch failed");
    if (hipDeviceSynchronize() != hipSuccess) return ec_fail("Gin GEMM failed");
    r->dev.gin = d_gin;
    r->dev.H = H; r->dev.J = J; r->dev.V = V; r->dev.max_consec = r->cfg.max_consec_dec_step;
    r->finalized = true;
    return 0;
}

size_t effconf_rnnt_workspace_bytes(const EcRnnt* r, int32_t batch, int32_t t_out) {
    if (!r || batch < 0 || t_out < 0) return 0;
    const int ncl = (batch + CU - 1) / CU;
    return (size_t)batch * t_out * r->cfg.dim_joint * 4 + 256 + cluster_exchange_bytes(ncl, r->cfg.dim_decoder, r->cfg.dim_joint);
}

int effconf_rnnt_set_option(EcRnnt* r, const char* name, int32_t value) {
    if (!r || !name) return ec_fail("null argument");
    if (!strcmp(name, "cluster_decode")) { r->cluster_mode = value; return 0; }
    return ec_fail("unknown option");
}

int32_t effconf_rnnt_max_tokens(const EcRnnt* r, int32_t t_out) {
    if (!r || t_out < 0) return 0;
    const int m = r->cfg.max_consec_dec_step * t_out;
    return m > 1 ? m : 1;
}

int effconf_rnnt_greedy(EcRnnt* r, const float* enc_out, const int64_t* out_len, int32_t batch, int32_t t_out,
                        int32_t* tokens, int32_t* token_len, int32_t max_tokens, void* workspace, size_t workspace_bytes, void* stream) {
    if (!r || !r->finalized) return ec_fail("rnnt handle not finalized");
    if (batch == 0) return 0;
    if (!enc_out || !out_len || !tokens || !token_len || !workspace) return ec_fail("null argument");
    if (batch < 0 || t_out <= 0 || max_tokens < effconf_rnnt_max_tokens(r, t_out)) return ec_fail("bad shape / token buffer smaller than max_consec_dec_step * T_out");
    if (workspace_bytes < effconf_rnnt_workspace_bytes(r, batch, t_out)) return ec_fail("workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const int H = r->cfg.dim_decoder, J = r->cfg.dim_joint, De = r->cfg.dim_encoder;
    float* fe = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    // linear_encoder(f) for every frame of the batch, once (joint_networks.py:82 recomputes it per decision)
    if (launch_sgemm_nt(enc_out, De, r->we, De, r->be, fe, J, batch * t_out, J, De, s) != 0) return ec_fail("linear_encoder GEMM launch failed");
    // auto: clusters only while they need at most half of the CUs (two decodes on two streams must not starve each other's
    // cluster-mates: a spinning workgroup keeps its CU)
    const bool cluster = cluster_supported(r->cfg) && (r->cluster_mode == 1 || (r->cluster_mode < 0 && batch >= 2 * CU && ((batch + CU - 1) / CU) * CW <= 128));
    if (cluster) {
        const int ncl = (batch + CU - 1) / CU;
        if (ncl * CW > 256) return ec_fail("cluster decode needs every workgroup resident: batch <= 256");
        char* ex = reinterpret_cast<char*>(fe) + (size_t)batch * t_out * J * 4;
        ex = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(ex) + 255) & ~(uintptr_t)255);
        ClusterArgs a{};
        a.w = r->dev; a.fe = fe; a.lens = out_len; a.T = t_out; a.B = batch; a.tokens = tokens; a.counts = token_len; a.max_tok = max_tokens;
        a.ncl = ncl;
        a.cnt = reinterpret_cast<unsigned*>(ex); a.status = reinterpret_cast<int*>(ex + (size_t)ncl * 256);
        char* p = ex + (size_t)ncl * 256 + 256;
        a.xh = reinterpret_cast<float*>(p); p += (size_t)ncl * CU * H * 4;
        a.xgd = reinterpret_cast<float*>(p); p += (size_t)ncl * CU * J * 4;
        a.xav = reinterpret_cast<float*>(p); p += (size_t)ncl * CW * CU * CKF * 4;
        a.xai = reinterpret_cast<int*>(p);
        if (hipMemsetAsync(ex, 0, (size_t)ncl * 256 + 256, s) != hipSuccess) return ec_fail("memset failed");
        const int HU = H / CW;
        const size_t lds = (size_t)(CU * H + CU * J + CU * CKF * J + CU * HU + CU * 4 * HU + CU * CKF * 16) * 4;
        static LdsAttr attr;
        ensure_dynamic_lds(reinterpret_cast<const void*>(&rnnt_cluster_kernel), (int)lds, attr);
        hipLaunchKernelGGL(rnnt_cluster_kernel, dim3(ncl * CW), dim3(CNT), lds, s, a);
        return hipGetLastError() == hipSuccess ? 0 : ec_fail("rnnt cluster launch failed");
    }
    const size_t lds = (size_t)(2 * H + 4 * H + J + KF * J + 2 * KF * (NT / 64)) * 4;
    hipLaunchKernelGGL(rnnt_greedy_kernel, dim3(batch), dim3(NT), lds, s, r->dev, fe, out_len, t_out, tokens, token_len, max_tokens);
    return hipGetLastError() == hipSuccess ? 0 : ec_fail("rnnt_greedy launch failed");
}

}  // extern "C"
