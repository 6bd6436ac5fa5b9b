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
#include <type_traits>

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
    const float4 *whh16, *wd16, *wj16;   // the same three in the MFMA order: [K/16][4][N] (kperm16), null if K % 16
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
// replicated: every workgroup takes the same decisions from the same exchanged argmax candidates.  The three mat-vec phases run on
// the fp32 matrix pipe (mfma_rows16 below) with the k order of every dot product ascending as in rnnt_greedy_kernel, the rest (libm
// tanhf / expf, the decisions) is that kernel's code: both produce identical tokens (tests: the reference's goldens and 165 k tokens
// of tools/rnnt_diag.py).  Round 3, per round of a 256-utterance decode: 206 k -> 112 k cycles (140 -> 46 ms per decode).
// ------------------------------------------------------------------------------------------------------------------
// Two cluster shapes <CW workgroups, CU utterances, CKF encoder frames per joint pass>, CU * CKF = 16 = the columns of an MFMA tile:
//   <8, 8, 2>    round 3's: 32 clusters at B = 256, every one of them streams the whole 10.7 MB (Transducer-Medium) per round - 342 MB per round through the
//                L2s, which is what a round costs; the gate / decoder-projection tiles use 8 of their 16 columns
//   <16, 16, 1>  round 6 (option cluster_shape = 1): 16 clusters - half the bytes per round and per workgroup, all 16 columns of every tile used; the state of 16
//                utterances (h, linear_decoder(h), one joint input each) is 3 x 16 x 640 x 4 B = 123 KB of LDS, so no second speculative frame.
//                MEASURED (profiles/r6_33_rnnt_cluster_shapes.txt): 41.9 ms against 42.4 ms at B = 256 - and B = 64 (8 clusters instead of 32) takes the same
//                40 ms: a decode is ~1090 lockstep rounds of ~38 us whatever the bytes per round, i.e. the round is a LATENCY chain (three cluster barriers +
//                exchange reloads + three dependent 160-MFMA accumulator chains), not the L2 stream the byte count suggests.  Kept as an option, not the default
//                (a blank-only decode costs 3.4 ms instead of 1.9: one frame per joint pass).
constexpr int CNT = 512;
constexpr int CLB = 8;         // weight loads kept in flight per thread (H/4 and J/4 must be multiples)

struct ClusterArgs {
    RnntDev w;
    const float* fe; const int64_t* lens; int T, B;
    int* tokens; int* counts; int max_tok;
    float* xh; float* xgd; float* xav; int* xai; unsigned* cnt; int* status;     // exchange buffers (per cluster), barrier counters
    int ncl, by_slice;
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

template <int CW>
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

// ---- the three mat-vec phases on the fp32 matrix pipe -----------------------------------------------------------------------------------
// A round multiplies a slice of a weight matrix with the CU = 8 (joint: CU * CKF = 16) state vectors of the cluster's utterances.  As VALU
// code that is 32 FMAs + 8 broadcast LDS reads per weight float4 and thread: the phases were issue-bound (s_memtime: 71k + 37k + 77k of a
// round's 206k cycles; the three cluster barriers 1.5k each).  v_mfma_f32_16x16x4_f32 takes 16 weight rows x 4 k (A) against 4 k x 16 state
// vectors (B): one weight float4 and one state float4 per lane feed four MFMAs (4096 MACs).  The k order of every dot product stays
// ASCENDING: lane group g = lane / 16 is k-slot g of an MFMA, so the float4 a lane loads for 16-block q must hold k = 16q + g, 16q + 4 + g,
// 16q + 8 + g, 16q + 12 + g (MFMA c of the block then covers k = 16q + 4c .. 16q + 4c + 3) - the weights get a second image in that order
// (kperm16) and the state vectors sit in LDS with k permuted the same way (kperm).
__host__ __device__ __forceinline__ int kperm(int k) { return (k & ~15) | ((k & 3) << 2) | ((k >> 2) & 3); }
constexpr int CMB = 4;          // 16-blocks of weight loads in flight per tile and lane (8 when a wave owns at most two tiles and K allows)

template <int NTL, int CMB>
__device__ __forceinline__ void mfma_rows16_t(const float4* __restrict__ W16, int N, int K16, const int (&nrow)[NTL], const float* xs_lane, f32x4 (&acc)[NTL]) {
    const int g = (threadIdx.x & 63) >> 4;
    const size_t bs = (size_t)4 * N;                    // float4s per 16-block: [g][n]
    const float4* wp[NTL];
    float4 wn[NTL][CMB];
#pragma unroll
    for (int i = 0; i < NTL; ++i) {
        wp[i] = W16 + (size_t)g * N + nrow[i];
#pragma unroll
        for (int b = 0; b < CMB; ++b) wn[i][b] = wp[i][b * bs];
    }
    for (int q0 = 0; q0 < K16; q0 += CMB) {
        float4 wv[NTL][CMB];
#pragma unroll
        for (int i = 0; i < NTL; ++i)
#pragma unroll
            for (int b = 0; b < CMB; ++b) wv[i][b] = wn[i][b];
        const int qn = q0 + CMB < K16 ? q0 + CMB : q0;                      // last pass: harmless re-load
#pragma unroll
        for (int i = 0; i < NTL; ++i)
#pragma unroll
            for (int b = 0; b < CMB; ++b) wn[i][b] = wp[i][(size_t)(qn + b) * bs];
#pragma unroll
        for (int b = 0; b < CMB; ++b) {
            const float4 x = *reinterpret_cast<const float4*>(xs_lane + 16 * (q0 + b));
#pragma unroll
            for (int i = 0; i < NTL; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[i][b].x, x.x, acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[i][b].y, x.y, acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[i][b].z, x.z, acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[i][b].w, x.w, acc[i], 0, 0, 0);
            }
        }
    }
}

template <int NTL>
__device__ __forceinline__ void mfma_rows16(const float4* __restrict__ W16, int N, int K16, const int (&nrow)[NTL], const float* xs_lane, f32x4 (&acc)[NTL]) {
    if (NTL <= 2 && (K16 & 7) == 0) mfma_rows16_t<NTL, 8>(W16, N, K16, nrow, xs_lane, acc);
    else mfma_rows16_t<NTL, 4>(W16, N, K16, nrow, xs_lane, acc);
}

// exchange buffers -> registers, three 16-byte loads per lane in flight (sc1: served at the coherent level, as xload; plain loads instead of
// atomics so that they pipeline - the relaxed atomic loads were issued one at a time, 13k cycles per round for two 20 KB reloads).  The
// cluster barrier in front orders them against the producers' stores.
__device__ __forceinline__ void xload3(const float* p0, const float* p1, const float* p2, float4& a, float4& b, float4& c) {
    asm volatile("global_load_dwordx4 %0, %3, off sc1\n\tglobal_load_dwordx4 %1, %4, off sc1\n\tglobal_load_dwordx4 %2, %5, off sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(a), "=&v"(b), "=&v"(c) : "v"(p0), "v"(p1), "v"(p2) : "memory");
}

template <int CW, int CU, int CKF>
__global__ __launch_bounds__(CNT) void rnnt_cluster_kernel(const ClusterArgs a) {
    static_assert(CU * CKF == 16 && CU <= 16, "one MFMA column tile of joint inputs");
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
    // workgroup -> (cluster, slice).  by_slice: consecutive workgroups are the CW = 8 slices of one cluster, i.e. slice k of EVERY cluster
    // runs on XCD k (workgroups are dealt to the 8 XCDs round-robin) and that XCD's L2 only ever holds slice k of the weights (1.3 of the
    // 10.7 MB: resident); otherwise the members of a cluster share an XCD (ncl % 8 == 0) and its L2 sees all 10.7 MB per round.
    const int cl = a.by_slice ? blockIdx.x / CW : blockIdx.x % a.ncl, wg = a.by_slice ? blockIdx.x % CW : blockIdx.x / a.ncl;
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
            // ---- phase 1: gates of this workgroup's units (4 HU rows of W_hh as 16-row tiles, dealt to the waves), cell update, h slice out
            {
                const int ntile = 4 * HU / 16;                             // HU % 4 == 0 (cluster_supported)
                const int j = lane & 15, g = lane >> 4;
                const float* xl = sh + (j % CU) * H + 4 * g;                // columns 8 .. 15 of B repeat the 8 utterances (results unused)
                auto run = [&](auto ntl_c) __attribute__((always_inline)) {
                    constexpr int NTL = decltype(ntl_c)::value;
                    int nrow[NTL];
                    f32x4 acc[NTL];
#pragma unroll
                    for (int i = 0; i < NTL; ++i) {
                        const int r = 16 * (wave + 8 * i) + j;             // local gate row of this lane's A operand
                        nrow[i] = (r / HU) * H + wg * HU + r % HU;
                        acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                    mfma_rows16<NTL>(w.whh16, 4 * H, H / 16, nrow, xl, acc);
                    if (j < CU) {
#pragma unroll
                        for (int i = 0; i < NTL; ++i)
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const int r = 16 * (wave + 8 * i) + 4 * g + e;
                                const int n = (r / HU) * H + wg * HU + r % HU;
                                sg[j * 4 * HU + r] = acc[i][e] + w.gin[(size_t)s_y[j] * 4 * H + n];
                            }
                    }
                };
                const int mine = (ntile - wave + 7) / 8;                   // tiles wave, wave + 8, ...
                if (mine == 1) run(std::integral_constant<int, 1>{});
                else if (mine == 2) run(std::integral_constant<int, 2>{});
                else if (mine == 3) run(std::integral_constant<int, 3>{});
                else if (mine >= 4) run(std::integral_constant<int, 4>{});
            }
            __syncthreads();
            for (int i = tid; i < CU * HU; i += CNT) {
                const int u = i / HU, unit = i - u * HU;
                float hv = sh[u * H + kperm(wg * HU + unit)];
                if (s_need[u] && s_step[u] < s_T[u]) {
                    const float* g = sg + u * 4 * HU + unit;
                    const float ig = sigmoid_precise(g[0]), fg = sigmoid_precise(g[HU]), gg = tanhf(g[2 * HU]), og = sigmoid_precise(g[3 * HU]);
                    const float c = fg * sc[i] + ig * gg;
                    sc[i] = c;
                    hv = og * tanhf(c);
                }
                xstore(xh + u * H + wg * HU + unit, hv);
            }
            ok = cluster_sync<CW>(cnt, epoch, a.status);
            for (int i0 = tid; i0 < CU * H / 4; i0 += 3 * CNT) {         // h of every utterance, k-permuted (see mfma_rows16)
                const int n4 = CU * H / 4, i1 = i0 + CNT, i2 = i0 + 2 * CNT;
                float4 v[3];
                xload3(xh + 4 * i0, xh + 4 * (i1 < n4 ? i1 : i0), xh + 4 * (i2 < n4 ? i2 : i0), v[0], v[1], v[2]);
#pragma unroll
                for (int e = 0; e < 3; ++e) {
                    const int i = 4 * (i0 + e * CNT);
                    if (i < CU * H) {
                        float* d = sh + (i / H) * H + kperm(i % H);           // elements i .. i + 3 sit 4 floats apart
                        d[0] = v[e].x; d[4] = v[e].y; d[8] = v[e].z; d[12] = v[e].w;
                    }
                }
            }
            __syncthreads();
            // ---- phase 2: this workgroup's columns of linear_decoder(h): JU rows of W_d as 16-row tiles
            {
                const int ntile = (JU + 15) / 16;
                const int j = lane & 15, g = lane >> 4;
                for (int t = wave; t < ntile; t += 8) {
                    int nrow[1];
                    const int lr = 16 * t + j;
                    nrow[0] = wg * JU + (lr < JU ? lr : JU - 1);
                    f32x4 acc[1] = {f32x4{0.f, 0.f, 0.f, 0.f}};
                    mfma_rows16<1>(w.wd16, J, H / 16, nrow, sh + (j % CU) * H + 4 * g, acc);
                    if (j < CU) {
                        const int u = j;
                        const bool upd = s_need[u] && s_step[u] < s_T[u];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int lc = 16 * t + 4 * g + e, n = wg * JU + lc;
                            if (lc < JU) xstore(xgd + u * J + n, upd ? acc[0][e] + w.bd[n] : sgd[u * J + n]);
                        }
                    }
                }
            }
            ok = cluster_sync<CW>(cnt, epoch, a.status) && ok;
            for (int i0 = tid; i0 < CU * J / 4; i0 += 3 * CNT) {
                const int n4 = CU * J / 4, i1 = i0 + CNT, i2 = i0 + 2 * CNT;
                float4 v[3];
                xload3(xgd + 4 * i0, xgd + 4 * (i1 < n4 ? i1 : i0), xgd + 4 * (i2 < n4 ? i2 : i0), v[0], v[1], v[2]);
                *reinterpret_cast<float4*>(sgd + 4 * i0) = v[0];
                if (i1 < n4) *reinterpret_cast<float4*>(sgd + 4 * i1) = v[1];
                if (i2 < n4) *reinterpret_cast<float4*>(sgd + 4 * i2) = v[2];
            }
            __syncthreads();
        }
        // ---- phase 3: joint on CKF frames per utterance, this workgroup's slice of the vocabulary
        for (int q0 = tid; q0 < CU * CKF * J / 4; q0 += 4 * CNT) {       // four 16-byte loads of linear_encoder(f) in flight per lane
            const int n4 = CU * CKF * J / 4, J4 = J / 4;
            float4 fv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int q = q0 + e * CNT < n4 ? q0 + e * CNT : q0;
                const int row = q / J4, u = row / CKF, kf = row - u * CKF;
                const int b = u0 + u < a.B ? u0 + u : a.B - 1;
                int t = s_step[u] + kf;
                t = t < s_T[u] ? t : (s_T[u] > 0 ? s_T[u] - 1 : 0);
                fv[e] = *reinterpret_cast<const float4*>(a.fe + ((size_t)b * a.T + t) * J + 4 * (q - row * J4));
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int q = q0 + e * CNT;
                if (q < n4) {
                    const int row = q / J4, j = 4 * (q - row * J4), u = row / CKF;
                    const float4 gd = *reinterpret_cast<const float4*>(sgd + u * J + j);
                    float* d = sz + row * J + kperm(j);                       // elements j .. j + 3 sit 4 floats apart
                    d[0] = tanhf(fv[e].x + gd.x); d[4] = tanhf(fv[e].y + gd.y); d[8] = tanhf(fv[e].z + gd.z); d[12] = tanhf(fv[e].w + gd.w);
                }
            }
        }
        __syncthreads();
        {   // VU vocabulary rows of W_j as 16-row tiles (one per wave) against the CU * CKF = 16 joint inputs; partial argmax per wave
            const int ntile = (VU + 15) / 16;                              // <= 8 (cluster_supported)
            const int j = lane & 15, g = lane >> 4;
            float bv = -INFINITY; int bi = 0x7fffffff;
            if (wave < ntile) {
                int nrow[1];
                const int lr = 16 * wave + j;
                const int nn = wg * VU + lr;
                nrow[0] = (lr < VU && nn < V) ? nn : V - 1;
                f32x4 acc[1] = {f32x4{0.f, 0.f, 0.f, 0.f}};
                mfma_rows16<1>(w.wj16, V, J / 16, nrow, sz + j * J + 4 * g, acc);
#pragma unroll
                for (int e = 0; e < 4; ++e) {                              // rows 4 g + e of the tile, column j = (utterance, frame)
                    const int lc = 16 * wave + 4 * g + e, n = wg * VU + lc;
                    if (lc < VU && n < V) {
                        const float v = acc[0][e] + w.bj[n];
                        if (v > bv || (v == bv && n < bi)) { bv = v; bi = n; }
                    }
                }
            }
#pragma unroll
            for (int o = 16; o <= 32; o <<= 1) {                           // the 4 lane groups hold different rows of the same column
                const float ov = __shfl_xor(bv, o); const int oi = __shfl_xor(bi, o);
                if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
            }
            if (lane < 16) { sred[lane * 8 + wave] = bv; sredi[lane * 8 + wave] = bi; }
        }
        __syncthreads();
        if (tid < CU * CKF) {
            float bv = sred[tid * 8]; int bi = sredi[tid * 8];
            for (int q = 1; q < 8; ++q) {
                const float ov = sred[tid * 8 + q]; const int oi = sredi[tid * 8 + q];
                if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
            }
            xstore(xav + wg * CU * CKF + tid, bv); xstorei(xai + wg * CU * CKF + tid, bi);
        }
        ok = cluster_sync<CW>(cnt, epoch, a.status) && ok;
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

template <int CW, int CU, int CKF>
struct ClusterShape {
    static size_t exchange_bytes(int ncl, int H, int J) { return (size_t)ncl * ((size_t)CU * H * 4 + (size_t)CU * J * 4 + (size_t)CW * CU * CKF * 8 + 256) + 256; }
    static size_t lds_bytes(int H, int J) { const int HU = H / CW; return (size_t)(CU * H + CU * J + CU * CKF * J + CU * HU + CU * 4 * HU + CU * CKF * 16) * 4; }
    static bool supported(const EcRnntConfig& c) {
        // 16-blocks of k in groups of CMB; whole 16-row tiles of gate rows; at most 4 gate tiles per wave and one vocabulary tile per wave; the state in LDS
        return c.dim_decoder % (16 * CMB) == 0 && c.dim_joint % (16 * CMB) == 0 && c.dim_decoder % (4 * CW) == 0 && c.dim_joint % CW == 0 &&
               4 * (c.dim_decoder / CW) <= 16 * 8 * 4 && (c.vocab_size + CW - 1) / CW <= 128 && lds_bytes(c.dim_decoder, c.dim_joint) <= 160 * 1024 - 512;
    }
};
using ShapeA = ClusterShape<8, 8, 2>;
using ShapeB = ClusterShape<16, 16, 1>;

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
    int cluster_by_slice = 1;   // cluster decode: workgroup -> XCD mapping (see rnnt_cluster_kernel)
    int cluster_mode = -1;   // -1 auto (cluster decode for batches >= 2 x the cluster's utterances), 0 per-utterance kernel, 1 force cluster
    int cluster_shape = 0;   // 0 <8, 8, 2> (default), 1 <16, 16, 1> where supported (measured: no faster - see the shapes' comment - and a blank costs a round of its own)
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

// W [N][K] row-major -> float4 image [K/16][4][N]: entry (q, g, n) = W[n][16q + g], W[n][16q + 4 + g], W[n][16q + 8 + g], W[n][16q + 12 + g]
std::vector<float> kperm16(const std::vector<float>& W, int N, int K) {
    std::vector<float> o((size_t)N * K);
    for (int q = 0; q < K / 16; ++q)
        for (int g = 0; g < 4; ++g)
            for (int n = 0; n < N; ++n)
                for (int e = 0; e < 4; ++e) o[(((size_t)q * 4 + g) * N + n) * 4 + e] = W[(size_t)n * K + 16 * q + 4 * e + g];
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
    r->dev.whh16 = r->dev.wd16 = r->dev.wj16 = nullptr;
    if (ShapeA::supported(r->cfg) || ShapeB::supported(r->cfg)) {
        const std::vector<float> a16 = kperm16(whh->data, 4 * H, H), b16 = kperm16(wd->data, J, H), c16 = kperm16(wj->data, V, J);
        r->dev.whh16 = (const float4*)upload(r, a16.data(), a16.size() * 4);
        r->dev.wd16 = (const float4*)upload(r, b16.data(), b16.size() * 4);
        r->dev.wj16 = (const float4*)upload(r, c16.data(), c16.size() * 4);
        if (!r->dev.whh16 || !r->dev.wd16 || !r->dev.wj16) return ec_fail("device allocation / upload failed");
    }
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
    const size_t ex = std::max(ShapeA::exchange_bytes((batch + 7) / 8, r->cfg.dim_decoder, r->cfg.dim_joint), ShapeB::exchange_bytes((batch + 15) / 16, r->cfg.dim_decoder, r->cfg.dim_joint));
    return (size_t)batch * t_out * r->cfg.dim_joint * 4 + 256 + ex;
}

int effconf_rnnt_set_option(EcRnnt* r, const char* name, int32_t value) {
    if (!r || !name) return ec_fail("null argument");
    if (!strcmp(name, "cluster_decode")) { r->cluster_mode = value; return 0; }
    if (!strcmp(name, "cluster_by_slice")) { r->cluster_by_slice = value != 0; return 0; }
    if (!strcmp(name, "cluster_shape")) { r->cluster_shape = value; return 0; }
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
    // auto: clusters only while they need at most half of the CUs with the by-cluster mapping (two decodes on two streams must not starve
    // each other's cluster-mates: a spinning workgroup keeps its CU).
    // With the by-slice mapping a cluster is 8 CONSECUTIVE workgroups: whatever part of a launch is resident consists of whole clusters plus
    // at most one split at the dispatch frontier, so decodes sharing the GPU cannot starve each other and a launch may use every CU.
    auto run_cluster = [&](auto shape, auto cw_c, auto cu_c, auto ckf_c) -> int {
        using Shape = decltype(shape);
        constexpr int CW = decltype(cw_c)::value, CU = decltype(cu_c)::value, CKF = decltype(ckf_c)::value;
        const int ncl = (batch + CU - 1) / CU;
        if (ncl * CW > 256) return ec_fail("cluster decode needs every workgroup resident: batch <= 256");
        char* ex = reinterpret_cast<char*>(fe) + (size_t)batch * t_out * J * 4;
        ex = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(ex) + 255) & ~(uintptr_t)255);
        ClusterArgs a{};
        a.w = r->dev; a.fe = fe; a.lens = out_len; a.T = t_out; a.B = batch; a.tokens = tokens; a.counts = token_len; a.max_tok = max_tokens;
        a.ncl = ncl; a.by_slice = r->cluster_by_slice;
        a.cnt = reinterpret_cast<unsigned*>(ex); a.status = reinterpret_cast<int*>(ex + (size_t)ncl * 256);
        char* p = ex + (size_t)ncl * 256 + 256;
        a.xh = reinterpret_cast<float*>(p); p += (size_t)ncl * CU * H * 4;
        a.xgd = reinterpret_cast<float*>(p); p += (size_t)ncl * CU * J * 4;
        a.xav = reinterpret_cast<float*>(p); p += (size_t)ncl * CW * CU * CKF * 4;
        a.xai = reinterpret_cast<int*>(p);
        if (hipMemsetAsync(ex, 0, (size_t)ncl * 256 + 256, s) != hipSuccess) return ec_fail("memset failed");
        const size_t lds = Shape::lds_bytes(H, J);
        static LdsAttr attr;
        ensure_dynamic_lds(reinterpret_cast<const void*>(&rnnt_cluster_kernel<CW, CU, CKF>), (int)lds, attr);
        hipLaunchKernelGGL((rnnt_cluster_kernel<CW, CU, CKF>), dim3(ncl * CW), dim3(CNT), lds, s, a);
        return hipGetLastError() == hipSuccess ? 0 : ec_fail("rnnt cluster launch failed");
    };
    // a cluster shape runs when the batch fills two of its clusters and all its workgroups are resident (by-cluster mapping: at most half of the CUs)
    auto fits = [&](int cw, int cu) { return batch >= 2 * cu && ((batch + cu - 1) / cu) * cw <= (r->cluster_by_slice ? 256 : 128); };
    if (r->cluster_mode != 0) {
        const bool okB = ShapeB::supported(r->cfg) && r->cluster_shape == 1 && (r->cluster_mode == 1 ? ((batch + 15) / 16) * 16 <= 256 : fits(16, 16));
        const bool okA = ShapeA::supported(r->cfg) && r->cluster_shape != 1 && (r->cluster_mode == 1 || fits(8, 8));
        if (okB) return run_cluster(ShapeB{}, std::integral_constant<int, 16>{}, std::integral_constant<int, 16>{}, std::integral_constant<int, 1>{});
        if (okA) return run_cluster(ShapeA{}, std::integral_constant<int, 8>{}, std::integral_constant<int, 8>{}, std::integral_constant<int, 2>{});
    }
    const size_t lds = (size_t)(2 * H + 4 * H + J + KF * J + 2 * KF * (NT / 64)) * 4;
    hipLaunchKernelGGL(rnnt_greedy_kernel, dim3(batch), dim3(NT), lds, s, r->dev, fe, out_len, t_out, tokens, token_len, max_tokens);
    return hipGetLastError() == hipSuccess ? 0 : ec_fail("rnnt_greedy launch failed");
}

}  // extern "C"
