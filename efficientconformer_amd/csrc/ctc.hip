// Length bookkeeping and the CTC head (gfx950).
//
//  * lengths_kernel      — x_len // hop + 1 (modules.py:100), (len-1)//2+1 per subsampling layer
//                           (modules.py:243) and per strided block (encoders.py:139), all stages at once.
//  * ctc_argmax_kernel   — logits = x fc^T + b in fp32 on the vector ALUs (model_ctc.py:49, 66, 96) and the
//                           per-frame argmax (model_ctc.py:99; log_softmax is monotone) fused; the head is
//                           0.2 % of the FLOPs, so it stays exact fp32 to keep greedy labels stable.
//  * ctc_collapse_kernel — drop blanks (id 0), collapse repeats not separated by a blank, stop at len[b]
//                           (the reference's Python loop with .item() per token, model_ctc.py:105-133).
#include "kernels.h"

namespace {

__global__ void lengths_kernel(const int64_t* __restrict__ x_len, int B, int from_audio, int hop, int sub_layers,
                               const int* __restrict__ block_stride, int n_blocks, int* stage_lens, int64_t* out_len) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    // floor division as torch's `//` (modules.py:100, 243; encoders.py:139): an empty row (mel length 0) stays 0, it does not become 1
    auto fdiv = [](long long a, long long d) { long long q = a / d; return (a % d != 0 && ((a < 0) != (d < 0))) ? q - 1 : q; };
    long long l = x_len[b];
    if (from_audio) l = fdiv(l, hop) + 1;
    for (int i = 0; i < sub_layers; ++i) l = fdiv(l - 1, 2) + 1;
    stage_lens[b] = (int)l;                                  // lengths seen by block 0
    for (int k = 0; k < n_blocks; ++k) {
        const int s = block_stride[k];
        if (s > 1) l = fdiv(l - 1, s) + 1;
        stage_lens[(size_t)(k + 1) * B + b] = (int)l;        // lengths after block k (= seen by block k+1)
    }
    if (out_len) out_len[b] = l;
}


// Ragged batches (see kernels.h): lengths of every stage, then the prefix sums the ragged kernels index with.  One workgroup of sixteen
// waves (four until round 4: the 48 scans of a 15-block encoder ran twelve deep per wave, 20 us at the head of every range's stream; three
// deep now): thread b computes utterance b's lengths; then every (position, array) pair is one wave-parallel exclusive scan (64 utterances per
// step, __shfl_up ladder) - a serial scan per thread cost ~0.1 ms of dependent global accesses at the head of every forward.
__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(v, o); if (lane >= o) v += t; }
    return v;
}

__global__ __launch_bounds__(1024) void lengths_ragged_kernel(const int64_t* __restrict__ x_len, int B, int from_audio, int hop, int sub_layers,
                                                             const int* __restrict__ block_stride, const int* __restrict__ group,
                                                             const int* __restrict__ heads, int n_blocks, int* stage_lens, int* mel_len,
                                                             int* row_off, int* wg_off, int* tile_off, int64_t* out_len) {
    auto fdiv = [](long long a, long long d) { long long q = a / d; return (a % d != 0 && ((a < 0) != (d < 0))) ? q - 1 : q; };
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        long long l = x_len[b];
        if (from_audio) l = fdiv(l, hop) + 1;
        mel_len[b] = (int)l;
        for (int i = 0; i < sub_layers; ++i) l = fdiv(l - 1, 2) + 1;
        stage_lens[b] = (int)l;
        for (int k = 0; k < n_blocks; ++k) {
            const int s = block_stride[k];
            if (s > 1) l = fdiv(l - 1, s) + 1;
            stage_lens[(size_t)(k + 1) * B + b] = (int)l;
        }
        if (out_len) out_len[b] = l;
    }
    __threadfence_block();
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
    const int q8 = B >> 3, r8 = B & 7;
    // job = 3 * k + a: a = 0 row offsets of position k (k <= n_blocks), 1 attention workgroups of block k, 2 depthwise-conv tiles of block k
    for (int job = wave; job < 3 * (n_blocks + 1); job += nwave) {
        const int k = job / 3, a = job - 3 * k;
        if (k == n_blocks && a != 0) continue;
        const int G = k < n_blocks ? group[k] : 1;
        const int* len = stage_lens + (size_t)k * B;
        int* dst = (a == 0 ? row_off + (size_t)k * (B + 1) : (a == 1 ? wg_off : tile_off) + (size_t)k * (B + 1));
        int carry = 0;
        for (int base = 0; base < B; base += 64) {
            const int u = base + lane;
            int v = 0;
            if (u < B) {
                if (a == 0) v = (len[u] + G - 1) / G * G;
                else if (a == 1) {                               // list position u <-> utterance x + 8 j (the attention kernels' order)
                    int x, j;
                    if (u < r8 * (q8 + 1)) { x = u / (q8 + 1); j = u - x * (q8 + 1); }
                    else { const int u2 = u - r8 * (q8 + 1); x = u2 / q8; j = u2 - x * q8; x += r8; }
                    const int tg = (len[x + 8 * j] + G - 1) / G;
                    v = heads[k] * ((tg + 63) / 64);
                } else {                                         // block k's output rows are padded for block k + 1's attention
                    const int Gn = k + 1 < n_blocks ? group[k + 1] : 1;
                    const int lo = stage_lens[(size_t)(k + 1) * B + u];
                    v = ((lo + Gn - 1) / Gn * Gn + 127) / 128;
                }
            }
            const int inc = wave_incl_scan(v, lane);
            if (u < B) dst[u] = carry + inc - v;
            carry += __shfl(inc, 63);
        }
        if (lane == 0) dst[B] = carry;
    }
}

// One workgroup = CTC_ROWS frames x the whole vocabulary (thread = vocabulary column).  Every workgroup reads all of fc^T from L2, so
// the rows per workgroup set the L2 traffic (8 rows: 3200 workgroups x 245 KB = 0.8 GB per launch, the whole 150 us of the first
// version); the frame tile sits in LDS and is read as float4 broadcasts along k (one LDS read per 4 FMAs).
template <int CTC_ROWS>
__global__ __launch_bounds__(256) void ctc_argmax_kernel(const float* __restrict__ x, int M, int D,
                                                         const float* __restrict__ Wt, const float* __restrict__ bias,
                                                         int V, int* __restrict__ preds, float* __restrict__ logits) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sx = reinterpret_cast<float*>(smem);                         // [CTC_ROWS][D]   (D % 4 == 0)
    float* sbest = sx + CTC_ROWS * D;                                   // [CTC_ROWS][4]
    int* sidx = reinterpret_cast<int*>(sbest + CTC_ROWS * 4);           // [CTC_ROWS][4]
    const int m0 = blockIdx.x * CTC_ROWS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid * 4; i < CTC_ROWS * D; i += 256 * 4) {
        const int r = i / D;
        const int mr = m0 + r < M ? m0 + r : M - 1;
        *reinterpret_cast<float4*>(sx + i) = *reinterpret_cast<const float4*>(x + (size_t)mr * D + (i - r * D));
    }
    __syncthreads();
    float best[CTC_ROWS];
    int bidx[CTC_ROWS];
#pragma unroll
    for (int r = 0; r < CTC_ROWS; ++r) { best[r] = -INFINITY; bidx[r] = 0x7fffffff; }
    for (int v0 = 0; v0 < V; v0 += 256) {
        const int v = v0 + tid, vc = v < V ? v : V - 1;
        float acc[CTC_ROWS];
#pragma unroll
        for (int r = 0; r < CTC_ROWS; ++r) acc[r] = 0.f;
        for (int k = 0; k < D; k += 4) {
            const float w0 = Wt[(size_t)k * V + vc], w1 = Wt[(size_t)(k + 1) * V + vc], w2 = Wt[(size_t)(k + 2) * V + vc], w3 = Wt[(size_t)(k + 3) * V + vc];
#pragma unroll
            for (int r = 0; r < CTC_ROWS; ++r) {
                const float4 xv = *reinterpret_cast<const float4*>(sx + r * D + k);
                acc[r] = fmaf(xv.w, w3, fmaf(xv.z, w2, fmaf(xv.y, w1, fmaf(xv.x, w0, acc[r]))));      // k ascending, as the first version
            }
        }
        const float bz = bias[vc];
#pragma unroll
        for (int r = 0; r < CTC_ROWS; ++r) {
            const float val = acc[r] + bz;
            if (logits && v < V && m0 + r < M) logits[(size_t)(m0 + r) * V + v] = val;
            if (v < V && val > best[r]) { best[r] = val; bidx[r] = v; }   // strictly greater: first max wins
        }
    }
    // reduce (max value, then lowest index) across the wave, then across the 4 waves
#pragma unroll
    for (int r = 0; r < CTC_ROWS; ++r) {
        float bv = best[r]; int bi = bidx[r];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o); const int oi = __shfl_xor(bi, o);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (lane == 0) { sbest[r * 4 + wave] = bv; sidx[r * 4 + wave] = bi; }
    }
    __syncthreads();
    if (tid < CTC_ROWS && m0 + tid < M) {
        float bv = sbest[tid * 4]; int bi = sidx[tid * 4];
        for (int w = 1; w < 4; ++w) {
            const float ov = sbest[tid * 4 + w]; const int oi = sidx[tid * 4 + w];
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        preds[m0 + tid] = bi;
    }
}

// fc + argmax on the fp32 MFMA (v_mfma_f32_32x32x2_f32: a k-ordered fmaf chain, so logits are bit-identical to ctc_argmax_kernel's).
// One workgroup = ROWS frames x 256 vocabulary columns per pass (wave = 64 columns: RT x 2 accumulator tiles); the frame tile sits in
// LDS (row pitch D + 1 floats: conflict-free column reads), fc^T streams from L2 as MFMA B fragments (two rows of 32 consecutive
// floats per load).  The logits tile goes back through LDS for the per-row argmax (first maximum wins) and the optional row-major
// logits output.  6.3 GFLOP per 256 x 201 frames: 222 us on the VALU kernel above, fp32-MFMA bound here.
template <int ROWS>
__global__ __launch_bounds__(256) void ctc_argmax_mfma_kernel(const float* __restrict__ x, int M, int D, const float* __restrict__ Wt,
                                                              const float* __restrict__ bias, int V, int* __restrict__ preds,
                                                              float* __restrict__ logits, int tile_off) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int RT = ROWS / 32, TPR = 256 / ROWS, CPT = 256 / TPR, LDT = 257;
    float* sx = reinterpret_cast<float*>(smem);                         // [ROWS][D + 1]
    float* st = sx + tile_off;                                          // [ROWS][LDT] logits tile (aliases sx when V <= 256: one pass)
    const int LDX = D + 1;
    const int m0 = blockIdx.x * ROWS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid * 4; i < ROWS * D; i += 1024) {
        const int r = i / D, k = i - r * D;
        const int mr = m0 + r < M ? m0 + r : M - 1;
        const float4 v = *reinterpret_cast<const float4*>(x + (size_t)mr * D + k);
        float* d = sx + r * LDX + k;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    __syncthreads();
    const int prow = tid / TPR, part = tid - prow * TPR;                // argmax: TPR adjacent threads per row, CPT columns each
    float best = -INFINITY;
    int bidx = 0x7fffffff;
    const int kh = lane >> 5, lc = lane & 31;
    for (int v0 = 0; v0 < V; v0 += 256) {
        f32x16 acc[RT][2];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[rt][ct][r] = 0.f;
        const int c0 = v0 + wave * 64 + lc, c1 = c0 + 32;
        const float* w0 = Wt + (c0 < V ? c0 : V - 1) + (size_t)kh * V;
        const float* w1 = Wt + (c1 < V ? c1 : V - 1) + (size_t)kh * V;
        const float* xa = sx + lc * LDX + kh;
#pragma unroll 4
        for (int k = 0; k < D; k += 2) {
            const float b0 = w0[(size_t)k * V], b1 = w1[(size_t)k * V];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const float a = xa[rt * 32 * LDX + k];
                acc[rt][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc[rt][0], 0, 0, 0);
                acc[rt][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc[rt][1], 0, 0, 0);
            }
        }
        if (tile_off == 0 || v0 > 0) __syncthreads();                  // aliasing tile: every wave is done with the frame tile; later passes: the previous tile was consumed
        const float bz0 = bias[c0 < V ? c0 : V - 1], bz1 = bias[c1 < V ? c1 : V - 1];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                st[row * LDT + wave * 64 + lc] = acc[rt][0][r] + bz0;
                st[row * LDT + wave * 64 + 32 + lc] = acc[rt][1][r] + bz1;
            }
        __syncthreads();
        if (logits) {
            for (int i = tid; i < ROWS * 256; i += 256) {
                const int r = i >> 8, cc = i & 255;
                if (m0 + r < M && v0 + cc < V) logits[(size_t)(m0 + r) * V + v0 + cc] = st[r * LDT + cc];
            }
        }
#pragma unroll 8
        for (int j = 0; j < CPT; ++j) {
            const int cc = part * CPT + j;
            const float val = st[prow * LDT + cc];
            if (v0 + cc < V && val > best) { best = val; bidx = v0 + cc; }   // ascending columns, strictly greater: first maximum wins
        }
    }
#pragma unroll
    for (int o = 1; o < TPR; o <<= 1) {
        const float ov = __shfl_xor(best, o); const int oi = __shfl_xor(bidx, o);
        if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
    }
    if (part == 0 && m0 + prow < M) preds[m0 + prow] = bidx;
}

// fc + argmax with SPLIT-bf16 operands on the bf16 matrix pipe (the default of the bf16 path): x = x_hi + x_lo, W = W_hi + W_lo (two bf16 each:
// 16 significant bits), logits = x_hi W_hi + x_hi W_lo + x_lo W_hi, fp32 accumulation - relative error ~2^-16 per product, four orders below the
// 3e-2 the bf16 encoder output already carries, at 3 bf16 MFMAs (96 cycles) per 16 k where the fp32 pipe needs 8 fp32 MFMAs (512 cycles).
// The fp32 head was 2 % of the Small step and, with the north_star's all-gather BEFORE the head, runs on N x the frames on every rank.
// Same tiling as the fp32 kernel (ROWS frames x 256 columns per pass, wave = 64 columns); the frame tile is split once into two bf16 LDS
// images (row pitch 2 Kp + 16 bytes: the 16-byte fragment reads of 16 consecutive rows fall on disjoint banks), the weight halves are
// packed at finalize as MFMA B fragments ([k / 16][column][k-half][8] bf16: a wave's load is 1 KiB contiguous).
// XBF = true (round 4): the frame rows arrive as bf16 (the gathered encoder outputs of the multi-rank path, bf16 wire): x_hi is the input itself,
// x_lo = 0, so the x_lo W_hi term and the lo image of the frame tile vanish (2 MFMAs per 16 k, no conversion pass over the gathered chunk, half
// the input bytes) - and the logits equal, bit for bit, those of the fp32-input kernel fed with the same values widened to fp32.
template <int ROWS, bool XBF = false>
__global__ __launch_bounds__(256) void ctc_argmax_bf16x3_kernel(const float* __restrict__ x, int M, int D, int Kp, const bf16_t* __restrict__ Whi,
                                                                const bf16_t* __restrict__ Wlo, const float* __restrict__ bias, int V, int Vp,
                                                                int* __restrict__ preds, float* __restrict__ logits) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int RT = ROWS / 32, TPR = 256 / ROWS, CPT = 256 / TPR, LDT = 257;
    const int pitch = Kp * 2 + 16;                                       // bytes per frame row of one half
    char* xh = smem;                                                     // [ROWS][pitch] hi
    char* xl = smem + ROWS * pitch;                                      // [ROWS][pitch] lo
    float* st = reinterpret_cast<float*>(smem + 2 * ROWS * pitch);      // [ROWS][LDT] logits tile
    const int m0 = blockIdx.x * ROWS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid * 4; i < ROWS * Kp; i += 1024) {
        const int r = i / Kp, k = i - r * Kp;
        const int mr = m0 + r < M ? m0 + r : M - 1;
        if constexpr (XBF) {
            uint2 hv = make_uint2(0u, 0u);
            if (k < D) hv = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(x) + (size_t)mr * D + k);
            *reinterpret_cast<uint2*>(xh + r * pitch + k * 2) = hv;
            continue;
        }
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < D) v = *reinterpret_cast<const float4*>(x + (size_t)mr * D + k);        // D % 4 == 0: a quad is valid or pad as a whole
        const uint16_t h0 = f2bf(v.x), h1 = f2bf(v.y), h2 = f2bf(v.z), h3 = f2bf(v.w);
        *reinterpret_cast<uint2*>(xh + r * pitch + k * 2) = make_uint2(h0 | ((uint32_t)h1 << 16), h2 | ((uint32_t)h3 << 16));
        *reinterpret_cast<uint2*>(xl + r * pitch + k * 2) = make_uint2(pack_bf2(v.x - bf2f(h0), v.y - bf2f(h1)), pack_bf2(v.z - bf2f(h2), v.w - bf2f(h3)));
    }
    __syncthreads();
    const int prow = tid / TPR, part = tid - prow * TPR;
    float best = -INFINITY;
    int bidx = 0x7fffffff;
    const int kh = lane >> 5, lc = lane & 31;
    const int nks = Kp / 16;
    for (int v0 = 0; v0 < V; v0 += 256) {
        f32x16 acc[RT][2];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[rt][ct][r] = 0.f;
        const int c0 = v0 + wave * 64 + lc, c1 = c0 + 32;               // < Vp: the packed images are padded to whole 256-column passes
        const size_t o0 = ((size_t)c0 * 2 + kh) * 8, o1 = ((size_t)c1 * 2 + kh) * 8, ks = (size_t)Vp * 16;
        bf16x8 bh0 = *reinterpret_cast<const bf16x8*>(Whi + o0), bl0 = *reinterpret_cast<const bf16x8*>(Wlo + o0);
        bf16x8 bh1 = *reinterpret_cast<const bf16x8*>(Whi + o1), bl1 = *reinterpret_cast<const bf16x8*>(Wlo + o1);
        for (int s = 0; s < nks; ++s) {
            const bf16x8 ch0 = bh0, cl0 = bl0, ch1 = bh1, cl1 = bl1;
            const size_t nx = (size_t)(s + 1 < nks ? s + 1 : s) * ks;        // next k-step's fragments under this step's MFMAs
            bh0 = *reinterpret_cast<const bf16x8*>(Whi + nx + o0); bl0 = *reinterpret_cast<const bf16x8*>(Wlo + nx + o0);
            bh1 = *reinterpret_cast<const bf16x8*>(Whi + nx + o1); bl1 = *reinterpret_cast<const bf16x8*>(Wlo + nx + o1);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const int off = (rt * 32 + lc) * pitch + (16 * s + 8 * kh) * 2;
                const bf16x8 ah = *reinterpret_cast<const bf16x8*>(xh + off);
                acc[rt][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, ch0, acc[rt][0], 0, 0, 0);
                acc[rt][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, ch1, acc[rt][1], 0, 0, 0);
                acc[rt][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, cl0, acc[rt][0], 0, 0, 0);
                acc[rt][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, cl1, acc[rt][1], 0, 0, 0);
                if constexpr (!XBF) {                      // bf16 rows: x_lo = 0, the term contributes exact zeros
                    const bf16x8 al = *reinterpret_cast<const bf16x8*>(xl + off);
                    acc[rt][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, ch0, acc[rt][0], 0, 0, 0);
                    acc[rt][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, ch1, acc[rt][1], 0, 0, 0);
                }
            }
        }
        if (v0 > 0) __syncthreads();                                    // later passes: the previous logits tile was consumed
        const float bz0 = bias[c0 < V ? c0 : V - 1], bz1 = bias[c1 < V ? c1 : V - 1];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                st[row * LDT + wave * 64 + lc] = acc[rt][0][r] + bz0;
                st[row * LDT + wave * 64 + 32 + lc] = acc[rt][1][r] + bz1;
            }
        __syncthreads();
        if (logits) {
            for (int i = tid; i < ROWS * 256; i += 256) {
                const int r = i >> 8, cc = i & 255;
                if (m0 + r < M && v0 + cc < V) logits[(size_t)(m0 + r) * V + v0 + cc] = st[r * LDT + cc];
            }
        }
#pragma unroll 8
        for (int j = 0; j < CPT; ++j) {
            const int cc = part * CPT + j;
            const float val = st[prow * LDT + cc];
            if (v0 + cc < V && val > best) { best = val; bidx = v0 + cc; }
        }
    }
#pragma unroll
    for (int o = 1; o < TPR; o <<= 1) {
        const float ov = __shfl_xor(best, o); const int oi = __shfl_xor(bidx, o);
        if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
    }
    if (part == 0 && m0 + prow < M) preds[m0 + prow] = bidx;
}

// one wave per utterance: 64 frames per step, keep = (c != 0 && c != previous frame's c), output position = running count +
// prefix popcount of the keep ballot
__global__ __launch_bounds__(64) void ctc_collapse_kernel(const int* __restrict__ preds, const int64_t* __restrict__ lens, int B, int T,
                                                          int* __restrict__ labels, int* __restrict__ label_len) {
    const int b = blockIdx.x, lane = threadIdx.x;
    long long len = lens[b];
    if (len > T) len = T;
    if (len < 0) len = 0;
    int n = 0, carry = 0;                                      // carry = prediction of the frame before this step (0 at the start)
    for (int t0 = 0; t0 < (int)len; t0 += 64) {
        const int t = t0 + lane;
        const int c = t < (int)len ? preds[(size_t)b * T + t] : 0;
        int prev = __shfl_up(c, 1);
        if (lane == 0) prev = carry;
        const bool keep = t < (int)len && c != 0 && c != prev;
        const unsigned long long m = __ballot(keep);
        if (keep) labels[(size_t)b * T + n + __popcll(m & ((1ull << lane) - 1))] = c;
        n += __popcll(m);
        carry = __shfl(c, 63);
    }
    if (lane == 0) label_len[b] = n;
    for (int t = n + lane; t < T; t += 64) labels[(size_t)b * T + t] = 0;
}

}  // namespace

int launch_lengths_ragged(const int64_t* x_len, int B, int from_audio, int hop, int sub_layers, const int* block_stride, const int* group,
                          const int* heads, int n_blocks, int* stage_lens, int* mel_len, int* row_off, int* wg_off, int* tile_off,
                          int64_t* out_len, hipStream_t s) {
    if (B <= 0) return 0;
    if (B > 4096) return -2;
    hipLaunchKernelGGL(lengths_ragged_kernel, dim3(1), dim3(1024), 0, s, x_len, B, from_audio, hop, sub_layers, block_stride, group, heads,
                       n_blocks, stage_lens, mel_len, row_off, wg_off, tile_off, out_len);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_lengths(const int64_t* x_len, int B, int from_audio, int hop, int sub_layers, const int* block_stride,
                   int n_blocks, int* stage_lens, int64_t* out_len, hipStream_t s) {
    if (B <= 0) return 0;
    hipLaunchKernelGGL(lengths_kernel, dim3((B + 63) / 64), dim3(64), 0, s, x_len, B, from_audio, hop, sub_layers,
                       block_stride, n_blocks, stage_lens, out_len);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

template <int ROWS>
int launch_ctc_mfma(const float* x, int M, int D, const float* Wt, const float* bias, int V, int* preds, float* logits, hipStream_t s) {
    const size_t frame = (size_t)ROWS * (D + 1) * 4, tile = (size_t)ROWS * 257 * 4;
    const bool alias = V <= 256;                                       // one pass: the logits tile may reuse the frame tile
    const size_t lds = alias ? (frame > tile ? frame : tile) : frame + tile;
    if (lds > 160 * 1024) return -2;
    static LdsAttr attr;
    ensure_dynamic_lds(reinterpret_cast<const void*>(&ctc_argmax_mfma_kernel<ROWS>), (int)lds, attr);
    hipLaunchKernelGGL((ctc_argmax_mfma_kernel<ROWS>), dim3((M + ROWS - 1) / ROWS), dim3(256), lds, s, x, M, D, Wt, bias, V, preds, logits,
                       alias ? 0 : (int)(frame / 4));
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_ctc_split(const float* x, int M, int D, const bf16_t* Whi, const bf16_t* Wlo, const float* bias, int V, int* preds, float* logits,
                     hipStream_t s, int x_is_bf16) {
    if (M <= 0) return 0;
    if (D % 4 || !Whi || !Wlo) return -2;
    if (x_is_bf16) {
        const int Kp = (D + 15) / 16 * 16, Vp = (V + 255) / 256 * 256;
        auto lds_b = [&](int rows) { return (size_t)2 * rows * (Kp * 2 + 16) + (size_t)rows * 257 * 4; };      // same layout (the lo image stays unused)
        static LdsAttr b64, b32;
        if (lds_b(64) <= 160 * 1024) {
            ensure_dynamic_lds(reinterpret_cast<const void*>(&ctc_argmax_bf16x3_kernel<64, true>), (int)lds_b(64), b64);
            hipLaunchKernelGGL((ctc_argmax_bf16x3_kernel<64, true>), dim3((M + 63) / 64), dim3(256), lds_b(64), s, x, M, D, Kp, Whi, Wlo, bias, V, Vp, preds, logits);
        } else if (lds_b(32) <= 160 * 1024) {
            ensure_dynamic_lds(reinterpret_cast<const void*>(&ctc_argmax_bf16x3_kernel<32, true>), (int)lds_b(32), b32);
            hipLaunchKernelGGL((ctc_argmax_bf16x3_kernel<32, true>), dim3((M + 31) / 32), dim3(256), lds_b(32), s, x, M, D, Kp, Whi, Wlo, bias, V, Vp, preds, logits);
        } else {
            return -2;
        }
        return hipGetLastError() == hipSuccess ? 0 : -1;
    }
    const int Kp = (D + 15) / 16 * 16, Vp = (V + 255) / 256 * 256;     // = finalize's packing (encoder.hip): a pass covers 256 columns
    auto lds_for = [&](int rows) { return (size_t)2 * rows * (Kp * 2 + 16) + (size_t)rows * 257 * 4; };
    static LdsAttr attr64, attr32;
    if (lds_for(64) <= 160 * 1024) {
        ensure_dynamic_lds(reinterpret_cast<const void*>(&ctc_argmax_bf16x3_kernel<64>), (int)lds_for(64), attr64);
        hipLaunchKernelGGL((ctc_argmax_bf16x3_kernel<64>), dim3((M + 63) / 64), dim3(256), lds_for(64), s, x, M, D, Kp, Whi, Wlo, bias, V, Vp, preds, logits);
    } else if (lds_for(32) <= 160 * 1024) {          // wide last stages (D = 720: Large)
        ensure_dynamic_lds(reinterpret_cast<const void*>(&ctc_argmax_bf16x3_kernel<32>), (int)lds_for(32), attr32);
        hipLaunchKernelGGL((ctc_argmax_bf16x3_kernel<32>), dim3((M + 31) / 32), dim3(256), lds_for(32), s, x, M, D, Kp, Whi, Wlo, bias, V, Vp, preds, logits);
    } else {
        return -2;
    }
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_ctc_argmax(const float* x, int M, int D, const float* Wt, const float* bias, int V,
                      int* preds, float* logits_or_null, hipStream_t s, int use_mfma) {
    if (M <= 0) return 0;
    if (D % 4) return -2;
    if (use_mfma && D <= 256) return launch_ctc_mfma<64>(x, M, D, Wt, bias, V, preds, logits_or_null, s);
    if (use_mfma && D <= 1024 && V <= 256) return launch_ctc_mfma<32>(x, M, D, Wt, bias, V, preds, logits_or_null, s);
    // rows per workgroup: as many as keep the frame tile within the default 64 KB of dynamic LDS
    const int rows = D <= 384 ? 32 : (D <= 768 ? 16 : 8);
    const size_t lds = (size_t)rows * D * sizeof(float) + rows * 4 * (sizeof(float) + sizeof(int));
    if (lds > 64 * 1024) return -2;
    const dim3 grid((M + rows - 1) / rows);
    if (rows == 32) hipLaunchKernelGGL(ctc_argmax_kernel<32>, grid, dim3(256), lds, s, x, M, D, Wt, bias, V, preds, logits_or_null);
    else if (rows == 16) hipLaunchKernelGGL(ctc_argmax_kernel<16>, grid, dim3(256), lds, s, x, M, D, Wt, bias, V, preds, logits_or_null);
    else hipLaunchKernelGGL(ctc_argmax_kernel<8>, grid, dim3(256), lds, s, x, M, D, Wt, bias, V, preds, logits_or_null);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_ctc_collapse(const int* preds, const int64_t* lens, int B, int T, int* labels, int* label_len, hipStream_t s) {
    if (B <= 0) return 0;
    hipLaunchKernelGGL(ctc_collapse_kernel, dim3(B), dim3(64), 0, s, preds, lens, B, T, labels, label_len);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
