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
    long long l = x_len[b];
    if (from_audio) l = l / hop + 1;
    for (int i = 0; i < sub_layers; ++i) l = (l - 1) / 2 + 1;
    stage_lens[b] = (int)l;                                  // lengths seen by block 0
    for (int k = 0; k < n_blocks; ++k) {
        const int s = block_stride[k];
        if (s > 1) l = (l - 1) / s + 1;
        stage_lens[(size_t)(k + 1) * B + b] = (int)l;        // lengths after block k (= seen by block k+1)
    }
    if (out_len) out_len[b] = l;
}

constexpr int CTC_ROWS = 8;

__global__ __launch_bounds__(256) void ctc_argmax_kernel(const float* __restrict__ x, int M, int D,
                                                         const float* __restrict__ Wt, const float* __restrict__ bias,
                                                         int V, int* __restrict__ preds, float* __restrict__ logits) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sx = reinterpret_cast<float*>(smem);                         // [CTC_ROWS][D]
    float* sbest = sx + CTC_ROWS * D;                                   // [CTC_ROWS][4]
    int* sidx = reinterpret_cast<int*>(sbest + CTC_ROWS * 4);           // [CTC_ROWS][4]
    const int m0 = blockIdx.x * CTC_ROWS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < CTC_ROWS * D; i += 256) {
        const int r = i / D, c = i - r * D;
        sx[i] = (m0 + r < M) ? x[(size_t)(m0 + r) * D + c] : 0.f;
    }
    __syncthreads();
    float best[CTC_ROWS];
    int bidx[CTC_ROWS];
#pragma unroll
    for (int r = 0; r < CTC_ROWS; ++r) { best[r] = -INFINITY; bidx[r] = 0x7fffffff; }
    for (int v = tid; v < V; v += 256) {
        float acc[CTC_ROWS];
#pragma unroll
        for (int r = 0; r < CTC_ROWS; ++r) acc[r] = 0.f;
        for (int k = 0; k < D; ++k) {
            const float w = Wt[(size_t)k * V + v];
#pragma unroll
            for (int r = 0; r < CTC_ROWS; ++r) acc[r] = fmaf(sx[r * D + k], w, acc[r]);
        }
        const float bz = bias[v];
#pragma unroll
        for (int r = 0; r < CTC_ROWS; ++r) {
            const float val = acc[r] + bz;
            if (logits && m0 + r < M) logits[(size_t)(m0 + r) * V + v] = val;
            if (val > best[r]) { best[r] = val; bidx[r] = v; }          // strictly greater: first max wins
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

__global__ void ctc_collapse_kernel(const int* __restrict__ preds, const int64_t* __restrict__ lens, int B, int T,
                                    int* __restrict__ labels, int* __restrict__ label_len) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    int n = 0, prev = 0;
    long long len = lens[b];
    if (len > T) len = T;
    for (int t = 0; t < len; ++t) {
        const int c = preds[(size_t)b * T + t];
        if (c != 0 && c != prev) labels[(size_t)b * T + n++] = c;
        prev = c;
    }
    label_len[b] = n;
    for (int t = n; t < T; ++t) labels[(size_t)b * T + t] = 0;
}

}  // namespace

int launch_lengths(const int64_t* x_len, int B, int from_audio, int hop, int sub_layers, const int* block_stride,
                   int n_blocks, int* stage_lens, int64_t* out_len, hipStream_t s) {
    if (B <= 0) return 0;
    hipLaunchKernelGGL(lengths_kernel, dim3((B + 63) / 64), dim3(64), 0, s, x_len, B, from_audio, hop, sub_layers,
                       block_stride, n_blocks, stage_lens, out_len);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_ctc_argmax(const float* x, int M, int D, const float* Wt, const float* bias, int V,
                      int* preds, float* logits_or_null, hipStream_t s) {
    if (M <= 0) return 0;
    const size_t lds = (size_t)CTC_ROWS * D * sizeof(float) + CTC_ROWS * 4 * (sizeof(float) + sizeof(int));
    hipLaunchKernelGGL(ctc_argmax_kernel, dim3((M + CTC_ROWS - 1) / CTC_ROWS), dim3(256), lds, s, x, M, D, Wt, bias, V,
                       preds, logits_or_null);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_ctc_collapse(const int* preds, const int64_t* lens, int B, int T, int* labels, int* label_len, hipStream_t s) {
    if (B <= 0) return 0;
    hipLaunchKernelGGL(ctc_collapse_kernel, dim3((B + 63) / 64), dim3(64), 0, s, preds, lens, B, T, labels, label_len);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
