// Log-mel frontend (gfx950): framing with centre reflect padding, Hann(win) centred in n_fft,
// 512-point FFT in LDS, power spectrum, sparse HTK triangular filterbank, log(x + 1e-9).
//
// Restates what the reference obtains from torchaudio (models/modules.py:81-82, 90-96:
// Spectrogram(n_fft, win_length, hop_length, power=2) -> MelScale(n_mels, sr, 0, 8000) -> log).
// The arithmetic of that dependency is not in the reference repository ("parity unpinned", see
// oracle/ref_encoder.py); this kernel is validated against the oracle's torch.stft restatement.
//
// One workgroup = 16 consecutive frames of one utterance (4 waves x 4 frames), so that the
// (B, n_mels, Tm) output is written as 64-byte row segments; HBM-bound (640 B in, 320 B out per frame).
#include "kernels.h"

namespace {

constexpr int NFFT = 512;
constexpr int LOGN = 9;
constexpr int FRAMES_PER_WAVE = 4;
constexpr int FRAMES_PER_BLOCK = 16;
constexpr int MAX_MELS = 128;

__device__ __forceinline__ int bitrev9(int n) { return (int)(__brev((unsigned)n) >> (32 - LOGN)); }

__global__ __launch_bounds__(256) void mel_kernel(const float* __restrict__ audio, int L, MelTables tb, int hop,
                                                  int n_mels, int Tm, int normalize, float mean, float inv_std,
                                                  float* __restrict__ mel) {
    __shared__ float2 sx[4][NFFT];                               // per-wave FFT buffer (power spectrum aliases .x)
    __shared__ float sout[MAX_MELS][FRAMES_PER_BLOCK + 1];
    const int tiles = (Tm + FRAMES_PER_BLOCK - 1) / FRAMES_PER_BLOCK;
    const int b = blockIdx.x / tiles, t0 = (blockIdx.x % tiles) * FRAMES_PER_BLOCK;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float2* x = sx[wave];
    const float* a = audio + (size_t)b * L;

    for (int fi = 0; fi < FRAMES_PER_WAVE; ++fi) {
        const int fl = wave * FRAMES_PER_WAVE + fi;
        const int t = t0 + fl;
        const bool live = t < Tm;
        // ---- frame gather (reflect at both ends of the *padded* waveform, torch.stft center=True) + window
#pragma unroll
        for (int i = 0; i < NFFT / 64; ++i) {
            const int n = lane + 64 * i;
            float v = 0.f;
            if (live) {
                int s = t * hop - NFFT / 2 + n;
                if (s < 0) s = -s;
                if (s >= L) s = 2 * (L - 1) - s;
                v = a[s] * tb.window[n];
            }
            x[bitrev9(n)] = make_float2(v, 0.f);
        }
        __syncthreads();
        // ---- radix-2 decimation-in-time FFT, 9 stages, 256 butterflies per stage (4 per lane)
#pragma unroll
        for (int st = 0; st < LOGN; ++st) {
            const int half = 1 << st;
#pragma unroll
            for (int i = 0; i < NFFT / 2 / 64; ++i) {
                const int q = lane + 64 * i;
                const int pos = q & (half - 1);
                const int i0 = ((q >> st) << (st + 1)) + pos, i1 = i0 + half;
                const float2 w = tb.twiddle[pos << (LOGN - 1 - st)];
                const float2 u = x[i0], v = x[i1];
                const float2 tv = make_float2(w.x * v.x - w.y * v.y, w.x * v.y + w.y * v.x);
                x[i0] = make_float2(u.x + tv.x, u.y + tv.y);
                x[i1] = make_float2(u.x - tv.x, u.y - tv.y);
            }
            __syncthreads();
        }
        // ---- power spectrum of bins 0..256 (kept in registers across the barrier, then aliased onto x[].x)
        float pw[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int k = lane + 64 * i;
            pw[i] = 0.f;
            if (k <= NFFT / 2) { const float2 z = x[k]; pw[i] = z.x * z.x + z.y * z.y; }
        }
        __syncthreads();
        float* P = reinterpret_cast<float*>(x);
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int k = lane + 64 * i;
            if (k <= NFFT / 2) P[k] = pw[i];
        }
        __syncthreads();
        // ---- sparse triangular filterbank + log
        for (int m = lane; m < n_mels; m += 64) {
            const int s0 = tb.fb_start[m], cnt = tb.fb_count[m];
            const float* w = tb.fb_weight + tb.fb_offset[m];
            float acc = 0.f;
            for (int j = 0; j < cnt; ++j) acc = fmaf(P[s0 + j], w[j], acc);
            float y = logf(acc + 1e-9f);
            if (normalize) y = (y - mean) * inv_std;
            sout[m][fl] = y;
        }
        __syncthreads();
    }
    // ---- coalesced store: rows of FRAMES_PER_BLOCK consecutive frames
    for (int i = threadIdx.x; i < n_mels * FRAMES_PER_BLOCK; i += 256) {
        const int m = i / FRAMES_PER_BLOCK, fl = i - m * FRAMES_PER_BLOCK;
        if (t0 + fl < Tm) mel[((size_t)b * n_mels + m) * Tm + t0 + fl] = sout[m][fl];
    }
}

}  // namespace

int launch_mel(const float* audio, int B, int L, const MelTables& t, int n_fft, int hop, int n_mels, int Tm,
               int normalize, float mean, float std, float* mel, hipStream_t s) {
    if (B <= 0) return 0;
    if (n_fft != NFFT || n_mels > MAX_MELS || L <= n_fft / 2) return -2;
    const int tiles = (Tm + FRAMES_PER_BLOCK - 1) / FRAMES_PER_BLOCK;
    hipLaunchKernelGGL(mel_kernel, dim3(B * tiles), dim3(256), 0, s, audio, L, t, hop, n_mels, Tm, normalize, mean,
                       1.0f / std, mel);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
