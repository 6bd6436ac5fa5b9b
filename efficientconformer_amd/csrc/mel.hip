// Log-mel frontend (gfx950): framing with centre reflect padding, Hann(win) centred in n_fft,
// 512-point FFT, power spectrum, sparse HTK triangular filterbank, log(x + 1e-9).
//
// Restates what the reference obtains from torchaudio (models/modules.py:81-82, 90-96:
// Spectrogram(n_fft, win_length, hop_length, power=2) -> MelScale(n_mels, sr, 0, 8000) -> log).
// The arithmetic of that dependency is not in the reference repository ("parity unpinned", see
// oracle/ref_encoder.py); this kernel is validated against the oracle's torch.stft restatement.
//
// HBM-bound by design (640 B in, 320 B out per frame), so the FFT is organised to stay out of the way:
//   * one wave transforms TWO real frames at once as one complex 512-point FFT (z = a + i b) and separates them
//     with X_a[k] = (Z[k] + conj Z[N-k]) / 2, X_b[k] = (Z[k] - conj Z[N-k]) / 2i;
//   * 512 = 8 x 8 x 8: three radix-8 passes entirely in registers (8 complex values per lane) with two
//     exchanges through a per-wave LDS buffer whose pitches (72 / 68 float2) make every access conflict-free;
//     synchronisation is wave-local (LDS operations of one wave execute in order) — no workgroup barriers
//     inside the transform (the first version used 9 block-wide barriers per frame: profiles/r1_02_*);
//   * a workgroup produces 32 consecutive frames of one utterance, staged in LDS and written as 128-byte rows
//     of the (B, n_mels, Tm) output.
#include "kernels.h"

// libeffconf is compiled WITHOUT packed-fp32 VALU instructions (_build.py: -target-feature -packed-fp32-ops).  Measured on MI355X
// (tools/mel_repro.py, tools/mel_repro2.py, profiles/r2_mel_packed_fp32_hazard.txt): v_pk_add_f32 / v_pk_mul_f32 whose `op_sel`
// selects the HIGH half of a source pair for the low result lane - the forms the complex butterflies of this kernel compile to -
// return wrong values while a bf16 MFMA of ANOTHER wave executes on the same SIMD.  That was round 1's "mel kernel next to another
// stream's subsampling kernels" corruption.  MEL_PK_BUILD = a second compilation of this file WITH packed fp32, kept as diagnostic
// variant 8 so that the reproducer can show the hazard; the product never launches it.
#ifdef MEL_PK_BUILD
#define launch_mel launch_mel_pk_unused
#define launch_mel_debug launch_mel_debug_pk
#define MEL_NS mel_pk_build
#else
#define MEL_NS
#endif

namespace MEL_NS {

constexpr int NFFT = 512;
constexpr int FRAMES_PER_BLOCK = 32;     // 4 waves x 4 iterations x 2 frames
constexpr int MAX_MELS = 128;
constexpr int P1 = 72;                   // float2 pitch of the step-1 exchange buffer [k1][n2*8+n3]
constexpr int P2 = 68;                   // float2 pitch of the step-2 exchange buffer [n3][k1+8*k2]
constexpr int MAX_FBW = 768;             // filterbank weights kept in LDS (every bin belongs to <= 2 triangles: ~2 * 257)

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 mul_mi(float2 a) { return make_float2(a.y, -a.x); }   // a * (-i)
__device__ __forceinline__ float2 mul_pi(float2 a) { return make_float2(-a.y, a.x); }   // a * (+i)

// in-place 8-point DFT: v[k] <- sum_j v[j] W8^{jk}
__device__ __forceinline__ void dft8(float2 (&v)[8]) {
    const float r = 0.70710678118654752f;
    float2 a[4], b[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { a[j] = cadd(v[j], v[j + 4]); b[j] = csub(v[j], v[j + 4]); }
    b[1] = make_float2(r * (b[1].x + b[1].y), r * (b[1].y - b[1].x));      // * W8^1 = (1 - i)/sqrt2
    b[2] = mul_mi(b[2]);                                                   // * W8^2 = -i
    b[3] = make_float2(r * (b[3].y - b[3].x), -r * (b[3].x + b[3].y));     // * W8^3 = (-1 - i)/sqrt2
    {
        const float2 s0 = cadd(a[0], a[2]), s1 = cadd(a[1], a[3]), d0 = csub(a[0], a[2]), d1 = csub(a[1], a[3]);
        v[0] = cadd(s0, s1); v[4] = csub(s0, s1); v[2] = cadd(d0, mul_mi(d1)); v[6] = cadd(d0, mul_pi(d1));
    }
    {
        const float2 s0 = cadd(b[0], b[2]), s1 = cadd(b[1], b[3]), d0 = csub(b[0], b[2]), d1 = csub(b[1], b[3]);
        v[1] = cadd(s0, s1); v[5] = csub(s0, s1); v[3] = cadd(d0, mul_mi(d1)); v[7] = cadd(d0, mul_pi(d1));
    }
}

// LDS operations of one wave execute in order; this only stops the compiler from reordering across the hand-off
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Diagnostic variants (V != 0, tools/mel_repro.py; the product launches V = 0, whose code and LDS layout are unchanged):
//   V & 1  canary words in front of and behind the per-wave exchange buffers, checked when the workgroup exits (dbg[0])
//   V & 2  every wave-level hand-off verifies itself: own writes read back (dbg[1], [3], [5]) and every exchanged value read twice
//          (dbg[2], [4], [6])
//   V & 4  workgroup barriers instead of the wave-level hand-off
__device__ __forceinline__ float2 reread(const float2* p) {           // a second, un-mergeable read of the same LDS word pair
    const volatile float* q = reinterpret_cast<const volatile float*>(p);
    return make_float2(q[0], q[1]);
}

// MM = rows of the output staging tile (80 for the shipped n_mels = 80: with the exchange buffers at their own pitches and the twiddle table at 256 entries the
// workgroup's LDS image is 53.6 KB - THREE workgroups per CU instead of two at 63 KB; the kernel is a dependent chain per wave, so occupancy is its speed)
template <int V, int MM>
__global__ __launch_bounds__(256) void mel_kernel(const float* __restrict__ audio, int L, MelTables tb, int hop,
                                                  int n_mels, int Tm, int normalize, float mean, float inv_std,
                                                  float* __restrict__ mel, unsigned int* __restrict__ dbg,
                                                  const int64_t* __restrict__ rag_len = nullptr) {
    constexpr int CAN = (V & 1) ? 512 : 0;                       // canary words on each side of the exchange buffers
    constexpr int WBUF = 8 * P1 * 8;                             // bytes of one wave's exchange buffer: A [8][P1] float2 and B [8][P2] float2 are the SAME bytes (see below)
    constexpr int O_SBUF = CAN * 4, O_POST = O_SBUF + 4 * WBUF, O_STW = O_POST + CAN * 4, O_SWIN = O_STW + (NFFT / 2) * 8,
                  O_SOUT = O_SWIN + NFFT * 4, O_SFW = O_SOUT + MM * (FRAMES_PER_BLOCK + 1) * 4, O_END = O_SFW + MAX_FBW * 4;
    static_assert(8 * P1 >= NFFT && P1 >= P2 && 8 * P1 * 2 >= 264 + NFFT / 2 + 1, "one buffer holds Z[512], the step-2 layout and the two power spectra");
    __shared__ __attribute__((aligned(16))) char lds[O_END];
    float2* stw = reinterpret_cast<float2*>(lds + O_STW);        // W512^m, m < 256 (W^(m + 256) = -W^m: the sign goes on by one XOR per component)
    float* swin = reinterpret_cast<float*>(lds + O_SWIN);
    float (*sout)[FRAMES_PER_BLOCK + 1] = reinterpret_cast<float (*)[FRAMES_PER_BLOCK + 1]>(lds + O_SOUT);
    float* sfw = reinterpret_cast<float*>(lds + O_SFW);          // packed triangular weights (a global load per tap made the
                                                                 // rolled per-mel loop one L1 round trip per bin)
    if constexpr ((V & 1) != 0) {
        unsigned int* c0 = reinterpret_cast<unsigned int*>(lds), *c1 = reinterpret_cast<unsigned int*>(lds + O_POST);
        for (int i = threadIdx.x; i < CAN; i += 256) { c0[i] = 0xC0FFEE00u + i; c1[i] = 0xBADC0DE0u + i; }
    }
    auto hand_off = [&]() __attribute__((always_inline)) { if constexpr ((V & 4) != 0) __syncthreads(); else wave_sync(); };
    unsigned int bad[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto neq = [](float2 a, float2 b) { return (__float_as_uint(a.x) != __float_as_uint(b.x)) | (__float_as_uint(a.y) != __float_as_uint(b.y)); };
    const int tiles = (Tm + FRAMES_PER_BLOCK - 1) / FRAMES_PER_BLOCK;
    const int b = blockIdx.x / tiles, t0 = (blockIdx.x % tiles) * FRAMES_PER_BLOCK;
    // ragged batch (rag_len != null): the utterance runs at ITS OWN length Lb - reflect padding at its own ends, Lb / hop + 1 frames -
    // exactly what the reference computes for it alone; the row pitch of audio and of the mel image stay L and Tm
    const int Lb = rag_len ? (int)rag_len[b] : L;
    const int Tmb = rag_len ? Lb / hop + 1 : Tm;
    if (t0 >= Tmb) return;                                       // workgroup-uniform: a tile past the utterance's last frame
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool fw_lds = tb.fb_nnz <= MAX_FBW;
    {   // tables -> LDS: every global load of the set-up is issued before the first wait (the rolled copy loops paid one memory
        // round trip per iteration, ~7 per workgroup of 32 frames)
        static_assert(NFFT == 512 && MAX_FBW == 768, "two / three passes of 256 threads");
        const float2 w = tb.twiddle[tid];                        // table holds W512^k for k < 256; W^(k+256) = -W^k
        const float wn0 = tb.window[tid], wn1 = tb.window[tid + 256];
        float f[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) { const int i = tid + 256 * j; f[j] = tb.fb_weight[i < tb.fb_nnz ? i : (tb.fb_nnz > 0 ? tb.fb_nnz - 1 : 0)]; }
        stw[tid] = w;
        swin[tid] = wn0; swin[tid + 256] = wn1;
        if (fw_lds) {
#pragma unroll
            for (int j = 0; j < 3; ++j) if (tid + 256 * j < tb.fb_nnz) sfw[tid + 256 * j] = f[j];
        }
    }
    // this lane's mel bins (m = lane, lane + 64): filter descriptors fetched once, not once per frame pair inside the loop
    int fs0[2], fcnt[2], foff[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int m = (threadIdx.x & 63) + 64 * j, mc = m < n_mels ? m : n_mels - 1;
        fs0[j] = tb.fb_start[mc]; fcnt[j] = m < n_mels ? tb.fb_count[mc] : 0; foff[j] = tb.fb_offset[mc];
    }
    __syncthreads();
    float2* A = reinterpret_cast<float2*>(lds + O_SBUF + wave * WBUF);      // per wave: exchange buffers A (step 1, Z) and B (step 2, power)
    // B aliases A: every phase reads its 8 values per lane into registers (all lanes, one instruction at a time) before it writes the next layout, and the LDS
    // operations of one wave execute in order - with two buffers the workgroup's LDS image was 53.6 KB (three workgroups per CU), with one it is 36 KB (four)
    float2* B = A;
    auto tw = [&](int m) __attribute__((always_inline)) {            // W512^m for m in [0, 512)
        float2 w = stw[m & 255];
        const uint32_t sg = (uint32_t)(m & 256) << 23;
        w.x = __uint_as_float(__float_as_uint(w.x) ^ sg); w.y = __uint_as_float(__float_as_uint(w.y) ^ sg);
        return w;
    };
    const float* a = audio + (size_t)b * L;

    // gather of one frame pair (reflect at the ends of the padded waveform, torch.stft center=True): unconditional clamped loads
    auto gather = [&](int ta, float (&xa)[8], float (&xb)[8]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int n = lane + 64 * j;
            int sa = ta * hop - NFFT / 2 + n, sb = sa + hop;
            sa = sa < 0 ? -sa : sa; sa = sa >= Lb ? 2 * (Lb - 1) - sa : sa; sa = sa < 0 ? 0 : (sa >= Lb ? Lb - 1 : sa);
            sb = sb < 0 ? -sb : sb; sb = sb >= Lb ? 2 * (Lb - 1) - sb : sb; sb = sb < 0 ? 0 : (sb >= Lb ? Lb - 1 : sb);
            xa[j] = a[sa]; xb[j] = a[sb];
        }
    };
    float nxa[8], nxb[8];                                        // the NEXT pair's samples are in flight during this pair's transform
    gather(t0 + wave * (FRAMES_PER_BLOCK / 4), nxa, nxb);
    for (int it = 0; it < FRAMES_PER_BLOCK / 8; ++it) {
        const int fl = wave * (FRAMES_PER_BLOCK / 4) + 2 * it;   // local index of the first frame of the pair
        const int ta = t0 + fl, tbb = ta + 1;
        // ---- window both frames
        float2 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float w = swin[lane + 64 * j];
            v[j] = make_float2(ta < Tmb ? nxa[j] * w : 0.f, tbb < Tmb ? nxb[j] * w : 0.f);
        }
        if (it + 1 < FRAMES_PER_BLOCK / 8) gather(ta + 2, nxa, nxb);
        // ---- step 1: lane = 8*n2 + n3 holds x[64*n1 + lane]; DFT over n1, twiddle W64^{n2 k1}
        {
            const int n2 = lane >> 3;
            dft8(v);
#pragma unroll
            for (int k1 = 0; k1 < 8; ++k1) { v[k1] = cmul(v[k1], tw((8 * n2 * k1) & 511)); A[k1 * P1 + lane] = v[k1]; }
        }
        hand_off();
        if constexpr ((V & 2) != 0) {
#pragma unroll
            for (int k1 = 0; k1 < 8; ++k1) bad[1] += neq(A[k1 * P1 + lane], v[k1]);
        }
        // ---- step 2: lane = 8*k1 + n3; DFT over n2, twiddle W512^{n3 (k1 + 8 k2)}
        {
            const int k1 = lane >> 3, n3 = lane & 7;
#pragma unroll
            for (int n2 = 0; n2 < 8; ++n2) v[n2] = A[k1 * P1 + n2 * 8 + n3];
            if constexpr ((V & 2) != 0) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                for (int n2 = 0; n2 < 8; ++n2) bad[2] += neq(reread(&A[k1 * P1 + n2 * 8 + n3]), v[n2]);
            }
            dft8(v);
#pragma unroll
            for (int k2 = 0; k2 < 8; ++k2) { v[k2] = cmul(v[k2], tw((n3 * (k1 + 8 * k2)) & 511)); B[n3 * P2 + k1 + 8 * k2] = v[k2]; }
        }
        hand_off();
        if constexpr ((V & 2) != 0) {
            const int k1 = lane >> 3, n3 = lane & 7;
#pragma unroll
            for (int k2 = 0; k2 < 8; ++k2) bad[3] += neq(B[n3 * P2 + k1 + 8 * k2], v[k2]);
        }
        // ---- step 3: lane = k1 + 8*k2; DFT over n3 -> Z[lane + 64*k3]
#pragma unroll
        for (int n3 = 0; n3 < 8; ++n3) v[n3] = B[n3 * P2 + lane];
        if constexpr ((V & 2) != 0) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int n3 = 0; n3 < 8; ++n3) bad[4] += neq(reread(&B[n3 * P2 + lane]), v[n3]);
        }
        dft8(v);
#pragma unroll
        for (int k3 = 0; k3 < 8; ++k3) A[lane + 64 * k3] = v[k3];
        hand_off();
        if constexpr ((V & 2) != 0) {
#pragma unroll
            for (int k3 = 0; k3 < 8; ++k3) bad[5] += neq(A[lane + 64 * k3], v[k3]);
        }
        // ---- separate the two real spectra, power for bins 0..256 -> B (as floats: [0..256] frame a, [264..520] frame b)
        float* Pa = reinterpret_cast<float*>(B);
        float* Pb = Pa + 264;
        float2 zz[5], zcc[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) {                                // all reads of Z first: the power spectra overwrite it (B aliases A)
            const int k = lane + 64 * j, kc = k <= NFFT / 2 ? k : NFFT / 2;
            zz[j] = A[kc]; zcc[j] = A[(NFFT - kc) & (NFFT - 1)];
            if constexpr ((V & 2) != 0) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                bad[6] += neq(reread(&A[kc]), zz[j]) + neq(reread(&A[(NFFT - kc) & (NFFT - 1)]), zcc[j]);
            }
        }
        hand_off();
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int k = lane + 64 * j;
            if (k <= NFFT / 2) {
                const float2 z = zz[j], zc = zcc[j];
                const float ar = 0.5f * (z.x + zc.x), ai = 0.5f * (z.y - zc.y);     // X_a = (Z[k] + conj Z[N-k]) / 2
                const float br = 0.5f * (z.y + zc.y), bi = 0.5f * (zc.x - z.x);     // X_b = (Z[k] - conj Z[N-k]) / 2i
                Pa[k] = ar * ar + ai * ai;
                Pb[k] = br * br + bi * bi;
            }
        }
        hand_off();
        // ---- sparse triangular filterbank + log for both frames
#pragma unroll
        for (int mj = 0; mj < 2; ++mj) {
            const int m = lane + 64 * mj;
            if (m >= n_mels) break;
            const int s0 = fs0[mj], cnt = fcnt[mj], off = foff[mj];
            float ya = 0.f, yb = 0.f;
            if (fw_lds) {
                for (int j0 = 0; j0 < cnt; j0 += 4) {             // 4 taps per trip, same summation order; taps >= cnt weigh 0
                    float wj[4], pa[4], pb[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int j = j0 + e, jc = j < cnt ? j : cnt - 1;
                        wj[e] = j < cnt ? sfw[off + jc] : 0.f;
                        pa[e] = Pa[s0 + jc]; pb[e] = Pb[s0 + jc];
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) { ya = fmaf(pa[e], wj[e], ya); yb = fmaf(pb[e], wj[e], yb); }
                }
            } else {
                const float* w = tb.fb_weight + off;
                for (int j = 0; j < cnt; ++j) { const float wj = w[j]; ya = fmaf(Pa[s0 + j], wj, ya); yb = fmaf(Pb[s0 + j], wj, yb); }
            }
            ya = logf(ya + 1e-9f); yb = logf(yb + 1e-9f);
            if (normalize) { ya = (ya - mean) * inv_std; yb = (yb - mean) * inv_std; }
            sout[m][fl] = ya;
            sout[m][fl + 1] = yb;
        }
        hand_off();
    }
    __syncthreads();
    // ---- coalesced store: rows of FRAMES_PER_BLOCK consecutive frames
    for (int i = tid; i < n_mels * FRAMES_PER_BLOCK; i += 256) {
        const int m = i / FRAMES_PER_BLOCK, fl = i - m * FRAMES_PER_BLOCK;
        if (t0 + fl < Tmb) mel[((size_t)b * n_mels + m) * Tm + t0 + fl] = sout[m][fl];
    }
    if constexpr ((V & 1) != 0) {
        const unsigned int* c0 = reinterpret_cast<const unsigned int*>(lds), *c1 = reinterpret_cast<const unsigned int*>(lds + O_POST);
        for (int i = threadIdx.x; i < CAN; i += 256) bad[0] += (c0[i] != 0xC0FFEE00u + i) + (c1[i] != 0xBADC0DE0u + i);
    }
    if constexpr (V != 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) if (bad[i]) atomicAdd(dbg + i, bad[i]);
    }
}

}  // namespace
#ifdef MEL_PK_BUILD
using namespace mel_pk_build;
#endif

int launch_mel(const float* audio, int B, int L, const MelTables& t, int n_fft, int hop, int n_mels, int Tm,
               int normalize, float mean, float std, float* mel, hipStream_t s, const int64_t* ragged_len) {
    if (B <= 0) return 0;
    if (n_fft != NFFT || n_mels > MAX_MELS || L <= n_fft / 2) return -2;
    const int tiles = (Tm + FRAMES_PER_BLOCK - 1) / FRAMES_PER_BLOCK;
    if (n_mels <= 80)
        hipLaunchKernelGGL((mel_kernel<0, 80>), dim3(B * tiles), dim3(256), 0, s, audio, L, t, hop, n_mels, Tm, normalize, mean,
                           1.0f / std, mel, (unsigned int*)nullptr, ragged_len);
    else
        hipLaunchKernelGGL((mel_kernel<0, MAX_MELS>), dim3(B * tiles), dim3(256), 0, s, audio, L, t, hop, n_mels, Tm, normalize, mean,
                           1.0f / std, mel, (unsigned int*)nullptr, ragged_len);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

#if defined(EFFCONF_DEBUG_ABI) || defined(MEL_PK_BUILD)        // libeffconf_debug.so only: the product library instantiates mel_kernel<0> alone
// diagnostics only (tools/mel_repro.py): kernel variant V, `extra_lds` bytes of unused dynamic LDS per workgroup, counters dbg[8]
int launch_mel_debug(int variant, int extra_lds, const float* audio, int B, int L, const MelTables& t, int n_fft, int hop, int n_mels,
                     int Tm, int normalize, float mean, float std, float* mel, unsigned int* dbg, hipStream_t s) {
    if (B <= 0) return 0;
    if (n_fft != NFFT || n_mels > MAX_MELS || L <= n_fft / 2) return -2;
#ifndef MEL_PK_BUILD
    if (variant & 8) return launch_mel_debug_pk(variant & 7, extra_lds, audio, B, L, t, n_fft, hop, n_mels, Tm, normalize, mean, std, mel, dbg, s);
#endif
    const int tiles = (Tm + FRAMES_PER_BLOCK - 1) / FRAMES_PER_BLOCK;
#define MEL_DBG_CASE(VV) case VV: { static LdsAttr attr; ensure_dynamic_lds(reinterpret_cast<const void*>(&mel_kernel<VV, MAX_MELS>), extra_lds, attr); \
        hipLaunchKernelGGL((mel_kernel<VV, MAX_MELS>), dim3(B * tiles), dim3(256), extra_lds, s, audio, L, t, hop, n_mels, Tm, normalize, mean, 1.0f / std, mel, dbg, (const int64_t*)nullptr); break; }
    switch (variant) {
        MEL_DBG_CASE(0) MEL_DBG_CASE(1) MEL_DBG_CASE(2) MEL_DBG_CASE(3) MEL_DBG_CASE(4) MEL_DBG_CASE(5) MEL_DBG_CASE(6) MEL_DBG_CASE(7)
        default: return -2;
    }
#undef MEL_DBG_CASE
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
#endif
