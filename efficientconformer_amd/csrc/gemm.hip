// bf16 MFMA GEMM with fused epilogues for the Conformer encoder (gfx950).
//
//   C[m, n] = sum_k A[m, k] * W[n, k] + bias[n]
//
// Replaces every nn.Linear / pointwise Conv1d on the hot path of the reference
// (models/layers.py:57-67, 122-136; call sites modules.py:378-382, 502-508, attentions.py:57-60, 471;
// encoders.py:116) together with the element-wise op that follows it (Swish, GLU, residual add,
// +u/+v and the (B,T,D)->(B,H,T/G,d) head split of attentions.py:674-686).
//
// Tiling: 128 x BN x 64 block tile, 4 waves (2x2), each wave 64 x BN/2 built from
// v_mfma_f32_32x32x16_bf16; register-staged global->LDS double buffer with one barrier per k-tile;
// LDS rows padded by 16 B (row stride 144 B = 9 x 16 B slots, conflict-free for ds_read_b128).
// M is large (B*T), N and K are small/odd multiples of 4: A tail chunks are masked in registers,
// weights are zero-padded once at pack time.
#include "kernels.h"

namespace {

constexpr int BM = 128;
constexpr int BK = 64;
constexpr int LROW = BK * 2 + 16;   // bytes per LDS row

struct FastDiv {   // exact for n * d < 2^32
    uint32_t mul, d;
    __host__ __device__ FastDiv() : mul(0), d(1) {}
    __host__ explicit FastDiv(uint32_t dd) : mul(dd > 1 ? (uint32_t)((1ull << 32) / dd + 1) : 0), d(dd) {}
    __device__ __forceinline__ uint32_t div(uint32_t n) const { return d == 1 ? n : __umulhi(n, mul); }
};

struct GemmDev {
    GemmParams p;
    FastDiv fG, fD, fd;
    bool staged;          // bf16 row outputs leave through the LDS-staged 16-byte-piece epilogue (alignment checked by launch_gemm)
};

template <int BN, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const GemmDev gd) {
    const GemmParams& p = gd.p;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sA = smem;                       // [2][BM][LROW]
    char* sB = smem + 2 * BM * LROW;       // [2][BN][LROW]
    constexpr int NB = BN / 32;            // B chunks per thread
    constexpr int NF = BN / 64;            // 32-col fragments per wave

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int n_tiles = (p.N + BN - 1) / BN;
    const int m_tiles = (p.M + BM - 1) / BM;
    const int id = xcd_remap(blockIdx.x, m_tiles * n_tiles);
    const int tm = id / n_tiles, tn = id - tm * n_tiles;
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- staging addresses (constant over the k loop)
    const int srow = tid >> 3, kc = tid & 7;
    const bf16_t* a_ptr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int m = m0 + srow + 32 * i;
        m = m < p.M ? m : p.M - 1;
        size_t src = m;
        if (p.a_rows > 0) {
            int b = m / p.a_rows, r = m - b * p.a_rows;
            src = (size_t)b * p.a_pitch + (size_t)r * p.a_stride;
        }
        a_ptr[i] = p.A + src * p.lda;
    }
    const bf16_t* b_ptr[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) b_ptr[i] = p.W + (size_t)(n0 + srow + 32 * i) * p.ldw + kc * 8;

    f32x16 acc[2][NF];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < NF; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    const int nk = (p.K + BK - 1) / BK;
    const int frag_off = (lane & 31) * LROW + (lane >> 5) * 16;
    // Two register staging sets: the global loads of k-tile kt+2 are issued at the top of iteration kt and published (written to
    // LDS) at the bottom of iteration kt+1, so they have a whole iteration to arrive.  (One set - issue at the top, publish at the
    // bottom of the SAME iteration - left only one 16-MFMA compute phase to cover the L2 / HBM latency.)
    struct Stage { uint4 ra[4], rb[NB]; int valid, tail; };
    Stage s0, s1;
#pragma unroll
    for (int i = 0; i < 4; ++i) { s0.ra[i] = make_uint4(0, 0, 0, 0); s1.ra[i] = s0.ra[i]; }
#pragma unroll
    for (int i = 0; i < NB; ++i) { s0.rb[i] = make_uint4(0, 0, 0, 0); s1.rb[i] = s0.rb[i]; }
    s0.valid = s1.valid = 0;
    s0.tail = s1.tail = 1;
    auto load_tile = [&](Stage& st, int t) __attribute__((always_inline)) {
        const int k = t * BK + kc * 8;
        const int valid = p.K - k;
        const int kcl = k < p.lda - 8 ? k : p.lda - 8;          // unconditional, clamped (in-bounds) loads; masked below
        const int tb = t < nk ? t : nk - 1;                     // weight tiles past the end: harmless re-read, never published
#pragma unroll
        for (int i = 0; i < 4; ++i) st.ra[i] = *reinterpret_cast<const uint4*>(a_ptr[i] + (kcl < 0 ? 0 : kcl));
#pragma unroll
        for (int i = 0; i < NB; ++i) st.rb[i] = *reinterpret_cast<const uint4*>(b_ptr[i] + tb * BK);
        st.tail = t >= nk - 1;
        st.valid = valid;          // the K tail is masked at PUBLISH time: masking here made every iteration wait for the loads it had
                                   // just issued (s_waitcnt vmcnt right after them: one L2 round trip per k-tile, 20 % of MFMA peak)
    };
    auto publish = [&](const Stage& st, int buf) __attribute__((always_inline)) {
        char* a = sA + buf * BM * LROW;
        char* b = sB + buf * BN * LROW;
        // only the last k-tile can hold columns >= K: wave-uniform branch (the 64 mask instructions were ~15 % of a k-tile's issue slots)
        if (__builtin_amdgcn_readfirstlane(st.tail)) {
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<uint4*>(a + (srow + 32 * i) * LROW + kc * 16) = mask_chunk(st.ra[i], st.valid);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<uint4*>(a + (srow + 32 * i) * LROW + kc * 16) = st.ra[i];
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) *reinterpret_cast<uint4*>(b + (srow + 32 * i) * LROW + kc * 16) = st.rb[i];
    };
    auto compute = [&](int buf) __attribute__((always_inline)) {
        const char* a = sA + buf * BM * LROW + (wm * 64) * LROW + frag_off;
        const char* b = sB + buf * BN * LROW + (wn * (BN / 2)) * LROW + frag_off;
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
            bf16x8 af[2], bf[NF];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) af[mi] = *reinterpret_cast<const bf16x8*>(a + mi * 32 * LROW + kk * 32);
#pragma unroll
            for (int ni = 0; ni < NF; ++ni) bf[ni] = *reinterpret_cast<const bf16x8*>(b + ni * 32 * LROW + kk * 32);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < NF; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mi], bf[ni], acc[mi][ni], 0, 0, 0);
        }
    };
    load_tile(s0, 0);
    load_tile(s1, 1);
    publish(s0, 0);
    __syncthreads();
    for (int kt = 0; kt < nk; kt += 2) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {            // h = 0: tile kt in buffer 0, tile kt+1 waiting in s1, tile kt+2 -> s0;  h = 1: roles swapped
            const int t = kt + h;
            if (t < nk) {
                // UNCONDITIONAL (tiles past the end re-read a clamped tile and are never published): with `if (t + 2 < nk)` the
                // compiler has to cover the path without these 8 younger loads and waits vmcnt(7..0) at the publish below instead
                // of vmcnt(15..8) - i.e. for the loads just issued, which removes the second stage of the prefetch
                if (h == 0) load_tile(s0, t + 2); else load_tile(s1, t + 2);
                compute(h);
                if (t + 1 < nk) { if (h == 0) publish(s1, 1); else publish(s0, 0); }
                __syncthreads();
            }
        }
    }

    // ---- epilogue.  C layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    const int lcol = lane & 31, lrow = 4 * (lane >> 5);
    if ((EPI == EPI_BF16 || EPI == EPI_SWISH_BF16 || EPI == EPI_QKV_NAT) && gd.staged) {
        // bf16 row outputs: the wave's 64 x BN/2 tile goes through LDS (the operand buffers are free now) and leaves as 16-byte
        // pieces, 64-128 contiguous bytes per row and 8-16 rows per store instruction.  One 2-byte global store per element (64
        // store instructions per wave, each touching two 64-byte row fragments) was half of the kernel (s_memtime phases: 51 %).
        constexpr int WCOLS = BN / 2, PITCH = WCOLS * 2 + 16, PPR = WCOLS / 8, RPI = 64 / PPR;
        __syncthreads();                                   // every wave is done with the last k-tile
        char* wbuf = smem + wave * 64 * PITCH;
        static_assert(4 * 64 * PITCH <= 2 * (BM + BN) * LROW, "staging fits the operand buffers");
#pragma unroll
        for (int ni = 0; ni < NF; ++ni) {
            const int n = n0 + wn * WCOLS + ni * 32 + lcol;
            const int nc = n < p.N ? n : p.N - 1;
            float add = p.bias[nc];
            if constexpr (EPI == EPI_QKV_NAT) { const int which = gd.fD.div(nc); if (which == 0) add += p.u[nc]; }
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float val = acc[mi][ni][r] + add;
                    if constexpr (EPI == EPI_SWISH_BF16) val = swishf_(val);
                    if (n >= p.N) val = 0.f;               // pad columns of the row buffers stay zero
                    const int row = mi * 32 + (r & 3) + 8 * (r >> 2) + lrow;
                    *reinterpret_cast<bf16_t*>(wbuf + row * PITCH + (ni * 32 + lcol) * 2) = f2bf(val);
                }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int piece = lane % PPR, prow = lane / PPR;
        const int nb = n0 + wn * WCOLS + piece * 8;       // first column of this lane's 8-column piece
        const int Tp = p.Tg * p.G;
#pragma unroll
        for (int it = 0; it < 64 / RPI; ++it) {
            const int row = it * RPI + prow, m = m0 + wm * 64 + row;
            const uint4 v = *reinterpret_cast<const uint4*>(wbuf + row * PITCH + piece * 16);
            if (m >= p.M) continue;
            if constexpr (EPI == EPI_QKV_NAT) {
                if (nb >= p.N) continue;                   // D % 8 == 0 (checked by the launcher): a piece never straddles Q | K | V
                const int which = gd.fD.div(nb), nn = nb - which * p.D;
                const int b = m / p.T, t = m - b * p.T;
                bf16_t* dst = which == 1 ? p.kh : (which == 2 ? p.vt : p.qu);
                *reinterpret_cast<uint4*>(dst + ((size_t)b * Tp + t) * p.D + nn) = v;
            } else {
                if (nb < p.ldc) *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.C) + (size_t)m * p.ldc + nb) = v;
            }
        }
        return;
    }
    if (EPI == EPI_GLU_BF16 && gd.staged) {
        // a wave's 64 columns are (a | b) of 32 output channels: stage 64 rows x 32 channels of bf16, leave as 16-byte pieces
        constexpr int PITCH = 32 * 2 + 16, PPR = 4, RPI = 16;
        __syncthreads();
        char* wbuf = smem + wave * 64 * PITCH;
        const int jc = (n0 + wn * 64) / 2;                 // first output channel of this wave
        const float ba = p.bias[n0 + wn * 64 + lcol], bb = p.bias[n0 + wn * 64 + 32 + lcol];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = mi * 32 + (r & 3) + 8 * (r >> 2) + lrow;
                *reinterpret_cast<bf16_t*>(wbuf + row * PITCH + lcol * 2) = f2bf((acc[mi][0][r] + ba) * sigmoidf_(acc[mi][NF - 1][r] + bb));
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int piece = lane % PPR, prow = lane / PPR, jb = jc + piece * 8;
#pragma unroll
        for (int it = 0; it < 64 / RPI; ++it) {
            const int row = it * RPI + prow, m = m0 + wm * 64 + row;
            const uint4 v = *reinterpret_cast<const uint4*>(wbuf + row * PITCH + piece * 16);
            if (m < p.M && jb < p.ldc) *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.C) + (size_t)m * p.ldc + jb) = v;
        }
        return;
    }
    if (EPI == EPI_RESID_F32 && gd.staged) {
        // fp32 residual rows: stage the wave's 64 x BN/2 tile of (acc + bias), then per 16-byte piece: residual in, sum out
        constexpr int WCOLS = BN / 2, PITCH = WCOLS * 4 + 16, PPR = WCOLS / 4, RPI = 64 / PPR, NIT = 64 / RPI;
        static_assert(4 * 64 * PITCH <= 2 * (BM + BN) * LROW, "staging fits the operand buffers");
        __syncthreads();
        char* wbuf = smem + wave * 64 * PITCH;
#pragma unroll
        for (int ni = 0; ni < NF; ++ni) {
            const int n = n0 + wn * WCOLS + ni * 32 + lcol;
            const float bias = p.bias[n < p.N ? n : p.N - 1];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = mi * 32 + (r & 3) + 8 * (r >> 2) + lrow;
                    *reinterpret_cast<float*>(wbuf + row * PITCH + (ni * 32 + lcol) * 4) = p.alpha * (acc[mi][ni][r] + bias);
                }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int piece = lane % PPR, prow = lane / PPR, nb = n0 + wn * WCOLS + piece * 4;
        const int nbc = nb < p.N - 4 ? nb : p.N - 4;
        float4 rv[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {                 // all residual pieces first (unconditional, clamped)
            const int m = m0 + wm * 64 + it * RPI + prow;
            rv[it] = *reinterpret_cast<const float4*>(p.R + (size_t)(m < p.M ? m : p.M - 1) * p.ldr + nbc);
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) asm volatile("" : "+v"(rv[it].x), "+v"(rv[it].y), "+v"(rv[it].z), "+v"(rv[it].w));
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int row = it * RPI + prow, m = m0 + wm * 64 + row;
            const float4 a = *reinterpret_cast<const float4*>(wbuf + row * PITCH + piece * 16);
            if (m < p.M && nb < p.N)
                *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.C) + (size_t)m * p.ldc + nb) = make_float4(rv[it].x + a.x, rv[it].y + a.y, rv[it].z + a.z, rv[it].w + a.w);
        }
        return;
    }
    if constexpr (EPI == EPI_GLU_BF16) {
        static_assert(BN == 128, "GLU needs both halves in one wave");
        const int j = (n0 + wn * 64) / 2 + lcol;           // output channel
        const float ba = p.bias[n0 + wn * 64 + lcol], bb = p.bias[n0 + wn * 64 + 32 + lcol];
        bf16_t* C = reinterpret_cast<bf16_t*>(p.C);
        if (j < p.ldc) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int m = m0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + lrow;
                    if (m < p.M) C[(size_t)m * p.ldc + j] = f2bf((acc[mi][0][r] + ba) * sigmoidf_(acc[mi][1][r] + bb));
                }
        }
    } else if constexpr (EPI == EPI_QKV || EPI == EPI_HEADS) {
        // rows m = (b, t); grouped view: t = G*tq + toff; flat = toff*D + nn; head h = flat / d, x = flat % d
        int b0 = 0, t0 = m0 + wm * 64;
        if (EPI == EPI_QKV) { b0 = t0 / p.T; t0 -= b0 * p.T; }
#pragma unroll
        for (int ni = 0; ni < NF; ++ni) {
            const int n = n0 + wn * (BN / 2) + ni * 32 + lcol;
            if (n >= p.N) continue;
            int which = 1, nn = n;
            if (EPI == EPI_QKV) { which = gd.fD.div(n); nn = n - which * p.D; }
            const float bias = p.bias[n];
            float bu = 0.f;
            if (EPI == EPI_QKV && which == 0) bu = p.u[nn];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rr = mi * 32 + (r & 3) + 8 * (r >> 2) + lrow;
                    if (m0 + wm * 64 + rr >= p.M) continue;
                    int b = b0, t = t0 + rr;
                    if (EPI == EPI_QKV) while (t >= p.T) { t -= p.T; ++b; }
                    const int tq = gd.fG.div(t), toff = t - tq * p.G;
                    const int flat = toff * p.D + nn;
                    const int h = gd.fd.div(flat), x = flat - h * p.d;
                    const float val = acc[mi][ni][r] + bias;
                    const size_t idx = ((size_t)(b * p.H + h) * p.Tg + tq) * p.dpad + x;
                    if (which == 0) p.qu[idx] = f2bf(val + bu);
                    else if (which == 1) p.kh[idx] = f2bf(val);
                    else p.vt[idx] = f2bf(val);
                }
        }
    } else if constexpr (EPI == EPI_QKV_NAT) {
        // natural layout: out[which][(b*Tp + t)][nn], which = n / D; rows m = (b, t)
        const int Tp = p.Tg * p.G;
        int b0 = (m0 + wm * 64) / p.T, t0 = (m0 + wm * 64) - b0 * p.T;
#pragma unroll
        for (int ni = 0; ni < NF; ++ni) {
            const int n = n0 + wn * (BN / 2) + ni * 32 + lcol;
            if (n >= p.N) continue;
            const int which = gd.fD.div(n), nn = n - which * p.D;
            const float bias = p.bias[n];
            const float bu = which == 0 ? p.u[nn] : 0.f;
            bf16_t* dst = which == 1 ? p.kh : (which == 2 ? p.vt : p.qu);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rr = mi * 32 + (r & 3) + 8 * (r >> 2) + lrow;
                    if (m0 + wm * 64 + rr >= p.M) continue;
                    int b = b0, t = t0 + rr;
                    while (t >= p.T) { t -= p.T; ++b; }
                    const size_t idx = ((size_t)b * Tp + t) * p.D + nn;
                    const float val = acc[mi][ni][r] + bias;
                    if (which == 0) p.qu[idx] = f2bf(val + bu);
                    else dst[idx] = f2bf(val);
                }
        }
    } else {
#pragma unroll
        for (int ni = 0; ni < NF; ++ni) {
            const int n = n0 + wn * (BN / 2) + ni * 32 + lcol;
            const bool f32out = (EPI == EPI_F32 || EPI == EPI_RESID_F32);
            const int nlim = f32out ? p.N : p.ldc;   // bf16 outputs also write their (zero) pad columns
            if (n >= nlim) continue;
            const float bias = p.bias[n];
            if constexpr (EPI == EPI_RESID_F32) {
                // residual epilogue: all 32 residual loads of this column tile first (unconditional, clamped rows), then the stores.
                // As one guarded load -> add -> store per element the loads were waited for one by one: 47 % of a wave's life
                // (s_memtime phases) for K = 2048.
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) {
                    float rv[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = m0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + lrow;
                        rv[r] = p.R[(size_t)(m < p.M ? m : p.M - 1) * p.ldr + n];
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(rv[r]));
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = m0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + lrow;
                        if (m < p.M) reinterpret_cast<float*>(p.C)[(size_t)m * p.ldc + n] = rv[r] + p.alpha * (acc[mi][ni][r] + bias);
                    }
                }
                continue;
            }
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + lrow;
                    if (m >= p.M) continue;
                    float val = acc[mi][ni][r] + bias;
                    if constexpr (EPI == EPI_F32) {
                        reinterpret_cast<float*>(p.C)[(size_t)m * p.ldc + n] = val;
                    } else if constexpr (EPI == EPI_RESID_F32) {
                        reinterpret_cast<float*>(p.C)[(size_t)m * p.ldc + n] = p.R[(size_t)m * p.ldr + n] + p.alpha * val;
                    } else if constexpr (EPI == EPI_SWISH_BF16) {
                        reinterpret_cast<bf16_t*>(p.C)[(size_t)m * p.ldc + n] = f2bf(swishf_(val));
                    } else {
                        reinterpret_cast<bf16_t*>(p.C)[(size_t)m * p.ldc + n] = f2bf(val);
                    }
                }
        }
    }
}

template <int BN, int EPI>
int launch_t(const GemmDev& gd, hipStream_t s) {
    const GemmParams& p = gd.p;
    const int n_tiles = (p.N + BN - 1) / BN, m_tiles = (p.M + BM - 1) / BM;
    const size_t lds = 2 * (BM + BN) * LROW;
    static LdsAttr attr;
    ensure_dynamic_lds(reinterpret_cast<const void*>(&gemm_kernel<BN, EPI>), (int)lds, attr);
    hipLaunchKernelGGL((gemm_kernel<BN, EPI>), dim3(m_tiles * n_tiles), dim3(256), lds, s, gd);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

template <int EPI>
int launch_bn(const GemmDev& gd, hipStream_t s) {
    const int n = gd.p.N;
    // the 64-wide tile wastes less when N just exceeds a multiple of 128 by <= 64
    if constexpr (EPI != EPI_GLU_BF16) {
        if (ec_round_up(n, 64) < ec_round_up(n, 128)) return launch_t<64, EPI>(gd, s);
    }
    return launch_t<128, EPI>(gd, s);
}

}  // namespace

constexpr long GEMM256_MIN_TILES = 100;     // Large 35.4 -> 34.3 ms per step, ConformerCTC-Large 48.9 -> 47.1, Medium unchanged (profiles/r5_22_gemm256_tile_threshold.txt); 200 until round 4

int launch_gemm(const GemmParams& p, int epi, hipStream_t s) {
    if (p.M <= 0 || p.N <= 0 || p.K <= 0) return 0;
    if (p.lda % 8 || p.ldw % 64) return -2;
    if (p.wide != 1 && gemm256_supported(p, epi)) {
        // gemm256.hip when its 256 x 256 tiles fill the chip (256 CUs, one 8-wave workgroup each).  Measured on the Large layer shapes
        // (tools/gemm_bench.py, in situ with bench.py --wide-gemm): >= 200 tiles: up to 1.4x over the 128 x 128 kernel (FFN, D = 720),
        // never behind; fewer tiles, or its 256 x 128 tile at any size: behind.  `wide` 2 / 3 force a tile (kernel-level tests).
        // Round 5: the threshold was tuned on kernels running ALONE (a launch of 141 tiles leaves 45 % of the CUs idle and loses to the 128 x 128 kernel's
        // 564 workgroups); in the forward three row ranges run on three streams and the chip is saturated (throughput flat from B = 256 to 1024,
        // profiles/r5_03_*), so what a launch costs is its CU time, not its latency - and per CU the 256 x 256 LDS-DMA kernel is the efficient one.
        // `wide` >= 16 = the tile threshold itself (option wide_gemm; default GEMM256_MIN_TILES)
        const long t256 = (long)((p.M + 255) / 256) * ((p.N + 255) / 256);
        const long min_tiles = p.wide >= 16 ? p.wide : GEMM256_MIN_TILES;
        if (p.wide == 2 || ((p.wide == 0 || p.wide >= 16) && p.N >= 192 && t256 >= min_tiles)) return launch_gemm256(p, epi, 256, s);
        if (p.wide == 3) return launch_gemm256(p, epi, 128, s);
    }
    GemmDev gd;
    gd.p = p;
    {
        auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
        gd.staged = false;
        if (epi == EPI_BF16 || epi == EPI_SWISH_BF16) gd.staged = p.ldc % 8 == 0 && al16(p.C);
        if (epi == EPI_QKV_NAT) gd.staged = p.D % 8 == 0 && al16(p.qu) && al16(p.kh) && al16(p.vt);
        if (epi == EPI_GLU_BF16) gd.staged = p.ldc % 8 == 0 && al16(p.C);
        if (epi == EPI_RESID_F32) gd.staged = p.N % 4 == 0 && p.N >= 4 && p.ldc % 4 == 0 && p.ldr % 4 == 0 && al16(p.C) && al16(p.R);
    }
    if (epi == EPI_QKV || epi == EPI_HEADS || epi == EPI_QKV_NAT) {
        gd.fG = FastDiv(p.G); gd.fD = FastDiv(p.D); gd.fd = FastDiv(p.d);
    }
    switch (epi) {
        case EPI_F32: return launch_bn<EPI_F32>(gd, s);
        case EPI_BF16: return launch_bn<EPI_BF16>(gd, s);
        case EPI_SWISH_BF16: return launch_bn<EPI_SWISH_BF16>(gd, s);
        case EPI_RESID_F32: return launch_bn<EPI_RESID_F32>(gd, s);
        case EPI_GLU_BF16: return launch_bn<EPI_GLU_BF16>(gd, s);
        case EPI_QKV: return launch_bn<EPI_QKV>(gd, s);
        case EPI_HEADS: return launch_bn<EPI_HEADS>(gd, s);
        case EPI_QKV_NAT: return launch_bn<EPI_QKV_NAT>(gd, s);
    }
    return -3;
}
