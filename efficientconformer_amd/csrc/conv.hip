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

#include <cstring>

namespace {

constexpr int SUB_TT = 8;     // output frames per workgroup

__global__ __launch_bounds__(256) void subsample_conv_kernel(const float* __restrict__ mel, int F, int Tm, int T1,
                                                             const float* __restrict__ w9, const float* __restrict__ bias,
                                                             int C, bf16_t* out, int ldo, const int* __restrict__ rag_tm,
                                                             const int* __restrict__ rag_off, const int* __restrict__ rag_t1) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TW = 2 * SUB_TT + 1;                 // input frames needed: 2t-1 .. 2t+1
    float* sm = reinterpret_cast<float*>(smem);        // [F + 1][TW + 1]  row 0 = frequency -1 (zero pad)
    float* sw = sm + (F + 1) * (TW + 1);               // [C][10]: 9 taps + bias
    const int tiles = (T1 + SUB_TT - 1) / SUB_TT;
    const int b = blockIdx.x / tiles, t0 = (blockIdx.x % tiles) * SUB_TT;
    const int tid = threadIdx.x;
    const int F2 = F / 2;
    const int tmb = rag_tm ? rag_tm[b] : Tm;          // ragged batch: the conv's zero padding starts at the utterance's own last mel frame
    // rag_off != null (round 4): the rows go straight into the RAGGED row space - utterance b owns rows [rag_off[b], rag_off[b + 1]) (its
    // rag_t1[b] frames + the group-padding rows, written as zeros) - and the tiles behind an utterance's own end do nothing: no pad frames are
    // computed or written, the Linear runs on the valid rows only and the gather pass is gone (Large: 31 % of the rectangle was padding).
    int t1b = T1, tpb = T1;
    size_t orow0 = (size_t)b * T1;
    if (rag_off) { t1b = rag_t1[b]; tpb = rag_off[b + 1] - rag_off[b]; orow0 = (size_t)rag_off[b]; if (t0 >= tpb) return; }
    for (int i = tid; i < (F + 1) * TW; i += 256) {
        const int fr = i / TW, tc = i - fr * TW;
        const int f = fr - 1, t = 2 * t0 - 1 + tc;
        const bool ok = f >= 0 && t >= 0 && t < tmb;
        const float v = mel[((size_t)b * F + (f < 0 ? 0 : f)) * Tm + (t < 0 ? 0 : (t < Tm ? t : Tm - 1))];   // clamped, unconditional
        sm[fr * (TW + 1) + tc] = ok ? v : 0.f;
    }
    for (int i = tid; i < C * 10; i += 256) {
        const int c = i / 10, j = i - c * 10;
        sw[i] = j < 9 ? w9[c * 9 + j] : bias[c];
    }
    __syncthreads();
    // a thread owns (channel c, two adjacent output frequencies f, f+1) for ALL frames of the tile: the channel's 9 taps + bias live in
    // registers and every LDS row of the 5 x (2 SUB_TT + 1) input window is read once (85 LDS reads per 16 outputs).  The first version
    // looped frames outermost and fetched taps and window per output: 19 LDS reads per output, 70 % of the wave cycles in LDS waits
    // (profiles/r2_03_large_sq_counters.txt), 1.03 ms per launch on Large's C = 360 front end.
    const int pairs = C * F2 / 2;                      // F2 is even for F = 80
    const int nt = (tpb - t0) < SUB_TT ? (tpb - t0) : SUB_TT;
    for (int q = tid; q < pairs; q += 256) {
        const int c = q / (F2 / 2), f = 2 * (q - c * (F2 / 2));
        float w[10];
#pragma unroll
        for (int i = 0; i < 10; ++i) w[i] = sw[c * 10 + i];
        float acc[2][SUB_TT];
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int tl = 0; tl < SUB_TT; ++tl) acc[e][tl] = w[9];
        // input rows 2f-1 .. 2f+3 = LDS rows 2f .. 2f+4: row i feeds output f with tap row i (i < 3) and output f+1 with tap row i-2 (i >= 2)
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const float* m = sm + (2 * f + i) * (TW + 1);
            float row[TW];
#pragma unroll
            for (int j = 0; j < TW; ++j) row[j] = m[j];
#pragma unroll
            for (int tl = 0; tl < SUB_TT; ++tl)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    if (i < 3) acc[0][tl] = fmaf(w[i * 3 + j], row[2 * tl + j], acc[0][tl]);
                    if (i >= 2) acc[1][tl] = fmaf(w[(i - 2) * 3 + j], row[2 * tl + j], acc[1][tl]);
                }
        }
        bf16_t* ocol = out + (orow0 + t0) * ldo + c * F2 + f;
#pragma unroll
        for (int tl = 0; tl < SUB_TT; ++tl)
            if (tl < nt) *reinterpret_cast<uint32_t*>(ocol + (size_t)tl * ldo) = t0 + tl < t1b ? pack_bf2(swishf_(acc[0][tl]), swishf_(acc[1][tl])) : 0u;
    }
}

// ---- depthwise conv: block = 128 output frames x 64 channels; thread = one channel PAIR (one dword of a bf16 row) x 16
// consecutive output frames.  The pair's KSZ x 2 folded tap weights live in registers and the input rows stream through once:
// row r is read from LDS as one dword, unpacked once, and feeds every output whose window contains it (static tap indices, the
// whole nest unrolled: 16*KSZ packed FMAs, (15*STRIDE + KSZ) LDS dwords and as many unpacks per thread).  An earlier version
// (8 channels x 4 outputs per thread, tap loop rolled) re-read and re-unpacked every row once per (tap, output) and fetched the
// tap weights from LDS per tap: ~2x the VALU work and 12x the LDS bytes, and it was the VALU / LDS pipes, not HBM, that bounded
// it (profiles/r1_12_kernel_stats.txt: 22 us for 36 MB).
constexpr int DW_NT = 16;                      // outputs per thread
constexpr int DW_TT = 8 * DW_NT;               // output frames per workgroup
constexpr int DW_CC = 64;                      // channels per workgroup
constexpr int DW_PITCH = DW_CC * 2 + 16;       // bytes per LDS row

typedef float dwf2 __attribute__((ext_vector_type(2)));

template <int KSZ, int STRIDE>
__global__ __launch_bounds__(256) void dwconv_kernel(const bf16_t* __restrict__ g, int T, int To, int C, int ld,
                                                     const float* __restrict__ w_kc, const float* __restrict__ bias, bf16_t* out,
                                                     RaggedConv rc, int causal) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ROWS = (DW_TT - 1) * STRIDE + KSZ;
    constexpr int TROWS = (DW_NT - 1) * STRIDE + KSZ;                       // input rows one thread consumes
    char* sg = smem;                                                        // [ROWS][DW_PITCH] bf16 rows
    const int ctiles = (C + DW_CC - 1) / DW_CC;
    const int ttiles = (To + DW_TT - 1) / DW_TT;
    int id = blockIdx.x;
    const int ct = id % ctiles; id /= ctiles;
    int tt, b, Top = To;                                                    // Top: output rows of the utterance including its group padding
    size_t grow0, orow0;                                                    // first input / output row of the utterance
    if (rc.tile_off) {
        // ragged batch: utterance b has its own T / To ("same" zero padding at ITS ends), its rows start at in_off[b] / out_off[b] of the
        // concatenated row spaces; rc.tile_off = prefix sums of the utterances' time tiles.  Output rows To .. Top - 1 pad the utterance
        // to a multiple of the NEXT stage's attention group size and are written as zeros.
        b = ragged_find_wave(rc.tile_off, rc.n, id); tt = id - rc.tile_off[b];
        T = rc.in_len[b]; To = rc.out_len[b];
        grow0 = (size_t)rc.in_off[b]; orow0 = (size_t)rc.out_off[b]; Top = rc.out_off[b + 1] - rc.out_off[b];
    } else { tt = id % ttiles; b = id / ttiles; grow0 = (size_t)b * T; orow0 = (size_t)b * To; }
    const int c0 = ct * DW_CC, to0 = tt * DW_TT;
    // zero pre-padding of reference layers.py:97-101: "same" = ((k - 1) / 2 on both sides), "causal" = (k - 1, 0)
    const int tin0 = to0 * STRIDE - (causal ? KSZ - 1 : (KSZ - 1) / 2);
    const int tid = threadIdx.x;
    const int cp = tid & 31, tg = tid >> 5;
    const int c = c0 + 2 * cp;                                              // this thread's channel pair
    // the pair's folded taps + bias: clamped unconditional loads (tiny, L2-resident), issued with the tile loads
    dwf2 w[KSZ], bz;
    {
        const unsigned ca = c < C ? c : C - 1, cb = c + 1 < C ? c + 1 : C - 1;     // unsigned 32-bit offsets: base + offset addressing
#pragma unroll
        for (int j = 0; j < KSZ; ++j) {
            const float wa = w_kc[(unsigned)(j * C) + ca], wb = w_kc[(unsigned)(j * C) + cb];
            w[j] = dwf2{c < C ? wa : 0.f, c + 1 < C ? wb : 0.f};
        }
        const float ba = bias[ca], bb = bias[cb];
        bz = dwf2{c < C ? ba : 0.f, c + 1 < C ? bb : 0.f};
    }
    {   // all loads of the tile first (one memory latency instead of one per pass), then the LDS writes
        constexpr int NL = (ROWS * (DW_CC / 8) + 255) / 256;
        const bf16_t* gb = g + grow0 * ld;                                  // uniform base; one utterance is < 2^31 elements
        uint4 v[NL];
#pragma unroll
        for (int n = 0; n < NL; ++n) {
            const int i = tid + 256 * n;
            const int r = i >> 3, ch = (i & 7) * 8;
            const int t = tin0 + (r < ROWS ? r : ROWS - 1), cq = c0 + ch;
            const int tc = t < 0 ? 0 : (t < T ? t : T - 1), cc = cq < ld - 8 ? cq : ld - 8;    // clamped, unconditional load
            v[n] = *reinterpret_cast<const uint4*>(gb + (unsigned)(tc * ld + cc));
        }
#pragma unroll
        for (int n = 0; n < NL; ++n) {
            const int i = tid + 256 * n;
            const int r = i >> 3, ch = (i & 7) * 8;
            const int t = tin0 + r, cq = c0 + ch;
            if (r < ROWS) *reinterpret_cast<uint4*>(sg + r * DW_PITCH + ch * 2) = mask_chunk(v[n], (t >= 0 && t < T && cq < ld) ? C - cq : 0);
        }
    }
    __syncthreads();
    dwf2 acc[DW_NT];
#pragma unroll
    for (int o = 0; o < DW_NT; ++o) acc[o] = bz;
    const char* base = sg + (tg * DW_NT * STRIDE) * DW_PITCH + cp * 4;
#pragma unroll
    for (int r = 0; r < TROWS; ++r) {
        const uint32_t raw = *reinterpret_cast<const uint32_t*>(base + r * DW_PITCH);
        const dwf2 x = dwf2{__uint_as_float(raw << 16), __uint_as_float(raw & 0xFFFF0000u)};
#pragma unroll
        for (int o = 0; o < DW_NT; ++o) {
            const int tap = r - o * STRIDE;                                 // static after unrolling
            if (tap >= 0 && tap < KSZ) acc[o] = __builtin_elementwise_fma(w[tap], x, acc[o]);
        }
    }
    if (c < ld) {   // ld is even, so the pair is inside the row; channels >= C have zero taps and bias: swish(0) = 0 keeps the pad columns zero
        bf16_t* ob = out + orow0 * ld;
#pragma unroll
        for (int o = 0; o < DW_NT; ++o) {
            const int to = to0 + tg * DW_NT + o;
            if (to < Top) *reinterpret_cast<uint32_t*>(ob + (unsigned)(to * ld + c)) = to < To ? pack_bf2(swishf_(acc[o].x), swishf_(acc[o].y)) : 0u;
        }
    }
}

// ---- depthwise conv on the MATRIX pipe (round 5; stride 1).  dwconv_kernel above is bound by the VALU: 16 * KSZ * 2 fp32 FMAs per thread are half of its
// issue slots (the Swish of the 32 outputs is a third; tools/experiments/probes/valu_rate_probe.hip).  A depthwise convolution of one channel is a product with
// a Toeplitz matrix, and v_mfma_f32_4x4x4_16b_bf16 computes 16 INDEPENDENT 4 x 4 x 4 products per wave - one block per channel:
//     out[16 S + 4 j + i] = sum_q sum_k A_q[i][k] * X[4 S + j + q][k],    A_q[i][k] = w[4 q + k - i] (0 outside the taps),   X[n][k] = x[4 n + k]
// (S = 16-frame set, q = 0 .. NQ - 1 groups of 4 taps).  Block b = lanes 4 b .. 4 b + 3 holds channel b: the A operand of lane 4 b + i is row i of A_q (packed
// on the host: bf16 hi + lo halves of the fp32 folded weight - two MFMAs per q, the products are exact in fp32), the B operand of lane 4 b + j is four
// CONSECUTIVE frames of the channel (block 4 S + j + q), the accumulators of lane 4 b + j are four consecutive output frames.  Operands that are runs of frames
// per lane need the tile as [channel][frame] in LDS: every thread loads the 16-byte chunk (8 channels) of four consecutive frame rows - the global access pattern of
// dwconv_kernel - builds the channels' 4-frame blocks with two v_perm each and writes eight 8-byte blocks; the outputs (Swish, bf16) go back IN PLACE (output block 4 S + j over input block
// 4 S + j: every read of set S precedes it in the wave's program order, later sets start at block 4 S + 4; a wave only touches its own 16 channel rows), and a last
// pass transposes back (16-byte stores).  Per wave and 2048 outputs: 16 NQ MFMAs (8 cycles each, beside other waves' VALU work), ~70 staging + ~60
// epilogue VALU instructions and the Swish - against ~1000 VALU instructions before.
template <int KSZ>
struct DwM {
    static constexpr int NQ = (KSZ + 6) / 4;              // groups of 4 taps: 4 q + k - i covers -3 .. KSZ + 2
    static constexpr int NBLK = DW_TT / 4 + NQ - 1;       // 4-frame input blocks of a tile: 4 S + j + q
    static constexpr int PITCH = NBLK * 8 + 8;            // bytes per channel row (74 dwords at KSZ = 15: staging writes and operand reads spread over the banks)
    static constexpr int LDS = DW_CC * PITCH;
};
typedef short dw_s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ dw_s16x4 dw_as_s16x4(uint32_t a, uint32_t b) { union { uint2 u; dw_s16x4 s; } c; c.u = make_uint2(a, b); return c.s; }

template <int KSZ>
__global__ __launch_bounds__(256) void dwconv_mfma_kernel(const bf16_t* __restrict__ g, int T, int To, int C, int ld, const uint4* __restrict__ wa,
                                                          const float* __restrict__ bias, bf16_t* out, RaggedConv rc, int causal) {
    using M = DwM<KSZ>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int ctiles = (C + DW_CC - 1) / DW_CC;
    const int ttiles = (To + DW_TT - 1) / DW_TT;
    int id = blockIdx.x;
    const int ct = id % ctiles; id /= ctiles;
    int tt, b, Top = To;
    size_t grow0, orow0;
    if (rc.tile_off) {                                                      // ragged batch: see dwconv_kernel
        b = ragged_find_wave(rc.tile_off, rc.n, id); tt = id - rc.tile_off[b];
        T = rc.in_len[b]; To = rc.out_len[b];
        grow0 = (size_t)rc.in_off[b]; orow0 = (size_t)rc.out_off[b]; Top = rc.out_off[b + 1] - rc.out_off[b];
    } else { tt = id % ttiles; b = id / ttiles; grow0 = (size_t)b * T; orow0 = (size_t)b * To; }
    const int c0 = ct * DW_CC, to0 = tt * DW_TT;
    const int tin0 = to0 - (causal ? KSZ - 1 : (KSZ - 1) / 2);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // ---- this lane's Toeplitz rows (A operands) and bias: block = channel 16 wave + (lane >> 2), row / column index lane & 3
    const int chl = lane >> 2, xi = lane & 3;
    const int chw = c0 + 16 * wave + chl, chc = chw < C ? chw : C - 1;
    uint4 aw[M::NQ];
#pragma unroll
    for (int q = 0; q < M::NQ; ++q) aw[q] = wa[(size_t)(chc * 4 + xi) * M::NQ + q];
    float bz = bias[chc];
    // ---- stage the tile transposed: thread = 8 channels (one 16-byte chunk of a frame row, as dwconv_kernel loads them) x input block t32 (+ 32)
    const int c8 = tid & 7, t32 = tid >> 3;
    const int cq = c0 + 8 * c8;
    {
        const bf16_t* gb = g + grow0 * ld;
        const unsigned cc = cq < ld - 8 ? cq : ld - 8;                      // clamped, unconditional loads
        constexpr int NQD = M::NBLK > 32 ? 2 : 1;
        uint4 v[NQD][4];
#pragma unroll
        for (int n = 0; n < NQD; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int t = tin0 + 4 * (t32 + 32 * n) + r;
                const int tc = t < 0 ? 0 : (t < T ? t : T - 1);
                v[n][r] = *reinterpret_cast<const uint4*>(gb + ((unsigned)(tc * ld) + cc));
            }
#pragma unroll
        for (int n = 0; n < NQD; ++n) {
            const int Q = t32 + 32 * n;
            if (n == 0 || Q < M::NBLK) {
                uint32_t w[4][4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {                               // "same" / causal zero padding at the utterance's own ends; channels >= C: zero
                    const int t = tin0 + 4 * Q + r;
                    const uint4 x = mask_chunk(v[n][r], (t >= 0 && t < T && cq < ld) ? C - cq : 0);
                    w[r][0] = x.x; w[r][1] = x.y; w[r][2] = x.z; w[r][3] = x.w;
                }
                char* dst = smem + (8 * c8) * M::PITCH + Q * 8;
#pragma unroll
                for (int pq = 0; pq < 4; ++pq) {                            // dword pq of a row = channels 2 pq (low half), 2 pq + 1
                    *reinterpret_cast<uint2*>(dst + (2 * pq) * M::PITCH) =
                        make_uint2(__builtin_amdgcn_perm(w[1][pq], w[0][pq], 0x05040100u), __builtin_amdgcn_perm(w[3][pq], w[2][pq], 0x05040100u));
                    *reinterpret_cast<uint2*>(dst + (2 * pq + 1) * M::PITCH) =
                        make_uint2(__builtin_amdgcn_perm(w[1][pq], w[0][pq], 0x07060302u), __builtin_amdgcn_perm(w[3][pq], w[2][pq], 0x07060302u));
                }
            }
        }
    }
    if (chw >= C) {                                                         // pad channels: zero taps and bias -> swish(0) = 0 keeps the pad columns zero
        bz = 0.f;
#pragma unroll
        for (int q = 0; q < M::NQ; ++q) aw[q] = make_uint4(0, 0, 0, 0);
    }
    __syncthreads();
    // ---- 8 sets of 16 frames: NQ x 2 MFMAs, Swish, in-place bf16 blocks
    char* xrow = smem + (16 * wave + chl) * M::PITCH + xi * 8;
#pragma unroll
    for (int S = 0; S < DW_TT / 16; ++S) {
        f32x4 acc = f32x4{bz, bz, bz, bz};
        uint2 bx[M::NQ];
#pragma unroll
        for (int q = 0; q < M::NQ; ++q) bx[q] = *reinterpret_cast<const uint2*>(xrow + (4 * S + q) * 8);
#pragma unroll
        for (int q = 0; q < M::NQ; ++q) {
            const dw_s16x4 xb = dw_as_s16x4(bx[q].x, bx[q].y);
            acc = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(dw_as_s16x4(aw[q].x, aw[q].y), xb, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(dw_as_s16x4(aw[q].z, aw[q].w), xb, acc, 0, 0, 0);
        }
        *reinterpret_cast<uint2*>(xrow + (4 * S) * 8) = make_uint2(pack_bf2(swishf_(acc[0]), swishf_(acc[1])), pack_bf2(swishf_(acc[2]), swishf_(acc[3])));
    }
    __syncthreads();
    // ---- back to (frame, channel) rows: thread = 8 channels x output block t32: four 16-byte stores
    if (cq < ld) {
        bf16_t* ob = out + orow0 * ld;
        const char* src = smem + (8 * c8) * M::PITCH + t32 * 8;
        uint32_t w[4][4];
#pragma unroll
        for (int pq = 0; pq < 4; ++pq) {
            const uint2 ya = *reinterpret_cast<const uint2*>(src + (2 * pq) * M::PITCH);
            const uint2 yb = *reinterpret_cast<const uint2*>(src + (2 * pq + 1) * M::PITCH);
            w[0][pq] = __builtin_amdgcn_perm(yb.x, ya.x, 0x05040100u); w[1][pq] = __builtin_amdgcn_perm(yb.x, ya.x, 0x07060302u);
            w[2][pq] = __builtin_amdgcn_perm(yb.y, ya.y, 0x05040100u); w[3][pq] = __builtin_amdgcn_perm(yb.y, ya.y, 0x07060302u);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int to = to0 + 4 * t32 + r;
            if (to < Top) *reinterpret_cast<uint4*>(ob + ((unsigned)(to * ld) + (unsigned)cq)) = to < To ? make_uint4(w[r][0], w[r][1], w[r][2], w[r][3]) : make_uint4(0, 0, 0, 0);
        }
    }
}

template <int KSZ>
int launch_dw_mfma(const bf16_t* g, int B, int T, int To, int C, int ld, const void* wa, const float* bias, bf16_t* out, hipStream_t s,
                   const RaggedConv& rc, int causal) {
    const int ctiles = (C + DW_CC - 1) / DW_CC, ttiles = (To + DW_TT - 1) / DW_TT;
    const int nwg = rc.tile_off ? rc.tiles * ctiles : B * ttiles * ctiles;
    if (nwg <= 0) return 0;
    static LdsAttr attr;
    ensure_dynamic_lds(reinterpret_cast<const void*>(&dwconv_mfma_kernel<KSZ>), DwM<KSZ>::LDS, attr);
    hipLaunchKernelGGL((dwconv_mfma_kernel<KSZ>), dim3(nwg), dim3(256), DwM<KSZ>::LDS, s, g, T, To, C, ld, reinterpret_cast<const uint4*>(wa), bias, out, rc, causal);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

template <int KSZ, int STRIDE>
int launch_dw_t(const bf16_t* g, int B, int T, int To, int C, int ld, const float* w_kc, const float* bias, bf16_t* out, hipStream_t s,
                const RaggedConv& rc, int causal) {
    const int ctiles = (C + DW_CC - 1) / DW_CC, ttiles = (To + DW_TT - 1) / DW_TT;
    const int nwg = rc.tile_off ? rc.tiles * ctiles : B * ttiles * ctiles;
    if (nwg <= 0) return 0;
    constexpr int ROWS = (DW_TT - 1) * STRIDE + KSZ;
    const size_t lds = (size_t)ROWS * DW_PITCH;
    static LdsAttr attr;
    ensure_dynamic_lds(reinterpret_cast<const void*>(&dwconv_kernel<KSZ, STRIDE>), (int)lds, attr);
    hipLaunchKernelGGL((dwconv_kernel<KSZ, STRIDE>), dim3(nwg), dim3(256), lds, s, g, T, To, C, ld, w_kc, bias, out, rc, causal);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace

int launch_subsample_conv(const float* mel, int B, int F, int Tm, int T1, const float* w9, const float* bias, int C,
                          bf16_t* out, int ldo, hipStream_t s, const int* rag_tm, const RaggedRows* rg) {
    if (B <= 0 || T1 <= 0) return 0;
    if (F % 4 || (C * (F / 2)) % 2) return -2;
    if (rg && (!rag_tm || !rg->off || !rg->len)) return -2;
    const int tiles = (T1 + SUB_TT - 1) / SUB_TT;      // ragged rows: T1 = an upper bound of every utterance's padded row count
    const size_t lds = ((size_t)(F + 1) * (2 * SUB_TT + 2) + (size_t)C * 10) * sizeof(float);
    hipLaunchKernelGGL(subsample_conv_kernel, dim3(B * tiles), dim3(256), lds, s, mel, F, Tm, T1, w9, bias, C, out, ldo, rag_tm,
                       rg ? rg->off : nullptr, rg ? rg->len : nullptr);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// Host side of dwconv_mfma_kernel's A operands: [channel][row i][group q] x 16 bytes = bf16 hi (4 taps) | bf16 lo (4 taps) of w[4 q + k - i]
int dwconv_mfma_groups(int ksize) { return (ksize + 6) / 4; }
bool dwconv_mfma_supported(int ksize, int stride) { return stride == 1 && (ksize == 15 || ksize == 31 || ksize == 7); }
void pack_dwconv_mfma(const float* w_kc, int ksize, int C, uint16_t* dst) {
    const int nq = dwconv_mfma_groups(ksize);
    auto bf = [](float f) { uint32_t u; memcpy(&u, &f, 4); const uint32_t r = u + 0x7FFFu + ((u >> 16) & 1u); return (uint16_t)(r >> 16); };   // RNE (finite weights)
    auto fl = [](uint16_t h) { const uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; };
    for (int ch = 0; ch < C; ++ch)
        for (int i = 0; i < 4; ++i)
            for (int q = 0; q < nq; ++q) {
                uint16_t* o = dst + ((size_t)(ch * 4 + i) * nq + q) * 8;
                for (int k = 0; k < 4; ++k) {
                    const int tap = 4 * q + k - i;
                    const float w = tap >= 0 && tap < ksize ? w_kc[(size_t)tap * C + ch] : 0.f;
                    const uint16_t hi = bf(w);
                    o[k] = hi; o[4 + k] = bf(w - fl(hi));
                }
            }
}

int launch_dwconv(const bf16_t* g, int B, int T, int To, int C, int ld, const float* w_kc, const float* bias,
                  int ksize, int stride, bf16_t* out, hipStream_t s, const RaggedConv* rcp, int causal, const void* w_mfma) {
    if (B <= 0 || To <= 0) return 0;
    if (ld % 8 || ld < C) return -2;
    RaggedConv rc{};
    if (rcp) rc = *rcp;
    if (w_mfma && stride == 1) {
        if (ksize == 15) return launch_dw_mfma<15>(g, B, T, To, C, ld, w_mfma, bias, out, s, rc, causal);
        if (ksize == 31) return launch_dw_mfma<31>(g, B, T, To, C, ld, w_mfma, bias, out, s, rc, causal);
        if (ksize == 7) return launch_dw_mfma<7>(g, B, T, To, C, ld, w_mfma, bias, out, s, rc, causal);
    }
    // taps are fully unrolled per (kernel size, stride); the shipped configs use k = 15 (Efficient Conformer) and 31 (Conformer)
    if (ksize == 15 && stride == 1) return launch_dw_t<15, 1>(g, B, T, To, C, ld, w_kc, bias, out, s, rc, causal);
    if (ksize == 15 && stride == 2) return launch_dw_t<15, 2>(g, B, T, To, C, ld, w_kc, bias, out, s, rc, causal);
    if (ksize == 31 && stride == 1) return launch_dw_t<31, 1>(g, B, T, To, C, ld, w_kc, bias, out, s, rc, causal);
    if (ksize == 31 && stride == 2) return launch_dw_t<31, 2>(g, B, T, To, C, ld, w_kc, bias, out, s, rc, causal);
    if (ksize == 7 && stride == 1) return launch_dw_t<7, 1>(g, B, T, To, C, ld, w_kc, bias, out, s, rc, causal);
    if (ksize == 7 && stride == 2) return launch_dw_t<7, 2>(g, B, T, To, C, ld, w_kc, bias, out, s, rc, causal);
    return -3;
}
