// Split-precision mode, second generation (round 6): the label-exact forward on FUSED kernels that take ragged batches and causal / streaming
// configurations.  Arithmetic as in split.hip (fp32 tensors, every product on the fp16 matrix pipe with operands x = h + l / 2048, three MFMAs per
// product, ~2^-21 relative); reference: models/attentions.py:549-718 (483-547 rel_to_abs, 506-529 causal skew, 1326-1403 masks), modules.py:511-525,
// layers.py:97-101, blocks.py:106-110.
//
//   sxf_attention_kernel   one workgroup = 64 grouped queries of one (utterance, head), all key tiles: S^T = K Qu^T + skew(E Qu^T) on the matrix pipe
//                          (TRANSPOSED scores: a lane owns a query column, so the softmax statistics are lane-local), online softmax in fp32, O^T += V^T P^T with
//                          the probabilities going from accumulator registers straight into B fragments - no score ever leaves the CU (split.hip wrote
//                          (B, H, Tg, Tg) fp32 scores to HBM and read them twice: 15.9 of its 31.4 ms).  Q + v is not formed: (Q + v) E^T = (Q + u) E^T + (v - u) E^T,
//                          and the second term is one number per (head, relative position) - sxf_posbias_kernel - that rides in a spare k column of E against a
//                          constant 1 in Q + u.  Ragged batches: per-utterance row offsets / lengths; chunk-padding rows (attentions.py:107-138) are substituted
//                          while staging (K = V = 0, Q = 0 -> Q + u = u), so no pad-row pass exists.  Streaming contexts / causal: band limits per query,
//                          causal relative tables indexed Tg - 1 + j - i with j <= i.
//   sxf_dwconv_kernel      depthwise conv + folded BatchNorm + Swish in fp32 on per-utterance row ranges, "same" or causal pre-padding.
//   sxf_decimate_kernel    frames 0, s, 2s, .. of every utterance (input of the conv_res 1 x 1 convolution of the transition blocks).
#include "kernels.h"
#include "sx_common.h"

namespace {

using namespace sx;

// Same-scale operand split (sx_common.h split2s): x S = h + l, one fp32 accumulator per product sum at the scale S S' - Q + u, K, E, V at 2^8, the probabilities
// at 2^10; the scales leave through constants that exist anyway (the score scale, the softmax normalisation).  The (h, l / 2048) form needed a correction
// accumulator and a fold per tile: a third of this kernel's VALU instructions.
constexpr float SQK = 256.0f, SV_ = 256.0f, SP_ = 1024.0f;
constexpr int SS_ROWS = 34;                                      // 32 key rows + a dump row on either side (band rows no pair of the wave's tile reads)
constexpr int SS_LD = 36;                                        // floats per key row of a wave's skew buffer (conflict-free: see the writer below; ds_*_b32 bank = dword mod 32 inside a 32-lane group)
constexpr int VROW = 64 * 2 + 16;                                // bytes per row of the V^T tiles (64 keys)

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// workgroup barrier that orders LDS traffic only: __syncthreads() also waits for every outstanding GLOBAL load (s_waitcnt vmcnt(0)), i.e. for the next
// tile's prefetch the moment it was issued
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int KS, int NT>
struct AttnLds {
    static constexpr int PK = 16 * KS, ROW = PK * 2 + 16;
    static constexpr int HALF = 2 * 64 * ROW;                    // one band half / the K tile: hi [64][ROW] | lo [64][ROW]
    static constexpr int VT = 2 * 32 * NT * VROW;                // V^T hi | lo
    static constexpr int SKEW = 4 * SS_ROWS * SS_LD * 4;         // one skew buffer per wave
    static constexpr bool VALIAS = 3 * HALF + VT + SKEW + PK * 4 > 150 * 1024;    // wide heads: V^T shares the K tile + the consumed band half
    static constexpr int BYTES = 3 * HALF + (VALIAS ? 0 : VT) + SKEW + PK * 4;
};

template <int KS, int NT>
__global__ __launch_bounds__(256, (AttnLds<KS, NT>::BYTES <= 80 * 1024 ? 2 : 1)) void sxf_attention_kernel(const SxfAttnParams p) {
    using L = AttnLds<KS, NT>;
    constexpr int PK = L::PK, ROW = L::ROW, CPR = PK / 4, HALF = L::HALF;
    extern __shared__ __attribute__((aligned(16))) char sm[];
    // [band half A][K tile][band half B]: the K tile and the band half a key tile has consumed are contiguous whichever half that is (V^T alias of the wide heads)
    char* const sEA = sm;
    char* const sK = sm + HALF;
    char* const sEB = sm + 2 * HALF;
    char* const tail = sm + 3 * HALF;
    char* const sVfix = tail;                                    // !VALIAS
    float* const sS = reinterpret_cast<float*>(tail + (L::VALIAS ? 0 : L::VT));
    float* const suv = sS + 4 * SS_ROWS * SS_LD;                      // [PK]: u of this head's columns | 1 at column d | 0
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wq = wave >> 1, wk = wave & 1, lr = lane & 31, kh = lane >> 5;
    const int b = blockIdx.y / p.H, h = blockIdx.y % p.H;
    const int d = p.d, G = p.G, D = p.D;
    // ---- this utterance: first row, frames that carry projections (rows behind them are chunk padding), key-mask length, grouped length
    long long row0; int nfr, klen, Tg;
    if (p.off) { row0 = p.off[b]; nfr = p.lens[b]; klen = nfr; Tg = (nfr + G - 1) / G; }
    else { row0 = (long long)b * p.Tp; nfr = p.T; klen = p.lens[b]; Tg = p.Tg; }
    const int i0 = blockIdx.x * 64;
    if (i0 >= Tg) return;                                        // ragged launches are sized for the longest utterance (whole workgroup leaves)
    const size_t hb = (size_t)h * d, gd = (size_t)G * D;
    const float* qbase = p.q + row0 * D + hb;
    const float* kbase = p.k + row0 * D + hb;
    const float* vbase = p.v + row0 * D + hb;
    const int erows = p.causal ? p.Tg : 2 * p.Tg - 1;            // rows of E (built for the LONGEST utterance: an utterance's own table is a centred slice of it)
    // valid elements of grouped row i's head span: natural row G i + (hb + x) / D carries a projection iff it is < nfr
    auto dspan = [&](int i) { const long long xl = (long long)(nfr - G * i) * D - (long long)hb; return (int)(xl < 0 ? 0 : (xl > d ? d : xl)); };
    for (int x = tid; x < PK; x += 256) suv[x] = x < d ? p.u[(int)((hb + x) % D)] : (x == d ? 1.0f : 0.f);
    __syncthreads();
    // ---- this lane's query as B fragments of Q + u (split), column d = 1 (the positional bias column of the staged E rows)
    f16x8 quh[KS], qul[KS];
    {
        int i = i0 + 32 * wq + lr;
        i = i < Tg ? i : Tg - 1;
        const float* qr = qbase + gd * i;
        const int de = dspan(i);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int x = 16 * s + 8 * kh;
            const float4 a = ld_span4(qr, x, de), c = ld_span4(qr, x + 4, de);
            const float4 u0 = *reinterpret_cast<const float4*>(suv + x), u1 = *reinterpret_cast<const float4*>(suv + x + 4);
            uint32_t uh[4], ul[4];
            split2s((a.x + u0.x) * SQK, (a.y + u0.y) * SQK, uh[0], ul[0]); split2s((a.z + u0.z) * SQK, (a.w + u0.w) * SQK, uh[1], ul[1]);
            split2s((c.x + u1.x) * SQK, (c.y + u1.y) * SQK, uh[2], ul[2]); split2s((c.z + u1.z) * SQK, (c.w + u1.w) * SQK, uh[3], ul[3]);
            quh[s] = as_f16x8(make_uint4(uh[0], uh[1], uh[2], uh[3])); qul[s] = as_f16x8(make_uint4(ul[0], ul[1], ul[2], ul[3]));
        }
    }
    // ---- key tiles this workgroup has to visit.  A tile all of whose keys are masked for all of the workgroup's queries contributes exp(-1e9 - max) = 0
    //      exactly, PROVIDED every query row has an unmasked key (its own): true for ragged batches and for query tiles in front of the key-mask length;
    //      pad-frame queries of a rectangular batch can have every key masked - the reference's softmax is then uniform over ALL keys - so those walk everything
    const int bl = p.band_l < (1 << 24) ? p.band_l : (1 << 24), br = p.causal ? 0 : (p.band_r < (1 << 24) ? p.band_r : (1 << 24));
    int jt_lo = 0, jt_hi = (Tg + 63) / 64;
    {
        const int ilast = (i0 + 63 < Tg ? i0 + 63 : Tg - 1);
        if (p.off || G * ilast < klen) {
            const int kmax = (klen + G - 1) / G;                 // first key masked by the padding mask
            int hi = (kmax + 63) / 64;
            const int hr = (ilast + br) / 64 + 1;
            hi = hi < hr ? hi : hr;
            jt_hi = jt_hi < hi ? jt_hi : hi;
            if (i0 - bl > 0) jt_lo = (i0 - bl) / 64;
        }
    }
    const float c2 = 1.4426950408889634f / sqrtf((float)d) / (SQK * SQK);      // accumulators (scale 2^16) -> scores in log2 units
    const int kvis = min(Tg, (klen + G - 1) / G);                // keys >= kvis are masked or do not exist
    // ---- staging.  K, V and E arrive PRE-SPLIT (sxf_pack_kv_kernel / sxf_pack_e_kernel: fp16 (h, l) planes, head spans zero padded to PK columns, chunk-padding rows
    //      zeroed, V transposed, the positional bias in column d of E): a key tile is KS + KS + 2 NT 16-byte copies per thread, requested for the NEXT tile before
    //      the current tile's products and written to LDS after them.  (First version: fp32 operands split, masked and transposed HERE, once per (query tile, key tile)
    //      pair - ~1900 VALU instructions per tile and wave against 72 MFMAs, and 18 dependent L2 round trips: 40 k cycles per tile for 2.3 k of matrix work.)
    typedef uint32_t u4v __attribute__((ext_vector_type(4)));
    constexpr int CP = PK / 8;                                   // 16-byte chunks per plane row
    // band row w (0 .. 127) of key tile j0 = E row Tgmax - 1 + j0 - i0 - 63 + w, clamped (rows no visible pair touches)
    auto erel = [&](int w, int j0) { int rel = p.Tg - 1 + j0 - i0 - 63 + w; return rel < 0 ? 0 : (rel > erows - 1 ? erows - 1 : rel); };
    const long long rowg0 = p.off ? row0 / G : (long long)b * p.Tg;      // first GROUPED row of this utterance in the packed K image
    const uint16_t* kpk = p.kp + ((size_t)rowg0 * p.H + h) * 2 * PK;
    const uint16_t* epk = p.ep + (size_t)h * 2 * PK;
    const uint16_t* vpk = p.vp + ((size_t)b * p.H + h) * 2 * 32 * NT * (size_t)p.vpitch;
    u4v kreg[KS], ereg[KS], vreg[2 * NT];
    auto fetch_ke = [&](int j0, int wbase, bool want_k) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < KS; ++it) {
            const int c = tid + 256 * it, r = c / (2 * CP), rem = c - r * 2 * CP;      // rem: chunk of the row's hi | lo pair (contiguous 2 PK halfs)
            if (want_k) { int j = j0 + r; j = j < Tg ? j : Tg - 1; kreg[it] = *reinterpret_cast<const u4v*>(kpk + (size_t)j * p.H * 2 * PK + rem * 8); }
            ereg[it] = *reinterpret_cast<const u4v*>(epk + (size_t)erel(wbase + r, j0) * p.H * 2 * PK + rem * 8);
        }
    };
    auto fetch_v = [&](int j0) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < 2 * NT; ++it) {
            const int c = tid + 256 * it, xr = c >> 3, ch = c & 7;                   // xr: plane * 32 NT + column
            vreg[it] = *reinterpret_cast<const u4v*>(vpk + (size_t)xr * p.vpitch + j0 + ch * 8);
        }
    };
    auto put_rows = [&](char* dst, const u4v (&reg)[KS]) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < KS; ++it) {
            const int c = tid + 256 * it, r = c / (2 * CP), rem = c - r * 2 * CP, pl = rem >= CP ? 1 : 0, ch = rem - pl * CP;
            *reinterpret_cast<u4v*>(dst + pl * 64 * ROW + r * ROW + ch * 16) = reg[it];
        }
    };
    float m_run = -INFINITY, l_run = 0.f;
    f32x16 oacc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[t][r] = 0.f;
    float* sSw = sS + wave * SS_ROWS * SS_LD + SS_LD;            // row 0 of the wave's 32 key rows (rows -1 and 32 are the dump rows)
    const int iq = i0 + 32 * wq + lr;                            // this lane's query (may lie behind the utterance: computed on clamped data, never stored)
    int cur = 0;                                                 // 0: band rows 0 .. 63 of the current key tile live in half A
    fetch_ke(jt_lo * 64, 0, false);
    put_rows(sEA, ereg);
    fetch_ke(jt_lo * 64, 64, true);
    if (!L::VALIAS) fetch_v(jt_lo * 64);
    for (int jt = jt_lo; jt < jt_hi; ++jt, cur ^= 1) {
        const int j0 = jt * 64;
        const int jn = (jt + 1 < jt_hi ? jt + 1 : jt) * 64;      // the tile to prefetch (the last tile prefetches itself: never published)
        char* const eLo = cur ? sEB : sEA;
        char* const eHi = cur ? sEA : sEB;
        char* const sV = L::VALIAS ? (cur ? sK : sEA) : sVfix;   // alias: consumed band half + K tile (contiguous either way)
        put_rows(sK, kreg);
        put_rows(eHi, ereg);
        auto put_v = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int it = 0; it < 2 * NT; ++it) {
                const int c = tid + 256 * it, xr = c >> 3, ch = c & 7;
                *reinterpret_cast<u4v*>(sV + xr * VROW + ch * 16) = vreg[it];
            }
        };
        // wide heads (V^T aliased): the K / E registers and the V registers are never live together - V of THIS tile is requested here and lands under the score
        // products, the next tile's K / E are requested behind the V^T write and land under the P V products (a third of the register file less: no scratch)
        if (!L::VALIAS) { put_v(); fetch_v(jn); fetch_ke(jn, 64, true); }      // land under this tile's products
        else fetch_v(j0);
        lds_barrier();
        // ---- S1^T = K (Q + u)^T on this wave's 32 keys x 32 queries
        f32x16 s1h, s1x;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s1h[r] = 0.f; s1x[r] = 0.f; }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int ko = (16 * s + 8 * kh) * 2;
            const f16x8 ah = *reinterpret_cast<const f16x8*>(sK + (32 * wk + lr) * ROW + ko), al = *reinterpret_cast<const f16x8*>(sK + 64 * ROW + (32 * wk + lr) * ROW + ko);
            s1h = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, quh[s], s1h, 0, 0, 0);
            s1x = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, qul[s], s1x, 0, 0, 0);
            s1x = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, quh[s], s1x, 0, 0, 0);
        }
        // ---- band product E_band (Q + u)^T on the 64 band rows this wave's (key, query) pairs touch, realigned by the writer:
        //      pair (jj, ii) of the wave's tile reads band row jj - ii + 31 of those 64; value of band row wr for query ii goes to S[wr - 31 + ii][ii].
        //      Banks: writer 40 wr + 41 ii (41 odd: 32 queries on 32 banks, the kh halves 160 = 32 banks apart), reader 40 jj + ii likewise
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int wb = 32 * (wk - wq + 1 + t) + lr;          // band row of the workgroup tile this lane loads as the A operand
            const char* eb = ((wb >> 6) ? eHi : eLo) + (wb & 63) * ROW;
            f32x16 ph, px;
#pragma unroll
            for (int r = 0; r < 16; ++r) { ph[r] = 0.f; px[r] = 0.f; }
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int ko = (16 * s + 8 * kh) * 2;
                const f16x8 eh = *reinterpret_cast<const f16x8*>(eb + ko), el = *reinterpret_cast<const f16x8*>(eb + 64 * ROW + ko);
                ph = __builtin_amdgcn_mfma_f32_32x32x16_f16(eh, quh[s], ph, 0, 0, 0);
                px = __builtin_amdgcn_mfma_f32_32x32x16_f16(eh, qul[s], px, 0, 0, 0);
                px = __builtin_amdgcn_mfma_f32_32x32x16_f16(el, quh[s], px, 0, 0, 0);
            }
            // band row wr of the wave's 64 -> key row wr - 31 + lr; t = 0: always < 32, t = 1: always >= 0 - the other side is clamped to the dump row (no branch:
            // 32 predicated stores were 32 exec-mask round trips per tile)
            const int jb = 4 * kh - 31 + lr;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int wr = 32 * t + (r & 3) + 8 * (r >> 2);
                const int jj = t == 0 ? max(wr + jb, -1) : min(wr + jb, 32);
                sSw[jj * SS_LD + lr] = ph[r] + px[r];
            }
        }
        wave_sync();
        // ---- scores of this lane's query against its 16 keys in log2 units (one multiply by log2(e) / sqrt(d): the exponentials below are bare v_exp_f32),
        //      additive mask (attentions.py:692-701: ONE mask = max(padding, streaming); -1e9 log2(e) absorbs any score exactly as -1e9 does), online softmax.
        //      Interior tiles - every key of the wave's 32 exists and is visible to every one of its 32 queries - skip the mask arithmetic (wave-uniform branch)
        float sc[16], tmax = -INFINITY;
        const int jw0 = j0 + 32 * wk, iw0 = i0 + 32 * wq;
        const bool edge = jw0 + 31 >= kvis || jw0 + 31 - iw0 > br || iw0 + 31 - jw0 > bl;
        if (!edge) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int jj = (r & 3) + 8 * (r >> 2) + 4 * kh;
                sc[r] = ((s1h[r] + s1x[r]) + sSw[jj * SS_LD + lr]) * c2;
                tmax = fmaxf(tmax, sc[r]);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int jj = (r & 3) + 8 * (r >> 2) + 4 * kh, j = jw0 + jj;
                float sv = ((s1h[r] + s1x[r]) + sSw[jj * SS_LD + lr]) * c2;
                if (G * j >= klen || j - iq > br || iq - j > bl) sv += -1.4426950e9f;
                if (j >= Tg) sv = -INFINITY;                     // no such key (ragged: behind this utterance; the last tile's tail)
                sc[r] = sv;
                tmax = fmaxf(tmax, sv);
            }
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
        const float m_new = fmaxf(m_run, tmax);
        const float m_use = m_new < -1e30f ? 0.f : m_new;       // nothing seen yet (a wave whose keys all lie behind the utterance): exp2(-inf - 0) = 0 below
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);
        m_run = m_new;
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sc[r] = __builtin_amdgcn_exp2f(sc[r] - m_use); psum += sc[r]; }
        l_run = fmaf(l_run, alpha, psum);                        // the two kh halves keep partial sums (same alpha): added once at the end
        if (__ballot(alpha != 1.0f)) {                           // the running maximum of some query moved (rare after the first tiles)
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[t][r] *= alpha;
        }
        // probabilities as B fragments: accumulator registers 8 s .. 8 s + 7 ARE k positions 8 kh .. 8 kh + 7 of k-step s (keys 4 kh + (e & 3) + 8 (e >> 2) + 16 s)
        f16x8 pbh[2], pbl[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            uint32_t hh[4], ll[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) split2s(sc[8 * s + 2 * e] * SP_, sc[8 * s + 2 * e + 1] * SP_, hh[e], ll[e]);
            pbh[s] = as_f16x8(make_uint4(hh[0], hh[1], hh[2], hh[3])); pbl[s] = as_f16x8(make_uint4(ll[0], ll[1], ll[2], ll[3]));
        }
        if (L::VALIAS) { lds_barrier(); put_v(); fetch_ke(jn, 64, true); lds_barrier(); }      // every wave is done with the K tile and the lower band half
        // ---- O^T += V^T P^T over this wave's 32 keys (one accumulator per column tile at the scale 2^18; kind-major over the tiles: consecutive MFMAs hit
        //      different accumulators)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            f16x8 vh[NT], vl[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const char* vr = sV + (32 * t + lr) * VROW + (32 * wk + 16 * s + 4 * kh) * 2;
                const uint2 a0 = *reinterpret_cast<const uint2*>(vr), a1 = *reinterpret_cast<const uint2*>(vr + 16);
                const uint2 b0 = *reinterpret_cast<const uint2*>(vr + 32 * NT * VROW), b1 = *reinterpret_cast<const uint2*>(vr + 32 * NT * VROW + 16);
                vh[t] = as_f16x8(make_uint4(a0.x, a0.y, a1.x, a1.y)); vl[t] = as_f16x8(make_uint4(b0.x, b0.y, b1.x, b1.y));
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) oacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh[t], pbh[s], oacc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NT; ++t) oacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh[t], pbl[s], oacc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NT; ++t) oacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl[t], pbh[s], oacc[t], 0, 0, 0);
        }
        lds_barrier();                                         // the next tile's staging overwrites the K tile, the lower band half and V^T
    }
    // ---- merge the two key halves of every query (wave wk = 1 -> wave wk = 0 through LDS), normalise, un-group (attentions.py:707-712)
    l_run += __shfl_xor(l_run, 32);
    float* mrg = reinterpret_cast<float*>(sm) + (size_t)wq * (16 * NT + 2) * 64;
    if (wk == 1) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) mrg[(16 * t + r) * 64 + lane] = oacc[t][r];
        mrg[16 * NT * 64 + lane] = m_run;
        mrg[(16 * NT + 1) * 64 + lane] = l_run;
    }
    __syncthreads();
    if (wk == 0 && iq < Tg) {
        const float m1 = mrg[16 * NT * 64 + lane], l1 = mrg[(16 * NT + 1) * 64 + lane];
        const float mm = fmaxf(m_run, m1);
        const float a0 = __builtin_amdgcn_exp2f(m_run - mm), a1 = __builtin_amdgcn_exp2f(m1 - mm);      // mm is finite: this wave saw key 0 .. 31 of some tile
        const float inv = sx_rcp(fmaf(l_run, a0, l1 * a1)) * (1.0f / (SP_ * SV_));      // the scale of the P V accumulators leaves with the normalisation
        float* orow = p.out + row0 * D + gd * iq + hb;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int x = 32 * t + 8 * rq + 4 * kh;
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = fmaf(oacc[t][4 * rq + e], a0, mrg[(16 * t + 4 * rq + e) * 64 + lane] * a1) * inv;
                if (x + 3 < d) {
                    typedef float f32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
                    *reinterpret_cast<f32x4_a4*>(orow + x) = f32x4_a4{o[0], o[1], o[2], o[3]};
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (x + e < d) orow[x + e] = o[e];
                }
            }
    }
}

// ---- operand images of the attention kernel
// K image  [grouped row][head][hi PK | lo PK] fp16: head span of the grouped row, zero behind the span (columns >= d) and for natural rows that are chunk
//          padding (attentions.py:107-138, 671: zeros AFTER the projection); grouped rows of utterance b start at off[b] / G (ragged) or b Tg.
// V image  [utterance][head][hi | lo][32 NT columns][vpitch keys] fp16: TRANSPOSED (the P V product contracts over keys), zero for pad rows / columns / keys
//          behind the utterance inside its last 64-key tile.
// One workgroup = 64 grouped rows of one (utterance, head).
template <int KS, int NT>
__global__ __launch_bounds__(256) void sxf_pack_kv_kernel(const SxfAttnParams p) {
    constexpr int PK = 16 * KS, CPR = PK / 4, VX = 32 * NT;
    __shared__ __attribute__((aligned(16))) char sT[2 * VX * VROW];
    const int tid = threadIdx.x;
    const int b = blockIdx.y / p.H, h = blockIdx.y % p.H, d = p.d, G = p.G, D = p.D;
    long long row0; int nfr, Tg;
    if (p.off) { row0 = p.off[b]; nfr = p.lens[b]; Tg = (nfr + G - 1) / G; }
    else { row0 = (long long)b * p.Tp; nfr = p.T; Tg = p.Tg; }
    const int j0 = blockIdx.x * 64;
    if (j0 >= Tg) return;
    const size_t hb = (size_t)h * d, gd = (size_t)G * D;
    const float* kbase = p.k + row0 * D + hb;
    const float* vbase = p.v + row0 * D + hb;
    auto dspan = [&](int i) { const long long xl = (long long)(nfr - G * i) * D - (long long)hb; return (int)(xl < 0 ? 0 : (xl > d ? d : xl)); };
    const long long rowg0 = p.off ? row0 / G : (long long)b * p.Tg;
    uint16_t* kout = const_cast<uint16_t*>(p.kp) + ((size_t)(rowg0 + j0) * p.H + h) * 2 * PK;
#pragma unroll
    for (int it = 0; it < KS; ++it) {
        const int c = tid + 256 * it, r = c / CPR, x = (c - r * CPR) * 4, j = j0 + r;
        if (j >= Tg) continue;
        const float4 v = ld_span4(kbase + gd * j, x, dspan(j));
        uint32_t h0, l0, h1, l1;
        split2s(v.x * SQK, v.y * SQK, h0, l0); split2s(v.z * SQK, v.w * SQK, h1, l1);
        uint16_t* o = kout + (size_t)r * p.H * 2 * PK + x;
        *reinterpret_cast<uint2*>(o) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(o + PK) = make_uint2(l0, l1);
    }
    // V: consecutive lanes <-> consecutive KEYS of one column quad (the transposing 2-byte stores of a wave fall on 32 consecutive dwords)
#pragma unroll
    for (int it = 0; it < 2 * NT; ++it) {
        const int c = tid + 256 * it, r = c & 63, x = (c >> 6) * 4, j = j0 + r;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (j < Tg) v = ld_span4(vbase + gd * j, x, dspan(j));
        const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; e += 2) {
            uint32_t hh, ll;
            split2s(vv[e] * SV_, vv[e + 1] * SV_, hh, ll);
            *reinterpret_cast<uint16_t*>(sT + (x + e) * VROW + r * 2) = (uint16_t)(hh & 0xFFFFu);
            *reinterpret_cast<uint16_t*>(sT + (x + e + 1) * VROW + r * 2) = (uint16_t)(hh >> 16);
            *reinterpret_cast<uint16_t*>(sT + VX * VROW + (x + e) * VROW + r * 2) = (uint16_t)(ll & 0xFFFFu);
            *reinterpret_cast<uint16_t*>(sT + VX * VROW + (x + e + 1) * VROW + r * 2) = (uint16_t)(ll >> 16);
        }
    }
    __syncthreads();
    uint16_t* vout = const_cast<uint16_t*>(p.vp) + ((size_t)b * p.H + h) * 2 * VX * (size_t)p.vpitch + j0;
#pragma unroll
    for (int it = 0; it < 2 * NT; ++it) {
        const int c = tid + 256 * it, xr = c >> 3, ch = c & 7;
        *reinterpret_cast<uint4*>(vout + (size_t)xr * p.vpitch + ch * 8) = *reinterpret_cast<const uint4*>(sT + xr * VROW + ch * 16);
    }
}

// E image  [grouped relative row][head][hi PK | lo PK] fp16 of E = pos_layer(R) (fp32 [rows][D]); column d carries the positional bias
//   cb[r][h] = sum_x (v - u)[(h d + x) mod D] E[r][h d + x] = what (Q + v) E^T has over (Q + u) E^T - one number per (relative position, head) - against the
//   constant 1 in column d of the query fragments.  Input independent.  One wave per (row, head).
__global__ __launch_bounds__(256) void sxf_pack_e_kernel(const float* __restrict__ e, const float* __restrict__ u, const float* __restrict__ vb, int erows, int H,
                                                         int G, int D, int d, int PK, uint16_t* __restrict__ ep) {
    const int wid = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wid >= erows * H) return;
    const int r = wid / H, h = wid % H;
    const float* er = e + (size_t)r * G * D + (size_t)h * d;
    float acc = 0.f;
    for (int x = lane; x < d; x += 64) {
        const int n = (int)(((size_t)h * d + x) % D);
        acc = fmaf(vb[n] - u[n], er[x], acc);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    uint16_t* out = ep + (size_t)wid * 2 * PK;
    for (int x = 2 * lane; x < PK; x += 128) {
        const float a = x < d ? er[x] : (x == d ? acc : 0.f), c = x + 1 < d ? er[x + 1] : (x + 1 == d ? acc : 0.f);
        uint32_t hh, ll;
        split2s(a * SQK, c * SQK, hh, ll);
        *reinterpret_cast<uint32_t*>(out + x) = hh;
        *reinterpret_cast<uint32_t*>(out + PK + x) = ll;
    }
}

// depthwise conv (k taps, "same" or causal zero pre-padding at the UTTERANCE's own ends, stride S) + folded BatchNorm + Swish, fp32 (modules.py:516-518;
// layers.py:97-101): 8 output frames of one channel per thread from a register window, taps in ascending order (the order of exact.hip's kernels).
// Utterance b: input rows in_off[b] .. + in_len[b], output rows out_off[b] .. + out_len[b]; output rows up to out_off[b + 1] (group padding) are zero filled.
// Rectangular batches: offsets b T / b To, lengths T / To (pad frames are live: SURVEY.md 8a).
template <int KSZ, int S>
__global__ __launch_bounds__(256) void sxf_dwconv_kernel(const float* __restrict__ g, int B, int T, int To, int C, const float* __restrict__ w_kc,
                                                         const float* __restrict__ bias, float* __restrict__ out, RaggedConv rc, int ragged, int causal, int ntile) {
    constexpr int TO = 8, W = S * (TO - 1) + KSZ;
    const int pre = causal ? KSZ - 1 : (KSZ - 1) / 2;
    const long long total = (long long)B * ntile * C;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int c = (int)(idx % C);
        const long long q = idx / C;
        const int tile = (int)(q % ntile), b = (int)(q / ntile);
        long long ibase, obase; int tin, tout, tcap;
        if (ragged) { ibase = rc.in_off[b]; obase = rc.out_off[b]; tin = rc.in_len[b]; tout = rc.out_len[b]; tcap = rc.out_off[b + 1] - rc.out_off[b]; }
        else { ibase = (long long)b * T; obase = (long long)b * To; tin = T; tout = To; tcap = To; }
        const int to0 = tile * TO;
        if (to0 >= tcap) continue;
        const int t0 = S * to0 - pre;
        const float* gp = g + ibase * C + c;
        float x[W], w[KSZ];
#pragma unroll
        for (int i = 0; i < W; ++i) {
            const int t = t0 + i;
            const bool ok = t >= 0 && t < tin;
            x[i] = ok ? gp[(size_t)(ok ? t : 0) * C] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < KSZ; ++j) w[j] = w_kc[(size_t)j * C + c];
        const float bz = bias[c];
#pragma unroll
        for (int o = 0; o < TO; ++o) {
            if (to0 + o >= tcap) break;
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < KSZ; ++j) acc = fmaf(w[j], x[S * o + j], acc);
            const float y = acc + bz;
            out[(obase + to0 + o) * C + c] = to0 + o < tout ? y * sx_rcp(1.0f + sx_expf(fminf(-y, 87.0f))) : 0.f;
        }
    }
}

// out[out_off[b] + t][:] = x[in_off[b] + stride t][:] for t < out_len[b], zero rows up to out_off[b + 1]  (blocks.py:106-110: the 1 x 1 strided conv_res reads frames 0, s, 2s, ..)
__global__ __launch_bounds__(256) void sxf_decimate_kernel(const float* __restrict__ x, int D4, int stride, RaggedConv rc, float* __restrict__ out) {
    const long long total = (long long)rc.out_rows * D4;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int c = (int)(idx % D4);
        const int m = (int)(idx / D4);
        const int b = ragged_find(rc.out_off, rc.n, m);
        const int t = m - rc.out_off[b];
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t < rc.out_len[b]) v = reinterpret_cast<const float4*>(x)[((size_t)rc.in_off[b] + (size_t)stride * t) * D4 + c];
        reinterpret_cast<float4*>(out)[idx] = v;
    }
}

// a[m][n] * sigmoid(a[m][N + n]) on fp32 rows (GLU over channels, activations.py:37-39) with the accurate transcendental forms of the split kernels
__global__ __launch_bounds__(256) void sxf_glu_kernel(const float* __restrict__ in, long long M, int N4, float* __restrict__ out) {
    const long long total = M * N4;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const long long m = idx / N4; const int n = (int)(idx % N4);
        const float4 a = reinterpret_cast<const float4*>(in)[m * 2 * N4 + n], gt = reinterpret_cast<const float4*>(in)[m * 2 * N4 + N4 + n];
        float4 o;
        o.x = a.x * sx_rcp(1.0f + sx_expf(fminf(-gt.x, 87.0f))); o.y = a.y * sx_rcp(1.0f + sx_expf(fminf(-gt.y, 87.0f)));
        o.z = a.z * sx_rcp(1.0f + sx_expf(fminf(-gt.z, 87.0f))); o.w = a.w * sx_rcp(1.0f + sx_expf(fminf(-gt.w, 87.0f)));
        reinterpret_cast<float4*>(out)[idx] = o;
    }
}

template <int KS, int NT>
int launch_attn(const SxfAttnParams& p, int what, hipStream_t s) {
    using L = AttnLds<KS, NT>;
    static_assert(L::BYTES <= 160 * 1024, "LDS image of the fused split attention");
    static_assert((16 * NT + 2) * 64 * 4 * 2 <= 3 * L::HALF, "merge buffer");
    if (what == 1) {                                            // the K / V images of this block
        hipLaunchKernelGGL((sxf_pack_kv_kernel<KS, NT>), dim3((p.Tg + 63) / 64, p.B * p.H), dim3(256), 0, s, p);
        return hipGetLastError() == hipSuccess ? 0 : -1;
    }
    static LdsAttr attr;
    ensure_dynamic_lds(reinterpret_cast<const void*>(&sxf_attention_kernel<KS, NT>), L::BYTES, attr);
    hipLaunchKernelGGL((sxf_attention_kernel<KS, NT>), dim3((p.Tg + 63) / 64, p.B * p.H), dim3(256), L::BYTES, s, p);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int dispatch_attn(const SxfAttnParams& p, int what, hipStream_t s) {
    const int ks = sxf_attention_pk(p.d) / 16;
    switch (ks) {
        case 1: return launch_attn<1, 1>(p, what, s);
        case 2: return launch_attn<2, 1>(p, what, s);
        case 3: return p.d <= 32 ? launch_attn<3, 1>(p, what, s) : launch_attn<3, 2>(p, what, s);
        case 4: return launch_attn<4, 2>(p, what, s);
        case 5: return p.d <= 64 ? launch_attn<5, 2>(p, what, s) : launch_attn<5, 3>(p, what, s);
        case 6: return launch_attn<6, 3>(p, what, s);
        case 7: return p.d <= 96 ? launch_attn<7, 3>(p, what, s) : launch_attn<7, 4>(p, what, s);
        case 8: return launch_attn<8, 4>(p, what, s);
        case 9: return p.d <= 128 ? launch_attn<9, 4>(p, what, s) : launch_attn<9, 5>(p, what, s);
        default: return launch_attn<10, 5>(p, what, s);
    }
}

template <int KSZ, int S>
int launch_dw(const float* g, int B, int T, int To, int C, const float* w, const float* bias, float* out, const RaggedConv* rc, int causal, int tcap_max, hipStream_t s) {
    const int ntile = (tcap_max + 7) / 8;
    const long long total = (long long)B * ntile * C;
    if (total <= 0) return 0;
    const int grid = (int)std::min<long long>((total + 255) / 256, 1 << 20);
    RaggedConv r{};
    if (rc) r = *rc;
    hipLaunchKernelGGL((sxf_dwconv_kernel<KSZ, S>), dim3(grid), dim3(256), 0, s, g, B, T, To, C, w, bias, out, r, rc ? 1 : 0, causal, ntile);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace

bool sxf_attention_supported(int d) { return d >= 1 && d <= 144; }

int sxf_attention_pk(int d) { return (d + 1 + 15) / 16 * 16; }                  // head width + the positional bias column, in 16-wide k-steps
int sxf_attention_vx(int d) { return (d + 31) / 32 * 32; }

int launch_sxf_pack_e(const float* e, const float* u, const float* vb, int erows, int H, int G, int D, int d, uint16_t* ep, hipStream_t s) {
    const long long waves = (long long)erows * H;
    if (waves <= 0) return 0;
    hipLaunchKernelGGL(sxf_pack_e_kernel, dim3((unsigned)((waves * 64 + 255) / 256)), dim3(256), 0, s, e, u, vb, erows, H, G, D, d, sxf_attention_pk(d), ep);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_sxf_pack_kv(const SxfAttnParams& p, hipStream_t s) {
    if (p.B <= 0 || p.Tg <= 0) return 0;
    if (!sxf_attention_supported(p.d) || (long long)p.B * p.H > 65535 || !p.kp || !p.vp || p.vpitch % 64 || p.vpitch < p.Tg) return -2;
    return dispatch_attn(p, 1, s);
}

int launch_sxf_attention(const SxfAttnParams& p, hipStream_t s) {
    if (p.B <= 0 || p.Tg <= 0) return 0;
    if (!sxf_attention_supported(p.d) || (long long)p.B * p.H > 65535 || !p.kp || !p.vp || !p.ep || p.vpitch % 64) return -2;
    return dispatch_attn(p, 0, s);
}

int launch_sxf_dwconv(const float* g, int B, int T, int To, int C, const float* w_kc, const float* bias, int ks, int stride, float* out, hipStream_t s,
                      const RaggedConv* rc, int causal, int tcap_max) {
    if (!rc) tcap_max = To;
    if (ks == 15 && stride == 1) return launch_dw<15, 1>(g, B, T, To, C, w_kc, bias, out, rc, causal, tcap_max, s);
    if (ks == 15 && stride == 2) return launch_dw<15, 2>(g, B, T, To, C, w_kc, bias, out, rc, causal, tcap_max, s);
    if (ks == 31 && stride == 1) return launch_dw<31, 1>(g, B, T, To, C, w_kc, bias, out, rc, causal, tcap_max, s);
    if (ks == 31 && stride == 2) return launch_dw<31, 2>(g, B, T, To, C, w_kc, bias, out, rc, causal, tcap_max, s);
    if (ks == 7 && stride == 1) return launch_dw<7, 1>(g, B, T, To, C, w_kc, bias, out, rc, causal, tcap_max, s);
    if (ks == 7 && stride == 2) return launch_dw<7, 2>(g, B, T, To, C, w_kc, bias, out, rc, causal, tcap_max, s);
    if (ks == 3 && stride == 1) return launch_dw<3, 1>(g, B, T, To, C, w_kc, bias, out, rc, causal, tcap_max, s);
    if (ks == 3 && stride == 2) return launch_dw<3, 2>(g, B, T, To, C, w_kc, bias, out, rc, causal, tcap_max, s);
    return -2;
}

int launch_sxf_decimate(const float* x, int D, int stride, const RaggedConv& rc, float* out, hipStream_t s) {
    if (D % 4) return -2;
    const long long total = (long long)rc.out_rows * (D / 4);
    if (total <= 0) return 0;
    hipLaunchKernelGGL(sxf_decimate_kernel, dim3((unsigned)std::min<long long>((total + 255) / 256, 1 << 20)), dim3(256), 0, s, x, D / 4, stride, rc, out);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_sxf_glu(const float* in, long long M, int N, float* out, hipStream_t s) {
    if (N % 4) return -2;
    const long long total = M * (N / 4);
    if (total <= 0) return 0;
    hipLaunchKernelGGL(sxf_glu_kernel, dim3((unsigned)std::min<long long>((total + 255) / 256, 1 << 20)), dim3(256), 0, s, in, M, N / 4, out);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
