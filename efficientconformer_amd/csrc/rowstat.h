// Shared device helpers of the row-stationary kernels (rsgemm.hip, chain.hip): LDS-DMA weight ring, counted waits,
// coalesced row-tile staging.  See rsgemm.hip for the design notes.
#pragma once
#include "kernels.h"

namespace {

constexpr int CH = 32;   // weight rows per LDS chunk (== hidden units per FFN step)

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;

// One wave-wide LDS-DMA: lane i's 16 bytes at g land at lds_wave_base + 16*i (lds_wave_base wave-uniform).
// Written as inline assembly on purpose.  Through __builtin_amdgcn_global_load_lds the compiler knows the instruction writes LDS
// and, unable to prove which bytes, makes the next LDS read it considers aliasing wait for vmcnt(0) - i.e. for the prefetch that
// was just issued, three chunks ahead of its use (found in the ISA of every chain kernel: a drained DMA queue per chunk).  The
// ring protocol already orders every DMA against its readers with counted s_waitcnt vmcnt(N) + s_barrier (wait_chunks /
// wg_barrier), so the instruction is made opaque; "memory" keeps the compiler from moving LDS or global accesses across it.
__device__ __forceinline__ void glds16(const void* g, char* lds_wave_base) {
    const uint32_t l = (uint32_t)(uintptr_t)(lds_void_t*)lds_wave_base;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(l), "v"(g) : "memory", "m0");
}
// same with a wave-uniform base and a per-lane 32-bit byte offset (saddr form: the offset register can live for the whole kernel)
__device__ __forceinline__ void glds16(const char* base, uint32_t off, char* lds_wave_base) {
    const uint32_t l = (uint32_t)(uintptr_t)(lds_void_t*)lds_wave_base;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(l), "v"(off), "s"(base) : "memory", "m0");
}
// The same instruction WITHOUT the "memory" clobber: for refills placed between the MFMAs of an iteration (chain.hip).  The ring protocol orders
// the DMA against the accesses that matter - it follows the barrier that released its buffer and precedes the next one (volatile asm statements
// keep their order among themselves) - and the LDS reads of the iteration touch other buffers, so the compiler may schedule them across it.
__device__ __forceinline__ void glds16_nc(const char* base, uint32_t off, char* lds_wave_base) {
    const uint32_t l = (uint32_t)(uintptr_t)(lds_void_t*)lds_wave_base;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(l), "v"(off), "s"(base) : "m0");
}
template <int N> __device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt immediate is 6 bits");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// runtime count (wave-uniform, clamped to the 6-bit immediate): used where global stores share the FIFO with the weight DMAs
__device__ __forceinline__ void wait_vmcnt_dyn(int n) {
    n = n < 0 ? 0 : (n > 63 ? 63 : n);
    switch (n) {
#define EC_VM(k) case k: wait_vmcnt<k>(); break;
#define EC_VM8(b) EC_VM(b) EC_VM(b + 1) EC_VM(b + 2) EC_VM(b + 3) EC_VM(b + 4) EC_VM(b + 5) EC_VM(b + 6) EC_VM(b + 7)
        EC_VM8(0) EC_VM8(8) EC_VM8(16) EC_VM8(24) EC_VM8(32) EC_VM8(40) EC_VM8(48) EC_VM8(56)
#undef EC_VM8
#undef EC_VM
    }
}
// wait until at most `allowed` chunks (of PER DMA instructions each, issued by this wave) are still in flight
template <int PER, int MAXC> __device__ __forceinline__ void wait_chunks(int allowed) {
    if (allowed >= MAXC) wait_vmcnt<PER * MAXC>();
    else if (MAXC >= 2 && allowed == 1) wait_vmcnt<PER>();
    else wait_vmcnt<0>();
}
__device__ __forceinline__ void wg_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// DMA shapes of the weight ring.  A kernel computes the per-lane BYTE offsets of its wave's DMA instructions once, keeps them in
// registers, and per chunk only moves the wave-uniform base it hands to glds16 (a division, a modulo and a 64-bit multiply-add
// per instruction and chunk otherwise: ~80 cycles per DMA in the EFFCONF_FFN_PHASES profile).
//  * [32 rows][P pieces of 16 B] weight chunk (row r at base + r*ld elements).  LDS image: piece pc of row r sits at slot
//    r*P + (pc + r) % P.  Wave-instruction i (32*P/64 per chunk) covers slots [64i, 64i+64) -> LDS bytes [1024 i, 1024 i + 1024).
template <int P> __device__ __forceinline__ uint32_t dma_rows32_off(int ld, int i, int lane) {
    const int L = 64 * i + lane;
    const int r = L / P, q = L - r * P;
    int pc = q - (r % P);
    pc += pc < 0 ? P : 0;
    return (uint32_t)(r * ld + pc * 8) * 2u;
}
//  * FFN second weight chunk: [R rows][4 pieces] (32 hidden units), piece pc of row n at slot n*4 + ((pc + (n>>2)) & 3)
__device__ __forceinline__ uint32_t dma_w2_off(int ld, int j, int lane) {
    const int L = 64 * j + lane;
    const int n = L >> 2, q = L & 3;
    const int pc = (q - (n >> 2)) & 3;
    return (uint32_t)(n * ld + pc * 8) * 2u;
}


// ---- coalesced row-tile staging ----------------------------------------------------------------------------------------
// Row-stationary kernels want "lane = row": a lane's 16-byte global access then touches its own cache line and one wave
// instruction covers 32-64 partial lines (the TA processes them one by one: the load/store phases, not the MFMA loop,
// dominated these kernels: EFFCONF_FFN_PHASES profile, profiles/r1_09_*).  Instead a wave moves its 32-row tile in
// 256-byte-per-row windows with lane = (row 4i + lane/16, piece lane%16) - four 256-byte row segments per instruction -
// and transposes through a private LDS region (row pitch 272 B = 17 pieces: conflict-free for both access patterns).
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));   // native vector: HIP's uint4 struct copies become memcpys that stay in scratch
constexpr int STG_ROW = 272;
constexpr int STG_BYTES = 32 * STG_ROW;

__device__ __forceinline__ void wave_sync() {          // LDS operations of one wave execute in order; this pins the compiler
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// window = bytes [wbyte, wbyte + 256) of rows m_base .. m_base + 31 (row pitch `pitch` bytes, `row_bytes` readable bytes per
// row, a multiple of 16); out-of-range rows / pieces are clamped to valid addresses (their contents are never used)
template <int OFF, int N>
__device__ __forceinline__ void stage_load(const char* base, size_t pitch, int row_bytes, int m_base, int M, int wbyte, int lane,
                                           u32x4 (&v)[N]) {
    int cb = wbyte + 16 * (lane & 15);
    cb = cb < row_bytes - 16 ? cb : row_bytes - 16;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int m = m_base + 4 * i + (lane >> 4);
        v[OFF + i] = *reinterpret_cast<const u32x4*>(base + (size_t)(m < M ? m : M - 1) * pitch + cb);
    }
}
template <int OFF, int N>
__device__ __forceinline__ void stage_put(char* stg, int lane, const u32x4 (&v)[N]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) *reinterpret_cast<u32x4*>(stg + (4 * i + (lane >> 4)) * STG_ROW + 16 * (lane & 15)) = v[OFF + i];
}
__device__ __forceinline__ void stage_store(const char* stg, char* base, size_t pitch, int row_bytes, int m_base, int M, int wbyte, int lane) {
    const int cb = wbyte + 16 * (lane & 15);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int m = m_base + 4 * i + (lane >> 4);
        const u32x4 v = *reinterpret_cast<const u32x4*>(stg + (4 * i + (lane >> 4)) * STG_ROW + 16 * (lane & 15));
        if (m < M && cb < row_bytes) *reinterpret_cast<u32x4*>(base + (size_t)m * pitch + cb) = v;
    }
}
// this lane's fp32 row fragments (columns 16s + 8*half + 0..7 of row lr) for all s < KS, fetched through the staging region
template <int KS, int W>
__device__ __forceinline__ void take_window_f32(const char* stg, int lr, int half, float4 (&ra)[KS], float4 (&rb)[KS]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        constexpr int dummy = 0; (void)dummy;
        if (4 * W + j < KS) {
            const char* src = stg + lr * STG_ROW + (16 * j + 8 * half) * 4;
            ra[4 * W + j] = *reinterpret_cast<const float4*>(src);
            rb[4 * W + j] = *reinterpret_cast<const float4*>(src + 16);
        }
    }
}
template <int KS, int W0>
__device__ __forceinline__ void fetch_pair_f32(const char* X, size_t pitch, int row_bytes, int m_base, int M, char* stg, int lane,
                                               float4 (&ra)[KS], float4 (&rb)[KS]) {
    constexpr int NWIN = (KS + 3) / 4;
    if constexpr (W0 < NWIN) {
        u32x4 v[16] = {};
        stage_load<0>(X, pitch, row_bytes, m_base, M, 256 * W0, lane, v);
        if constexpr (W0 + 1 < NWIN) stage_load<8>(X, pitch, row_bytes, m_base, M, 256 * (W0 + 1), lane, v);
        wave_sync();
        stage_put<0>(stg, lane, v);
        wave_sync();
        take_window_f32<KS, W0>(stg, lane & 31, lane >> 5, ra, rb);
        if constexpr (W0 + 1 < NWIN) {
            wave_sync();
            stage_put<8>(stg, lane, v);
            wave_sync();
            take_window_f32<KS, W0 + 1>(stg, lane & 31, lane >> 5, ra, rb);
        }
        fetch_pair_f32<KS, W0 + 2>(X, pitch, row_bytes, m_base, M, stg, lane, ra, rb);
    }
}
// this lane's fp32 row fragments (columns 16s + 8*half + 0..7 of row lr) for all s < KS, fetched through the staging region
template <int KS>
__device__ __forceinline__ void fetch_row_f32(const float* X, int ldx, int D, int m_base, int M, char* stg, int lane,
                                              float4 (&ra)[KS], float4 (&rb)[KS]) {
    fetch_pair_f32<KS, 0>(reinterpret_cast<const char*>(X), (size_t)ldx * 4, D * 4, m_base, M, stg, lane, ra, rb);
}
// same for a bf16 row-major operand: raw 16-byte fragments (8 bf16 at column 16s + 8*half)
template <int KS>
__device__ __forceinline__ void fetch_row_bf16(const bf16_t* A, int lda, int m_base, int M, char* stg, int lane, uint4 (&raw)[KS]) {
    constexpr int NWIN = (KS + 7) / 8;                    // 128 columns per window
    const int lr = lane & 31, half = lane >> 5;
#pragma unroll
    for (int w = 0; w < NWIN; ++w) {
        u32x4 v[8] = {};
        stage_load<0>(reinterpret_cast<const char*>(A), (size_t)lda * 2, lda * 2, m_base, M, 256 * w, lane, v);
        wave_sync();
        stage_put<0>(stg, lane, v);
        wave_sync();
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int s = 8 * w + j;
            if (s < KS) { const u32x4 t = *reinterpret_cast<const u32x4*>(stg + lr * STG_ROW + (16 * j + 8 * half) * 2); raw[s] = make_uint4(t.x, t.y, t.z, t.w); }
        }
    }
}


struct FastDiv32 {   // exact for n * d < 2^32
    uint32_t mul, d;
    __host__ __device__ FastDiv32() : mul(0), d(1) {}
    __host__ explicit FastDiv32(uint32_t dd) : mul(dd > 1 ? (uint32_t)((1ull << 32) / dd + 1) : 0), d(dd) {}
    __device__ __forceinline__ uint32_t div(uint32_t n) const { return d == 1 ? n : __umulhi(n, mul); }
};

}  // namespace
