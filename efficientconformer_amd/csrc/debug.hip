// Diagnostics (not on the product path): synthetic "neighbour" kernels that load ONE hardware resource of a compute unit each.
// tools/mel_repro.py runs them on one HIP stream next to mel_kernel variants on another to name what mel_kernel is sensitive to
// (HISTORY.md section 5; profiles/r2_mel_repro.txt).
#include "kernels.h"

namespace {

template <int KIND>
__global__ __launch_bounds__(256) void neighbour_kernel(float* __restrict__ buf, size_t n, int iters, int lds_bytes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    float acc = (float)tid * 1e-3f;
    if constexpr (KIND == 0) {            // LDS hammer: 16-byte writes and reads all over the workgroup's own LDS, no barriers
        uint4* l = reinterpret_cast<uint4*>(smem);
        const int nq = lds_bytes / 16;
        for (int it = 0; it < iters; ++it)
            for (int i = tid; i < nq; i += 256) {
                l[i] = make_uint4(i, it, tid, 0x5A5A5A5Au);
                const uint4 r = l[(i * 7 + 3) % nq];
                acc += (float)(r.x & 1);
            }
    } else if constexpr (KIND == 1) {     // VALU + transcendental unit only
        for (int it = 0; it < iters * 64; ++it) acc = acc * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-acc)) + 0.25f;
    } else if constexpr (KIND == 2) {     // global loads only (streams through `buf`)
        const size_t stride = (size_t)gridDim.x * 256;
        for (int it = 0; it < iters; ++it)
            for (size_t i = (size_t)blockIdx.x * 256 + tid; i < n; i += stride * 64) acc += buf[i];
    } else if constexpr (KIND == 3) {     // global stores only
        const size_t stride = (size_t)gridDim.x * 256;
        for (int it = 0; it < iters; ++it)
            for (size_t i = (size_t)blockIdx.x * 256 + tid; i < n; i += stride * 64) buf[i] = acc + (float)it;
    } else if constexpr (KIND == 4) {     // MFMA only
        f32x16 c;
#pragma unroll
        for (int r = 0; r < 16; ++r) c[r] = 0.f;
        bf16x8 a, b;
#pragma unroll
        for (int r = 0; r < 8; ++r) { a[r] = (__bf16)(0.001f * (tid + r)); b[r] = (__bf16)(0.002f * (tid - r)); }
        for (int it = 0; it < iters * 16; ++it) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
        acc += c[0] + c[15];
    } else if constexpr (KIND == 5) {     // LDS publish + workgroup barrier loop (the tiled-GEMM shape: 16-byte writes, barrier, 16-byte reads)
        uint4* l = reinterpret_cast<uint4*>(smem);
        const int nq = lds_bytes / 16;
        for (int it = 0; it < iters; ++it) {
            for (int i = tid; i < nq; i += 256) l[i] = make_uint4(i, it, tid, 0xA5A5A5A5u);
            __syncthreads();
            for (int i = tid; i < nq; i += 256) acc += (float)(l[(i * 5 + 1) % nq].y & 1);
            __syncthreads();
        }
    } else if constexpr (KIND == 6) {     // 4-byte LDS writes / reads (no wide accesses), no barriers
        unsigned int* l = reinterpret_cast<unsigned int*>(smem);
        const int nq = lds_bytes / 4;
        for (int it = 0; it < iters; ++it)
            for (int i = tid; i < nq; i += 256) { l[i] = i ^ it; acc += (float)(l[(i * 7 + 3) % nq] & 1); }
    }
    else if constexpr (KIND == 7) {     // fp32 MFMA only
        f32x16 c;
#pragma unroll
        for (int r = 0; r < 16; ++r) c[r] = 0.f;
        const float a = 0.001f * tid, b = 0.002f * (tid - 7);
        for (int it = 0; it < iters * 16; ++it) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
        acc += c[0] + c[15];
    } else if constexpr (KIND == 8) {     // packed-fp32 VALU hammer: 8 independent v_pk_fma_f32 chains per lane, nothing else
        typedef float f2 __attribute__((ext_vector_type(2)));
        f2 x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = f2{0.1f * j + 1e-3f * tid, 0.2f * j};
        const f2 a = f2{0.99f, 0.98f}, b = f2{0.01f, 0.02f};
        for (int it = 0; it < iters * 32; ++it) {
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[j]) : "v"(a), "v"(b));
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += x[j].x + x[j].y;
    } else if constexpr (KIND == 9) {     // 16x16x32 bf16 MFMA only
        f32x4 c = {0.f, 0.f, 0.f, 0.f};
        bf16x8 a, b;
#pragma unroll
        for (int r = 0; r < 8; ++r) { a[r] = (__bf16)(0.001f * (tid + r)); b[r] = (__bf16)(0.002f * (tid - r)); }
        for (int it = 0; it < iters * 32; ++it) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
        acc += c[0] + c[3];
    } else if constexpr (KIND == 10) {    // bf16 MFMA at ~50 % duty: bursts of 8 MFMAs separated by s_sleep
        f32x16 c;
#pragma unroll
        for (int r = 0; r < 16; ++r) c[r] = 0.f;
        bf16x8 a, b;
#pragma unroll
        for (int r = 0; r < 8; ++r) { a[r] = (__bf16)(0.001f * (tid + r)); b[r] = (__bf16)(0.002f * (tid - r)); }
        for (int it = 0; it < iters * 2; ++it) {
#pragma unroll
            for (int j = 0; j < 8; ++j) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
            __builtin_amdgcn_s_sleep(8);
        }
        acc += c[0] + c[15];
    }
    else if constexpr (KIND == 11 || KIND == 12 || KIND == 13 || KIND == 14) {
        // matrix-pipe rate probes (tools/mfma_rate_probe.py): 11 / 12 = 32x32x16 bf16 / f16 on FOUR independent accumulators, 13 / 14 = the same on
        // ONE accumulator (a dependent chain); iters * 16 MFMAs per wave
        typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
        f32x16 c[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) c[j][r] = 0.f;
        bf16x8 a, b;
        f16x8_t ah, bh;
#pragma unroll
        for (int r = 0; r < 8; ++r) { a[r] = (__bf16)(0.001f * (tid + r)); b[r] = (__bf16)(0.002f * (tid - r)); ah[r] = (_Float16)(0.001f * (tid + r)); bh[r] = (_Float16)(0.002f * (tid - r)); }
        for (int it = 0; it < iters * 4; ++it) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = (KIND >= 13) ? 0 : j;
                if constexpr (KIND == 11 || KIND == 13) c[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[k], 0, 0, 0);
                else c[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c[k], 0, 0, 0);
            }
        }
        acc += c[0][0] + c[1][15] + c[2][3] + c[3][7];
    }
    if (acc == 123.456f) buf[0] = acc;    // keeps the work alive
}

// Self-contained victims: every lane runs 8 independent chains of ONE instruction class and stores the results; the host compares
// a run next to an aggressor with a run alone.
template <int KIND>
__global__ __launch_bounds__(256) void victim_kernel(float* __restrict__ out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef float f2 __attribute__((ext_vector_type(2)));
    const int tid = threadIdx.x;
    const size_t gid = (size_t)blockIdx.x * 256 + tid;
    float r[16];
    const float seed = 1e-3f * (float)(gid % 9973);
    if constexpr (KIND == 0) {            // v_fma_f32
        float x[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) x[j] = 0.1f * j + seed;
        const float a = 0.99f, b = 0.013f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 16; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[j]) : "v"(a), "v"(b));
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) r[j] = x[j];
    } else if constexpr (KIND == 1) {     // v_pk_fma_f32
        f2 x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = f2{0.1f * j + seed, 0.1f * j + 0.05f + seed};
        const f2 a = f2{0.99f, 0.985f}, b = f2{0.013f, 0.017f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[j]) : "v"(a), "v"(b));
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) { r[2 * j] = x[j].x; r[2 * j + 1] = x[j].y; }
    } else if constexpr (KIND == 2) {     // v_pk_mul_f32 + v_pk_add_f32
        f2 x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = f2{0.1f * j + seed, 0.1f * j + 0.05f + seed};
        const f2 a = f2{0.99f, 0.985f}, b = f2{0.013f, 0.017f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x[j]) : "v"(a)); asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x[j]) : "v"(b)); }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) { r[2 * j] = x[j].x; r[2 * j + 1] = x[j].y; }
    } else if constexpr (KIND == 3) {     // v_log_f32 / v_exp_f32 (transcendental unit)
        float x[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) x[j] = 1.5f + 0.1f * j + seed;
        for (int it = 0; it < iters / 4; ++it) {
#pragma unroll
            for (int j = 0; j < 16; ++j) { asm volatile("v_log_f32 %0, %0" : "+v"(x[j])); asm volatile("s_nop 1\n\tv_exp_f32 %0, %0\n\ts_nop 1" : "+v"(x[j])); }
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) r[j] = x[j];
    } else if constexpr (KIND == 4) {     // v_mul_f32 + v_add_f32
        float x[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) x[j] = 0.1f * j + seed;
        const float a = 0.99f, b = 0.013f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 16; ++j) { asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[j]) : "v"(a)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[j]) : "v"(b)); }
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) r[j] = x[j];
    } else if constexpr (KIND == 5) {     // integer VALU
        unsigned int x[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) x[j] = (unsigned int)(gid * 16 + j);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 16; ++j) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(x[j]) : "v"(0x10DCDu), "v"(0x3039u));
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) r[j] = __uint_as_float((x[j] & 0x007FFFFFu) | 0x3F800000u);
    } else if constexpr (KIND == 6) {     // wave-local LDS exchange (the mel kernel's hand-off shape), float2, no workgroup barrier
        float2* l = reinterpret_cast<float2*>(smem) + (tid >> 6) * 1152;
        const int lane = tid & 63;
        float2 x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = make_float2(0.1f * j + seed, 0.2f * j + seed);
        for (int it = 0; it < iters / 8; ++it) {
#pragma unroll
            for (int j = 0; j < 8; ++j) l[j * 72 + lane] = x[j];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = l[(lane >> 3) * 72 + j * 8 + (lane & 7)];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) { r[2 * j] = x[j].x; r[2 * j + 1] = x[j].y; }
    }
    else if constexpr (KIND >= 7 && KIND <= 13) {     // packed-fp32 forms with operand modifiers, as mel_kernel's complex arithmetic uses them
        f2 x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = f2{0.1f * j + seed, 0.1f * j + 0.05f + seed};
        const f2 a = f2{0.99f, 0.985f}, b = f2{0.013f, 0.017f}, c = f2{1.0f, 1.0f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if constexpr (KIND == 7) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,1]" : "+v"(x[j]) : "v"(a), "v"(b));
                if constexpr (KIND == 8) { asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x[j]) : "v"(a)); asm volatile("v_pk_add_f32 %0, %1, %0 neg_lo:[0,1] neg_hi:[0,1]" : "+v"(x[j]) : "v"(c)); }
                if constexpr (KIND == 9) { asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x[j]) : "v"(a)); asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[0,1]" : "+v"(x[j]) : "v"(b)); }
                if constexpr (KIND == 10) { asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0]" : "+v"(x[j]) : "v"(a)); asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x[j]) : "v"(b)); }
                if constexpr (KIND == 11) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 neg_lo:[0,0,1] neg_hi:[0,0,1]" : "+v"(x[j]) : "v"(a), "v"(b));
                if constexpr (KIND == 12) { asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[j]) : "v"(a), "v"(b)); asm volatile("v_pk_mov_b32 %0, %0, %0 op_sel:[1,0]" : "+v"(x[j])); }
                if constexpr (KIND == 13) { asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "+v"(x[j]) : "v"(b)); asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[0,0]" : "+v"(x[j]) : "v"(a)); }
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) { r[2 * j] = x[j].x; r[2 * j + 1] = x[j].y; }
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) out[gid * 16 + j] = r[j];
}

template <int KIND>
int launch_n(int blocks, int lds, int iters, float* buf, size_t n, hipStream_t s) {
    static LdsAttr attr;
    ensure_dynamic_lds(reinterpret_cast<const void*>(&neighbour_kernel<KIND>), lds, attr);
    hipLaunchKernelGGL((neighbour_kernel<KIND>), dim3(blocks), dim3(256), lds, s, buf, n, iters, lds);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace

int launch_debug_neighbour(int kind, int blocks, int lds_bytes, int iters, float* buf, size_t n, hipStream_t s) {
    switch (kind) {
        case 0: return launch_n<0>(blocks, lds_bytes, iters, buf, n, s);
        case 1: return launch_n<1>(blocks, lds_bytes, iters, buf, n, s);
        case 2: return launch_n<2>(blocks, lds_bytes, iters, buf, n, s);
        case 3: return launch_n<3>(blocks, lds_bytes, iters, buf, n, s);
        case 4: return launch_n<4>(blocks, lds_bytes, iters, buf, n, s);
        case 5: return launch_n<5>(blocks, lds_bytes, iters, buf, n, s);
        case 6: return launch_n<6>(blocks, lds_bytes, iters, buf, n, s);
        case 7: return launch_n<7>(blocks, lds_bytes, iters, buf, n, s);
        case 8: return launch_n<8>(blocks, lds_bytes, iters, buf, n, s);
        case 9: return launch_n<9>(blocks, lds_bytes, iters, buf, n, s);
        case 10: return launch_n<10>(blocks, lds_bytes, iters, buf, n, s);
        case 11: return launch_n<11>(blocks, lds_bytes, iters, buf, n, s);
        case 12: return launch_n<12>(blocks, lds_bytes, iters, buf, n, s);
        case 13: return launch_n<13>(blocks, lds_bytes, iters, buf, n, s);
        case 14: return launch_n<14>(blocks, lds_bytes, iters, buf, n, s);
    }
    return -2;
}

// out: blocks * 256 * 16 floats
int launch_debug_victim(int kind, int blocks, int iters, float* out, hipStream_t s) {
#define VICTIM_CASE(K) case K: hipLaunchKernelGGL((victim_kernel<K>), dim3(blocks), dim3(256), 4 * 1152 * 8, s, out, iters); break;
    switch (kind) {
        VICTIM_CASE(0) VICTIM_CASE(1) VICTIM_CASE(2) VICTIM_CASE(3) VICTIM_CASE(4) VICTIM_CASE(5) VICTIM_CASE(6) VICTIM_CASE(7)
        VICTIM_CASE(8) VICTIM_CASE(9) VICTIM_CASE(10) VICTIM_CASE(11) VICTIM_CASE(12) VICTIM_CASE(13)
        default: return -2;
    }
#undef VICTIM_CASE
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// One wave that does nothing for `microseconds` (s_memrealtime: the 100 MHz constant counter): stands in for the DURATION of a transfer that
// a single GPU cannot perform (an xGMI all-gather) in tools/overlap_probe.py, without taking CUs or HBM bandwidth from the kernels beside it.
namespace {
__global__ __launch_bounds__(64) void spin_kernel(long long ticks) {
    const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
    while ((long long)__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
}  // namespace

int launch_debug_spin(double microseconds, hipStream_t s) {
    if (microseconds <= 0) return 0;
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, (long long)(microseconds * 100.0));
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// How fast does a CU fill its LDS from L2?  (round 4: the D = 240 chains stream a 32 KiB weight chunk per barrier interval through every workgroup; their
// phase profiles put the cost of a refill at ~30 bytes per cycle and CU - tools/lds_fill_rate_probe.py measures the path alone.)  Every workgroup walks the same
// `window` bytes of `src` (L2 resident after the first pass) in 1 KiB wave-instructions, `kib_per_wave` of them per wave and pass:
//   mode 0: global_load_lds_dwordx4 straight into a 4 x 32 KiB LDS ring (the chains' refill);
//   mode 1: global_load_dwordx4 into registers, 8 in flight per wave, then ds_write_b128.
// out[2 * workgroup] = cycles (s_memtime) of the whole walk as wave 0 saw it, out[2 * workgroup + 1] = bytes the workgroup moved.
namespace {
template <int MODE>
__global__ __launch_bounds__(512) void lds_fill_kernel(const char* __restrict__ src, size_t window, int kib_per_wave, int passes, unsigned long long* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    size_t pos = 0;
    for (int p = 0; p < passes; ++p) {
        for (int i0 = 0; i0 < kib_per_wave; i0 += 8) {
            typedef uint32_t u4 __attribute__((ext_vector_type(4)));
            u4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const size_t o = (pos + ((size_t)(i0 + j) * nw + wave) * 1024) % window;
                char* dst = smem + ((((i0 + j) * nw + wave) * 1024) & (128 * 1024 - 1));
                if constexpr (MODE == 0) {
                    const uint32_t l = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)dst;
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(l), "v"((uint32_t)(lane * 16)), "s"(src + o) : "memory", "m0");
                } else {
                    v[j] = *reinterpret_cast<const u4*>(src + o + lane * 16);
                }
            }
            if constexpr (MODE == 1) {
#pragma unroll
                for (int j = 0; j < 8; ++j) *reinterpret_cast<u4*>(smem + ((((i0 + j) * nw + wave) * 1024) & (128 * 1024 - 1)) + lane * 16) = v[j];
            }
        }
        pos += (size_t)kib_per_wave * nw * 1024;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = (unsigned long long)passes * kib_per_wave * nw * 1024ull; }
}
}  // namespace

int launch_debug_lds_fill(int mode, int blocks, int waves, const char* src, size_t window, int kib_per_wave, int passes, unsigned long long* out, hipStream_t s) {
    if (waves < 1 || waves > 8 || kib_per_wave % 8 || window % 1024 || window < (size_t)kib_per_wave * waves * 1024) return -2;
    static LdsAttr a0, a1;
    if (mode == 0) {
        ensure_dynamic_lds(reinterpret_cast<const void*>(&lds_fill_kernel<0>), 128 * 1024, a0);
        hipLaunchKernelGGL((lds_fill_kernel<0>), dim3(blocks), dim3(waves * 64), 128 * 1024, s, src, window, kib_per_wave, passes, out);
    } else {
        ensure_dynamic_lds(reinterpret_cast<const void*>(&lds_fill_kernel<1>), 128 * 1024, a1);
        hipLaunchKernelGGL((lds_fill_kernel<1>), dim3(blocks), dim3(waves * 64), 128 * 1024, s, src, window, kib_per_wave, passes, out);
    }
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
