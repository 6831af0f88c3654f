// Device/runtime vocabulary used by every kernel in this directory.
//
// Product build: hipcc --offload-arch=gfx950 (CDNA4 only; no other backend is supported).
// Test build (tests/emu): the same sources are compiled for the host with -DMTTS_EMU so the
// CPU test-suite can execute kernel and orchestration logic without a GPU (see
// tests/emu/hip_emu.h; that library is never loaded by the product).
#pragma once

#if defined(MTTS_EMU)
#include "hip_emu.h"
#else
#include <hip/hip_runtime.h>
#define MTTS_LAUNCH(kernel, grid, block, stream, ...) \
    hipLaunchKernelGGL(kernel, (grid), (block), 0, (stream), __VA_ARGS__)
#endif

#include <cstdint>

namespace mtts {

constexpr int kWave = 64;  // CDNA wavefront width

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// bf16 storage (the operand planes of the bf16 numerics mode: gemm_bf16.h).  Scalar round-to-nearest-even conversion, the rounding
// v_cvt_pk_bf16_f32 does (the staging passes of gemm_bf16.h use the packed instruction; producers that write one twin value per lane
// use this one).
typedef unsigned short bf16_t;
__device__ __forceinline__ bf16_t f32_to_bf16(float x) {
    unsigned u = __builtin_bit_cast(unsigned, x);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);   // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
__device__ __forceinline__ float bf16_to_f32(bf16_t h) { return __builtin_bit_cast(float, (unsigned)h << 16); }

// Wavefront reductions on the DPP cross-lane paths (no LDS traffic): quad_perm xor-1 / xor-2, row_half_mirror and
// row_mirror leave the sum of each 16-lane row in all of its lanes; four v_readlane + scalar adds combine the rows, so the
// result is wave-uniform.  (__shfl_xor lowers to six dependent ds_bpermute round trips per reduction, which made the row
// kernels latency- instead of HBM-bound.)
#if !defined(MTTS_EMU)
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float lane_bcast(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
#endif

__device__ __forceinline__ float wave_sum(float v) {
#if defined(MTTS_EMU)
    return emu_wave_sum(v);
#else
    v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);  // row_half_mirror
    v += dpp_mov<0x140>(v);  // row_mirror
    return (lane_bcast(v, 0) + lane_bcast(v, 16)) + (lane_bcast(v, 32) + lane_bcast(v, 48));
#endif
}

__device__ __forceinline__ float wave_max(float v) {
#if defined(MTTS_EMU)
    return emu_wave_max(v);
#else
    v = fmaxf(v, dpp_mov<0xB1>(v));
    v = fmaxf(v, dpp_mov<0x4E>(v));
    v = fmaxf(v, dpp_mov<0x141>(v));
    v = fmaxf(v, dpp_mov<0x140>(v));
    return fmaxf(fmaxf(lane_bcast(v, 0), lane_bcast(v, 16)), fmaxf(lane_bcast(v, 32), lane_bcast(v, 48)));
#endif
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// counter-based dropout mask shared by dropout_kernel and the LayerNorm kernels that apply it in passing: keep bits of
// the four elements (row, c .. c+3) of task z = 16-bit fields of splitmix64(seed, z, element id / 4)
__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
struct DropSpec { unsigned seed = 0, thr16 = 0; float scale = 1.f; };  // thr16 == 0: off
__device__ __forceinline__ float4 drop4(const DropSpec& d, int z, int row, int C, int c, float4 v) {
    const unsigned long long base = ((unsigned long long)d.seed << 32) ^ ((unsigned long long)z << 24);
    const unsigned long long h = splitmix64(base + ((unsigned long long)row * (unsigned)C + (unsigned)c) / 4ull);
    return make_float4(((h) & 0xFFFFu) >= d.thr16 ? v.x * d.scale : 0.f, ((h >> 16) & 0xFFFFu) >= d.thr16 ? v.y * d.scale : 0.f,
                       ((h >> 32) & 0xFFFFu) >= d.thr16 ? v.z * d.scale : 0.f, ((h >> 48) & 0xFFFFu) >= d.thr16 ? v.w * d.scale : 0.f);
}


}  // namespace mtts

// cross-workgroup hand-off primitives (split-K rendezvous in gemm.h)
#if defined(MTTS_EMU)
#define MTTS_UNIFORM(x) (x)
#define MTTS_OPAQUE_TID() ((int)threadIdx.x)
#define MTTS_WAIT_VMEM() ((void)0)
#define MTTS_SETPRIO_HIGH() ((void)0)
#define MTTS_FENCE_RELEASE_AGENT() __atomic_thread_fence(__ATOMIC_SEQ_CST)
#define MTTS_FENCE_ACQUIRE_AGENT() __atomic_thread_fence(__ATOMIC_SEQ_CST)
#define MTTS_ATOMIC_INC_AGENT(p) atomicAdd((p), 1)
#define MTTS_ATOMIC_LOAD_AGENT(p) __atomic_load_n((p), __ATOMIC_RELAXED)
#else
#define MTTS_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)   // a wave-uniform int held in a VGPR (an LDS read) -> SGPR
// threadIdx.x as a value the optimiser cannot see through: inside a persistent loop this keeps the per-thread address arithmetic of
// the loop body from being hoisted above the loop (where the three operand forms' worth of it would be live at once)
__device__ __forceinline__ int mtts_opaque_tid() { int t = (int)threadIdx.x; asm volatile("" : "+v"(t)); return t; }
#define MTTS_OPAQUE_TID() mtts_opaque_tid()
#define MTTS_WAIT_VMEM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define MTTS_SETPRIO_HIGH() __builtin_amdgcn_s_setprio(3)   // wavefront issue priority 3 (of 0..3) until the wavefront ends
#define MTTS_FENCE_RELEASE_AGENT() __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent")
#define MTTS_FENCE_ACQUIRE_AGENT() __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent")
#define MTTS_ATOMIC_INC_AGENT(p) __hip_atomic_fetch_add((p), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define MTTS_ATOMIC_LOAD_AGENT(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#endif
