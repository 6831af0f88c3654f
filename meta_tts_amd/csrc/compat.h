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

__device__ __forceinline__ float wave_sum(float v) {
#if defined(MTTS_EMU)
    return emu_wave_sum(v);
#else
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
#endif
}

__device__ __forceinline__ float wave_max(float v) {
#if defined(MTTS_EMU)
    return emu_wave_max(v);
#else
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
#endif
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }

}  // namespace mtts
