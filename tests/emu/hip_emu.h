// TEST INFRASTRUCTURE ONLY — a minimal SIMT emulator so that the *same* kernel and host
// orchestration sources under meta_tts_amd/csrc can be executed on CPU cores by the
// `-m "not gpu"` tests (index math, layouts, masks, backward formulas, MAML bookkeeping).
// It is never built into, loaded by, or selected from the product library (libmtts.so);
// tests load tests/emu/libmtts_emu.so explicitly.  Threads of a block run as ucontext fibers;
// blocks are distributed over OS threads.  MFMA itself is not emulated instruction-by-
// instruction: csrc/gemm.h's mma_chunk() has an MTTS_EMU arm that evaluates the tile product
// straight from the LDS images (so the operand->lane mapping is only validated on the GPU).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct emu_uint3 { unsigned x, y, z; };
extern thread_local emu_uint3 threadIdx, blockIdx;
extern thread_local dim3 blockDim, gridDim;

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static thread_local
#define __launch_bounds__(...)

struct alignas(16) float4 { float x, y, z, w; };
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
struct alignas(8) float2 { float x, y; };
static inline float2 make_float2(float x, float y) { return float2{x, y}; }

void emu_syncthreads();
#define __syncthreads() emu_syncthreads()
#define __threadfence() __atomic_thread_fence(__ATOMIC_SEQ_CST)
float emu_wave_sum(float v);
float emu_wave_max(float v);
float emu_shfl(float v, int src_lane);
float atomicAdd(float* p, float v);
int atomicAdd(int* p, int v);
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }

// ---- host runtime shims ("device" memory is host memory) --------------------------------
typedef int hipError_t;
typedef void* hipStream_t;
typedef void* hipEvent_t;
enum { hipSuccess = 0 };
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
hipError_t hipMalloc(void** p, size_t n);
hipError_t hipFree(void* p);
static inline hipError_t hipHostMalloc(void** p, size_t n) { return hipMalloc(p, n); }
static inline hipError_t hipHostFree(void* p) { return hipFree(p); }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return 0; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return 0; }
static inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return 0; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return 0; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
enum { hipStreamNonBlocking = 1 };
static inline hipError_t hipStreamCreate(hipStream_t* s) { *s = (hipStream_t)1; return 0; }   // launches execute at the call: one in-order "stream"
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = (hipStream_t)1; return 0; }
static inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { *s = (hipStream_t)1; return 0; }
static inline hipError_t hipDeviceGetStreamPriorityRange(int* lo, int* hi) { *lo = 0; *hi = 0; return 0; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return 0; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return 0; }
static inline hipError_t hipDeviceSynchronize() { return 0; }
static inline hipError_t hipGetLastError() { return 0; }
static inline hipError_t hipSetDevice(int) { return 0; }
static inline const char* hipGetErrorString(hipError_t) { return "emu"; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return 0; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return 0; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return 0; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return 0; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return 0; }

void emu_launch(dim3 grid, dim3 block, const std::function<void()>& body);
#define MTTS_LAUNCH(kernel, grid, block, stream, ...) \
    emu_launch((grid), (block), [=]() { kernel(__VA_ARGS__); })
