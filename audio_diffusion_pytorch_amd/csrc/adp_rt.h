// Runtime shim shared by every kernel source of libadp_hip.so (gfx950 / CDNA4 only).
// The product build is plain HIP.  -DADP_EMULATE (tests only, see tests/emul/) swaps in a
// host-side SIMT emulator header that is NOT part of this package.
#pragma once

#define ADP_OK 0
#define ADP_ERR_SHAPE (-1)
#define ADP_ERR_UNSUPPORTED (-2)
#define ADP_ERR_ALIGN (-3)
#define ADP_ERR_LAUNCH (-4)
#define ADP_ERR_NULL (-5)

#ifdef ADP_EMULATE
#include "adp_rt_emul.h"
#else
#include <hip/hip_runtime.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// exact-f32 matrix core ops (v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32)
__device__ __forceinline__ f32x16 adp_mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 adp_mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// compiler scheduling fence: nothing moves across (keeps prefetch loads ahead of the matrix work that hides them)
__device__ __forceinline__ void adp_sched_fence() { __builtin_amdgcn_sched_barrier(0); }

// v_rcp_f32 (1 ulp) instead of the IEEE division sequence
__device__ __forceinline__ float adp_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

#define ADP_LAUNCH(kern, grid, block, stream, ...)                                     \
  do {                                                                                 \
    adp_rt_note_launch(#kern, __PRETTY_FUNCTION__, (void*)(stream));                   \
    hipLaunchKernelGGL(kern, grid, block, 0, (hipStream_t)(stream), __VA_ARGS__);      \
    adp_rt_launch_done((void*)(stream));                                               \
  } while (0)
#define ADP_LAUNCH_OK() (hipGetLastError() == hipSuccess ? ADP_OK : ADP_ERR_LAUNCH)

// Workgroup barrier for a wave that only CONSUMES LDS data after it and has global stores in flight.
// __syncthreads() carries a workgroup-scope release fence, which on gfx9-family hardware waits (vmcnt) for every
// outstanding global STORE of the wave to be acknowledged by L2 -- microseconds of HBM write latency per loop
// iteration in a streaming kernel whose epilogue stores precede the next tile's barrier (measured in
// conv_stream.hip: 10 of 26 us).  The waves that PRODUCE the LDS data use __syncthreads() (their ds_writes are
// complete before they arrive); this variant is the bare s_barrier plus a compiler-level ordering point.
__device__ __forceinline__ void adp_barrier_consume() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

#endif

// Launch trace (introspection only, see adp_launch_trace / adp_launch_times in adp.h): when tracing is on, every
// ADP_LAUNCH appends "<kernel expression>@<launcher signature with its template arguments>" to a per-thread buffer
// and brackets the launch with a pair of HIP events recorded on the launch stream.  Defined in elementwise.hip.
void adp_rt_note_launch(const char* kern, const char* site, void* stream);
void adp_rt_launch_done(void* stream);

#include <stdint.h>

#define ADP_WAVE 64

__device__ __forceinline__ float adp_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float adp_wave_max(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float adp_sigmoid(float h) { return 1.0f / (1.0f + __expf(-h)); }
__device__ __forceinline__ float adp_silu(float h) { return h * adp_sigmoid(h); }
// hot-loop form: v_exp_f32 + v_rcp_f32, ~2 ulp (the 1e-3 parity contract has 4 orders of magnitude of slack)
__device__ __forceinline__ float adp_silu_fast(float h) { return h * adp_rcp(1.0f + __expf(-h)); }
// d silu(h) / dh
__device__ __forceinline__ float adp_dsilu(float h) {
  float s = adp_sigmoid(h);
  return s * (1.0f + h * (1.0f - s));
}
// block-wide sum of one value per thread (NW waves); sh needs NW floats; all threads get the result
template <int NW>
__device__ __forceinline__ float adp_block_sum(float v, float* sh) {
  v = adp_wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < NW; ++i) s += sh[i];
  return s;
}
__device__ __forceinline__ float adp_gelu(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float adp_dgelu(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}
static inline int64_t adp_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
