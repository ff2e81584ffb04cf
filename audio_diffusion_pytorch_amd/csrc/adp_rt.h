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

// nontemporal (streaming) global access: the data is not kept in the L2 / Infinity Cache
template <class T> __device__ __forceinline__ T adp_nt_load(const T* p) { return __builtin_nontemporal_load(p); }
template <class T> __device__ __forceinline__ void adp_nt_store(T v, T* p) { __builtin_nontemporal_store(v, p); }

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
// the round-2 streaming conv: 10 of 26 us).  The waves that PRODUCE the LDS data use __syncthreads() (their ds_writes are
// complete before they arrive); this variant is the bare s_barrier plus a compiler-level ordering point.
__device__ __forceinline__ void adp_barrier_consume() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// Workgroup barrier that publishes LDS writes only: waits for this wave's LDS operations (lgkmcnt) and nothing else, so
// global LOADS issued before it stay in flight across it (__syncthreads() carries an s_waitcnt vmcnt(0)).
__device__ __forceinline__ void adp_barrier_lds() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Ordering point for LDS data a wave hands to ITSELF (lane A writes, lane B of the same wave reads): the LDS executes one
// wave's instructions in order, so the hardware needs nothing; the compiler must not move the reads above the writes.
__device__ __forceinline__ void adp_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The value is materialised in registers HERE (optimisation barrier for code motion across this point, no instruction)
__device__ __forceinline__ void adp_pin(f32x2& v) { asm volatile("" : "+v"(v)); }

// Keeps two registers occupied up to this point without an instruction (register-allocation hint: see conv_mm's chunk-ahead reads)
__device__ __forceinline__ void adp_keep(float a, float b) { asm volatile("" ::"v"(a), "v"(b)); }

// Issue priority of this wave among the waves of its SIMD (0-3; s_setprio takes an immediate)
__device__ __forceinline__ void adp_setprio(int p) {
  if (p >= 3) __builtin_amdgcn_s_setprio(3);
  else if (p == 2) __builtin_amdgcn_s_setprio(2);
  else if (p == 1) __builtin_amdgcn_s_setprio(1);
  else __builtin_amdgcn_s_setprio(0);
}

// wave-uniform value as a scalar (v_readfirstlane): addresses built from it use SGPR bases instead of per-lane 64-bit math
__device__ __forceinline__ int adp_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
// value of lane `src` (compile-time constant) in every lane
__device__ __forceinline__ float adp_read_lane(float v, int src) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src));
}
// sum over each 16-lane row of the wave with four DPP adds; valid in EVERY lane of the row
__device__ __forceinline__ float adp_row16_sum(float v) {
#define ADP_DPP_ADD16(ctrl) \
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, false))
  ADP_DPP_ADD16(0xB1);   // quad_perm [1,0,3,2]
  ADP_DPP_ADD16(0x4E);   // quad_perm [2,3,0,1]
  ADP_DPP_ADD16(0x141);  // row_half_mirror
  ADP_DPP_ADD16(0x140);  // row_mirror
#undef ADP_DPP_ADD16
  return v;
}
// sum over the 32 lanes of each half-wave with five DPP adds (no LDS, no bpermute); VALID IN LANES 16-31 AND 48-63 only
__device__ __forceinline__ float adp_half_sum(float v) {
#define ADP_DPP_ADD(ctrl, rmask) \
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, rmask, 0xf, false))
  ADP_DPP_ADD(0xB1, 0xf);   // quad_perm [1,0,3,2]
  ADP_DPP_ADD(0x4E, 0xf);   // quad_perm [2,3,0,1]
  ADP_DPP_ADD(0x141, 0xf);  // row_half_mirror
  ADP_DPP_ADD(0x140, 0xf);  // row_mirror: every lane of a 16-lane row holds the row sum
  ADP_DPP_ADD(0x142, 0xa);  // row_bcast15 into rows 1 and 3: lanes 16-31 / 48-63 hold the half-wave sum
#undef ADP_DPP_ADD
  return v;
}
__device__ __forceinline__ float adp_exp2(float x) { return __builtin_amdgcn_exp2f(x); }  // v_exp_f32
// value of the previous / next lane of the wave (DPP wave_shr:1 / wave_shl:1, no LDS); lane 0 / lane 63 keep `edge`
__device__ __forceinline__ float adp_lane_prev(float edge, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge), __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float adp_lane_next(float edge, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge), __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false));
}
// value of the previous / next lane within the 16-lane DPP row (row_shr:1 / row_shl:1); 0 at the row's first / last lane
__device__ __forceinline__ float adp_row_prev(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));
}
__device__ __forceinline__ float adp_row_next(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x101, 0xf, 0xf, true));
}
// sum over each aligned group of 8 lanes with three DPP adds; valid in every lane of the group
__device__ __forceinline__ float adp_oct_sum(float v) {
#define ADP_DPP_ADD8(ctrl) \
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, false))
  ADP_DPP_ADD8(0xB1);   // quad_perm [1,0,3,2]
  ADP_DPP_ADD8(0x4E);   // quad_perm [2,3,0,1]
  ADP_DPP_ADD8(0x141);  // row_half_mirror
#undef ADP_DPP_ADD8
  return v;
}

// 100 MHz wall clock (s_memrealtime) and a parked wait on it: staggers the waves that share a SIMD without a barrier.
__device__ __forceinline__ long long adp_clock() { return (long long)wall_clock64(); }
__device__ __forceinline__ void adp_wait_until(long long t) {
  while ((long long)wall_clock64() < t) __builtin_amdgcn_s_sleep(2);
}

#endif

// Issue priority of the loader waves of the wave-specialised kernels (conv_mm4, wgrad_mm): compile-time A/B knob, default 0 =
// no instruction (tools/ktrace.py build <tag> -DADP_LOADER_PRIO=2 builds a measurement library with it).
#if defined(ADP_LOADER_PRIO) && !defined(ADP_EMULATE)
#define ADP_LOADER_PRIO_SET() __builtin_amdgcn_s_setprio(ADP_LOADER_PRIO)
#else
#define ADP_LOADER_PRIO_SET()
#endif

// In-kernel timeline marks for kernel work (tools/ktrace.py builds a SEPARATE measurement library with -DADP_KTRACE; in the
// product build every macro below is empty).  A wave's lane 0 drops the shader clock (s_memtime) into a per-wave row of 64
// LDS slots -- no global traffic inside the loops being measured -- and the rows of the first blocks are dumped at the end.
#if defined(ADP_KTRACE) && !defined(ADP_EMULATE)
#define ADP_KT_DECL(buf)                                           \
  __shared__ unsigned long long adp_kt_lds[16 * 64];               \
  if (threadIdx.x < 1024) adp_kt_lds[threadIdx.x & 1023] = 0ull;   \
  unsigned long long* const adp_kt_out = (buf);
#define ADP_KT(slot)                                                                                          \
  do {                                                                                                        \
    if ((threadIdx.x & 63) == 0) adp_kt_lds[(threadIdx.x >> 6) * 64 + (slot)] = __builtin_readcyclecounter(); \
  } while (0)
#ifndef ADP_KT_STRIDE
#define ADP_KT_STRIDE 1  /* trace every ADP_KT_STRIDE-th workgroup (64 of them) */
#endif
#define ADP_KT_DUMP(block_linear)                                                                                   \
  do {                                                                                                              \
    if (adp_kt_out && ((block_linear) % ADP_KT_STRIDE) == 0 && (block_linear) / ADP_KT_STRIDE < 64)                 \
      adp_kt_out[((int64_t)((block_linear) / ADP_KT_STRIDE) * 16 + (threadIdx.x >> 6)) * 64 + (threadIdx.x & 63)] = \
          adp_kt_lds[(threadIdx.x >> 6) * 64 + (threadIdx.x & 63)];                                                 \
  } while (0)
#else
#define ADP_KT_DECL(buf)
#define ADP_KT(slot)
#define ADP_KT_DUMP(block_linear)
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
// SiLU of two values: packed multiplies / adds around the two transcendental pairs (v_exp_f32, v_rcp_f32)
__device__ __forceinline__ f32x2 adp_silu2(f32x2 h) {
  const f32x2 t = h * -1.4426950408889634f;
  const f32x2 den = f32x2{adp_exp2(t[0]), adp_exp2(t[1])} + 1.0f;
  return h * f32x2{adp_rcp(den[0]), adp_rcp(den[1])};
}
// d silu(h) / dh
__device__ __forceinline__ float adp_dsilu(float h) {
  float s = adp_sigmoid(h);
  return s * (1.0f + h * (1.0f - s));
}
__device__ __forceinline__ float adp_dsilu_fast(float h) {
  const float sg = adp_rcp(1.0f + __expf(-h));
  return sg * fmaf(h, 1.0f - sg, 1.0f);
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
