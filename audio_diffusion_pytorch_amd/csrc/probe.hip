// Calibration probes (measurement infrastructure of bench.py, not on the denoising path): three kernels whose ideal
// rates are known, so that a bench line can say how fast THIS box streams, multiplies and launches --
//   adp_probe_copy    16-byte streaming copy                     -> HBM GB/s
//   adp_probe_mfma    register-only v_mfma_f32_32x32x2_f32 loop  -> exact-f32 matrix TFLOP/s
//   adp_probe_launch  an empty kernel                            -> dependent-launch gap inside a hipGraph
//   adp_probe_chase   one lane following a pointer chain         -> load-to-use latency of L2 / Infinity Cache / HBM
#include "adp_rt.h"
#include "adp.h"

namespace {

__global__ __launch_bounds__(256) void probe_copy_kernel(const float4* __restrict__ src, float4* __restrict__ dst, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) dst[i] = src[i];
}

// Four independent accumulator tiles per wave, operands in registers: nothing but the matrix pipe limits it.
// flops per wave = iters * 4 * 4096 (32 x 32 x 2 x 2).
__global__ __launch_bounds__(256) void probe_mfma_kernel(int64_t iters, float* out) {
  const int lane = threadIdx.x & 63;
  f32x16 acc[4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.0f;
  const float av = lane * 1e-3f, bv = lane * 2e-3f;
  for (int64_t it = 0; it < iters; ++it) {
#pragma unroll
    for (int a = 0; a < 4; ++a) acc[a] = adp_mfma32(av, bv, acc[a]);
  }
  float s = 0.0f;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[a][r];
  out[(int64_t)blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ void probe_empty_kernel() {}

// one lane, `steps` dependent loads: i = chain[i]; the chain is a random cycle over cache lines the host built
__global__ __launch_bounds__(64) void probe_chase_kernel(const int* __restrict__ chain, int64_t steps, int* out) {
  if (threadIdx.x != 0) return;
  int i = 0;
  for (int64_t s = 0; s < steps; ++s) i = chain[i];
  out[0] = i;
}

}  // namespace

extern "C" int adp_probe_copy(const float* src, float* dst, int64_t n, void* stream) {
  if (!src || !dst) return ADP_ERR_NULL;
  if (n <= 0 || n % 4) return ADP_ERR_SHAPE;
  if (((uintptr_t)src | (uintptr_t)dst) & 15) return ADP_ERR_ALIGN;
  const int64_t n4 = n / 4;
  const int64_t grid = n4 / 256 < 1 ? 1 : (n4 / 256 > 256 * 16 ? 256 * 16 : n4 / 256);
  ADP_LAUNCH(probe_copy_kernel, dim3((unsigned)grid), dim3(256), stream, (const float4*)src, (float4*)dst, n4);
  return ADP_LAUNCH_OK();
}

extern "C" int64_t adp_probe_mfma(int64_t iters, float* out, int64_t out_elems, void* stream) {
  if (!out) return ADP_ERR_NULL;
  const int64_t grid = 512;  // two 4-wave workgroups per CU: two waves per SIMD
  if (iters <= 0 || out_elems < grid * 256) return ADP_ERR_SHAPE;
  ADP_LAUNCH(probe_mfma_kernel, dim3((unsigned)grid), dim3(256), stream, iters, out);
  if (ADP_LAUNCH_OK() != ADP_OK) return ADP_ERR_LAUNCH;
  return grid * 4 * iters * 4 * 4096;  // flops of the launch
}

extern "C" int adp_probe_launch(int64_t workgroups, void* stream) {
  if (workgroups <= 0 || workgroups > 65535) return ADP_ERR_SHAPE;
  ADP_LAUNCH(probe_empty_kernel, dim3((unsigned)workgroups), dim3(64), stream);
  return ADP_LAUNCH_OK();
}

extern "C" int adp_probe_chase(const int32_t* chain, int64_t steps, int32_t* out, void* stream) {
  if (!chain || !out) return ADP_ERR_NULL;
  if (steps <= 0) return ADP_ERR_SHAPE;
  ADP_LAUNCH(probe_chase_kernel, dim3(1), dim3(64), stream, chain, steps, out);
  return ADP_LAUNCH_OK();
}
