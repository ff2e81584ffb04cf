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

// Copy variants for the ceiling search (MI355X_MICROARCH.md quotes 6.29 TB/s for a float4 copy): U independent 16-byte loads in
// flight per lane before the first store, a persistent grid (blocks x 256 lanes stride over the buffer), NT = nontemporal
// loads and stores (streaming data has no business in the L2 / Infinity Cache).
template <int U, bool NT>
__global__ __launch_bounds__(256) void probe_copy_u_kernel(const f32x4* __restrict__ src, f32x4* __restrict__ dst, int64_t n4) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + (U - 1) * stride < n4; i += U * stride) {
    f32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NT ? adp_nt_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (NT) adp_nt_store(v[u], dst + i + u * stride);
      else dst[i + u * stride] = v[u];
    }
  }
  for (; i < n4; i += stride) dst[i] = src[i];
}

// read-only / write-only halves of the copy (what the HBM does in ONE direction): sums of 16-byte nontemporal loads, four in
// flight per lane, one float per lane written at the end; 16-byte nontemporal stores of a constant
__global__ __launch_bounds__(256) void probe_read_kernel(const f32x4* __restrict__ src, float* __restrict__ dst, int64_t n4) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  f32x4 acc = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  for (; i + 3 * stride < n4; i += 4 * stride) {
    const f32x4 a = adp_nt_load(src + i), b = adp_nt_load(src + i + stride), c = adp_nt_load(src + i + 2 * stride),
                e = adp_nt_load(src + i + 3 * stride);
    acc = acc + ((a + b) + (c + e));
  }
  for (; i < n4; i += stride) acc = acc + src[i];
  dst[(int64_t)blockIdx.x * 256 + threadIdx.x] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
}
__global__ __launch_bounds__(256) void probe_write_kernel(f32x4* __restrict__ dst, int64_t n4) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  const f32x4 v = f32x4{1.0f, 2.0f, 3.0f, 4.0f};
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) adp_nt_store(v, dst + i);
}

// MFMA variants: UN rounds of the four independent accumulators per loop trip (fewer scalar loop instructions between MFMAs)
template <int UN>
__global__ __launch_bounds__(256) void probe_mfma_u_kernel(int64_t iters, float* out) {
  const int lane = threadIdx.x & 63;
  f32x16 acc[4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.0f;
  const float av = lane * 1e-3f, bv = lane * 2e-3f;
  for (int64_t it = 0; it < iters; it += UN) {
#pragma unroll
    for (int u = 0; u < UN; ++u)
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

extern "C" int adp_probe_copy_v(const float* src, float* dst, int64_t n, int variant, void* stream) {
  if (!src || !dst) return ADP_ERR_NULL;
  if (n <= 0 || n % 4) return ADP_ERR_SHAPE;
  if (((uintptr_t)src | (uintptr_t)dst) & 15) return ADP_ERR_ALIGN;
  const int64_t n4 = n / 4;
  const f32x4* s4 = (const f32x4*)src;
  f32x4* d4 = (f32x4*)dst;
  const dim3 blk(256);
  switch (variant) {
    case 0: return adp_probe_copy(src, dst, n, stream);
    case 1: ADP_LAUNCH((probe_copy_u_kernel<4, false>), dim3(2048), blk, stream, s4, d4, n4); break;
    case 2: ADP_LAUNCH((probe_copy_u_kernel<4, true>), dim3(2048), blk, stream, s4, d4, n4); break;
    case 3: ADP_LAUNCH((probe_copy_u_kernel<8, true>), dim3(2048), blk, stream, s4, d4, n4); break;
    case 4: ADP_LAUNCH((probe_copy_u_kernel<2, true>), dim3(4096), blk, stream, s4, d4, n4); break;
    case 5: ADP_LAUNCH((probe_copy_u_kernel<8, false>), dim3(1024), blk, stream, s4, d4, n4); break;
    case 6: ADP_LAUNCH((probe_copy_u_kernel<4, true>), dim3(1024), blk, stream, s4, d4, n4); break;
    case 7: ADP_LAUNCH((probe_copy_u_kernel<1, true>), dim3(8192), blk, stream, s4, d4, n4); break;
    case 8: if (n4 < 4096 * 256) return ADP_ERR_SHAPE;  // read only (dst receives one float per lane of the grid)
      ADP_LAUNCH(probe_read_kernel, dim3(4096), blk, stream, s4, dst, n4); break;
    case 9: ADP_LAUNCH(probe_write_kernel, dim3(4096), blk, stream, d4, n4); break;  // write only
    default: return ADP_ERR_UNSUPPORTED;
  }
  return ADP_LAUNCH_OK();
}

extern "C" int64_t adp_probe_mfma_v(int64_t iters, float* out, int64_t out_elems, int variant, void* stream) {
  if (!out) return ADP_ERR_NULL;
  if (variant == 0) return adp_probe_mfma(iters, out, out_elems, stream);
  // 1: the same grid (two waves per SIMD), 4 rounds per trip; 2: one wave per SIMD; 3: four waves per SIMD; 4: 8 rounds per trip
  const int64_t grid = variant == 2 ? 256 : (variant == 3 ? 1024 : 512);
  if (iters <= 0 || iters % 8 || out_elems < grid * 256 || variant < 0 || variant > 4) return ADP_ERR_SHAPE;
  if (variant == 4) ADP_LAUNCH((probe_mfma_u_kernel<8>), dim3((unsigned)grid), dim3(256), stream, iters, out);
  else ADP_LAUNCH((probe_mfma_u_kernel<4>), dim3((unsigned)grid), dim3(256), stream, iters, out);
  if (ADP_LAUNCH_OK() != ADP_OK) return ADP_ERR_LAUNCH;
  return grid * 4 * iters * 4 * 4096;
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
