// Probe of the Winograd MMA-wave inner loop on gfx950: how much of the VALU transforms, LDS fragment reads and chunk
// barriers hides under the f32 MFMAs?  build: hipcc --offload-arch=gfx950 -O3 -o wn_probe wn_probe.hip
// Each wave: ITER chunks x 4 channel pairs x { NLDS b64 fragment reads, NVALU dependent adds, 4 MFMAs (4 accumulators) }.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int NVALU, int NLDS, int BAR>
__global__ void probe(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[16384];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = 1e-6f * i;
  __syncthreads();
  f32x16 acc[4];
  for (int a = 0; a < 4; ++a)
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  float g0 = lane * 1e-3f, g1 = lane * 2e-3f, g2 = lane * 3e-3f;
  const int base = (wave * 576 + 2 * (lane & 31) + 72 * 4 * (lane >> 5)) & 8191;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
      float d0 = g0, d1 = g1, d2 = g2, d3 = g0;
      if (NLDS) {
        const float* xp = lds + ((base + cc * 72 + (it & 7) * 288) & 8191);
        const f32x2 p0 = *reinterpret_cast<const f32x2*>(xp);
        const f32x2 p1 = *reinterpret_cast<const f32x2*>(xp + 2);
        const f32x2 p2 = *reinterpret_cast<const f32x2*>(xp + 4);
        d0 = p0[1], d1 = p1[0], d2 = p1[1], d3 = p2[0];
      }
      float u0 = g0, u1 = g1, u2 = g2, u3 = g0, v0 = d0, v1 = d1, v2 = d2, v3 = d3;
      if (NVALU >= 7) {
        const float gs = g0 + g2;
        u1 = gs + g1, u2 = gs - g1;
        v0 = d0 - d2, v1 = d1 + d2, v2 = d2 - d1, v3 = d1 - d3;
      }
#pragma unroll
      for (int e = 7; e < NVALU; e += 4) {  // extra VALU work on the operands
        v0 = v0 * 1.0001f + d3, v1 = v1 * 1.0001f + d0, v2 = v2 * 1.0001f + d1, v3 = v3 * 1.0001f + d2;
      }
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(u0, v0, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(u1, v1, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(u2, v2, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(u3, v3, acc[3], 0, 0, 0);
    }
    if (BAR) __syncthreads();
  }
  float s = 0.f;
  for (int a = 0; a < 4; ++a)
    for (int r = 0; r < 16; ++r) s += acc[a][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// the same loop with the fragment reads of channel pair cc+1 issued BEFORE the MFMAs of pair cc (register double buffer)
template <int BAR>
__global__ void probe_pipe(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[16384];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = 1e-6f * i;
  __syncthreads();
  f32x16 acc[4];
  for (int a = 0; a < 4; ++a)
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  float g0 = lane * 1e-3f, g1 = lane * 2e-3f, g2 = lane * 3e-3f;
  const int base = (wave * 576 + 2 * (lane & 31) + 72 * 4 * (lane >> 5)) & 8191;
  f32x2 p[2][3];
  {
    const float* xp = lds + base;
    p[0][0] = *reinterpret_cast<const f32x2*>(xp), p[0][1] = *reinterpret_cast<const f32x2*>(xp + 2), p[0][2] = *reinterpret_cast<const f32x2*>(xp + 4);
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
      const int nx = (cc + 1) & 1, cu = cc & 1;
      const float* xp = lds + ((base + ((cc + 1) & 3) * 72 + ((it + (cc == 3)) & 7) * 288) & 8191);
      p[nx][0] = *reinterpret_cast<const f32x2*>(xp), p[nx][1] = *reinterpret_cast<const f32x2*>(xp + 2), p[nx][2] = *reinterpret_cast<const f32x2*>(xp + 4);
      __builtin_amdgcn_sched_barrier(0);  // the reads of the NEXT pair stay above the MFMAs of this one
      const float d0 = p[cu][0][1], d1 = p[cu][1][0], d2 = p[cu][1][1], d3 = p[cu][2][0];
      const float gs = g0 + g2;
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(g0, d0 - d2, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(gs + g1, d1 + d2, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(gs - g1, d2 - d1, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(g2, d1 - d3, acc[3], 0, 0, 0);
    }
    if (BAR) __syncthreads();
  }
  float s = 0.f;
  for (int a = 0; a < 4; ++a)
    for (int r = 0; r < 16; ++r) s += acc[a][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int BAR>
void run_pipe(int waves, int blocks_per_cu, float* out) {
  const int iters = 512;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int grid = 256 * blocks_per_cu;
  probe_pipe<BAR><<<grid, waves * 64>>>(out, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe_pipe<BAR><<<grid, waves * 64>>>(out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double nm = (double)grid * waves * iters * 16.0;
  printf("PIPELINED lds reads, barrier/16mfma %d  waves/blk %2d blk/CU %d : %7.3f ms  %6.1f TF executed  %5.1f cyc/MFMA/SIMD @2.0GHz\n", BAR, waves,
         blocks_per_cu, ms, nm * 4096.0 / ms / 1e9, ms * 1e-3 * 2.0e9 / (nm / 1024.0));
}

// cost model of the fragment reads: per 4 MFMAs, NR reads of WIDTH dwords each (issued one pair ahead)
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NR, int WIDTH>
__global__ void probe_w(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[16384];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = 1e-6f * i;
  __syncthreads();
  f32x16 acc[4];
  for (int a = 0; a < 4; ++a)
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  float g0 = lane * 1e-3f, g1 = lane * 2e-3f, g2 = lane * 3e-3f;
  const int base = (wave * 1024 + WIDTH * lane) & 8191;
  float cur[NR * WIDTH], nxt[NR * WIDTH];
  for (int k = 0; k < NR * WIDTH; ++k) cur[k] = g0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
      const float* xp = lds + ((base + cc * 256 + (it & 3) * 2048) & 8191);
#pragma unroll
      for (int q = 0; q < NR; ++q) {
        if (WIDTH == 4) {
          const f32x4 t = *reinterpret_cast<const f32x4*>(xp + q * 1024);
          for (int k = 0; k < 4; ++k) nxt[q * 4 + k] = t[k];
        } else if (WIDTH == 2) {
          const f32x2 t = *reinterpret_cast<const f32x2*>(xp + q * 1024);
          nxt[q * 2] = t[0], nxt[q * 2 + 1] = t[1];
        } else {
          nxt[q] = xp[q * 1024];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      float s0 = 0, s1 = 0, s2 = 0, s3 = 0;
#pragma unroll
      for (int k = 0; k < NR * WIDTH; k += 4) {
        s0 += cur[k];
        if (k + 1 < NR * WIDTH) s1 += cur[k + 1];
        if (k + 2 < NR * WIDTH) s2 += cur[k + 2];
        if (k + 3 < NR * WIDTH) s3 += cur[k + 3];
      }
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(g0, s0, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(g1, s1 + g0, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(g2, s2 + g1, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(g0, s3 + g2, acc[3], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < NR * WIDTH; ++k) cur[k] = nxt[k];
    }
  }
  float s = 0.f;
  for (int a = 0; a < 4; ++a)
    for (int r = 0; r < 16; ++r) s += acc[a][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NR, int WIDTH>
void run_w(int waves, float* out) {
  const int iters = 512;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  probe_w<NR, WIDTH><<<256, waves * 64>>>(out, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe_w<NR, WIDTH><<<256, waves * 64>>>(out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double nm = 256.0 * waves * iters * 16.0;
  printf("per 4 MFMAs: %d reads x %d dwords  waves/blk %2d : %7.3f ms  %6.1f TF executed  %5.1f cyc/MFMA/SIMD @2.0GHz\n", NR, WIDTH, waves, ms,
         nm * 4096.0 / ms / 1e9, ms * 1e-3 * 2.0e9 / (nm / 1024.0));
}

template <int NVALU, int NLDS, int BAR>
void run(int waves, int blocks_per_cu, float* out) {
  const int iters = 512;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int grid = 256 * blocks_per_cu;
  probe<NVALU, NLDS, BAR><<<grid, waves * 64>>>(out, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<NVALU, NLDS, BAR><<<grid, waves * 64>>>(out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double nm = (double)grid * waves * iters * 16.0;
  printf("valu/4mfma %2d lds %d barrier/16mfma %d  waves/blk %2d blk/CU %d : %7.3f ms  %6.1f TF executed  %5.1f cyc/MFMA/SIMD @2.0GHz\n", NVALU, NLDS,
         BAR, waves, blocks_per_cu, ms, nm * 4096.0 / ms / 1e9, ms * 1e-3 * 2.0e9 / (nm / 1024.0));
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 4 * 1024 * sizeof(float));
  run<0, 0, 0>(4, 1, out);
  run<0, 0, 0>(8, 1, out);
  run<7, 0, 0>(4, 1, out);
  run<7, 0, 0>(8, 1, out);
  run<15, 0, 0>(8, 1, out);
  run<31, 0, 0>(8, 1, out);
  run<7, 1, 0>(4, 1, out);
  run<7, 1, 0>(8, 1, out);
  run<7, 1, 1>(4, 1, out);
  run<7, 1, 1>(8, 1, out);
  run<7, 1, 1>(12, 1, out);
  run<7, 1, 1>(4, 2, out);
  run<7, 1, 1>(8, 2, out);
  run<0, 0, 1>(8, 1, out);
  for (int w = 4; w <= 8; w += 4) {
    run_w<1, 1>(w, out);
    run_w<2, 1>(w, out);
    run_w<4, 1>(w, out);
    run_w<8, 1>(w, out);
    run_w<1, 2>(w, out);
    run_w<2, 2>(w, out);
    run_w<4, 2>(w, out);
    run_w<1, 4>(w, out);
    run_w<2, 4>(w, out);
    run_w<3, 4>(w, out);
  }
  run_pipe<0>(4, 1, out);
  run_pipe<0>(8, 1, out);
  run_pipe<1>(4, 1, out);
  run_pipe<1>(8, 1, out);
  run_pipe<1>(12, 1, out);
  run_pipe<1>(8, 2, out);
  return 0;
}
