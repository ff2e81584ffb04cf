// Micro-probes of the f32 MFMA issue behaviour on gfx950 (build: hipcc --offload-arch=gfx950 -O3 -o mfma_probe mfma_probe.hip).
// Each probe: grid = 256 blocks (1 per CU) x WAVES waves; every wave runs ITER rounds of NMF MFMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, int LDSR, int BAR>
__global__ void probe(float* out, int iters) {
  __shared__ float lds[8192];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = 1e-6f * i;
  __syncthreads();
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; ++a)
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  float av = lane * 1e-3f, bv = lane * 2e-3f;
  int off = lane + (threadIdx.x >> 6) * 64;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      float b = bv;
      if (LDSR) b = lds[(off + k * 67 + it) & 8191];
#pragma unroll
      for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b, acc[a], 0, 0, 0);
    }
    if (BAR) __syncthreads();
  }
  float s = 0.f;
  for (int a = 0; a < NACC; ++a)
    for (int r = 0; r < 16; ++r) s += acc[a][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC, int LDSR, int BAR>
void run(const char* name, int waves, int blocks_per_cu, float* out) {
  const int iters = 2000 / NACC;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int grid = 256 * blocks_per_cu;
  probe<NACC, LDSR, BAR><<<grid, waves * 64>>>(out, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<NACC, LDSR, BAR><<<grid, waves * 64>>>(out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)grid * waves * iters * 12.0 * NACC * 4096.0;
  printf("%-44s waves/blk %2d blk/CU %d acc %d lds %d bar %d : %8.3f ms  %7.1f TF\n", name, waves, blocks_per_cu, NACC, LDSR, BAR,
         ms, flops / ms / 1e9);
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 4 * 1024 * sizeof(float));
  run<1, 0, 0>("1 acc chain, regs only", 4, 1, out);
  run<1, 0, 0>("1 acc chain, regs only", 8, 1, out);
  run<1, 0, 0>("1 acc chain, regs only", 16, 1, out);
  run<2, 0, 0>("2 acc, regs only", 4, 1, out);
  run<4, 0, 0>("4 acc, regs only", 4, 1, out);
  run<4, 0, 0>("4 acc, regs only", 8, 1, out);
  run<1, 1, 0>("1 acc + ds_read per mfma", 16, 1, out);
  run<4, 1, 0>("4 acc + ds_read per 4 mfma", 4, 1, out);
  run<1, 1, 1>("1 acc + ds_read + barrier/12", 16, 1, out);
  run<1, 1, 1>("1 acc + ds_read + barrier/12", 8, 2, out);
  run<4, 1, 1>("4 acc + ds_read + barrier/48", 4, 1, out);
  run<4, 1, 1>("4 acc + ds_read + barrier/48", 4, 2, out);
  run<2, 1, 1>("2 acc + ds_read + barrier/24", 8, 1, out);
  return 0;
}
