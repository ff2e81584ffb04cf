// Do exact-f32 MFMAs and f32 VALU ops of one SIMD overlap or add (gfx950)?  And what does a wave64 VALU op cost?
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probe/alu_probe tools/probe/alu_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// mode per wave: 0 = idle, 1 = MFMA loop, 2 = v_fma loop, 3 = v_exp loop, 4 = v_pk_fma loop
__global__ void probe(long long* out, int iters, int modeA, int modeB) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int mode = wave < 4 ? modeA : modeB;
  f32x16 acc[4];
  for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  float v[8];
  f32x2 p[8];
  for (int i = 0; i < 8; ++i) { v[i] = lane * 1e-3f + i; p[i] = f32x2{v[i], v[i] + 1.f}; }
  __syncthreads();
  const long long t0 = clock64();
  if (mode == 1) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[0], v[1], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[2], v[3], acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[4], v[5], acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[6], v[7], acc[3], 0, 0, 0);
      }
    }
  } else if (mode == 2) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 16; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[i]) : "v"(1.0001f));
    }
  } else if (mode == 3) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 16; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
    }
  } else if (mode == 4) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 16; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));
    }
  }
  const long long t1 = clock64();
  float s = 0.f;
  for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
  for (int i = 0; i < 8; ++i) s += v[i] + p[i][0] + p[i][1];
  if (s == 1.2345f) out[1000] = 1;
  if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
}

int main() {
  long long* d; hipMalloc(&d, 1 << 20);
  const int iters = 200;
  const char* names[5] = {"idle", "mfma32x32x2f32", "v_fma_f32", "v_exp_f32", "v_pk_fma_f32"};
  const int per[5] = {0, 16, 128, 128, 128};
  for (int nw = 4; nw <= 8; nw += 4)
    for (int a = 1; a <= 4; ++a)
      for (int b = 0; b <= (nw == 8 ? 4 : 0); ++b) {
        if (nw == 4 && b) continue;
        probe<<<256, 64 * nw>>>(d, iters, a, b);
        hipDeviceSynchronize();
        long long h[8]; hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
        printf("waves/SIMD %d  A=%-16s B=%-16s : A %.1f cyc/op", nw / 4, names[a], nw == 8 ? names[b] : "-", (double)h[0] / (iters * per[a]));
        if (nw == 8 && b) printf("   B %.1f cyc/op", (double)h[4] / (iters * per[b]));
        printf("   (wave cycles A %lld B %lld)\n", h[0], nw == 8 ? h[4] : 0LL);
      }
  return 0;
}
