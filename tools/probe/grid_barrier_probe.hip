// Kernel work: cost of a device-wide barrier inside one launch (all workgroups co-resident) against two launches.
// hipcc --offload-arch=gfx950 -O3 tools/probe/grid_barrier_probe.hip -o /tmp/gbp && /tmp/gbp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

__device__ __forceinline__ void grid_barrier(unsigned* ctr, unsigned nblocks) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    atomicAdd(&ctr[0], 1u);
    while (__hip_atomic_load(&ctr[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < nblocks) __builtin_amdgcn_s_sleep(1);
    __threadfence();
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void fused(const float* x, float* part, float* y, unsigned* ctr, int n, int with_barrier) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  float v = x[i % n];
  // phase 1: one partial per block
  float s = v;
  for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) part[blockIdx.x * 4 + (threadIdx.x >> 6)] = s;
  if (with_barrier) grid_barrier(ctr, gridDim.x);
  // phase 2: every block reads a partial written by ANOTHER block
  const float p = part[((blockIdx.x + gridDim.x / 2) % gridDim.x) * 4 + (threadIdx.x >> 6)];
  y[i] = v + p;
  if (with_barrier) {
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned old = atomicAdd(&ctr[1], 1u);
      if (old == gridDim.x - 1) {
        ctr[0] = 0;
        ctr[1] = 0;
      }
    }
  }
}

int main() {
  const int maxb = 4096;
  float *x, *part, *y;
  unsigned* ctr;
  hipMalloc(&x, maxb * 256 * 4);
  hipMalloc(&part, maxb * 16);
  hipMalloc(&y, maxb * 256 * 4);
  hipMalloc(&ctr, 8);
  hipMemset(ctr, 0, 8);
  std::vector<float> h(maxb * 256, 1.0f);
  hipMemcpy(x, h.data(), maxb * 256 * 4, hipMemcpyHostToDevice);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int nb : {256, 512, 1024, 2048}) {
    for (int wb = 0; wb < 2; ++wb) {
      for (int it = 0; it < 5; ++it) fused<<<nb, 256>>>(x, part, y, ctr, nb * 256, wb);
      hipDeviceSynchronize();
      hipEventRecord(a);
      for (int it = 0; it < 200; ++it) fused<<<nb, 256>>>(x, part, y, ctr, nb * 256, wb);
      hipEventRecord(b);
      hipEventSynchronize(b);
      float ms;
      hipEventElapsedTime(&ms, a, b);
      std::vector<float> out(nb * 256);
      hipMemcpy(out.data(), y, nb * 256 * 4, hipMemcpyDeviceToHost);
      int bad = 0;
      for (int i = 0; i < nb * 256; ++i) bad += (out[i] != 65.0f);
      printf("blocks %4d barrier %d: %.2f us per launch, wrong %d\n", nb, wb, ms * 1000 / 200, bad);
    }
  }
  return 0;
}
