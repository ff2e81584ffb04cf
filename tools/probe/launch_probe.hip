// What does a tiny dependent kernel cost on MI355X (launch boundary + cold first access)?
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probe/launch_probe tools/probe/launch_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_empty() {}
// one load -> one store per thread, data set of `n` floats (grid covers it)
__global__ void k_copy(const float* __restrict__ a, float* __restrict__ b, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) b[i] = a[i] + 1.0f;
}
// two dependent round trips: index load, then data load
__global__ void k_dep(const int* __restrict__ idx, const float* __restrict__ a, float* __restrict__ b, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) b[i] = a[idx[i]] + 1.0f;
}
// float4 copy
__global__ void k_copy4(const float4* __restrict__ a, float4* __restrict__ b, int n4) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n4) { float4 v = a[i]; v.x += 1.f; b[i] = v; }
}
template <typename F>
float run(F launch, int reps) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 10; ++i) launch(i);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) launch(i);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1000.f / reps;
}
int main() {
  const int N = 1 << 24;
  float *a, *b; int* idx;
  hipMalloc(&a, N * 4); hipMalloc(&b, N * 4); hipMalloc(&idx, N * 4);
  hipMemset(a, 0, N * 4); hipMemset(idx, 0, N * 4);
  // stream launches (the host may be the bound) and the same inside a graph
  hipStream_t s; hipStreamCreate(&s);
  auto graph_time = [&](auto body, int n) {
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int i = 0; i < n; ++i) body(i);
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, s); hipStreamSynchronize(s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, s);
    for (int r = 0; r < 5; ++r) hipGraphLaunch(ge, s);
    hipEventRecord(e1, s); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.f / (5 * n);
  };
  printf("graph of 200 dependent launches, us per launch:\n");
  printf("  empty 1 WG            : %.2f\n", graph_time([&](int) { k_empty<<<1, 64, 0, s>>>(); }, 200));
  printf("  empty 256 WG          : %.2f\n", graph_time([&](int) { k_empty<<<256, 256, 0, s>>>(); }, 200));
  for (int n : {2048, 65536, 1 << 20, 1 << 22}) {
    printf("  copy  %8d floats (a->b, b->a alternating): %.2f\n", n, graph_time([&](int i) { if (i & 1) k_copy<<<(n + 255) / 256, 256, 0, s>>>(b, a, n); else k_copy<<<(n + 255) / 256, 256, 0, s>>>(a, b, n); }, 200));
    printf("  copy4 %8d floats                           : %.2f\n", n, graph_time([&](int i) { if (i & 1) k_copy4<<<(n / 4 + 255) / 256, 256, 0, s>>>((float4*)b, (float4*)a, n / 4); else k_copy4<<<(n / 4 + 255) / 256, 256, 0, s>>>((float4*)a, (float4*)b, n / 4); }, 200));
    printf("  dep   %8d floats (idx -> a -> b)           : %.2f\n", n, graph_time([&](int i) { if (i & 1) k_dep<<<(n + 255) / 256, 256, 0, s>>>(idx, b, a, n); else k_dep<<<(n + 255) / 256, 256, 0, s>>>(idx, a, b, n); }, 200));
  }
  return 0;
}
