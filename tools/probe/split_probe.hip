// Probe for an fp32-equivalent conv on the bf16 matrix cores of gfx950 (build: hipcc --offload-arch=gfx950 -O3 -o
// split_probe split_probe.hip).  An fp32 value x is split into three bf16 parts hi + mid + lo (24 mantissa bits); a
// product of two such values keeps the six partial products down to 2^-16 (hi*hi, hi*mid, mid*hi, mid*mid, hi*lo,
// lo*hi), each a v_mfma_f32_32x32x16_bf16 with fp32 accumulation: 6 x 32 cycles per K=16 step against 8 x 64 cycles of
// v_mfma_f32_32x32x2_f32 -- 2.67x the exact-f32 MFMA rate at fp32-level error (measured on the host: 2.2e-7 relative
// to fp64 over K = 3072, the exact-f32 chain 5.0e-7).
//   part 1: operand layout check of the bf16 MFMA (random A, B against a host reference);
//   part 2: MMA-wave-only throughput of the 6-product loop with LDS-resident operands laid out the way a conv kernel
//           would stage them ([tap][row][16 channels] weights, [position][16 channels] activations).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ void layout_kernel(const __bf16* A, const __bf16* B, float* C) {
  // A[32][16] row-major, B[16][32] row-major, C[32][32]
  const int l = threadIdx.x;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) {
    a[j] = A[(l & 31) * 16 + 8 * (l >> 5) + j];
    b[j] = B[(8 * (l >> 5) + j) * 32 + (l & 31)];
  }
  f32x16 c;
  for (int r = 0; r < 16; ++r) c[r] = 0.f;
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}

static float bf16_round(float x) { return (float)(__bf16)x; }

static int layout_check() {
  std::vector<__bf16> A(32 * 16), B(16 * 32);
  std::vector<float> Af(32 * 16), Bf(16 * 32), C(32 * 32), R(32 * 32, 0.f);
  srand(1);
  for (int i = 0; i < 32 * 16; ++i) {
    Af[i] = bf16_round((rand() % 2001 - 1000) / 500.0f);
    Bf[i] = bf16_round((rand() % 2001 - 1000) / 700.0f);
    A[i] = (__bf16)Af[i];
    B[i] = (__bf16)Bf[i];
  }
  for (int i = 0; i < 32; ++i)
    for (int j = 0; j < 32; ++j) {
      double s = 0;
      for (int k = 0; k < 16; ++k) s += (double)Af[i * 16 + k] * Bf[k * 32 + j];
      R[i * 32 + j] = (float)s;
    }
  __bf16 *dA, *dB;
  float* dC;
  hipMalloc(&dA, A.size() * 2);
  hipMalloc(&dB, B.size() * 2);
  hipMalloc(&dC, C.size() * 4);
  hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
  layout_kernel<<<1, 64>>>(dA, dB, dC);
  hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
  double err = 0, ref = 0;
  for (int i = 0; i < 32 * 32; ++i) {
    err = fmax(err, fabs(C[i] - R[i]));
    ref = fmax(ref, fabs(R[i]));
  }
  printf("layout check: A[i=l&31][k=8*(l>>5)+j], B[k=8*(l>>5)+j][n=l&31]: max err %.3g (max |ref| %.3g) -> %s\n", err, ref,
         err < 1e-4 * ref ? "OK" : "MISMATCH");
  return err < 1e-4 * ref ? 0 : 1;
}

// WM x WN accumulator subtiles (32 x 32) per wave; NWM x NWN MMA waves per workgroup; NPROD partial products (6 / 3 / 1).
template <int WM, int WN, int NWM, int NWN, int NPROD, int BAR>
__global__ __launch_bounds__(NWM* NWN * 64) void mma_loop(float* out, int iters) {
  constexpr int BM = NWM * WM * 32, BN = NWN * WN * 32;
  constexpr int NP = NPROD == 1 ? 1 : (NPROD == 3 ? 2 : 3);
  __shared__ __attribute__((aligned(16))) __bf16 Al[2][NP][3][BM][16];
  __shared__ __attribute__((aligned(16))) __bf16 Xl[2][NP][BN + 2][16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave % NWM) * WM * 32, wn0 = (wave / NWM) * WN * 32;
  for (int i = tid; i < 2 * NP * 3 * BM * 16; i += blockDim.x) (&Al[0][0][0][0][0])[i] = (__bf16)(1e-3f * (i % 977));
  for (int i = tid; i < 2 * NP * (BN + 2) * 16; i += blockDim.x) (&Xl[0][0][0][0])[i] = (__bf16)(1e-3f * (i % 911));
  __syncthreads();
  f32x16 acc[WM][WN];
  for (int i = 0; i < WM; ++i)
    for (int j = 0; j < WN; ++j)
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int l31 = lane & 31, kh = (lane >> 5) * 8;
  for (int it = 0; it < iters; ++it) {
    const int buf = it & 1;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      bf16x8 a[WM][NP], b[WN][NP];
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int p = 0; p < NP; ++p) a[i][p] = *reinterpret_cast<const bf16x8*>(&Al[buf][p][t][wm0 + i * 32 + l31][kh]);
#pragma unroll
      for (int j = 0; j < WN; ++j)
#pragma unroll
        for (int p = 0; p < NP; ++p) b[j][p] = *reinterpret_cast<const bf16x8*>(&Xl[buf][p][wn0 + j * 32 + l31 + t][kh]);
      // small terms first
#pragma unroll
      for (int o = 2 * (NP - 1); o >= 0; --o)
#pragma unroll
        for (int pa = 0; pa < NP; ++pa) {
          const int pb = o - pa;
          if (pb < 0 || pb >= NP) continue;
          if (NPROD == 6 && o > 2) continue;  // mid*lo, lo*mid, lo*lo dropped
          if (NPROD == 3 && o > 1) continue;  // lo*lo dropped
#pragma unroll
          for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][pa], b[j][pb], acc[i][j], 0, 0, 0);
        }
    }
    if (BAR) __syncthreads();
  }
  float s = 0.f;
  for (int i = 0; i < WM; ++i)
    for (int j = 0; j < WN; ++j)
      for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[(size_t)blockIdx.x * blockDim.x + tid] = s;
}

template <int WM, int WN, int NWM, int NWN, int NPROD, int BAR>
void run(const char* name, int blocks_per_cu, float* out) {
  constexpr int BM = NWM * WM * 32, BN = NWN * WN * 32;
  const int iters = 400;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int grid = 256 * blocks_per_cu;
  mma_loop<WM, WN, NWM, NWN, NPROD, BAR><<<grid, NWM * NWN * 64>>>(out, iters);
  if (hipDeviceSynchronize() != hipSuccess) {
    printf("%s: launch failed (%s)\n", name, hipGetErrorString(hipGetLastError()));
    return;
  }
  hipEventRecord(e0);
  mma_loop<WM, WN, NWM, NWN, NPROD, BAR><<<grid, NWM * NWN * 64>>>(out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)grid * iters * (double)BM * BN * 48.0 * 2.0;  // fp32-equivalent conv flops
  printf("%-34s tile %3dx%3d wave %dx%d waves %d blk/CU %d prod %d bar %d: %8.3f ms  %7.1f TF(fp32-equivalent)\n", name, BM, BN,
         WM * 32, WN * 32, NWM * NWN, blocks_per_cu, NPROD, BAR, ms, flops / ms / 1e9);
}

int main() {
  if (layout_check()) return 1;
  float* out;
  hipMalloc(&out, 256 * 2 * 1024 * sizeof(float));
  run<2, 2, 2, 2, 6, 1>("128x128, 4 waves of 64x64", 1, out);
  run<2, 2, 2, 2, 6, 0>("128x128, 4 waves of 64x64", 1, out);
  run<2, 2, 2, 2, 3, 1>("128x128, 4 waves, 3 products", 1, out);
  run<2, 2, 2, 2, 1, 1>("128x128, 4 waves, plain bf16", 1, out);
  run<1, 2, 2, 2, 6, 1>("64x128, 4 waves of 32x64", 1, out);
  run<1, 2, 2, 2, 6, 1>("64x128, 4 waves of 32x64", 2, out);
  run<1, 2, 4, 2, 6, 1>("128x128, 8 waves of 32x64", 1, out);
  run<2, 2, 2, 4, 6, 1>("128x256, 8 waves of 64x64", 1, out);
  run<1, 1, 2, 2, 6, 1>("64x64, 4 waves of 32x32", 2, out);
  run<1, 2, 1, 4, 6, 1>("32x256, 4 waves of 32x64", 1, out);
  return 0;
}
