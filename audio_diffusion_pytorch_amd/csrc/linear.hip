// Small-batch Linear layers of the conditioning path (gfx950, HBM-bound weight streaming):
// TimeConditioningPlugin's MLP (components.py:74-76) and the bank of every
// `Linear(modulation_features -> 2C | C)(SiLU(features))` that ModulationItem / SkipModulate own
// (components.py:90, :99, modulation_features :48).  With B <= 16 rows these are GEMV-shaped: the weight
// matrix is read exactly once with 16-byte lane loads, the few activation rows sit in LDS.
#include "adp_rt.h"
#include "adp.h"
#include "conv_internal.h"

namespace {

constexpr int LIN_BMAX = 16;
constexpr int LIN_KC = 1024;  // K chunk held in LDS: 16 rows x 1024 x 4 B = 64 KiB max
// output rows per wave: the activation rows are staged (and their SiLU evaluated) once per workgroup, so a workgroup
// should stream more than four weight rows behind them -- 8 per wave = 128 KiB of weights per 16 KiB staged
constexpr int LIN_RPW = 8;
static inline int64_t lin_rpw(int64_t N) { return N >= 4096 ? LIN_RPW : 1; }

__device__ __forceinline__ float lin_act(float x, int act) {
  return act == 1 ? adp_silu(x) : (act == 2 ? adp_gelu(x) : x);
}

// stage act(x[b, k0 : k0+kc]) for all rows into xs[b][LIN_KC]
__device__ __forceinline__ void lin_stage(const float* x, int64_t B, int64_t K, int64_t k0, int kc, int act,
                                          float* xs) {
  for (int e = threadIdx.x; e < (int)B * LIN_KC; e += 256) {
    const int b = e / LIN_KC, k = e - b * LIN_KC;
    xs[e] = (k < kc) ? lin_act(x[b * K + k0 + k], act) : 0.0f;
  }
}

// y[b, n] = post(bias[n] + sum_k act(x[b,k]) w[n,k]); one wave per output row n, lanes stride K by 4
template <int BT>
__global__ __launch_bounds__(256) void linear_fwd_kernel(const float* x, const float* w, const float* bias,
                                                         int64_t B, int64_t K, int64_t N, int act, int post,
                                                         int64_t ybstride, float* y, int rpw) {
  __shared__ float xs[BT * LIN_KC];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t nbase = ((int64_t)blockIdx.x * 4 + wave) * rpw;
  const bool vec = (K % 4 == 0);
  for (int64_t k0 = 0; k0 < K; k0 += LIN_KC) {
    const int kc = (int)((K - k0) < LIN_KC ? (K - k0) : LIN_KC);
    __syncthreads();
    lin_stage(x, B, K, k0, kc, act, xs);
    __syncthreads();
    for (int r = 0; r < rpw; ++r) {
      const int64_t n = nbase + r;
      if (n >= N) break;
      float acc[BT];
#pragma unroll
      for (int b = 0; b < BT; ++b) acc[b] = 0.0f;
      const float* wr = w + n * K + k0;
#pragma unroll 4
      for (int k = lane * 4; k < kc; k += 256) {  // a 1024-wide row is four independent 16-byte loads per lane
        float w0, w1, w2, w3;
        if (vec) {
          const float4 wv = *reinterpret_cast<const float4*>(wr + k);
          w0 = wv.x; w1 = wv.y; w2 = wv.z; w3 = wv.w;
        } else {
          w0 = wr[k];
          w1 = (k + 1 < kc) ? wr[k + 1] : 0.0f;
          w2 = (k + 2 < kc) ? wr[k + 2] : 0.0f;
          w3 = (k + 3 < kc) ? wr[k + 3] : 0.0f;
        }
#pragma unroll
        for (int b = 0; b < BT; ++b) {
          if (b < B) {
            const float* xr = xs + b * LIN_KC + k;
            acc[b] = fmaf(w0, xr[0], fmaf(w1, xr[1], fmaf(w2, xr[2], fmaf(w3, xr[3], acc[b]))));
          }
        }
      }
      // (K <= LIN_KC for every layer with rpw > 1; longer rows accumulate through y)
#pragma unroll
      for (int b = 0; b < BT; ++b) {
        const float sres = adp_wave_sum(acc[b]);
        if (lane == 0 && b < B) {
          float v = sres + ((k0 == 0 && bias) ? bias[n] : 0.0f);
          if (k0 > 0) v += y[b * ybstride + n];
          if (post == 2 && k0 + LIN_KC >= K) v = adp_gelu(v);
          y[b * ybstride + n] = v;
        }
      }
    }
  }
}

// partial[p][b][k] = sum_{n in block p's LIN_NR-row range} dy[b,n] w[n,k].  grid = (row ranges, 256-column
// slabs): a workgroup streams LIN_NR rows x 1 KB; its four waves take a quarter of the rows each (16 independent
// 16-byte loads in flight per lane) and meet in LDS.  [With 256 rows per workgroup and one workgroup per row range
// the 1024x1024 time-MLP layers ran on FOUR workgroups (132 us each) and the 184 MB bank at 0.96 TB/s.]
constexpr int LIN_NR = 64;

template <int BT>
__global__ __launch_bounds__(256) void linear_bwd_data_kernel(const float* dy, int64_t dybstride, const float* w,
                                                              int64_t B, int64_t K, int64_t N, float* ws) {
  constexpr int NR = LIN_NR, RPW = NR / 4;
  __shared__ float dys[BT * NR];
  __shared__ __attribute__((aligned(16))) float part[3][BT][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t n0 = (int64_t)blockIdx.x * NR;
  const int nr = (int)((N - n0) < NR ? (N - n0) : NR);
  for (int e = threadIdx.x; e < BT * NR; e += 256) {
    const int b = e / NR, j = e - b * NR;
    dys[e] = (b < B && j < nr) ? dy[b * dybstride + n0 + j] : 0.0f;
  }
  __syncthreads();
  const bool vec = (K % 4 == 0);
  const int64_t k = (int64_t)blockIdx.y * 256 + lane * 4;
  float acc[BT][4];
#pragma unroll
  for (int b = 0; b < BT; ++b) acc[b][0] = acc[b][1] = acc[b][2] = acc[b][3] = 0.0f;
  if (k < K) {
#pragma unroll 8
    for (int jj = 0; jj < RPW; ++jj) {
      const int j = wave * RPW + jj;
      if (j < nr) {
        const float* wr = w + (n0 + j) * K + k;
        float w0, w1, w2, w3;
        if (vec) {
          const float4 wv = *reinterpret_cast<const float4*>(wr);
          w0 = wv.x; w1 = wv.y; w2 = wv.z; w3 = wv.w;
        } else {
          w0 = wr[0];
          w1 = (k + 1 < K) ? wr[1] : 0.0f;
          w2 = (k + 2 < K) ? wr[2] : 0.0f;
          w3 = (k + 3 < K) ? wr[3] : 0.0f;
        }
#pragma unroll
        for (int b = 0; b < BT; ++b) {
          const float d = dys[b * NR + j];
          acc[b][0] = fmaf(d, w0, acc[b][0]);
          acc[b][1] = fmaf(d, w1, acc[b][1]);
          acc[b][2] = fmaf(d, w2, acc[b][2]);
          acc[b][3] = fmaf(d, w3, acc[b][3]);
        }
      }
    }
  }
  // fixed-order sum of the four waves (deterministic)
  if (wave > 0) {
#pragma unroll
    for (int b = 0; b < BT; ++b)
#pragma unroll
      for (int q = 0; q < 4; ++q) part[wave - 1][b][lane * 4 + q] = acc[b][q];
  }
  __syncthreads();
  if (wave == 0 && k < K) {
#pragma unroll
    for (int b = 0; b < BT; ++b) {
      if (b < B) {
        float* o = ws + ((int64_t)blockIdx.x * B + b) * K + k;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float v = ((acc[b][q] + part[0][b][lane * 4 + q]) + part[1][b][lane * 4 + q]) + part[2][b][lane * 4 + q];
          if (k + q < K) o[q] = v;
        }
      }
    }
  }
}

// dw[n,k] = sum_b dy[b,n] act(x[b,k]) ; dbias[n] = sum_b dy[b,n]; one wave per row n
template <int BT>
__global__ __launch_bounds__(256) void linear_bwd_weight_kernel(const float* dy, int64_t dybstride, const float* x,
                                                                int64_t B, int64_t K, int64_t N, int act,
                                                                int accumulate, float* dw, float* dbias, int rpw) {
  __shared__ __attribute__((aligned(16))) float xs[BT * LIN_KC];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t nbase = ((int64_t)blockIdx.x * 4 + wave) * rpw;
  const bool vec = (K % 4 == 0) && ((reinterpret_cast<uintptr_t>(dw) & 15) == 0);
  // the first row's dy values are requested before the activation rows are staged: the small layers (one row per wave) are a
  // chain of memory round trips (x -> LDS -> dy -> store), and this one need not wait for the barrier
  float dfirst[BT];
#pragma unroll
  for (int b = 0; b < BT; ++b) dfirst[b] = (b < B && nbase < N) ? dy[b * dybstride + nbase] : 0.0f;
  for (int64_t k0 = 0; k0 < K; k0 += LIN_KC) {
    const int kc = (int)((K - k0) < LIN_KC ? (K - k0) : LIN_KC);
    __syncthreads();
    lin_stage(x, B, K, k0, kc, act, xs);
    __syncthreads();
    for (int r = 0; r < rpw; ++r) {
      const int64_t n = nbase + r;
      if (n >= N) break;
      float d[BT];
#pragma unroll
      for (int b = 0; b < BT; ++b) d[b] = r == 0 ? dfirst[b] : ((b < B) ? dy[b * dybstride + n] : 0.0f);
      if (k0 == 0 && dbias && lane == 0) {
        float s = 0.0f;
#pragma unroll
        for (int b = 0; b < BT; ++b) s += d[b];
        dbias[n] = accumulate ? dbias[n] + s : s;
      }
      float* orow = dw + n * K + k0;
      if (vec) {
        for (int k = lane * 4; k < kc; k += 256) {  // 16-byte stores: the gradient rows are what bounds this kernel
          f32x4 sv = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
          for (int b = 0; b < BT; ++b)
            if (b < B) {
              const f32x4 xv = *reinterpret_cast<const f32x4*>(xs + b * LIN_KC + k);
#pragma unroll
              for (int j = 0; j < 4; ++j) sv[j] = fmaf(d[b], xv[j], sv[j]);
            }
          if (accumulate) {
            const f32x4 o = *reinterpret_cast<const f32x4*>(orow + k);
#pragma unroll
            for (int j = 0; j < 4; ++j) sv[j] += o[j];
          }
          *reinterpret_cast<f32x4*>(orow + k) = sv;
        }
      } else {
        for (int k = lane; k < kc; k += 64) {
          float s = 0.0f;
#pragma unroll
          for (int b = 0; b < BT; ++b)
            if (b < B) s = fmaf(d[b], xs[b * LIN_KC + k], s);
          orow[k] = accumulate ? orow[k] + s : s;
        }
      }
    }
  }
}

}  // namespace

#define LIN_DISPATCH(B, CALL)                   \
  do {                                          \
    if ((B) <= 4) { constexpr int BT = 4; CALL; }        \
    else if ((B) <= 8) { constexpr int BT = 8; CALL; }   \
    else { constexpr int BT = 16; CALL; }       \
  } while (0)

extern "C" int adp_linear_fwd(const float* x, const float* w, const float* bias, int64_t B, int64_t K, int64_t N,
                              int64_t act, int64_t post, float* y, int64_t y_bstride, void* stream) {
  if (!x || !w || !y) return ADP_ERR_NULL;
  if (B <= 0 || B > LIN_BMAX || K <= 0 || N <= 0) return ADP_ERR_SHAPE;
  if (act < 0 || act > 2 || (post != 0 && post != 2)) return ADP_ERR_UNSUPPORTED;
  if (y_bstride == 0) y_bstride = N;
  const int64_t rpw = lin_rpw(N);
  dim3 grid((unsigned)adp_cdiv(N, 4 * rpw));
  LIN_DISPATCH(B, ADP_LAUNCH((linear_fwd_kernel<BT>), grid, dim3(256), stream, x, w, bias, B, K, N, (int)act,
                             (int)post, y_bstride, y, (int)rpw));
  return ADP_LAUNCH_OK();
}

extern "C" int64_t adp_linear_bwd_data_ws_bytes(int64_t B, int64_t K, int64_t N) {
  if (B <= 0 || K <= 0 || N <= 0) return ADP_ERR_SHAPE;
  return adp_cdiv(N, LIN_NR) * B * K * (int64_t)sizeof(float);
}

extern "C" int adp_linear_bwd_data(const float* dy, int64_t dy_bstride, const float* w, int64_t B, int64_t K,
                                   int64_t N, int64_t accumulate, float* dxa, float* ws, void* stream) {
  if (!dy || !w || !dxa || !ws) return ADP_ERR_NULL;
  if (B <= 0 || B > LIN_BMAX || K <= 0 || N <= 0) return ADP_ERR_SHAPE;
  if (dy_bstride == 0) dy_bstride = N;
  const int64_t P = adp_cdiv(N, LIN_NR);
  if (P > 2147483647 || adp_cdiv(K, 256) > 65535) return ADP_ERR_SHAPE;
  LIN_DISPATCH(B, ADP_LAUNCH((linear_bwd_data_kernel<BT>), dim3((unsigned)P, (unsigned)adp_cdiv(K, 256)), dim3(256),
                             stream, dy, dy_bstride, w, B, K, N, ws));
  // second stage: the split-lane row reduction shared with the conv weight gradients (16 lanes per output)
  return adp_wgrad_reduce(ws, P, B * K, 0, dxa, nullptr, (int)accumulate, stream);
}

extern "C" int adp_linear_bwd_weight(const float* dy, int64_t dy_bstride, const float* x, int64_t B, int64_t K,
                                     int64_t N, int64_t act, int64_t accumulate, float* dw, float* dbias,
                                     void* stream) {
  if (!dy || !x || !dw) return ADP_ERR_NULL;
  if (B <= 0 || B > LIN_BMAX || K <= 0 || N <= 0) return ADP_ERR_SHAPE;
  if (act < 0 || act > 2) return ADP_ERR_UNSUPPORTED;
  if (dy_bstride == 0) dy_bstride = N;
  const int64_t rpw = lin_rpw(N);
  dim3 grid((unsigned)adp_cdiv(N, 4 * rpw));
  LIN_DISPATCH(B, ADP_LAUNCH((linear_bwd_weight_kernel<BT>), grid, dim3(256), stream, dy, dy_bstride, x, B, K, N,
                             (int)act, (int)accumulate, dw, dbias, (int)rpw));
  return ADP_LAUNCH_OK();
}
