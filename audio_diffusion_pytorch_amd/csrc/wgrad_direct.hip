// Weight (+bias) gradient of the narrow convolutions (M * R <= 256: the 2-8 channel ends of the U-Net) on the
// vector ALU -- the counterpart of conv_direct.hip for autograd's conv weight gradient
// (/root/reference/audio_diffusion_pytorch/components.py:84-99 at depth 0-1).
//
//   dw[m][r][t] = sum_{b,n} dy[b,m,n] * Xv[b, r, n*S + t - pad]        dbias[m] = sum_{b,n} dy[b,m,n]
//
// The output is tiny (<= 1024 numbers) and the reduction is long (B * 2**18 positions), so the kernel is a
// streaming reduction organised around HBM:
//   * a workgroup walks tiles of 256 output positions; dy [M x 256] and the activated / upsampled input
//     [R x (256*S + halo)] are staged in LDS with 16-byte global loads, every element fetched and activated once;
//   * a thread owns ONE (m, r) pair (all KT taps) and a 1/NSEG share of the tile's positions; it reads 16-byte
//     quads of its dy row and x row from LDS (lanes of a wave read 8-32 distinct rows, row stride 4 mod 8 dwords:
//     bank-conflict free; lanes sharing a row broadcast) and keeps its KT sums in registers for the whole kernel;
//   * at the end the NSEG position shares are summed through LDS and each workgroup writes one partial; the
//     partials are combined by adp_wgrad_reduce (fixed order: deterministic, no atomics).
#include "adp_rt.h"
#include "adp.h"
#include "conv_internal.h"

namespace {

constexpr int WD_TP = 256;        // output positions per tile
constexpr int WD_MAXBLOCKS = 1024;

template <int UP>
__device__ __forceinline__ f32x4 wd_load_xquad(const float* p) {
  f32x4 v;
  if (UP == 1) {
    v = *reinterpret_cast<const f32x4*>(p);
  } else if (UP == 2) {
    const f32x2 t = *reinterpret_cast<const f32x2*>(p);
    v[0] = t[0];
    v[1] = t[0];
    v[2] = t[1];
    v[3] = t[1];
  } else {
    const float t = *p;
    v[0] = t;
    v[1] = t;
    v[2] = t;
    v[3] = t;
  }
  return v;
}

// LDSF: floats of LDS (two size classes, so the 8 x 8 layers keep 8 workgroups per CU)
template <int KT, int S, int UP, int LDSF>
__global__ __launch_bounds__(256) void wgrad_direct_kernel(adp_wgrad_desc d, int NP2, int TPB, int ntiles) {
  constexpr int TP = WD_TP;
  constexpr int PAD = (S == 1) ? (KT - 1) / 2 : 0;
  constexpr int HALO = (S == 1 && KT > 1) ? 4 : 0;
  constexpr int XW = TP * S + 2 * HALO;      // staged (virtual) positions per x row
  constexpr int DS = TP + 4, XS = XW + 4;    // row strides (floats), 4 mod 8
  constexpr int NXQ = (S == 1) ? (KT > 1 ? 3 : 1) : S;  // 16-byte quads of x per 4 outputs
  __shared__ __attribute__((aligned(16))) float smem[LDSF];

  const int tid = threadIdx.x;
  const int M = (int)d.M, R = (int)d.R, R1 = (int)d.R1, L = (int)d.Lin, N = (int)d.N, G = (int)d.groups;
  const int Lv = L * UP;
  float* Dys = smem;
  float* Xs = smem + M * DS;

  const int pair = tid % NP2, seg = tid / NP2, nseg = 256 / NP2;
  const bool active = pair < M * R;
  const int m = active ? pair / R : 0, r = active ? pair % R : 0;
  const int qps = (TP / 4) / nseg;  // position quads per thread per tile

  float acc[KT];
#pragma unroll
  for (int t = 0; t < KT; ++t) acc[t] = 0.0f;
  float bsum = 0.0f;

  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int b = tile / TPB, p0 = (tile - b * TPB) * TP;
    // ---- stage dy [M][TP] (zero beyond N)
    for (int e = tid; e < M * (TP / 4); e += 256) {
      const int row = e / (TP / 4), q = e - row * (TP / 4);
      const int n = p0 + 4 * q;
      f32x4 v;
      if (n < N) {
        v = *reinterpret_cast<const f32x4*>(d.dy + ((int64_t)b * M + row) * N + n);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = 0.0f;
      }
      *reinterpret_cast<f32x4*>(Dys + row * DS + 4 * q) = v;
    }
    // ---- stage x [R][XW]: virtual positions p0*S - HALO ..., prologue applied, zero outside [0, Lv)
    for (int e = tid; e < R * (XW / 4); e += 256) {
      const int row = e / (XW / 4), q = e - row * (XW / 4);
      const int u = p0 * S - HALO + 4 * q;
      f32x4 v;
      if (u >= 0 && u < Lv) {
        const float* src = (row < R1) ? d.x + ((int64_t)b * R1 + row) * L
                                      : d.x2 + ((int64_t)b * (R - R1) + (row - R1)) * L;
        v = wd_load_xquad<UP>(src + u / UP);
        if (d.prologue == 1) {
          const int g = row / (R / G);
          const float mean = d.pro_stats[((int64_t)b * G + g) * 2];
          const float pa = (d.pro_gamma ? d.pro_gamma[row] : 1.0f) * d.pro_stats[((int64_t)b * G + g) * 2 + 1];
          const float pb = (d.pro_beta ? d.pro_beta[row] : 0.0f) - mean * pa;
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = adp_silu_fast(fmaf(v[j], pa, pb));
        }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = 0.0f;
      }
      *reinterpret_cast<f32x4*>(Xs + row * XS + 4 * q) = v;
    }
    __syncthreads();
    // ---- this thread's (m, r) pair over its share of the tile
    const float* dp = Dys + m * DS + seg * qps * 4;
    const float* xp = Xs + r * XS + seg * qps * 4 * S;
    for (int q = 0; q < qps; ++q) {
      const f32x4 dq = *reinterpret_cast<const f32x4*>(dp + 4 * q);
      float xq[4 * NXQ];
#pragma unroll
      for (int k = 0; k < NXQ; ++k) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(xp + 4 * q * S + 4 * k + ((S == 1 && KT == 1) ? HALO : 0));
#pragma unroll
        for (int j = 0; j < 4; ++j) xq[4 * k + j] = v[j];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int t = 0; t < KT; ++t) {
          const int xi = (S == 1) ? j + HALO - PAD + t : j * S + t;
          acc[t] = fmaf(dq[j], xq[xi], acc[t]);
        }
        bsum += dq[j];
      }
    }
    __syncthreads();
  }

  // ---- sum the position shares in a fixed order, one partial per workgroup
  float* red = smem;  // [KT + 1][256]
#pragma unroll
  for (int t = 0; t < KT; ++t) red[t * 256 + tid] = acc[t];
  red[KT * 256 + tid] = bsum;
  __syncthreads();
  if (seg == 0 && active) {
    const int64_t cnt = (int64_t)M * R * KT;
    float* wsw = d.ws + (int64_t)blockIdx.x * cnt;
#pragma unroll
    for (int t = 0; t < KT; ++t) {
      float s = 0.0f;
      for (int g = 0; g < nseg; ++g) s += red[t * 256 + g * NP2 + pair];
      wsw[((int64_t)m * R + r) * KT + t] = s;
    }
    if (r == 0 && d.dbias) {
      float s = 0.0f;
      for (int g = 0; g < nseg; ++g) s += red[KT * 256 + g * NP2 + pair];
      d.ws[(int64_t)gridDim.x * cnt + (int64_t)blockIdx.x * M + m] = s;
    }
  }
}

struct WdPlan {
  int np2, tpb, ntiles, blocks;
  size_t lds;
};

WdPlan wd_plan(const adp_wgrad_desc& d) {
  WdPlan p;
  int np2 = 4;
  while (np2 < d.M * d.R) np2 <<= 1;
  p.np2 = np2;
  p.tpb = (int)adp_cdiv(d.N, WD_TP);
  p.ntiles = (int)(d.B * p.tpb);
  p.blocks = p.ntiles < WD_MAXBLOCKS ? p.ntiles : WD_MAXBLOCKS;
  const int S = (int)d.stride, halo = (S == 1 && d.KT > 1) ? 4 : 0;
  const size_t stage = ((size_t)d.M * (WD_TP + 4) + (size_t)d.R * (WD_TP * S + 2 * halo + 4)) * sizeof(float);
  const size_t red = (size_t)(d.KT + 1) * 256 * sizeof(float);
  p.lds = stage > red ? stage : red;
  return p;
}

constexpr int WD_LDS_SMALL = 5120, WD_LDS_BIG = 16640;  // floats: 20 KiB / 65 KiB (the 8 -> 32 channel kernel-4 stride-4
                                                         // DownsampleItem needs 16544: two workgroups per CU either way)

template <int KT, int S, int UP>
int launch_wd(const adp_wgrad_desc& d, void* stream) {
  const WdPlan p = wd_plan(d);
  if (p.lds <= WD_LDS_SMALL * sizeof(float))
    ADP_LAUNCH((wgrad_direct_kernel<KT, S, UP, WD_LDS_SMALL>), dim3((unsigned)p.blocks), dim3(256), stream, d, p.np2,
               p.tpb, p.ntiles);
  else
    ADP_LAUNCH((wgrad_direct_kernel<KT, S, UP, WD_LDS_BIG>), dim3((unsigned)p.blocks), dim3(256), stream, d, p.np2,
               p.tpb, p.ntiles);
  if (ADP_LAUNCH_OK() != ADP_OK) return ADP_ERR_LAUNCH;
  return adp_wgrad_reduce(d.ws, p.blocks, d.M * d.R * d.KT, d.M, d.dw, d.dbias, (int)d.accumulate, stream);
}

}  // namespace

bool adp_wgrad_direct_eligible(const adp_wgrad_desc& d) {
  if (d.dil != 1 || d.M * d.R > 256 || (d.prologue != 0 && d.prologue != 1)) return false;
  const bool s1 = d.stride == 1 && (d.KT == 1 || d.KT == 3) && d.pad == (d.KT - 1) / 2 &&
                  (d.up == 1 || d.up == 2 || d.up == 4);
  const bool down = (d.stride == 2 || d.stride == 4) && d.KT == d.stride && d.pad == 0 && d.up == 1;
  if (!s1 && !down) return false;
  if (d.N % 4 != 0 || (d.Lin * d.up) % 4 != 0 || d.N * d.stride != d.Lin * d.up) return false;
  if ((reinterpret_cast<uintptr_t>(d.x) | reinterpret_cast<uintptr_t>(d.dy) | reinterpret_cast<uintptr_t>(d.x2)) & 15)
    return false;
  if (d.B * adp_cdiv(d.N, WD_TP) >= (int64_t)1 << 31 || d.Lin * d.up >= (int64_t)1 << 30) return false;
  return wd_plan(d).lds <= WD_LDS_BIG * sizeof(float);
}

int64_t adp_wgrad_direct_ws_floats(const adp_wgrad_desc& d) {
  return (int64_t)wd_plan(d).blocks * (d.M * d.R * d.KT + d.M);
}

int adp_wgrad_direct(const adp_wgrad_desc& d, void* stream) {
  if (d.stride == 2) return launch_wd<2, 2, 1>(d, stream);
  if (d.stride == 4) return launch_wd<4, 4, 1>(d, stream);
  if (d.KT == 3) {
    if (d.up == 2) return launch_wd<3, 1, 2>(d, stream);
    if (d.up == 4) return launch_wd<3, 1, 4>(d, stream);
    return launch_wd<3, 1, 1>(d, stream);
  }
  if (d.up == 2) return launch_wd<1, 1, 2>(d, stream);
  if (d.up == 4) return launch_wd<1, 1, 4>(d, stream);
  return launch_wd<1, 1, 1>(d, stream);
}
