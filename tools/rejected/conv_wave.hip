// Wave-tile Conv1d for gfx950: every wave owns one [32*MT output channels x 64 positions] tile and feeds the f32 matrix
// cores STRAIGHT from global memory (L2) -- no LDS staging, no loader waves, no barrier in the K loop.
// ResnetItem ConvBlock convs and their data gradients (kernel 1 / 3, stride 1; /root/reference/audio_diffusion_pytorch/
// components.py:89) at the channel counts where the staged kernel (conv_mm_impl.h) spends its time
// filling and draining: C = 64 .. 256 has 2-8 chunks of K per block, so a block lives for a few microseconds and
// most of that is the first load's latency, the barriers and the K-group exchange (round-1 profile: 33-57 TF).
//
// Why this shape fits MI355X: v_mfma_f32_32x32x2_f32 takes ONE float per lane per operand, and both operand patterns are
// natural global accesses --
//   A (weights)  lane (m = l31, kk = hi) needs w[m][r .. r+3][0 .. KT-1] for its 4 channels of an 8-channel group: 12
//                contiguous floats = three 16-byte loads per lane (forward view); the gradient view w[r][m][KT-1-t] is
//                a 12-byte run per lane, coalesced across the 32 lanes of a half-wave;
//   X            lane (kk = hi, n = l31) needs x[r][n0 + l31 + t - pad]: coalesced 128-byte half-wave rows, the KT taps
//                are overlapping rows served by the vector L1.
// A wave keeps TWO groups of operands in registers: while the 24*MT MFMAs of group g issue, the loads of group g+1
// are in flight; nothing else synchronises.  Weights (48-786 KB) and the activation tile are L2 resident and are read by
// exactly the waves that need them, once (no redundancy inside a workgroup: its 4 waves own 4 adjacent position tiles,
// or -- deep layers, NKW = 4 -- the 4 interleaved quarter-slices of K, summed through LDS at the very end).
// GroupNorm + SiLU (prologue 1) is applied in registers between the load and the MFMA from a per-(b, channel) table the
// workgroup builds once in LDS; zero padding is applied after the activation, like nn.Conv1d.
#include <stdlib.h>
#include "adp_rt.h"
#include "adp.h"
#include "conv_internal.h"

namespace {

constexpr int WV_RMAX = 1024;  // channels whose GroupNorm constants fit the LDS table

// KT: taps (1 | 3); TR: data gradient (transposed weight view); PRO: 0 | 1 (GroupNorm+SiLU); MT: 32-row tiles per wave;
// NKW: waves of a workgroup that split K (1: four independent position tiles per workgroup; 4: one tile, K in quarters)
template <int KT, bool TR, int PRO, int MT, int NKW>
__global__ __launch_bounds__(256) void conv_wave_kernel(adp_conv_desc d) {
  __shared__ float red[NKW > 1 ? 4 * MT * 2 * 1024 : 1];
  __shared__ float Pa[PRO == 1 ? WV_RMAX : 1], Pb[PRO == 1 ? WV_RMAX : 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hi = lane >> 5, l31 = lane & 31;
  const int M = (int)d.M, R = (int)d.R, L = (int)d.Lin, N = (int)d.N, pad = (int)d.pad;
  const int b = blockIdx.z, m0 = blockIdx.y * (32 * MT);
  const int nt = NKW > 1 ? blockIdx.x : blockIdx.x * 4 + wave;
  const int n0 = nt * 64;

  if (PRO == 1) {
    const int cpg = R / (int)d.groups;
    for (int r = tid; r < R; r += 256) {
      const int g = r / cpg;
      const float mean = d.pro_stats[((int64_t)b * d.groups + g) * 2];
      const float ga = (d.pro_gamma ? d.pro_gamma[r] : 1.0f) * d.pro_stats[((int64_t)b * d.groups + g) * 2 + 1];
      Pa[r] = ga;
      Pb[r] = (d.pro_beta ? d.pro_beta[r] : 0.0f) - mean * ga;
    }
    __syncthreads();
  }
  const bool active = n0 < N;  // a trailing wave of the last workgroup may have no tile (it still meets the barriers)

  f32x16 acc[MT][2];
#pragma unroll
  for (int mi = 0; mi < MT; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.0f;

  if (active) {
    const float* xb = d.x + (int64_t)b * R * L;
    // X: per (ni, t) a clamped position offset and a validity flag (zero padding / ragged end)
    int xoff[2][KT];
    bool xok[2][KT];
    bool interior = true;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int t = 0; t < KT; ++t) {
        const int p = n0 + 32 * ni + l31 + t - pad;
        xok[ni][t] = p >= 0 && p < L;
        xoff[ni][t] = p < 0 ? 0 : (p < L ? p : L - 1);
        interior = interior && xok[ni][t];
      }
    interior = __all(interior);  // wave-uniform: interior tiles skip the masking altogether
    // A: forward  w[m][r][t]      -> lane base (m0 + l31) * R * KT + 4 hi KT   (+ 32 mi R KT, + 8 g KT; 12 floats)
    //    gradient w[r][m][KT-1-t] -> lane base (4 hi * M + m0 + l31) * KT      (+ 32 mi KT, + (8 g + c) M KT; 3 floats)
    const float* wl = TR ? d.w + ((int64_t)4 * hi * M + m0 + l31) * KT : d.w + ((int64_t)(m0 + l31) * R + 4 * hi) * KT;
    const float* xl = xb + (int64_t)4 * hi * L;  // channel 8 g + c + 4 hi

    const int ngroups = R / 8;
    const int g_first = NKW > 1 ? wave : 0, g_step = NKW > 1 ? NKW : 1;

    auto load_group = [&](float (&a)[MT][4 * KT], float (&x)[2][4 * KT], int g) {
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) {
        if (!TR) {
          const float* ap = wl + (int64_t)mi * 32 * R * KT + g * 8 * KT;
          if (KT == 3) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
              const f32x4 q = *reinterpret_cast<const f32x4*>(ap + 4 * j);
#pragma unroll
              for (int k = 0; k < 4; ++k) a[mi][4 * j + k] = q[k];
            }
          } else {
            const f32x4 q = *reinterpret_cast<const f32x4*>(ap);
#pragma unroll
            for (int k = 0; k < 4; ++k) a[mi][k] = q[k];
          }
        } else {
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const float* ap = wl + ((int64_t)(g * 8 + c) * M + mi * 32) * KT;
#pragma unroll
            for (int t = 0; t < KT; ++t) a[mi][c * KT + t] = ap[KT - 1 - t];
          }
        }
      }
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int t = 0; t < KT; ++t) x[ni][c * KT + t] = xl[(int64_t)(g * 8 + c) * L + xoff[ni][t]];
    };
    auto mma_group = [&](const float (&a)[MT][4 * KT], float (&x)[2][4 * KT], int g) {
      if (PRO == 1) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float pa = Pa[g * 8 + c + 4 * hi], pb = Pb[g * 8 + c + 4 * hi];
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int t = 0; t < KT; ++t) x[ni][c * KT + t] = adp_silu_fast(fmaf(x[ni][c * KT + t], pa, pb));
        }
      }
      if (!interior) {  // zero padding is applied after the activation
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
          for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int t = 0; t < KT; ++t) x[ni][c * KT + t] = xok[ni][t] ? x[ni][c * KT + t] : 0.0f;
      }
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int t = 0; t < KT; ++t)
#pragma unroll
          for (int mi = 0; mi < MT; ++mi) {
            acc[mi][0] = adp_mfma32(a[mi][c * KT + t], x[0][c * KT + t], acc[mi][0]);
            acc[mi][1] = adp_mfma32(a[mi][c * KT + t], x[1][c * KT + t], acc[mi][1]);
          }
    };

    float a0[MT][4 * KT], x0[2][4 * KT], a1[MT][4 * KT], x1[2][4 * KT];
    if (g_first < ngroups) load_group(a0, x0, g_first);
    for (int g = g_first; g < ngroups; g += 2 * g_step) {  // two groups per trip: the register sets swap statically
      if (g + g_step < ngroups) load_group(a1, x1, g + g_step);
#ifndef ADP_EMULATE
      __builtin_amdgcn_sched_barrier(0);  // keep the next group's loads ahead of this group's matrix work
#endif
      mma_group(a0, x0, g);
      if (g + g_step < ngroups) {
        if (g + 2 * g_step < ngroups) load_group(a0, x0, g + 2 * g_step);
#ifndef ADP_EMULATE
        __builtin_amdgcn_sched_barrier(0);
#endif
        mma_group(a1, x1, g + g_step);
      }
    }
  }

  // ---- K-quarter exchange (NKW = 4): every wave parks its tiles, then finishes a quarter of the accumulator rows
  constexpr int RPW = NKW > 1 ? 16 / NKW : 16;
  if (NKW > 1) {
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        float* rp = red + ((wave * MT + mi) * 2 + ni) * 1024 + lane;
#pragma unroll
        for (int r = 0; r < 16; ++r) rp[r * 64] = acc[mi][ni][r];
      }
    __syncthreads();
  }
  if (!active) return;

  // ---- epilogue (adp_conv1d's store-0 contract: bias, out_pre, e_scale, residual)
  const int64_t ebs = d.e_bstride ? d.e_bstride : M;
#pragma unroll
  for (int mi = 0; mi < MT; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int n = n0 + ni * 32 + l31;
#pragma unroll
      for (int rr = 0; rr < RPW; ++rr) {
        const int r = NKW > 1 ? wave * RPW + rr : rr;
        float v;
        if (NKW > 1) {
          v = 0.0f;
#pragma unroll
          for (int k = 0; k < NKW; ++k) v += red[((k * MT + mi) * 2 + ni) * 1024 + r * 64 + lane];
        } else {
          v = acc[mi][ni][rr];
        }
        const int m = m0 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (n < N) {
          if (d.bias) v += d.bias[m];
          const int64_t o = ((int64_t)b * M + m) * N + n;
          if (d.out_pre) d.out_pre[o] = v;
          if (d.e_scale) v *= d.e_scale[b * ebs + m];
          if (d.res) v += d.res[o];
          d.out[o] = v;
        }
      }
    }
}

template <int KT, bool TR, int PRO, int MT>
int launch_wave(const adp_conv_desc& d, int nkw, void* stream) {
  const unsigned ntn = (unsigned)adp_cdiv(d.N, 64);
  const dim3 grid(nkw > 1 ? ntn : (ntn + 3) / 4, (unsigned)(d.M / (32 * MT)), (unsigned)d.B);
  if (nkw > 1)
    ADP_LAUNCH((conv_wave_kernel<KT, TR, PRO, MT, 4>), grid, dim3(256), stream, d);
  else
    ADP_LAUNCH((conv_wave_kernel<KT, TR, PRO, MT, 1>), grid, dim3(256), stream, d);
  return ADP_LAUNCH_OK();
}

// waves the launch would run with MT-row-tile waves and K split nkw ways
int64_t wave_count(const adp_conv_desc& d, int mt, int nkw) { return (d.M / (32 * mt)) * adp_cdiv(d.N, 64) * d.B * nkw; }

int wave_mode() {
  static const int mode = [] {
    const char* e = getenv("ADP_CONV_WAVE");
    return e ? atoi(e) : 1;   // 0: off, 1: short-K layers (C <= 256), 2: every eligible shape
  }();
  return mode;
}

}  // namespace

bool adp_conv_wave_eligible(const adp_conv_desc& d) {
  const int mode = wave_mode();
  if (mode == 0) return false;
  if (d.R1 != d.R || d.x2) return false;
  if (d.stride != 1 || d.up != 1 || d.dil != 1 || d.store != 0) return false;
  if (d.KT != 1 && d.KT != 3) return false;
  if (d.pad < 0 || d.pad > d.KT - 1) return false;
  if (d.N != d.Lin + 2 * d.pad - (d.KT - 1)) return false;
  if (d.prologue != 0 && d.prologue != 1) return false;
  if (d.prologue == 1 && (d.R > WV_RMAX || d.groups < 1 || d.R % d.groups != 0)) return false;
  if (d.R % 32 != 0 || d.M % 64 != 0) return false;
  if ((reinterpret_cast<uintptr_t>(d.w) & 15) != 0) return false;
  if (d.B * d.R * d.Lin >= (int64_t)1 << 31 || d.M * d.R * d.KT >= (int64_t)1 << 31) return false;
  if (d.B > 65535 || d.M / 64 > 65535) return false;
  if (mode == 1 && (d.R > 256 || wave_count(d, 2, 1) < 512)) return false;
  return true;
}

int adp_conv_wave(const adp_conv_desc& d, void* stream) {
  // enough independent tiles for >= 2 waves per CU: one wave per tile; otherwise the 4 waves of a workgroup split K
  const int nkw = (wave_count(d, 2, 1) >= 512 || d.R < 128) ? 1 : 4;
  const bool tr = d.transposed != 0;
  if (d.KT == 3) {
    if (d.prologue == 1)
      return tr ? launch_wave<3, true, 1, 2>(d, nkw, stream) : launch_wave<3, false, 1, 2>(d, nkw, stream);
    return tr ? launch_wave<3, true, 0, 2>(d, nkw, stream) : launch_wave<3, false, 0, 2>(d, nkw, stream);
  }
  if (d.prologue == 1)
    return tr ? launch_wave<1, true, 1, 2>(d, nkw, stream) : launch_wave<1, false, 1, 2>(d, nkw, stream);
  return tr ? launch_wave<1, true, 0, 2>(d, nkw, stream) : launch_wave<1, false, 0, 2>(d, nkw, stream);
}
