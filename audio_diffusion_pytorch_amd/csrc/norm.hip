// Normalisation family for gfx950 (HBM-bound streaming + wave-shuffle / LDS reductions):
//   * GroupNorm statistics and the backward of SiLU(GroupNorm(x))        (a_unet ConvBlock; components.py:89)
//   * Modulation: per-position LayerNorm over channels * (1+scale) + shift  (a_unet ModulationItem; components.py:90)
//   * LayerNorm-over-channels statistics / backward for the attention projections (components.py:92-93)
//   * SkipModulate backward                                              (components.py:99)
// Layout [B, C, L], L fastest: lanes always run along L so every global access is a coalesced 256-B wave load.
// All cross-workgroup reductions are two-stage through a caller-provided workspace (deterministic, no atomics).
#include <stdlib.h>
#include "adp_rt.h"
#include "adp.h"

namespace {

constexpr int GN_CHUNK = 4096;  // elements per partial-statistics workgroup (16 per thread, register resident)

// ---- GroupNorm statistics: chunk-local (mean, M2) then Chan's exact combination ------------------------
__global__ __launch_bounds__(256) void gn_partial_kernel(const float* x, int64_t NG, int64_t nchunks, float* ws) {
  __shared__ float sh[4];
  const int64_t bg = blockIdx.y, c = blockIdx.x;
  const float* base = x + bg * NG + c * GN_CHUNK;
  const int64_t cnt = (NG - c * GN_CHUNK) < GN_CHUNK ? (NG - c * GN_CHUNK) : GN_CHUNK;
  float v[16];
  float s = 0.0f;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int64_t i = (int64_t)j * 256 + threadIdx.x;
    v[j] = (i < cnt) ? base[i] : 0.0f;
    s += v[j];
  }
  const float mean = adp_block_sum<4>(s, sh) / (float)cnt;
  float q = 0.0f;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int64_t i = (int64_t)j * 256 + threadIdx.x;
    const float dlt = v[j] - mean;
    q += (i < cnt) ? dlt * dlt : 0.0f;
  }
  const float m2 = adp_block_sum<4>(q, sh);
  if (threadIdx.x == 0) {
    ws[(bg * nchunks + c) * 2] = mean;
    ws[(bg * nchunks + c) * 2 + 1] = m2;
  }
}

// (A "last workgroup combines the partials" ticket was measured here: the agent-scope release/acquire it needs
// per workgroup -- L2 write-back + invalidate, the per-XCD L2s are not coherent -- made the 8192-workgroup depth-1
// launch 4x slower than this second tiny launch.)
__global__ __launch_bounds__(64) void gn_final_kernel(const float* ws, int64_t NG, int64_t nchunks, float eps,
                                                      float* stats) {
  const int64_t bg = blockIdx.x;
  const int lane = threadIdx.x;
  float s = 0.0f;
  for (int64_t c = lane; c < nchunks; c += 64) {
    const int64_t cnt = (NG - c * GN_CHUNK) < GN_CHUNK ? (NG - c * GN_CHUNK) : GN_CHUNK;
    s += ws[(bg * nchunks + c) * 2] * (float)cnt;
  }
  const float mean = adp_wave_sum(s) / (float)NG;
  float q = 0.0f;
  for (int64_t c = lane; c < nchunks; c += 64) {
    const int64_t cnt = (NG - c * GN_CHUNK) < GN_CHUNK ? (NG - c * GN_CHUNK) : GN_CHUNK;
    const float dm = ws[(bg * nchunks + c) * 2] - mean;
    q += ws[(bg * nchunks + c) * 2 + 1] + (float)cnt * dm * dm;
  }
  q = adp_wave_sum(q);
  if (lane == 0) {
    stats[bg * 2] = mean;
    stats[bg * 2 + 1] = 1.0f / sqrtf(q / (float)NG + eps);
  }
}

// Second stage of the statistics fused with the activation: a = SiLU(GroupNorm(x)) materialised.  Used by the WIDE
// layers only (C >= 512): there an activation is staged by 8-16 workgroups of the implicit-GEMM conv / weight-gradient
// kernels, so recomputing exp + rcp in every one of their loaders costs more MFMA issue time (measured: 0.7 us of a
// 4.5 us weight-gradient chunk at depth 7) than writing the 2-8 MB tensor once.  One workgroup = 1024 elements of one
// (b, c) row; every workgroup re-derives its group's (mean, rstd) from the partials (a handful at these sizes) and
// the first workgroup of each group also publishes them for the backward pass.
__global__ __launch_bounds__(256) void gn_apply_kernel(const float* x, const float* ws, const float* gamma,
                                                       const float* beta, int64_t C, int64_t L, int64_t G,
                                                       int64_t nchunks, float eps, float* stats, float* a) {
  const int64_t row = blockIdx.y, b = row / C, c = row % C, Cg = C / G, g = c / Cg, bg = b * G + g;
  const int64_t NG = Cg * L;
  // Chan's combination of the group's partials, in gn_final_kernel's order of operations (wave 0's lanes; every
  // thread then reads the result from LDS)
  __shared__ float st[2];
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    float s = 0.0f;
    for (int64_t k = lane; k < nchunks; k += 64) {
      const int64_t n = (NG - k * GN_CHUNK) < GN_CHUNK ? (NG - k * GN_CHUNK) : GN_CHUNK;
      s += ws[(bg * nchunks + k) * 2] * (float)n;
    }
    const float mean = adp_wave_sum(s) / (float)NG;
    float q = 0.0f;
    for (int64_t k = lane; k < nchunks; k += 64) {
      const int64_t n = (NG - k * GN_CHUNK) < GN_CHUNK ? (NG - k * GN_CHUNK) : GN_CHUNK;
      const float dm = ws[(bg * nchunks + k) * 2] - mean;
      q += ws[(bg * nchunks + k) * 2 + 1] + (float)n * dm * dm;
    }
    q = adp_wave_sum(q);
    if (lane == 0) {
      const float rstd = 1.0f / sqrtf(q / (float)NG + eps);
      st[0] = mean;
      st[1] = rstd;
      if (c == g * Cg && blockIdx.x == 0) {
        stats[bg * 2] = mean;
        stats[bg * 2 + 1] = rstd;
      }
    }
  }
  __syncthreads();
  const float pa = gamma[c] * st[1], pb = beta[c] - st[0] * pa;
  const int64_t l0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  const float* xr = x + row * L;
  float* ar = a + row * L;
  if (l0 + 3 < L && (L & 3) == 0) {
    f32x4 v = *reinterpret_cast<const f32x4*>(xr + l0);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = adp_silu_fast(fmaf(v[j], pa, pb));
    *reinterpret_cast<f32x4*>(ar + l0) = v;
  } else {
    for (int64_t l = l0; l < L && l < l0 + 4; ++l) ar[l] = adp_silu_fast(fmaf(xr[l], pa, pb));
  }
}

// ---- statistics from producer-side partials ------------------------------------------------------------------------
// part[((b * C/4 + cq) * E + e) * 3 + {0,1,2}] = (mean, M2, count) of slice e of the 4-channel row quad cq of batch
// element b.  Chan's combination over the (C/G)/4 quads x E slices of a group (contiguous in memory), one wave.
__device__ __forceinline__ void gn_combine_wave(const float* p, int64_t cnt, int lane, float eps, float& mean,
                                                float& rstd) {
  // the first four entries of a lane stay in registers for the second pass (the deep layers have 64-256 entries per group:
  // one trip to memory instead of two)
  float pm[4], pq[4], pn[4];
  float s = 0.0f, n = 0.0f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int64_t i = lane + 64 * k;
    pm[k] = pq[k] = pn[k] = 0.0f;
    if (i < cnt) {
      pm[k] = p[i * 3];
      pq[k] = p[i * 3 + 1];
      pn[k] = p[i * 3 + 2];
    }
    s = fmaf(pm[k], pn[k], s);
    n += pn[k];
  }
  for (int64_t i = lane + 256; i < cnt; i += 64) {
    s = fmaf(p[i * 3], p[i * 3 + 2], s);
    n += p[i * 3 + 2];
  }
  s = adp_wave_sum(s);
  n = adp_wave_sum(n);
  mean = s / n;
  float q = 0.0f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float dm = pm[k] - mean;
    q += pq[k] + pn[k] * dm * dm;
  }
  for (int64_t i = lane + 256; i < cnt; i += 64) {
    const float dm = p[i * 3] - mean;
    q += p[i * 3 + 1] + p[i * 3 + 2] * dm * dm;
  }
  q = adp_wave_sum(q);
  rstd = 1.0f / sqrtf(q / n + eps);
}

__global__ __launch_bounds__(64) void gn_finalize_kernel(const float* part, int64_t C, int64_t E, int64_t G, float eps,
                                                         float* stats) {
  const int64_t bg = blockIdx.x, b = bg / G, g = bg % G, Qg = (C / G) / 4;
  float mean, rstd;
  gn_combine_wave(part + ((b * (C / 4) + g * Qg) * E) * 3, Qg * E, threadIdx.x, eps, mean, rstd);
  if (threadIdx.x == 0) {
    stats[bg * 2] = mean;
    stats[bg * 2 + 1] = rstd;
  }
}

// Statistics from the producer's partials AND the materialised activation in ONE launch (wide layers, few slices per
// row): every workgroup (1024 elements of one (b, c) row) combines its group's Cg * E entries itself -- a few
// hundred at these sizes -- and the first workgroup of each group publishes (mean, rstd) for the backward pass.
__global__ __launch_bounds__(256) void gn_finalize_act_kernel(const float* x, const float* part, const float* gamma,
                                                              const float* beta, int64_t C, int64_t L, int64_t G,
                                                              int64_t E, float eps, float* stats, float* a) {
  __shared__ float st[2];
  const int64_t row = blockIdx.y, b = row / C, c = row % C, Cg = C / G, g = c / Cg, bg = b * G + g;
  if (threadIdx.x < 64) {
    const int64_t Qg = Cg / 4;
    float mean, rstd;
    gn_combine_wave(part + ((b * (C / 4) + g * Qg) * E) * 3, Qg * E, threadIdx.x, eps, mean, rstd);
    if (threadIdx.x == 0) {
      st[0] = mean;
      st[1] = rstd;
      if (c == g * Cg && blockIdx.x == 0) {
        stats[bg * 2] = mean;
        stats[bg * 2 + 1] = rstd;
      }
    }
  }
  __syncthreads();
  const float pa = gamma[c] * st[1], pb = beta[c] - st[0] * pa;
  const int64_t l0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  const float* xr = x + row * L;
  float* ar = a + row * L;
  if (l0 + 3 < L && (L & 3) == 0) {
    f32x4 v = *reinterpret_cast<const f32x4*>(xr + l0);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = adp_silu_fast(fmaf(v[j], pa, pb));
    *reinterpret_cast<f32x4*>(ar + l0) = v;
  } else {
    for (int64_t l = l0; l < L && l < l0 + 4; ++l) ar[l] = adp_silu_fast(fmaf(xr[l], pa, pb));
  }
}

// a = SiLU(GroupNorm(x)) from finished statistics; one workgroup = 1024 elements of one (b, c) row
__global__ __launch_bounds__(256) void gn_act_kernel(const float* x, const float* stats, const float* gamma,
                                                     const float* beta, int64_t C, int64_t L, int64_t G, float* a) {
  const int64_t row = blockIdx.y, b = row / C, c = row % C, g = c / (C / G);
  const float mean = stats[(b * G + g) * 2], rstd = stats[(b * G + g) * 2 + 1];
  const float pa = gamma[c] * rstd, pb = beta[c] - mean * pa;
  const int64_t l0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  const float* xr = x + row * L;
  float* ar = a + row * L;
  if (l0 + 3 < L && (L & 3) == 0) {
    f32x4 v = *reinterpret_cast<const f32x4*>(xr + l0);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = adp_silu_fast(fmaf(v[j], pa, pb));
    *reinterpret_cast<f32x4*>(ar + l0) = v;
  } else {
    for (int64_t l = l0; l < L && l < l0 + 4; ++l) ar[l] = adp_silu_fast(fmaf(xr[l], pa, pb));
  }
}

// ---- the three materialising kernels above, slab form (L % 4 == 0) -------------------------------------------------------
// A group's Cg rows are one contiguous run of Cg * L floats, so a workgroup takes 256 * NV float4 of that run -- whatever
// the row length -- instead of one (mostly idle) workgroup per 1024-element piece of a row: the deep layers' rows are
// 128-256 floats long, where the row form ran 4096 workgroups with a quarter of their threads busy, each re-deriving the
// group's statistics (10.9 us for a 4 MB tensor).  SRC: 0 = finished statistics, 1 = the producer's (mean, M2, count)
// partials (gn_finalize_act), 2 = gn_partial_kernel's chunk partials (gn_stats_act).
template <int SRC, int NV>
__global__ __launch_bounds__(256) void gn_act_slab_kernel(const float* __restrict__ x, const float* __restrict__ src,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          int64_t C, int64_t L, int64_t G, int64_t E, float eps,
                                                          float* __restrict__ stats, float* __restrict__ a) {
  __shared__ float st[2];
  const int64_t bg = blockIdx.y, b = bg / G, g = bg % G, Cg = C / G, NG = Cg * L;
  // Every load of the kernel is requested BEFORE the statistics are combined: the deep layers' launches of this kernel move
  // 2-8 MB and are bound by memory round trips, not by bandwidth -- with the tile requested behind the combination (its
  // loads, a wave reduction, a barrier) a launch was two round trips long.
  const float* xs = x + bg * NG;
  float* as = a + bg * NG;
  f32x4 v[NV];
  int64_t e[NV];
  float gm[NV], bt[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    e[j] = (((int64_t)blockIdx.x * NV + j) * 256 + threadIdx.x) * 4;
    if (e[j] < NG) {
      v[j] = *reinterpret_cast<const f32x4*>(xs + e[j]);
      const int64_t c = g * Cg + e[j] / L;  // L % 4 == 0: a quad never straddles two rows
      gm[j] = gamma[c];
      bt[j] = beta[c];
    }
  }
  if (SRC == 0) {
    if (threadIdx.x == 0) {
      st[0] = src[bg * 2];
      st[1] = src[bg * 2 + 1];
    }
  } else if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    float mean, rstd;
    if (SRC == 1) {
      const int64_t Qg = Cg / 4;
      gn_combine_wave(src + ((b * (C / 4) + g * Qg) * E) * 3, Qg * E, lane, eps, mean, rstd);
    } else {  // Chan's combination of gn_partial_kernel's chunks, in gn_final_kernel's order of operations
      float s = 0.0f;
      for (int64_t k = lane; k < E; k += 64) {
        const int64_t n = (NG - k * GN_CHUNK) < GN_CHUNK ? (NG - k * GN_CHUNK) : GN_CHUNK;
        s += src[(bg * E + k) * 2] * (float)n;
      }
      mean = adp_wave_sum(s) / (float)NG;
      float q = 0.0f;
      for (int64_t k = lane; k < E; k += 64) {
        const int64_t n = (NG - k * GN_CHUNK) < GN_CHUNK ? (NG - k * GN_CHUNK) : GN_CHUNK;
        const float dm = src[(bg * E + k) * 2] - mean;
        q += src[(bg * E + k) * 2 + 1] + (float)n * dm * dm;
      }
      q = adp_wave_sum(q);
      rstd = 1.0f / sqrtf(q / (float)NG + eps);
    }
    if (lane == 0) {
      st[0] = mean;
      st[1] = rstd;
      if (blockIdx.x == 0) {
        stats[bg * 2] = mean;
        stats[bg * 2 + 1] = rstd;
      }
    }
  }
  __syncthreads();
  const float mean = st[0], rstd = st[1];
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    if (e[j] >= NG) continue;
    const float pa = gm[j] * rstd, pb = bt[j] - mean * pa;
#pragma unroll
    for (int k = 0; k < 4; ++k) v[j][k] = adp_silu_fast(fmaf(v[j][k], pa, pb));
    *reinterpret_cast<f32x4*>(as + e[j]) = v[j];
  }
}

template <int SRC>
static int launch_gn_act_slab(const float* x, const float* src, const float* gamma, const float* beta, int64_t B,
                              int64_t C, int64_t L, int64_t G, int64_t E, float eps, float* stats, float* a, void* stream) {
  const int64_t NG = (C / G) * L;
  // 4096 elements per workgroup unless that leaves the chip short of workgroups (the 2-4 MB tensors of the deep layers)
  if (B * G * adp_cdiv(NG, 4096) >= 1024)
    ADP_LAUNCH((gn_act_slab_kernel<SRC, 4>), dim3((unsigned)adp_cdiv(NG, 4096), (unsigned)(B * G)), dim3(256), stream, x,
               src, gamma, beta, C, L, G, E, eps, stats, a);
  else
    ADP_LAUNCH((gn_act_slab_kernel<SRC, 1>), dim3((unsigned)adp_cdiv(NG, 1024), (unsigned)(B * G)), dim3(256), stream, x,
               src, gamma, beta, C, L, G, E, eps, stats, a);
  return ADP_LAUNCH_OK();
}

// ---- backward of y = SiLU(GN(x)) ------------------------------------------------------------------------
// ab[b, c, split, {A,B}] : A = sum ds*xhat, B = sum ds over the split's slice of L, ds = dact * silu'(h)
__global__ __launch_bounds__(256) void gn_bwd_reduce_kernel(const float* x, const float* dact, const float* stats,
                                                            const float* gamma, const float* beta, int64_t C,
                                                            int64_t L, int64_t G, int64_t NS, int64_t CL, float* ab) {
  __shared__ float sh[4];
  const int64_t row = blockIdx.y, split = blockIdx.x;
  const int64_t b = row / C, c = row % C, g = c / (C / G);
  const float mean = stats[(b * G + g) * 2], rstd = stats[(b * G + g) * 2 + 1];
  const float ga = gamma[c], be = beta[c];
  const int64_t lo = split * CL, hi = (lo + CL < L) ? lo + CL : L;
  const float* xr = x + row * L;
  const float* dr = dact + row * L;
  float a = 0.0f, bs = 0.0f;
  for (int64_t l = lo + threadIdx.x; l < hi; l += 256) {
    const float xh = (xr[l] - mean) * rstd;
    const float h = fmaf(xh, ga, be);
    const float ds = dr[l] * adp_dsilu(h);
    a = fmaf(ds, xh, a);
    bs += ds;
  }
  a = adp_block_sum<4>(a, sh);
  bs = adp_block_sum<4>(bs, sh);
  if (threadIdx.x == 0) {
    ab[(row * NS + split) * 2] = a;
    ab[(row * NS + split) * 2 + 1] = bs;
  }
}

__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const float* x, const float* dact, const float* stats,
                                                           const float* gamma, const float* beta, const float* ab,
                                                           const float* dres, int64_t C, int64_t L, int64_t G,
                                                           int64_t NS, int64_t CL, float* dx, int64_t B,
                                                           float* dgamma, float* dbeta, int accumulate, int64_t NSab) {
  __shared__ float sh[4];
  const int64_t row = blockIdx.y, split = blockIdx.x;
  const int64_t b = row / C, c = row % C, Cg = C / G, g = c / Cg;
  if (dgamma && b == 0 && split == 0) {
    // parameter gradients of this channel (what adp_gn_param_grad computes): dgamma = sum_{b,s} A, dbeta = sum B
    float pa = 0.0f, pb = 0.0f;
    for (int64_t e = threadIdx.x; e < B * NSab; e += 256) {
      const int64_t bb = e / NSab, spx = e % NSab;
      pa += ab[((bb * C + c) * NSab + spx) * 2];
      pb += ab[((bb * C + c) * NSab + spx) * 2 + 1];
    }
    pa = adp_block_sum<4>(pa, sh);
    pb = adp_block_sum<4>(pb, sh);
    if (threadIdx.x == 0) {
      dgamma[c] = accumulate ? dgamma[c] + pa : pa;
      dbeta[c] = accumulate ? dbeta[c] + pb : pb;
    }
  }
  const float mean = stats[(b * G + g) * 2], rstd = stats[(b * G + g) * 2 + 1];
  float sa = 0.0f, sb = 0.0f;
  for (int64_t e = threadIdx.x; e < Cg * NSab; e += 256) {
    const int64_t cc = g * Cg + e / NSab, spx = e % NSab;
    const float gm = gamma[cc];
    sa = fmaf(gm, ab[((b * C + cc) * NSab + spx) * 2], sa);
    sb = fmaf(gm, ab[((b * C + cc) * NSab + spx) * 2 + 1], sb);
  }
  const float inv = 1.0f / ((float)Cg * (float)L);
  const float m2 = adp_block_sum<4>(sa, sh) * inv;
  const float m1 = adp_block_sum<4>(sb, sh) * inv;
  const float ga = gamma[c], be = beta[c];
  const int64_t lo = split * CL, hi = (lo + CL < L) ? lo + CL : L;
  const float* xr = x + row * L;
  const float* dr = dact + row * L;
  for (int64_t l = lo + threadIdx.x; l < hi; l += 256) {
    const float xh = (xr[l] - mean) * rstd;
    const float h = fmaf(xh, ga, be);
    const float ds = dr[l] * adp_dsilu(h);
    float v = rstd * (ga * ds - m1 - xh * m2);
    if (dres) v += dres[row * L + l];
    dx[row * L + l] = v;
  }
}

// The same two stages with 16-byte accesses and TPR threads per (row, split) segment (L % 4 == 0, segment length
// CL % 4 == 0): a segment of the deep layers is 128-256 floats, one float4 per lane of ONE wave (TPR = 32 / 64: no LDS,
// no workgroup barrier, four segments per workgroup) where the row form spent a 256-thread workgroup with two
// barrier-synchronised block sums on it; long segments take the whole workgroup (TPR = 256) as before.
template <int TPR>
__device__ __forceinline__ float gn_seg_sum(float v, float* sh) {
  if (TPR == 256) return adp_block_sum<4>(v, sh);
#pragma unroll
  for (int o = 1; o < TPR; o <<= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <int TPR>
__global__ __launch_bounds__(256) void gn_bwd_reduce_vec_kernel(const float* x, const float* dact, const float* stats,
                                                                const float* gamma, const float* beta, int64_t C,
                                                                int64_t L, int64_t G, int64_t NS, int64_t CL,
                                                                int64_t nseg, float* ab) {
  __shared__ float sh[4];
  const int sl = threadIdx.x / TPR, li = threadIdx.x % TPR;
  const int64_t seg = (int64_t)blockIdx.x * (256 / TPR) + sl;
  const bool live = seg < nseg;
  const int64_t sg = live ? seg : nseg - 1;  // idle tail segments shadow the last one (uniform control flow)
  const int64_t row = sg / NS, split = sg % NS;
  const int64_t b = row / C, c = row % C, g = c / (C / G);
  const float mean = stats[(b * G + g) * 2], rstd = stats[(b * G + g) * 2 + 1];
  const float ga = gamma[c] * rstd, be = beta[c] - mean * ga;  // h = x * ga + be
  const int64_t lo = split * CL, hi = (lo + CL < L) ? lo + CL : L;
  const float* xr = x + row * L;
  const float* dr = dact + row * L;
  float a = 0.0f, bs = 0.0f;
  for (int64_t l = lo + 4 * li; l < hi; l += 4 * TPR) {
    const f32x4 xv = *reinterpret_cast<const f32x4*>(xr + l);
    const f32x4 dv = *reinterpret_cast<const f32x4*>(dr + l);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float xh = (xv[k] - mean) * rstd;
      const float ds = dv[k] * adp_dsilu_fast(fmaf(xv[k], ga, be));
      a = fmaf(ds, xh, a);
      bs += ds;
    }
  }
  a = gn_seg_sum<TPR>(a, sh);
  bs = gn_seg_sum<TPR>(bs, sh);
  if (li == 0 && live) *reinterpret_cast<f32x2*>(ab + (row * NS + split) * 2) = f32x2{a, bs};
}

template <int TPR>
__global__ __launch_bounds__(256) void gn_bwd_apply_vec_kernel(const float* __restrict__ x, const float* __restrict__ dact,
                                                               const float* __restrict__ stats, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, const float* __restrict__ ab,
                                                               const float* __restrict__ dres, int64_t C, int64_t L, int64_t G,
                                                               int64_t NS, int64_t CL, int64_t nseg, float* __restrict__ dx,
                                                               int64_t B, float* __restrict__ dgamma,
                                                               float* __restrict__ dbeta, int accumulate, int64_t NSab) {
  __shared__ float sh[4];
  const int sl = threadIdx.x / TPR, li = threadIdx.x % TPR;
  const int64_t seg = (int64_t)blockIdx.x * (256 / TPR) + sl;
  const bool live = seg < nseg;
  const int64_t sg = live ? seg : nseg - 1;
  const int64_t row = sg / NS, split = sg % NS;
  const int64_t b = row / C, c = row % C, Cg = C / G, g = c / Cg;
  // The segment's first quads are requested before anything else: the deep layers' launches (one quad per lane) are bound
  // by memory round trips, and the sums below (two dependent trips to `ab`) used to come first.
  const int64_t lo = split * CL, hi = (lo + CL < L) ? lo + CL : L;
  const float* xr = x + row * L;
  const float* dr = dact + row * L;
  const int64_t l0 = lo + 4 * li;
  f32x4 xv0 = f32x4{0.0f, 0.0f, 0.0f, 0.0f}, dv0 = xv0, rv0 = xv0;
  if (live && l0 < hi) {
    xv0 = *reinterpret_cast<const f32x4*>(xr + l0);
    dv0 = *reinterpret_cast<const f32x4*>(dr + l0);
    if (dres) rv0 = *reinterpret_cast<const f32x4*>(dres + row * L + l0);
  }
  const float mean = stats[(b * G + g) * 2], rstd = stats[(b * G + g) * 2 + 1];
  const float gam = gamma[c], bet = beta[c];
  float sa = 0.0f, sb = 0.0f;
  // (NSab: slices per row of the first stage's sums; NS: this launch's own split.  The group's entries are contiguous: rows
  //  g * Cg .. + Cg - 1 x NSab slices; four independent requests per trip -- the trips are round trips to L2)
  {
    const f32x2* gab = reinterpret_cast<const f32x2*>(ab) + (b * C + g * Cg) * NSab;
    const float* ggm = gamma + g * Cg;
    const int64_t ne = Cg * NSab;
    int64_t e = li;
    for (; e + 3 * TPR < ne; e += 4 * TPR) {
      f32x2 p[4];
      float gm[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) p[u] = gab[e + u * TPR], gm[u] = ggm[(e + u * TPR) / NSab];
#pragma unroll
      for (int u = 0; u < 4; ++u) sa = fmaf(gm[u], p[u][0], sa), sb = fmaf(gm[u], p[u][1], sb);
    }
    for (; e < ne; e += TPR) {
      const f32x2 p = gab[e];
      const float gm = ggm[e / NSab];
      sa = fmaf(gm, p[0], sa);
      sb = fmaf(gm, p[1], sb);
    }
  }
  if (dgamma) {  // parameter gradients of this channel: dgamma = sum_{b,s} A, dbeta = sum_{b,s} B (batch 0's segments)
    const bool mine = (b == 0 && split == 0);
    float pa = 0.0f, pb = 0.0f;
    if (TPR == 256 ? mine : true) {  // (sub-wave segments: every lane walks, only batch 0's keep the result)
      for (int64_t e = li; e < B * NSab; e += TPR) {
        const int64_t bb = e / NSab, spx = e % NSab;
        const f32x2 p = *reinterpret_cast<const f32x2*>(ab + ((bb * C + c) * NSab + spx) * 2);
        pa += p[0];
        pb += p[1];
      }
      pa = gn_seg_sum<TPR>(pa, sh);
      pb = gn_seg_sum<TPR>(pb, sh);
      if (li == 0 && live && mine) {
        dgamma[c] = accumulate ? dgamma[c] + pa : pa;
        dbeta[c] = accumulate ? dbeta[c] + pb : pb;
      }
    }
  }
  const float inv = 1.0f / ((float)Cg * (float)L);
  const float m2 = gn_seg_sum<TPR>(sa, sh) * inv;
  const float m1 = gn_seg_sum<TPR>(sb, sh) * inv;
  const float ga = gam * rstd, be = bet - mean * ga;
  if (!live) return;
  for (int64_t l = l0; l < hi; l += 4 * TPR) {
    f32x4 xv = xv0, dv = dv0, rv = rv0;
    if (l != l0) {
      xv = *reinterpret_cast<const f32x4*>(xr + l);
      dv = *reinterpret_cast<const f32x4*>(dr + l);
      if (dres) rv = *reinterpret_cast<const f32x4*>(dres + row * L + l);
    }
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float xh = (xv[k] - mean) * rstd;
      const float ds = dv[k] * adp_dsilu_fast(fmaf(xv[k], ga, be));
      o[k] = rstd * (gam * ds - m1 - xh * m2);
    }
    if (dres) {
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] += rv[k];
    }
    *reinterpret_cast<f32x4*>(dx + row * L + l) = o;
  }
}

// threads per (row, split) segment of the vector form: one float4 per lane up to a wave, the whole workgroup beyond
static int gn_bwd_tpr(int64_t CL) { return CL <= 64 ? 16 : (CL <= 128 ? 32 : (CL <= 1024 ? 64 : 256)); }
static bool gn_bwd_vec_ok(const float* x, const float* dact, const float* dres, const float* dx, int64_t L, int64_t CL) {
  return (L & 3) == 0 && (CL & 3) == 0 &&
         ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dact) | reinterpret_cast<uintptr_t>(dres) |
           reinterpret_cast<uintptr_t>(dx)) & 15) == 0;
}

__global__ __launch_bounds__(256) void gn_param_grad_kernel(const float* ab, int64_t B, int64_t C, int64_t NS,
                                                            float* dgamma, float* dbeta, int accumulate) {
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float a = 0.0f, bs = 0.0f;
  for (int64_t b = 0; b < B; ++b)
    for (int64_t s = 0; s < NS; ++s) {
      a += ab[((b * C + c) * NS + s) * 2];
      bs += ab[((b * C + c) * NS + s) * 2 + 1];
    }
  dgamma[c] = accumulate ? dgamma[c] + a : a;
  dbeta[c] = accumulate ? dbeta[c] + bs : bs;
}

// ---- LayerNorm over channels (per position) ------------------------------------------------------------------
// One workgroup owns TL consecutive positions x ALL channels and keeps its [C x TL] slab in registers, so x (and
// dy) are read from HBM exactly once.  Lanes run along L first (TL = 16/32/64 positions = 64/128/256-byte row
// segments), the remaining lane bits and the waves stride the channels (CG = NT/TL channel groups, each thread
// holds channels cg, cg+CG, ...).  Deep layers have few positions (depth 8 at batch 4: 512) but 1024 channels:
// they take TL = 16 with 1024-thread workgroups, so even 32 workgroups keep 512 waves of loads in flight.
// mode: y = xhat * (1 + ss[b*bstride + c]) + ss[b*bstride + C + c]      (Modulation); y == NULL: statistics only;
//       gam != NULL: y = xhat * gam[c] + bet[c] (affine LayerNorm of the attention items, materialised once for the
//       q / kv projections) and optionally a second affine output y2 with (gam2, bet2) from the same statistics.
template <int TL, int NT, int VPT>
__global__ __launch_bounds__(NT) void chan_ln_fwd_kernel(const float* x, const float* ss, int64_t bstride, int C, int L,
                                                         float eps, float* y, float* stats, const float* gam,
                                                         const float* bet, const float* gam2, const float* bet2,
                                                         float* y2) {
  constexpr int CG = NT / TL, NW = NT / 64, GPW = 64 / TL;  // channel groups, waves, channel groups per wave
  __shared__ float red[NW][TL];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int p = lane % TL, cg = wave * GPW + lane / TL;
  const int l = blockIdx.x * TL + p, b = blockIdx.y;
  const bool valid = l < L;
  const float* xb = x + (int64_t)b * C * L + l;
  float v[VPT];
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int c = cg + i * CG;
    v[i] = (valid && c < C) ? xb[(int64_t)c * L] : 0.0f;
    s += v[i];
  }
#pragma unroll
  for (int o = TL; o < 64; o <<= 1) s += __shfl_xor(s, o, 64);
  if (lane < TL) red[wave][p] = s;
  __syncthreads();
  float tot = 0.0f;
#pragma unroll
  for (int w = 0; w < NW; ++w) tot += red[w][p];
  const float mean = tot / (float)C;
  __syncthreads();
  float q = 0.0f;
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int c = cg + i * CG;
    const float dlt = (c < C) ? v[i] - mean : 0.0f;
    q = fmaf(dlt, dlt, q);
  }
#pragma unroll
  for (int o = TL; o < 64; o <<= 1) q += __shfl_xor(q, o, 64);
  if (lane < TL) red[wave][p] = q;
  __syncthreads();
  tot = 0.0f;
#pragma unroll
  for (int w = 0; w < NW; ++w) tot += red[w][p];
  const float rstd = 1.0f / sqrtf(tot / (float)C + eps);
  if (tid < TL && valid) {
    stats[((int64_t)b * L + l) * 2] = mean;
    stats[((int64_t)b * L + l) * 2 + 1] = rstd;
  }
  if (y == nullptr) return;
  float* yb = y + (int64_t)b * C * L + l;
  if (gam != nullptr) {
    float* yb2 = y2 ? y2 + (int64_t)b * C * L + l : nullptr;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
      const int c = cg + i * CG;
      if (valid && c < C) {
        const float xh = (v[i] - mean) * rstd;
        yb[(int64_t)c * L] = fmaf(xh, gam[c], bet[c]);
        if (yb2) yb2[(int64_t)c * L] = fmaf(xh, gam2[c], bet2[c]);
      }
    }
    return;
  }
  const float* sb = ss + b * bstride;
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int c = cg + i * CG;
    if (valid && c < C) yb[(int64_t)c * L] = fmaf((v[i] - mean) * rstd, 1.0f + sb[c], sb[C + c]);
  }
}

// Backward of y = xhat * mul_c + add_c with mul_c = 1 + ss[b*bstride + c] (gamma == NULL) or gamma[c].
// dx = rstd * (g - mean_c(g) - xhat * mean_c(g * xhat)), g = dy * mul_c   (+ dres)
// partial per-channel sums over the tile: ws[b][0][tile][c] = sum dy*xhat, ws[b][1][tile][c] = sum dy (channel
// fastest: a workgroup's 2 * C partials are two contiguous runs -- the [c][tile] order scattered them over 2 * C cache
// lines per workgroup, 6x the algorithmic traffic at C = 1024)
template <int TL, int NT, int VPT>
__global__ __launch_bounds__(NT) void chan_ln_bwd_kernel(const float* x, const float* dy, const float* ss,
                                                         int64_t bstride, const float* gamma, const float* stats,
                                                         const float* dres, int C, int L, int NTL, float* dx,
                                                         float* ws) {
  constexpr int CG = NT / TL, NW = NT / 64, GPW = 64 / TL;
  __shared__ float red[2][NW][TL];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int p = lane % TL, cg = wave * GPW + lane / TL;
  const int tile = blockIdx.x, l = tile * TL + p, b = blockIdx.y;
  const bool valid = l < L;
  const int64_t boff = (int64_t)b * C * L + l;
  const float mean = valid ? stats[((int64_t)b * L + l) * 2] : 0.0f;
  const float rstd = valid ? stats[((int64_t)b * L + l) * 2 + 1] : 0.0f;
  float xh[VPT], g[VPT];
  float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int c = cg + i * CG;
    const bool ok = valid && c < C;
    const float mul = (c < C) ? (gamma ? gamma[c] : 1.0f + ss[b * bstride + c]) : 0.0f;
    const float d = ok ? dy[boff + (int64_t)c * L] : 0.0f;
    xh[i] = ok ? (x[boff + (int64_t)c * L] - mean) * rstd : 0.0f;
    g[i] = d * mul;
    s1 += g[i];
    s2 = fmaf(g[i], xh[i], s2);
    // per-channel sums over the tile's positions (lanes p = 0..TL-1 of this channel group)
    float pa = d * xh[i], pb = d;
#pragma unroll
    for (int o = 1; o < TL; o <<= 1) {
      pa += __shfl_xor(pa, o, 64);
      pb += __shfl_xor(pb, o, 64);
    }
    if (p == 0 && c < C) {
      ws[(((int64_t)b * 2 + 0) * NTL + tile) * C + c] = pa;
      ws[(((int64_t)b * 2 + 1) * NTL + tile) * C + c] = pb;
    }
  }
#pragma unroll
  for (int o = TL; o < 64; o <<= 1) {
    s1 += __shfl_xor(s1, o, 64);
    s2 += __shfl_xor(s2, o, 64);
  }
  if (lane < TL) {
    red[0][wave][p] = s1;
    red[1][wave][p] = s2;
  }
  __syncthreads();
  float m1 = 0.0f, m2 = 0.0f;
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    m1 += red[0][w][p];
    m2 += red[1][w][p];
  }
  m1 /= (float)C;
  m2 /= (float)C;
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int c = cg + i * CG;
    if (valid && c < C) {
      float v = rstd * (g[i] - m1 - xh[i] * m2);
      if (dres) v += dres[boff + (int64_t)c * L];
      dx[boff + (int64_t)c * L] = v;
    }
  }
}

// ---- the same two kernels with 16-byte accesses (L % 4 == 0, 16-byte aligned tensors) ------------------------------------
// A lane holds FOUR consecutive positions of a channel row (one float4), LPR lanes cover a row segment of TL = 4 * LPR
// positions, the remaining lane bits and the waves stride the channels (RPP = NT / LPR rows per pass, VPT passes, all in
// registers).  For the same tile width a wave instruction touches a quarter of the cache lines per byte of the 4-byte
// form, and the deep layers (1024 channels x 128-256 positions: 2-4 MB tensors) get 16- or 32-byte row segments
// from ONE lane each instead of 4 lanes x 4 bytes.  Reductions: per position over channels = xor-shuffles across the
// row groups of a wave + one LDS round across waves; per channel over the tile's positions (backward) = the lane's four
// components + xor-shuffles across the LPR lanes of the segment.
template <int LPR, int NT, int VPT>
__global__ __launch_bounds__(NT) void chan_lnv_fwd_kernel(const float* x, const float* ss, int64_t bstride, int C, int L,
                                                          float eps, float* y, float* stats, const float* gam,
                                                          const float* bet, const float* gam2, const float* bet2,
                                                          float* y2, float eps2, float* cy, float* cstats) {
  // CHAIN (cy != nullptr; ModulationItem followed by an AttentionItem / CrossAttentionItem): y = LN(x) * (1 + scale) + shift
  // from `ss` as usual, then -- the tile still in registers -- the attention's own LayerNorm of y: cy = LN(y) * gam + bet,
  // y2 = LN(y) * gam2 + bet2 (norm_context of a self-attention item), statistics of y to cstats.  One launch and one pass over
  // x instead of two launches and a re-read of y.
  constexpr int TL = 4 * LPR, RPP = NT / LPR, NW = NT / 64;
  __shared__ float red2[2][NW][TL];  // one array per reduction round: no barrier before a round's writes
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = tid % LPR, rg = tid / LPR;
  const int l0 = blockIdx.x * TL + 4 * lr, b = blockIdx.y;
  const bool valid = l0 < L;  // L % 4 == 0: a quad is inside or outside
  const bool chain = cy != nullptr, aff = gam != nullptr && !chain;
  const float* xb = x + (int64_t)b * C * L + l0;
  const float* sb = ss ? ss + b * bstride : nullptr;
  f32x4 v[VPT];
  float mulv[VPT], addv[VPT];  // per-channel scale / shift, requested WITH the tile (not behind the two reductions)
  float s[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int c = rg + i * RPP;
    mulv[i] = addv[i] = 0.0f;
    if (y != nullptr && c < C) {
      mulv[i] = aff ? gam[c] : 1.0f + sb[c];
      addv[i] = aff ? bet[c] : sb[C + c];
    }
    if (valid && c < C) {
      v[i] = *reinterpret_cast<const f32x4*>(xb + (int64_t)c * L);
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) v[i][k] = 0.0f;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) s[k] += v[i][k];
  }
  auto over_channels = [&](float (&t)[4], int round) {  // sum over every row group of the workgroup; result in all threads
#pragma unroll
    for (int o = LPR; o < 64; o <<= 1)
#pragma unroll
      for (int k = 0; k < 4; ++k) t[k] += __shfl_xor(t[k], o, 64);
    if (NW > 1) {
      float (*red)[TL] = red2[round];
      if (lane < LPR)
#pragma unroll
        for (int k = 0; k < 4; ++k) red[wave][4 * lr + k] = t[k];
      __syncthreads();
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float a = 0.0f;
#pragma unroll
        for (int w = 0; w < NW; ++w) a += red[w][4 * lr + k];
        t[k] = a;
      }
    }
  };
  over_channels(s, 0);
  float mean[4], q[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int k = 0; k < 4; ++k) mean[k] = s[k] / (float)C;
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const bool in = rg + i * RPP < C;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float dlt = in ? v[i][k] - mean[k] : 0.0f;
      q[k] = fmaf(dlt, dlt, q[k]);
    }
  }
  over_channels(q, 1);
  float rstd[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) rstd[k] = 1.0f / sqrtf(q[k] / (float)C + eps);
  if (tid < LPR && valid) {
    float* sp = stats + ((int64_t)b * L + l0) * 2;
    *reinterpret_cast<f32x4*>(sp) = f32x4{mean[0], rstd[0], mean[1], rstd[1]};
    *reinterpret_cast<f32x4*>(sp + 4) = f32x4{mean[2], rstd[2], mean[3], rstd[3]};
  }
  if (y == nullptr || (!valid && !chain)) return;  // (chain: every thread stays for the second LayerNorm's barriers)
  float* yb = y + (int64_t)b * C * L + l0;
  float* yb2 = y2 ? y2 + (int64_t)b * C * L + l0 : nullptr;
  float g1[VPT], b1[VPT], g2[VPT], b2[VPT];  // the second LayerNorm's affine maps, requested ahead of its reductions
  float s2[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int c = rg + i * RPP;
    g1[i] = b1[i] = g2[i] = b2[i] = 0.0f;
    if (chain && c < C) {
      g1[i] = gam[c], b1[i] = bet[c];
      if (yb2) g2[i] = gam2[c], b2[i] = bet2[c];
    }
    if (c >= C || !valid) {
      if (chain) v[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      continue;
    }
    const float mul = mulv[i], add = addv[i];
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = fmaf((v[i][k] - mean[k]) * rstd[k], mul, add);
    *reinterpret_cast<f32x4*>(yb + (int64_t)c * L) = o;
    if (chain) {
      v[i] = o;
#pragma unroll
      for (int k = 0; k < 4; ++k) s2[k] += o[k];
    } else if (yb2) {
      const float m2 = gam2[c], a2 = bet2[c];
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] = fmaf((v[i][k] - mean[k]) * rstd[k], m2, a2);
      *reinterpret_cast<f32x4*>(yb2 + (int64_t)c * L) = o;
    }
  }
  if (!chain) return;
  // ---- second LayerNorm over the channels of y (rounds 2 and 3 reuse the two LDS arrays: a barrier lies between a round's
  // reads and the next use of its array)
  over_channels(s2, 0);
  float q2[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int k = 0; k < 4; ++k) mean[k] = s2[k] / (float)C;
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const bool in = rg + i * RPP < C;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float dlt = in ? v[i][k] - mean[k] : 0.0f;
      q2[k] = fmaf(dlt, dlt, q2[k]);
    }
  }
  over_channels(q2, 1);
#pragma unroll
  for (int k = 0; k < 4; ++k) rstd[k] = 1.0f / sqrtf(q2[k] / (float)C + eps2);
  if (!valid) return;
  if (tid < LPR) {
    float* sp = cstats + ((int64_t)b * L + l0) * 2;
    *reinterpret_cast<f32x4*>(sp) = f32x4{mean[0], rstd[0], mean[1], rstd[1]};
    *reinterpret_cast<f32x4*>(sp + 4) = f32x4{mean[2], rstd[2], mean[3], rstd[3]};
  }
  float* cb = cy + (int64_t)b * C * L + l0;
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int c = rg + i * RPP;
    if (c >= C) continue;
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = fmaf((v[i][k] - mean[k]) * rstd[k], g1[i], b1[i]);
    *reinterpret_cast<f32x4*>(cb + (int64_t)c * L) = o;
    if (yb2) {
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] = fmaf((v[i][k] - mean[k]) * rstd[k], g2[i], b2[i]);
      *reinterpret_cast<f32x4*>(yb2 + (int64_t)c * L) = o;
    }
  }
}

template <int LPR, int NT, int VPT>
__global__ __launch_bounds__(NT) void chan_lnv_bwd_kernel(const float* x, const float* dy, const float* ss,
                                                          int64_t bstride, const float* gamma, const float* stats,
                                                          const float* dres, int C, int L, int NTL, float* dx, float* ws) {
  constexpr int TL = 4 * LPR, RPP = NT / LPR, NW = NT / 64;
  __shared__ float red[2][NW][TL];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = tid % LPR, rg = tid / LPR;
  const int tile = blockIdx.x, l0 = tile * TL + 4 * lr, b = blockIdx.y;
  const bool valid = l0 < L;
  const int64_t boff = (int64_t)b * C * L + l0;
  float mean[4] = {0.0f, 0.0f, 0.0f, 0.0f}, rstd[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  if (valid) {
    const float* sp = stats + ((int64_t)b * L + l0) * 2;
    const f32x4 a = *reinterpret_cast<const f32x4*>(sp), c4 = *reinterpret_cast<const f32x4*>(sp + 4);
    mean[0] = a[0], rstd[0] = a[1], mean[1] = a[2], rstd[1] = a[3];
    mean[2] = c4[0], rstd[2] = c4[1], mean[3] = c4[2], rstd[3] = c4[3];
  }
  f32x4 xh[VPT], g[VPT];
  float s1[4] = {0.0f, 0.0f, 0.0f, 0.0f}, s2[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int c = rg + i * RPP;
    const bool ok = valid && c < C;
    f32x4 d;
    if (ok) {
      d = *reinterpret_cast<const f32x4*>(dy + boff + (int64_t)c * L);
      xh[i] = *reinterpret_cast<const f32x4*>(x + boff + (int64_t)c * L);
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) d[k] = xh[i][k] = 0.0f;
    }
    const float mul = (c < C) ? (gamma ? gamma[c] : 1.0f + ss[b * bstride + c]) : 0.0f;
    float pa = 0.0f, pb = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      xh[i][k] = ok ? (xh[i][k] - mean[k]) * rstd[k] : 0.0f;
      g[i][k] = d[k] * mul;
      s1[k] += g[i][k];
      s2[k] = fmaf(g[i][k], xh[i][k], s2[k]);
      pa = fmaf(d[k], xh[i][k], pa);
      pb += d[k];
    }
    // per-channel sums over the tile's positions: the LPR lanes of this row segment
#pragma unroll
    for (int o = 1; o < LPR; o <<= 1) {
      pa += __shfl_xor(pa, o, 64);
      pb += __shfl_xor(pb, o, 64);
    }
    if (lr == 0 && c < C) {
      ws[(((int64_t)b * 2 + 0) * NTL + tile) * C + c] = pa;
      ws[(((int64_t)b * 2 + 1) * NTL + tile) * C + c] = pb;
    }
  }
#pragma unroll
  for (int o = LPR; o < 64; o <<= 1)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      s1[k] += __shfl_xor(s1[k], o, 64);
      s2[k] += __shfl_xor(s2[k], o, 64);
    }
  float m1[4], m2[4];
  if (NW > 1) {
    if (lane < LPR)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        red[0][wave][4 * lr + k] = s1[k];
        red[1][wave][4 * lr + k] = s2[k];
      }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float a = 0.0f, c2 = 0.0f;
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        a += red[0][w][4 * lr + k];
        c2 += red[1][w][4 * lr + k];
      }
      m1[k] = a / (float)C;
      m2[k] = c2 / (float)C;
    }
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      m1[k] = s1[k] / (float)C;
      m2[k] = s2[k] / (float)C;
    }
  }
  if (!valid) return;
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int c = rg + i * RPP;
    if (c >= C) continue;
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = rstd[k] * (g[i][k] - m1[k] - xh[i][k] * m2[k]);
    if (dres) {
      const f32x4 r = *reinterpret_cast<const f32x4*>(dres + boff + (int64_t)c * L);
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] += r[k];
    }
    *reinterpret_cast<f32x4*>(dx + boff + (int64_t)c * L) = o;
  }
}

// Backward of "ModulationItem -> LayerNorm of an attention item" as ONE pass (the backward of chan_lnv_fwd's chained mode):
//   stage A   d(xn) -> d(y): LayerNorm-with-affine backward over the channels of y (+ the residual gradient dres), y itself
//             rebuilt from x (y = xhat1 * (1 + scale) + shift: x is needed by stage B anyway, y is not read);
//   stage B   d(y) -> d(x): the Modulation's own backward.
// Per-channel sums over the tile's positions for both stages: ws2 (sum d(xn) * xhat2 | sum d(xn): dgamma / dbeta of the
// LayerNorm) and ws1 (sum d(y) * xhat1 | sum d(y): dscale / dshift), both [b][2][tile][c] like chan_lnv_bwd_kernel's.
// Three tensor reads + one write where the two launches read five and write two; the intermediate d(y) stays in registers.
template <int LPR, int NT, int VPT>
__global__ __launch_bounds__(NT) void chan_lnv_bwd_chain_kernel(const float* x, const float* ss, int64_t bstride,
                                                                const float* stats1, const float* dxn, const float* gamma,
                                                                const float* stats2, const float* dres, int C, int L,
                                                                int NTL, float* dx, float* ws1, float* ws2) {
  constexpr int TL = 4 * LPR, RPP = NT / LPR, NW = NT / 64;
  __shared__ float red[4][NW][TL];  // two arrays per stage: no barrier between a stage's reads and the next stage's writes
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = tid % LPR, rg = tid / LPR;
  const int tile = blockIdx.x, l0 = tile * TL + 4 * lr, b = blockIdx.y;
  const bool valid = l0 < L;
  const int64_t boff = (int64_t)b * C * L + l0;
  float mean1[4] = {0.0f, 0.0f, 0.0f, 0.0f}, rstd1[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  float mean2[4] = {0.0f, 0.0f, 0.0f, 0.0f}, rstd2[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  if (valid) {
    const float* sp = stats1 + ((int64_t)b * L + l0) * 2;
    const f32x4 a = *reinterpret_cast<const f32x4*>(sp), c4 = *reinterpret_cast<const f32x4*>(sp + 4);
    mean1[0] = a[0], rstd1[0] = a[1], mean1[1] = a[2], rstd1[1] = a[3];
    mean1[2] = c4[0], rstd1[2] = c4[1], mean1[3] = c4[2], rstd1[3] = c4[3];
    const float* sq = stats2 + ((int64_t)b * L + l0) * 2;
    const f32x4 e = *reinterpret_cast<const f32x4*>(sq), f4 = *reinterpret_cast<const f32x4*>(sq + 4);
    mean2[0] = e[0], rstd2[0] = e[1], mean2[1] = e[2], rstd2[1] = e[3];
    mean2[2] = f4[0], rstd2[2] = f4[1], mean2[3] = f4[2], rstd2[3] = f4[3];
  }
  const float* sb = ss + b * bstride;
  f32x4 xh[VPT], g[VPT];
  float mulv[VPT], addv[VPT];
  float s1[4] = {0.0f, 0.0f, 0.0f, 0.0f}, s2[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  auto xhat2 = [&](int i, int k) {  // normalised y of the LayerNorm at (channel pass i, position k), from xhat1
    return (fmaf(xh[i][k], mulv[i], addv[i]) - mean2[k]) * rstd2[k];
  };
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int c = rg + i * RPP;
    const bool ok = valid && c < C;
    f32x4 d;
    if (ok) {
      d = *reinterpret_cast<const f32x4*>(dxn + boff + (int64_t)c * L);
      xh[i] = *reinterpret_cast<const f32x4*>(x + boff + (int64_t)c * L);
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) d[k] = xh[i][k] = 0.0f;
    }
    mulv[i] = (c < C) ? 1.0f + sb[c] : 0.0f;
    addv[i] = (c < C) ? sb[C + c] : 0.0f;
    const float gm = (c < C) ? gamma[c] : 0.0f;
    float pa = 0.0f, pb = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      xh[i][k] = ok ? (xh[i][k] - mean1[k]) * rstd1[k] : 0.0f;
      const float x2 = ok ? xhat2(i, k) : 0.0f;
      g[i][k] = d[k] * gm;
      s1[k] += g[i][k];
      s2[k] = fmaf(g[i][k], x2, s2[k]);
      pa = fmaf(d[k], x2, pa);
      pb += d[k];
    }
#pragma unroll
    for (int o = 1; o < LPR; o <<= 1) {
      pa += __shfl_xor(pa, o, 64);
      pb += __shfl_xor(pb, o, 64);
    }
    if (lr == 0 && c < C) {
      ws2[(((int64_t)b * 2 + 0) * NTL + tile) * C + c] = pa;
      ws2[(((int64_t)b * 2 + 1) * NTL + tile) * C + c] = pb;
    }
  }
  auto over_channels = [&](float (&a)[4], float (&c2)[4], int stage) {  // channel means of two sums, in every thread
#pragma unroll
    for (int o = LPR; o < 64; o <<= 1)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        a[k] += __shfl_xor(a[k], o, 64);
        c2[k] += __shfl_xor(c2[k], o, 64);
      }
    if (NW > 1) {
      if (lane < LPR)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          red[2 * stage][wave][4 * lr + k] = a[k];
          red[2 * stage + 1][wave][4 * lr + k] = c2[k];
        }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float u = 0.0f, w2 = 0.0f;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
          u += red[2 * stage][w][4 * lr + k];
          w2 += red[2 * stage + 1][w][4 * lr + k];
        }
        a[k] = u, c2[k] = w2;
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      a[k] /= (float)C;
      c2[k] /= (float)C;
    }
  };
  over_channels(s1, s2, 0);
  // ---- stage A output d(y) (kept in g) and stage B's sums
  float t1[4] = {0.0f, 0.0f, 0.0f, 0.0f}, t2[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int c = rg + i * RPP;
    const bool ok = valid && c < C;
    f32x4 r = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    if (ok && dres) r = *reinterpret_cast<const f32x4*>(dres + boff + (int64_t)c * L);
    float pa = 0.0f, pb = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float o = ok ? rstd2[k] * (g[i][k] - s1[k] - xhat2(i, k) * s2[k]) + r[k] : 0.0f;  // d(y)
      pa = fmaf(o, xh[i][k], pa);
      pb += o;
      g[i][k] = o * mulv[i];
      t1[k] += g[i][k];
      t2[k] = fmaf(g[i][k], xh[i][k], t2[k]);
    }
#pragma unroll
    for (int o = 1; o < LPR; o <<= 1) {
      pa += __shfl_xor(pa, o, 64);
      pb += __shfl_xor(pb, o, 64);
    }
    if (lr == 0 && c < C) {
      ws1[(((int64_t)b * 2 + 0) * NTL + tile) * C + c] = pa;
      ws1[(((int64_t)b * 2 + 1) * NTL + tile) * C + c] = pb;
    }
  }
  over_channels(t1, t2, 1);
  if (!valid) return;
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int c = rg + i * RPP;
    if (c >= C) continue;
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = rstd1[k] * (g[i][k] - t1[k] - xh[i][k] * t2[k]);
    *reinterpret_cast<f32x4*>(dx + boff + (int64_t)c * L) = o;
  }
}

// vector-form tile by channel count: (lanes per row segment, threads, passes); RPP * VPT >= C
struct LnvCfg {
  int lpr, nt, vpt;
};
static LnvCfg lnv_cfg(int64_t C, int64_t B, int64_t L, bool bwd = false) {
  if (C <= 8) return {64, 256, 2};     // 1 KB row segments, 4 rows per pass
  if (C <= 32) return {32, 256, 4};    // 512 B, 8 rows per pass
  if (C <= 64) return {16, 256, 4};    // 256 B, 16 rows per pass
  if (C <= 128) return {8, 256, 4};    // 128 B, 32 rows per pass
  if (C <= 256) {  // 128 B segments; 512 threads x 4 passes (1024 x 2 before round 4: ADP_LNV_NT=1024)
    const char* e = getenv(bwd ? "ADP_LNV_NT_BWD" : "ADP_LNV_NT");
    if (e && atoi(e) == 1024) return {8, 1024, 2};
    return {8, 512, 4};
  }
  // 512 / 1024 channels, 1024-thread workgroups: few positions (depth 8 at batch 4: 512), so the segment narrows until
  // enough workgroups exist (measured at batch 4, step time: 16 positions 14.38 ms, 8 positions 14.42, 4 positions 14.53)
  int lpr = 4;
  while (lpr > 1 && B * adp_cdiv(L, 4 * lpr) < 32) lpr >>= 1;  // (batch 4: 16 positions at depths 5-8 measured best)
  // 512-thread workgroups, twice the passes in registers (round 4; hipGraph per-launch us at batch 4, 1024 -> 512 threads:
  // forward C=512 L=1024 9.7 -> 8.0, L=512 6.9 -> 5.6, C=1024 7.4 -> 6.6 / 7.1 -> 6.3; backward 14.0 -> 12.9, 10.9 -> 9.9, 12.5 ->
  // 12.3, 9.3 -> 9.1: an 8-wave barrier and 8 partials per reduction instead of 16; 256 threads x 16 passes lose again).
  // ADP_LNV_NT / ADP_LNV_NT_BWD = 1024: the previous shape (A/B).
  const char* e = getenv(bwd ? "ADP_LNV_NT_BWD" : "ADP_LNV_NT");
  if (!e || atoi(e) != 1024) {
    const int lp = (C <= 512 && lpr == 1) ? 2 : lpr;
    return {lp, 512, C <= 512 ? lp : 2 * lp};
  }
  if (C <= 512) return {lpr == 1 ? 2 : lpr, 1024, lpr == 4 ? 2 : 1};
  return {lpr, 1024, lpr};  // RPP = 1024 / lpr rows per pass -> lpr passes cover 1024 channels
}
static bool lnv_ok(int64_t L, const void* a, const void* b, const void* c, const void* d, const void* e) {
  return (L & 3) == 0 && ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c) |
                           reinterpret_cast<uintptr_t>(d) | reinterpret_cast<uintptr_t>(e)) & 15) == 0;
}

// tile shape by channel count (registers: VPT = ceil(C / CG) values per thread) and by how many tiles there are
struct LnCfg {
  int tl, nt;
};
LnCfg ln_cfg(int64_t C, int64_t B, int64_t L) {
  if (C <= 64) return {64, 256};     // CG 4,  VPT 2 / 8 / 16
  if (C <= 128) return {32, 256};    // CG 8,  VPT 16
  if (C <= 256) return {32, 1024};   // CG 32, VPT 8
  // C <= 1024, 1024-thread workgroups: the deep layers have few positions (depth 8 at batch 4: 512), so the tile
  // narrows until the grid covers the chip -- 16 positions (CG 64, VPT 16), 8 (CG 128, VPT 8) or 4 (CG 256, VPT 4)
  if (B * adp_cdiv(L, 16) >= 192) return {16, 1024};
  if (B * adp_cdiv(L, 8) >= 192) return {8, 1024};
  return {4, 1024};
}
constexpr int64_t LN_CMAX = 1024;

int launch_ln_fwd(const float* x, const float* ss, int64_t bstride, int64_t B, int64_t C, int64_t L, float eps, float* y,
                  float* stats, void* stream, const float* gam = nullptr, const float* bet = nullptr,
                  const float* gam2 = nullptr, const float* bet2 = nullptr, float* y2 = nullptr, float eps2 = 0.0f,
                  float* cy = nullptr, float* cstats = nullptr) {
  if (cy && !(lnv_ok(L, x, y, y2, stats, cy) && lnv_ok(L, cstats, nullptr, nullptr, nullptr, nullptr))) {
    // chained Modulation -> LayerNorm without the 16-byte form: the two launches it stands for
    const int rc = launch_ln_fwd(x, ss, bstride, B, C, L, eps, y, stats, stream);
    if (rc != ADP_OK) return rc;
    return launch_ln_fwd(y, nullptr, 0, B, C, L, eps2, cy, cstats, stream, gam, bet, gam2, bet2, y2);
  }
  if (lnv_ok(L, x, y, y2, stats, nullptr)) {
    const LnvCfg v = lnv_cfg(C, B, L);
    dim3 vgrid((unsigned)adp_cdiv(L, 4 * v.lpr), (unsigned)B);
#define ADP_LNV_FWD(LPR, NT, VPT)                                                                                     \
  if (v.lpr == LPR && v.nt == NT && v.vpt == VPT) {                                                                   \
    ADP_LAUNCH((chan_lnv_fwd_kernel<LPR, NT, VPT>), vgrid, dim3(NT), stream, x, ss, bstride, (int)C, (int)L, eps, y, \
               stats, gam, bet, gam2, bet2, y2, eps2, cy, cstats);                                                    \
    return ADP_LAUNCH_OK();                                                                                           \
  }
    ADP_LNV_FWD(64, 256, 2) ADP_LNV_FWD(32, 256, 4) ADP_LNV_FWD(16, 256, 4) ADP_LNV_FWD(8, 256, 4) ADP_LNV_FWD(8, 1024, 2)
    ADP_LNV_FWD(4, 1024, 2) ADP_LNV_FWD(2, 1024, 1) ADP_LNV_FWD(4, 1024, 4) ADP_LNV_FWD(2, 1024, 2) ADP_LNV_FWD(1, 1024, 1)
    ADP_LNV_FWD(4, 512, 4) ADP_LNV_FWD(4, 512, 8) ADP_LNV_FWD(2, 512, 2) ADP_LNV_FWD(2, 512, 4) ADP_LNV_FWD(1, 512, 2) ADP_LNV_FWD(8, 512, 4)
#undef ADP_LNV_FWD
  }
  const LnCfg k = ln_cfg(C, B, L);
  dim3 grid((unsigned)adp_cdiv(L, k.tl), (unsigned)B);
  if (k.tl == 8)
    ADP_LAUNCH((chan_ln_fwd_kernel<8, 1024, 8>), grid, dim3(1024), stream, x, ss, bstride, (int)C, (int)L, eps, y, stats, gam, bet, gam2, bet2, y2);
  else if (k.tl == 4)
    ADP_LAUNCH((chan_ln_fwd_kernel<4, 1024, 4>), grid, dim3(1024), stream, x, ss, bstride, (int)C, (int)L, eps, y, stats, gam, bet, gam2, bet2, y2);
  else if (k.tl == 64 && C <= 8)
    ADP_LAUNCH((chan_ln_fwd_kernel<64, 256, 2>), grid, dim3(256), stream, x, ss, bstride, (int)C, (int)L, eps, y, stats, gam, bet, gam2, bet2, y2);
  else if (k.tl == 64 && C <= 32)
    ADP_LAUNCH((chan_ln_fwd_kernel<64, 256, 8>), grid, dim3(256), stream, x, ss, bstride, (int)C, (int)L, eps, y, stats, gam, bet, gam2, bet2, y2);
  else if (k.tl == 64)
    ADP_LAUNCH((chan_ln_fwd_kernel<64, 256, 16>), grid, dim3(256), stream, x, ss, bstride, (int)C, (int)L, eps, y, stats, gam, bet, gam2, bet2, y2);
  else if (k.tl == 32 && k.nt == 256)
    ADP_LAUNCH((chan_ln_fwd_kernel<32, 256, 16>), grid, dim3(256), stream, x, ss, bstride, (int)C, (int)L, eps, y, stats, gam, bet, gam2, bet2, y2);
  else if (k.tl == 32)
    ADP_LAUNCH((chan_ln_fwd_kernel<32, 1024, 8>), grid, dim3(1024), stream, x, ss, bstride, (int)C, (int)L, eps, y, stats, gam, bet, gam2, bet2, y2);
  else
    ADP_LAUNCH((chan_ln_fwd_kernel<16, 1024, 16>), grid, dim3(1024), stream, x, ss, bstride, (int)C, (int)L, eps, y, stats, gam, bet, gam2, bet2, y2);
  return ADP_LAUNCH_OK();
}

// returns the number of position tiles the launch wrote partial channel sums for (ws [b][2][tile][c])
int launch_ln_bwd(const float* x, const float* dy, const float* ss, int64_t bstride, const float* gamma,
                  const float* stats, const float* dres, int64_t B, int64_t C, int64_t L, float* dx, float* ws,
                  void* stream) {
  if (lnv_ok(L, x, dy, dres, dx, stats)) {
    const LnvCfg v = lnv_cfg(C, B, L, true);
    const int VNTL = (int)adp_cdiv(L, 4 * v.lpr);
    dim3 vgrid((unsigned)VNTL, (unsigned)B);
#define ADP_LNV_BWD(LPR, NT, VPT)                                                                                      \
  if (v.lpr == LPR && v.nt == NT && v.vpt == VPT) {                                                                    \
    ADP_LAUNCH((chan_lnv_bwd_kernel<LPR, NT, VPT>), vgrid, dim3(NT), stream, x, dy, ss, bstride, gamma, stats, dres,  \
               (int)C, (int)L, VNTL, dx, ws);                                                                          \
    return VNTL;                                                                                                       \
  }
    ADP_LNV_BWD(64, 256, 2) ADP_LNV_BWD(32, 256, 4) ADP_LNV_BWD(16, 256, 4) ADP_LNV_BWD(8, 256, 4) ADP_LNV_BWD(8, 1024, 2)
    ADP_LNV_BWD(4, 1024, 2) ADP_LNV_BWD(2, 1024, 1) ADP_LNV_BWD(4, 1024, 4) ADP_LNV_BWD(2, 1024, 2) ADP_LNV_BWD(1, 1024, 1)
    ADP_LNV_BWD(4, 512, 4) ADP_LNV_BWD(4, 512, 8) ADP_LNV_BWD(2, 512, 2) ADP_LNV_BWD(2, 512, 4) ADP_LNV_BWD(1, 512, 2) ADP_LNV_BWD(8, 512, 4)
#undef ADP_LNV_BWD
  }
  const LnCfg k = ln_cfg(C, B, L);
  const int NTL = (int)adp_cdiv(L, k.tl);
  dim3 grid((unsigned)NTL, (unsigned)B);
  if (k.tl == 8)
    ADP_LAUNCH((chan_ln_bwd_kernel<8, 1024, 8>), grid, dim3(1024), stream, x, dy, ss, bstride, gamma, stats, dres,
               (int)C, (int)L, NTL, dx, ws);
  else if (k.tl == 4)
    ADP_LAUNCH((chan_ln_bwd_kernel<4, 1024, 4>), grid, dim3(1024), stream, x, dy, ss, bstride, gamma, stats, dres,
               (int)C, (int)L, NTL, dx, ws);
  else if (k.tl == 64 && C <= 8)
    ADP_LAUNCH((chan_ln_bwd_kernel<64, 256, 2>), grid, dim3(256), stream, x, dy, ss, bstride, gamma, stats, dres,
               (int)C, (int)L, NTL, dx, ws);
  else if (k.tl == 64 && C <= 32)
    ADP_LAUNCH((chan_ln_bwd_kernel<64, 256, 8>), grid, dim3(256), stream, x, dy, ss, bstride, gamma, stats, dres,
               (int)C, (int)L, NTL, dx, ws);
  else if (k.tl == 64)
    ADP_LAUNCH((chan_ln_bwd_kernel<64, 256, 16>), grid, dim3(256), stream, x, dy, ss, bstride, gamma, stats, dres,
               (int)C, (int)L, NTL, dx, ws);
  else if (k.tl == 32 && k.nt == 256)
    ADP_LAUNCH((chan_ln_bwd_kernel<32, 256, 16>), grid, dim3(256), stream, x, dy, ss, bstride, gamma, stats, dres,
               (int)C, (int)L, NTL, dx, ws);
  else if (k.tl == 32)
    ADP_LAUNCH((chan_ln_bwd_kernel<32, 1024, 8>), grid, dim3(1024), stream, x, dy, ss, bstride, gamma, stats, dres,
               (int)C, (int)L, NTL, dx, ws);
  else
    ADP_LAUNCH((chan_ln_bwd_kernel<16, 1024, 16>), grid, dim3(1024), stream, x, dy, ss, bstride, gamma, stats, dres,
               (int)C, (int)L, NTL, dx, ws);
  return NTL;
}

// out[b*bstride + j] (or out[j] summed over b) = sum_t ws[(b*W + j)*NT + t]; one wave per row
__global__ __launch_bounds__(256) void reduce_rows_kernel(const float* ws, int64_t B, int64_t W, int64_t NT,
                                                          int64_t bstride, int sum_over_b, int accumulate,
                                                          float* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t row = (int64_t)blockIdx.x * 4 + wave;
  const int64_t rows = sum_over_b ? W : B * W;
  if (row >= rows) return;
  float s = 0.0f;
  if (sum_over_b) {
    for (int64_t b = 0; b < B; ++b)
      for (int64_t t = lane; t < NT; t += 64) s += ws[(b * W + row) * NT + t];
  } else {
    for (int64_t t = lane; t < NT; t += 64) s += ws[row * NT + t];
  }
  s = adp_wave_sum(s);
  if (lane == 0) {
    const int64_t o = sum_over_b ? row : (row / W) * bstride + (row % W);
    out[o] = accumulate ? out[o] + s : s;
  }
}

// out[b*bstride + j] (or out[j] summed over b) = sum_t ws[((b*2 + j / C) * NT + t) * C + j % C], j < 2 C: second stage of
// chan_ln_bwd_kernel's per-tile channel sums.  One workgroup = 64 outputs x 16 tile groups: lanes run along the
// channels (coalesced), the 16 waves stride the tiles (short serial chains), LDS combines the groups in a fixed order.
__global__ __launch_bounds__(1024) void reduce_tiles_kernel(const float* ws, int64_t B, int64_t C, int64_t NT,
                                                            int64_t bstride, int sum_over_b, int accumulate,
                                                            float* out) {
  __shared__ float part[16][64];
  const int lane = threadIdx.x & 63, tg = threadIdx.x >> 6;
  const int64_t W = 2 * C;
  const int64_t j = (int64_t)blockIdx.x * 64 + lane;
  const int64_t bb = blockIdx.y;                       // batch element (0 when summing over the batch)
  float s = 0.0f;
  if (j < W) {
    const int64_t which = j / C, c = j % C;
    const int64_t b0 = sum_over_b ? 0 : bb, b1 = sum_over_b ? B : bb + 1;
    for (int64_t b = b0; b < b1; ++b) {
      const float* p = ws + ((b * 2 + which) * NT) * C + c;
      for (int64_t t = tg; t < NT; t += 16) s += p[t * C];
    }
  }
  part[tg][lane] = s;
  __syncthreads();
  if (tg == 0 && j < W) {
    float v = 0.0f;
#pragma unroll
    for (int k = 0; k < 16; ++k) v += part[k][lane];
    const int64_t o = sum_over_b ? j : bb * bstride + j;
    out[o] = accumulate ? out[o] + v : v;
  }
}

// same sums, one WAVE per (b, j) with the lanes striding the tiles: for the narrow layers (few channels, thousands of
// tiles), where a thread per output would walk the tiles serially
__global__ __launch_bounds__(256) void reduce_tiles_wave_kernel(const float* ws, int64_t B, int64_t C, int64_t NT,
                                                                int64_t bstride, int sum_over_b, int accumulate,
                                                                float* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * 4 + wave;
  const int64_t W = 2 * C;
  const int64_t total = sum_over_b ? W : B * W;
  if (i >= total) return;
  const int64_t j = i % W, which = j / C, c = j % C;
  float s = 0.0f;
  const int64_t b0 = sum_over_b ? 0 : i / W, b1 = sum_over_b ? B : b0 + 1;
  for (int64_t b = b0; b < b1; ++b) {
    const float* p = ws + ((b * 2 + which) * NT) * C + c;
    for (int64_t t = lane; t < NT; t += 64) s += p[t * C];
  }
  s = adp_wave_sum(s);
  if (lane == 0) {
    const int64_t o = sum_over_b ? j : b0 * bstride + j;
    out[o] = accumulate ? out[o] + s : s;
  }
}

static void launch_reduce_tiles(const float* ws, int64_t B, int64_t C, int64_t NT, int64_t bstride, int sum_over_b,
                                int accumulate, float* out, void* stream) {
  const int64_t total = sum_over_b ? 2 * C : B * 2 * C;
  if (NT >= 128 && C <= 64)   // narrow + long: a wave per output, lanes over the tiles (the stride-C reads are short)
    ADP_LAUNCH(reduce_tiles_wave_kernel, dim3((unsigned)adp_cdiv(total, 4)), dim3(256), stream, ws, B, C, NT, bstride,
               sum_over_b, accumulate, out);
  else
    ADP_LAUNCH(reduce_tiles_kernel, dim3((unsigned)adp_cdiv(2 * C, 64), (unsigned)(sum_over_b ? 1 : B)), dim3(1024),
               stream, ws, B, C, NT, bstride, sum_over_b, accumulate, out);
}

// The same second stage for up to MOD_BATCH Modulation backwards of ONE shape in one launch (blockIdx.z = which one): the
// Modulation items of a U-Net depth share (B, C, L), and nothing reads their scale / shift gradients before the depth's
// conditioning-bank rows are formed at the end of the block -- 42 launches of a step become 9.  Same sums, same order.
constexpr int MOD_BATCH = 8;
struct adp_mod_batch {
  const float* ws[MOD_BATCH];
  float* out[MOD_BATCH];
};

__global__ __launch_bounds__(1024) void reduce_tiles_batch_kernel(adp_mod_batch q, int64_t C, int64_t NT, int64_t bstride) {
  __shared__ float part[16][64];
  const int lane = threadIdx.x & 63, tg = threadIdx.x >> 6;
  const int64_t W = 2 * C;
  const int64_t j = (int64_t)blockIdx.x * 64 + lane;
  const int64_t b = blockIdx.y;
  const float* ws = q.ws[blockIdx.z];
  float s = 0.0f;
  if (j < W) {
    const int64_t which = j / C, c = j % C;
    const float* p = ws + ((b * 2 + which) * NT) * C + c;
    for (int64_t t = tg; t < NT; t += 16) s += p[t * C];
  }
  part[tg][lane] = s;
  __syncthreads();
  if (tg == 0 && j < W) {
    float v = 0.0f;
#pragma unroll
    for (int k = 0; k < 16; ++k) v += part[k][lane];
    q.out[blockIdx.z][b * bstride + j] = v;
  }
}

__global__ __launch_bounds__(256) void reduce_tiles_wave_batch_kernel(adp_mod_batch q, int64_t B, int64_t C, int64_t NT,
                                                                      int64_t bstride) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * 4 + wave;
  const int64_t W = 2 * C;
  if (i >= B * W) return;
  const int64_t j = i % W, which = j / C, c = j % C, b = i / W;
  const float* p = q.ws[blockIdx.z] + ((b * 2 + which) * NT) * C + c;
  float s = 0.0f;
  for (int64_t t = lane; t < NT; t += 64) s += p[t * C];
  s = adp_wave_sum(s);
  if (lane == 0) q.out[blockIdx.z][b * bstride + j] = s;
}

// ---- SkipModulate backward: dx = scale[b,c] * g ; partial dot(g, x) per (row, split) -------------------------
__global__ __launch_bounds__(256) void skipmod_bwd_kernel(const float* g, const float* x, const float* scale,
                                                          int64_t sbstride, int64_t C, int64_t L, int64_t NS,
                                                          int64_t CL, float* dx, float* ws) {
  __shared__ float sh[4];
  const int64_t row = blockIdx.y, split = blockIdx.x;
  const int64_t b = row / C, c = row % C;
  const float sc = scale[b * sbstride + c];
  const int64_t lo = split * CL, hi = (lo + CL < L) ? lo + CL : L;
  float dot = 0.0f;
  const float* gr = g + row * L;
  const float* xr = x + row * L;
  float* dr = dx + row * L;
  // 16-byte accesses when the slice is tileable by 4 and the rows are aligned (the 4-byte loop ran the depth-0 merge,
  // 8 rows of 262144, at 0.27 TB/s)
  const bool vec = ((L | CL) & 3) == 0 &&
                   ((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dx)) & 15) == 0;
  if (vec) {
    for (int64_t l = lo + 4 * (int64_t)threadIdx.x; l < hi; l += 1024) {
      const f32x4 gv = *reinterpret_cast<const f32x4*>(gr + l);
      const f32x4 xv = *reinterpret_cast<const f32x4*>(xr + l);
      f32x4 o;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        dot = fmaf(gv[k], xv[k], dot);
        o[k] = sc * gv[k];
      }
      *reinterpret_cast<f32x4*>(dr + l) = o;
    }
  } else {
    for (int64_t l = lo + threadIdx.x; l < hi; l += 256) {
      const float gv = gr[l];
      dot = fmaf(gv, xr[l], dot);
      dr[l] = sc * gv;
    }
  }
  dot = adp_block_sum<4>(dot, sh);
  if (threadIdx.x == 0) ws[row * NS + split] = dot;
}

int64_t row_nsplit(int64_t rows, int64_t L) {
  int64_t ns = adp_cdiv(1024, rows);
  const int64_t mx = L / 1024 > 1 ? L / 1024 : 1;
  if (ns > mx) ns = mx;
  if (ns < 1) ns = 1;
  if (ns > 65535) ns = 65535;
  return ns;
}

}  // namespace

extern "C" int64_t adp_gn_stats_ws_bytes(int64_t B, int64_t C, int64_t L, int64_t G) {
  if (B <= 0 || C <= 0 || L <= 0 || G <= 0 || C % G) return ADP_ERR_SHAPE;
  const int64_t NG = (C / G) * L;
  return B * G * adp_cdiv(NG, GN_CHUNK) * 2 * (int64_t)sizeof(float);
}

extern "C" int adp_gn_stats(const float* x, int64_t B, int64_t C, int64_t L, int64_t G, float eps, float* stats,
                            float* ws, void* stream) {
  if (!x || !stats || !ws) return ADP_ERR_NULL;
  if (B <= 0 || C <= 0 || L <= 0 || G <= 0 || C % G) return ADP_ERR_SHAPE;
  const int64_t NG = (C / G) * L, nchunks = adp_cdiv(NG, GN_CHUNK);
  if (B * G > 65535) return ADP_ERR_SHAPE;
  ADP_LAUNCH(gn_partial_kernel, dim3((unsigned)nchunks, (unsigned)(B * G)), dim3(256), stream, x, NG, nchunks, ws);
  ADP_LAUNCH(gn_final_kernel, dim3((unsigned)(B * G)), dim3(64), stream, (const float*)ws, NG, nchunks, eps, stats);
  return ADP_LAUNCH_OK();
}

extern "C" int adp_gn_stats_act(const float* x, int64_t B, int64_t C, int64_t L, int64_t G, float eps,
                                const float* gamma, const float* beta, float* stats, float* act, float* ws,
                                void* stream) {
  if (!x || !gamma || !beta || !stats || !act || !ws) return ADP_ERR_NULL;
  if (B <= 0 || C <= 0 || L <= 0 || G <= 0 || C % G) return ADP_ERR_SHAPE;
  const int64_t NG = (C / G) * L, nchunks = adp_cdiv(NG, GN_CHUNK);
  if (B * G > 65535 || B * C > 65535) return ADP_ERR_SHAPE;
  ADP_LAUNCH(gn_partial_kernel, dim3((unsigned)nchunks, (unsigned)(B * G)), dim3(256), stream, x, NG, nchunks, ws);
  if ((L & 3) == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(act)) & 15) == 0)
    return launch_gn_act_slab<2>(x, ws, gamma, beta, B, C, L, G, nchunks, eps, stats, act, stream);
  ADP_LAUNCH(gn_apply_kernel, dim3((unsigned)adp_cdiv(L, 1024), (unsigned)(B * C)), dim3(256), stream, x,
             (const float*)ws, gamma, beta, C, L, G, nchunks, eps, stats, act);
  return ADP_LAUNCH_OK();
}

extern "C" int64_t adp_row_nsplit(int64_t rows, int64_t L) { return row_nsplit(rows, L); }

extern "C" int adp_gn_silu_bwd_reduce(const float* x, const float* dact, const float* stats, const float* gamma,
                                      const float* beta, int64_t B, int64_t C, int64_t L, int64_t G, int64_t NS,
                                      float* ab, void* stream) {
  if (!x || !dact || !stats || !gamma || !beta || !ab) return ADP_ERR_NULL;
  if (B <= 0 || C <= 0 || L <= 0 || G <= 0 || C % G || NS < 1 || NS > 65535 || B * C > 65535) return ADP_ERR_SHAPE;
  int64_t CL = adp_cdiv(L, NS);
  if ((L & 3) == 0) CL = (CL + 3) & ~(int64_t)3;  // (both stages derive the same segment length from L and NS)
  if (gn_bwd_vec_ok(x, dact, nullptr, nullptr, L, CL) && (reinterpret_cast<uintptr_t>(ab) & 7) == 0) {
    const int64_t nseg = B * C * NS;
    const int tpr = gn_bwd_tpr(CL);
    const dim3 grid((unsigned)adp_cdiv(nseg, 256 / tpr));
#define ADP_GN_RED(T) \
  ADP_LAUNCH((gn_bwd_reduce_vec_kernel<T>), grid, dim3(256), stream, x, dact, stats, gamma, beta, C, L, G, NS, CL, nseg, ab)
    if (tpr == 16) ADP_GN_RED(16);
    else if (tpr == 32) ADP_GN_RED(32);
    else if (tpr == 64) ADP_GN_RED(64);
    else ADP_GN_RED(256);
#undef ADP_GN_RED
    return ADP_LAUNCH_OK();
  }
  ADP_LAUNCH(gn_bwd_reduce_kernel, dim3((unsigned)NS, (unsigned)(B * C)), dim3(256), stream, x, dact, stats, gamma,
             beta, C, L, G, NS, CL, ab);
  return ADP_LAUNCH_OK();
}

extern "C" int adp_gn_silu_bwd_apply_ab(const float* x, const float* dact, const float* stats, const float* gamma,
                                        const float* beta, const float* ab, const float* dres, int64_t B, int64_t C,
                                        int64_t L, int64_t G, int64_t NS, int64_t NSab, float* dx, float* dgamma,
                                        float* dbeta, int64_t accumulate, void* stream) {
  if (!x || !dact || !stats || !gamma || !beta || !ab || !dx || (!dgamma != !dbeta)) return ADP_ERR_NULL;
  if (B <= 0 || C <= 0 || L <= 0 || G <= 0 || C % G || NS < 1 || NS > 65535 || NSab < 1 || B * C > 65535) return ADP_ERR_SHAPE;
  int64_t CL = adp_cdiv(L, NS);
  if ((L & 3) == 0) CL = (CL + 3) & ~(int64_t)3;
  if (gn_bwd_vec_ok(x, dact, dres, dx, L, CL) && (reinterpret_cast<uintptr_t>(ab) & 7) == 0) {
    const int64_t nseg = B * C * NS;
    const int tpr = gn_bwd_tpr(CL);
    const dim3 grid((unsigned)adp_cdiv(nseg, 256 / tpr));
#define ADP_GN_APP(T)                                                                                                  \
  ADP_LAUNCH((gn_bwd_apply_vec_kernel<T>), grid, dim3(256), stream, x, dact, stats, gamma, beta, ab, dres, C, L, G, NS, \
             CL, nseg, dx, B, dgamma, dbeta, (int)accumulate, NSab)
    if (tpr == 16) ADP_GN_APP(16);
    else if (tpr == 32) ADP_GN_APP(32);
    else if (tpr == 64) ADP_GN_APP(64);
    else ADP_GN_APP(256);
#undef ADP_GN_APP
    return ADP_LAUNCH_OK();
  }
  ADP_LAUNCH(gn_bwd_apply_kernel, dim3((unsigned)NS, (unsigned)(B * C)), dim3(256), stream, x, dact, stats, gamma,
             beta, ab, dres, C, L, G, NS, CL, dx, B, dgamma, dbeta, (int)accumulate, NSab);
  return ADP_LAUNCH_OK();
}

extern "C" int adp_gn_silu_bwd_apply(const float* x, const float* dact, const float* stats, const float* gamma,
                                     const float* beta, const float* ab, const float* dres, int64_t B, int64_t C,
                                     int64_t L, int64_t G, int64_t NS, float* dx, float* dgamma, float* dbeta,
                                     int64_t accumulate, void* stream) {
  return adp_gn_silu_bwd_apply_ab(x, dact, stats, gamma, beta, ab, dres, B, C, L, G, NS, NS, dx, dgamma, dbeta, accumulate,
                                  stream);
}

extern "C" int adp_gn_param_grad(const float* ab, int64_t B, int64_t C, int64_t NS, float* dgamma, float* dbeta,
                                 int64_t accumulate, void* stream) {
  if (!ab || !dgamma || !dbeta) return ADP_ERR_NULL;
  if (B <= 0 || C <= 0 || NS < 1) return ADP_ERR_SHAPE;
  ADP_LAUNCH(gn_param_grad_kernel, dim3((unsigned)adp_cdiv(C, 256)), dim3(256), stream, ab, B, C, NS, dgamma, dbeta,
             (int)accumulate);
  return ADP_LAUNCH_OK();
}

extern "C" int adp_modulation_fwd(const float* x, const float* ss, int64_t ss_bstride, int64_t B, int64_t C,
                                  int64_t L, float eps, float* y, float* stats, void* stream) {
  if (!x || !ss || !y || !stats) return ADP_ERR_NULL;
  if (B <= 0 || C <= 0 || L <= 0 || B > 65535 || L >= (int64_t)1 << 31) return ADP_ERR_SHAPE;
  if (C > LN_CMAX) return ADP_ERR_UNSUPPORTED;
  return launch_ln_fwd(x, ss, ss_bstride, B, C, L, eps, y, stats, stream);
}

extern "C" int adp_modulation_ln_fwd(const float* x, const float* ss, int64_t ss_bstride, int64_t B, int64_t C, int64_t L,
                                     float eps, float* y, float* stats, float eps_ln, const float* gamma, const float* beta,
                                     float* xn, const float* gamma2, const float* beta2, float* xn2, float* ln_stats,
                                     void* stream) {
  if (!x || !ss || !y || !stats || !gamma || !beta || !xn || !ln_stats) return ADP_ERR_NULL;
  if (xn2 && (!gamma2 || !beta2)) return ADP_ERR_NULL;
  if (B <= 0 || C <= 0 || L <= 0 || B > 65535 || L >= (int64_t)1 << 31) return ADP_ERR_SHAPE;
  if (C > LN_CMAX) return ADP_ERR_UNSUPPORTED;
  return launch_ln_fwd(x, ss, ss_bstride, B, C, L, eps, y, stats, stream, gamma, beta, gamma2, beta2, xn2, eps_ln, xn,
                       ln_stats);
}

extern "C" int adp_gn_finalize(const float* part, int64_t B, int64_t C, int64_t E, int64_t G, float eps, float* stats,
                               void* stream) {
  if (!part || !stats) return ADP_ERR_NULL;
  if (B <= 0 || C <= 0 || E <= 0 || G <= 0 || C % G || (C / G) % 4) return ADP_ERR_SHAPE;
  ADP_LAUNCH(gn_finalize_kernel, dim3((unsigned)(B * G)), dim3(64), stream, part, C, E, G, eps, stats);
  return ADP_LAUNCH_OK();
}

extern "C" int adp_gn_finalize_act(const float* x, const float* part, int64_t B, int64_t C, int64_t L, int64_t E,
                                   int64_t G, float eps, const float* gamma, const float* beta, float* stats,
                                   float* act, void* stream) {
  if (!x || !part || !gamma || !beta || !stats || !act) return ADP_ERR_NULL;
  if (B <= 0 || C <= 0 || L <= 0 || E <= 0 || G <= 0 || C % G || (C / G) % 4 || B * C > 65535) return ADP_ERR_SHAPE;
  if ((L & 3) == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(act)) & 15) == 0)
    return launch_gn_act_slab<1>(x, part, gamma, beta, B, C, L, G, E, eps, stats, act, stream);
  ADP_LAUNCH(gn_finalize_act_kernel, dim3((unsigned)adp_cdiv(L, 1024), (unsigned)(B * C)), dim3(256), stream, x, part,
             gamma, beta, C, L, G, E, eps, stats, act);
  return ADP_LAUNCH_OK();
}

extern "C" int adp_gn_act(const float* x, const float* stats, const float* gamma, const float* beta, int64_t B,
                          int64_t C, int64_t L, int64_t G, float* act, void* stream) {
  if (!x || !stats || !gamma || !beta || !act) return ADP_ERR_NULL;
  if (B <= 0 || C <= 0 || L <= 0 || G <= 0 || C % G || B * C > 65535) return ADP_ERR_SHAPE;
  if ((L & 3) == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(act)) & 15) == 0)
    return launch_gn_act_slab<0>(x, stats, gamma, beta, B, C, L, G, 0, 0.0f, (float*)nullptr, act, stream);
  ADP_LAUNCH(gn_act_kernel, dim3((unsigned)adp_cdiv(L, 1024), (unsigned)(B * C)), dim3(256), stream, x, stats, gamma,
             beta, C, L, G, act);
  return ADP_LAUNCH_OK();
}

extern "C" int adp_ln_stats(const float* x, int64_t B, int64_t C, int64_t L, float eps, float* stats, void* stream) {
  if (!x || !stats) return ADP_ERR_NULL;
  if (B <= 0 || C <= 0 || L <= 0 || B > 65535 || L >= (int64_t)1 << 31) return ADP_ERR_SHAPE;
  if (C > LN_CMAX) return ADP_ERR_UNSUPPORTED;
  return launch_ln_fwd(x, (const float*)nullptr, (int64_t)0, B, C, L, eps, (float*)nullptr, stats, stream);
}

extern "C" int adp_ln_affine_fwd(const float* x, int64_t B, int64_t C, int64_t L, float eps, const float* gamma,
                                 const float* beta, float* y, const float* gamma2, const float* beta2, float* y2,
                                 float* stats, void* stream) {
  if (!x || !gamma || !beta || !y || !stats) return ADP_ERR_NULL;
  if (y2 && (!gamma2 || !beta2)) return ADP_ERR_NULL;
  if (B <= 0 || C <= 0 || L <= 0 || B > 65535 || L >= (int64_t)1 << 31) return ADP_ERR_SHAPE;
  if (C > LN_CMAX) return ADP_ERR_UNSUPPORTED;
  return launch_ln_fwd(x, (const float*)nullptr, (int64_t)0, B, C, L, eps, y, stats, stream, gamma, beta, gamma2, beta2,
                       y2);
}

extern "C" int64_t adp_chan_ln_bwd_ws_bytes(int64_t B, int64_t C, int64_t L) {
  if (B <= 0 || C <= 0 || L <= 0) return ADP_ERR_SHAPE;
  // (room for whichever form the launch takes: the 16-byte form needs aligned tensors, which are not known here)
  const int64_t t_old = adp_cdiv(L, ln_cfg(C, B, L).tl), t_vec = adp_cdiv(L, 4 * lnv_cfg(C, B, L).lpr);
  return B * 2 * C * (t_old > t_vec ? t_old : t_vec) * (int64_t)sizeof(float);
}

extern "C" int adp_modulation_bwd(const float* x, const float* dy, const float* ss, int64_t ss_bstride,
                                  const float* stats, int64_t B, int64_t C, int64_t L, float* dx, float* dss,
                                  int64_t dss_bstride, float* ws, void* stream) {
  if (!x || !dy || !ss || !stats || !dx || !dss || !ws) return ADP_ERR_NULL;
  if (B <= 0 || C <= 0 || L <= 0 || B > 65535 || L >= (int64_t)1 << 31) return ADP_ERR_SHAPE;
  if (C > LN_CMAX) return ADP_ERR_UNSUPPORTED;
  const int64_t NT = launch_ln_bwd(x, dy, ss, ss_bstride, (const float*)nullptr, stats, (const float*)nullptr, B, C, L, dx,
                                   ws, stream);
  launch_reduce_tiles((const float*)ws, B, C, NT, dss_bstride, 0, 0, dss, stream);
  return ADP_LAUNCH_OK();
}

extern "C" int64_t adp_modulation_bwd_partial(const float* x, const float* dy, const float* ss, int64_t ss_bstride,
                                              const float* stats, int64_t B, int64_t C, int64_t L, float* dx, float* ws,
                                              void* stream) {
  if (!x || !dy || !ss || !stats || !dx || !ws) return ADP_ERR_NULL;
  if (B <= 0 || C <= 0 || L <= 0 || B > 65535 || L >= (int64_t)1 << 31) return ADP_ERR_SHAPE;
  if (C > LN_CMAX) return ADP_ERR_UNSUPPORTED;
  const int64_t NT = launch_ln_bwd(x, dy, ss, ss_bstride, (const float*)nullptr, stats, (const float*)nullptr, B, C, L, dx,
                                   ws, stream);
  return ADP_LAUNCH_OK() == ADP_OK ? NT : (int64_t)ADP_ERR_LAUNCH;
}

extern "C" int adp_modulation_bwd_reduce(const float* const* ws, float* const* dss, int64_t n, int64_t B, int64_t C,
                                         int64_t NT, int64_t dss_bstride, void* stream) {
  if (!ws || !dss) return ADP_ERR_NULL;
  if (n < 1 || B <= 0 || C <= 0 || NT <= 0 || B > 65535) return ADP_ERR_SHAPE;
  for (int64_t at = 0; at < n; at += MOD_BATCH) {
    adp_mod_batch q;
    const int m = (int)(n - at < MOD_BATCH ? n - at : MOD_BATCH);
    for (int i = 0; i < MOD_BATCH; ++i) {
      q.ws[i] = ws[at + (i < m ? i : 0)];
      q.out[i] = dss[at + (i < m ? i : 0)];
      if (!q.ws[i] || !q.out[i]) return ADP_ERR_NULL;
    }
    if (NT >= 128 && C <= 64)  // (the choice launch_reduce_tiles makes)
      ADP_LAUNCH(reduce_tiles_wave_batch_kernel, dim3((unsigned)adp_cdiv(B * 2 * C, 4), 1, (unsigned)m), dim3(256), stream, q,
                 B, C, NT, dss_bstride);
    else
      ADP_LAUNCH(reduce_tiles_batch_kernel, dim3((unsigned)adp_cdiv(2 * C, 64), (unsigned)B, (unsigned)m), dim3(1024),
                 stream, q, C, NT, dss_bstride);
  }
  return ADP_LAUNCH_OK();
}

extern "C" int adp_ln_bwd(const float* x, const float* dxn, const float* stats, const float* gamma, const float* dres,
                          int64_t B, int64_t C, int64_t L, int64_t accumulate, float* dx, float* dgamma_dbeta,
                          float* ws, void* stream) {
  if (!x || !dxn || !stats || !gamma || !dx || !dgamma_dbeta || !ws) return ADP_ERR_NULL;
  if (B <= 0 || C <= 0 || L <= 0 || B > 65535 || L >= (int64_t)1 << 31) return ADP_ERR_SHAPE;
  if (C > LN_CMAX) return ADP_ERR_UNSUPPORTED;
  const int64_t NT = launch_ln_bwd(x, dxn, (const float*)nullptr, (int64_t)0, gamma, stats, dres, B, C, L, dx, ws, stream);
  // dgamma_dbeta = [dgamma (C) | dbeta (C)]
  launch_reduce_tiles((const float*)ws, B, C, NT, (int64_t)0, 1, (int)accumulate, dgamma_dbeta, stream);
  return ADP_LAUNCH_OK();
}

extern "C" int64_t adp_modulation_ln_bwd_partial(const float* x, const float* ss, int64_t ss_bstride, const float* stats,
                                                 const float* y, const float* dxn, const float* gamma, const float* ln_stats,
                                                 const float* dres, int64_t B, int64_t C, int64_t L, int64_t accumulate,
                                                 float* dx, float* ws, float* dgamma_dbeta, float* ws_ln, void* stream) {
  if (!x || !ss || !stats || !y || !dxn || !gamma || !ln_stats || !dx || !ws || !dgamma_dbeta || !ws_ln) return ADP_ERR_NULL;
  if (B <= 0 || C <= 0 || L <= 0 || B > 65535 || L >= (int64_t)1 << 31) return ADP_ERR_SHAPE;
  if (C > LN_CMAX) return ADP_ERR_UNSUPPORTED;
  int64_t NT = -1;
  if (lnv_ok(L, x, dxn, dres, dx, stats) && lnv_ok(L, ln_stats, nullptr, nullptr, nullptr, nullptr)) {
    const LnvCfg v = lnv_cfg(C, B, L, true);
    const int VNTL = (int)adp_cdiv(L, 4 * v.lpr);
    dim3 vgrid((unsigned)VNTL, (unsigned)B);
#define ADP_LNV_CHAIN(LPR, NT_, VPT)                                                                                    \
  if (NT < 0 && v.lpr == LPR && v.nt == NT_ && v.vpt == VPT) {                                                          \
    ADP_LAUNCH((chan_lnv_bwd_chain_kernel<LPR, NT_, VPT>), vgrid, dim3(NT_), stream, x, ss, ss_bstride, stats, dxn,    \
               gamma, ln_stats, dres, (int)C, (int)L, VNTL, dx, ws, ws_ln);                                             \
    NT = VNTL;                                                                                                          \
  }
    ADP_LNV_CHAIN(64, 256, 2) ADP_LNV_CHAIN(32, 256, 4) ADP_LNV_CHAIN(16, 256, 4) ADP_LNV_CHAIN(8, 256, 4) ADP_LNV_CHAIN(8, 1024, 2)
    ADP_LNV_CHAIN(4, 1024, 2) ADP_LNV_CHAIN(2, 1024, 1) ADP_LNV_CHAIN(4, 1024, 4) ADP_LNV_CHAIN(2, 1024, 2) ADP_LNV_CHAIN(1, 1024, 1)
    ADP_LNV_CHAIN(4, 512, 4) ADP_LNV_CHAIN(4, 512, 8) ADP_LNV_CHAIN(2, 512, 2) ADP_LNV_CHAIN(2, 512, 4) ADP_LNV_CHAIN(1, 512, 2) ADP_LNV_CHAIN(8, 512, 4)
#undef ADP_LNV_CHAIN
    if (NT > 0) launch_reduce_tiles((const float*)ws_ln, B, C, NT, (int64_t)0, 1, (int)accumulate, dgamma_dbeta, stream);
  }
  if (NT < 0) {
    // without the 16-byte form: the two launches it stands for (d(y) passes through dx: every thread reads its own elements of
    // the incoming gradient before it writes them)
    const int64_t NA = launch_ln_bwd(y, dxn, (const float*)nullptr, (int64_t)0, gamma, ln_stats, dres, B, C, L, dx, ws_ln, stream);
    launch_reduce_tiles((const float*)ws_ln, B, C, NA, (int64_t)0, 1, (int)accumulate, dgamma_dbeta, stream);
    NT = launch_ln_bwd(x, dx, ss, ss_bstride, (const float*)nullptr, stats, (const float*)nullptr, B, C, L, dx, ws, stream);
  }
  return ADP_LAUNCH_OK() == ADP_OK ? NT : (int64_t)ADP_ERR_LAUNCH;
}

extern "C" int64_t adp_skipmod_bwd_ws_bytes(int64_t B, int64_t C, int64_t L) {
  if (B <= 0 || C <= 0 || L <= 0) return ADP_ERR_SHAPE;
  return B * C * row_nsplit(B * C, L) * (int64_t)sizeof(float);
}

extern "C" int adp_skipmod_bwd(const float* g, const float* x, const float* scale, int64_t scale_bstride, int64_t B,
                               int64_t C, int64_t L, float* dx, float* dscale, int64_t dscale_bstride, float* ws,
                               void* stream) {
  if (!g || !x || !scale || !dx || !dscale || !ws) return ADP_ERR_NULL;
  if (B <= 0 || C <= 0 || L <= 0 || B * C > 65535) return ADP_ERR_SHAPE;
  const int64_t NS = row_nsplit(B * C, L), CL = adp_cdiv(L, NS);
  ADP_LAUNCH(skipmod_bwd_kernel, dim3((unsigned)NS, (unsigned)(B * C)), dim3(256), stream, g, x, scale,
             scale_bstride, C, L, NS, CL, dx, ws);
  ADP_LAUNCH(reduce_rows_kernel, dim3((unsigned)adp_cdiv(B * C, 4)), dim3(256), stream, (const float*)ws, B, C, NS,
             dscale_bstride, 0, 0, dscale);
  return ADP_LAUNCH_OK();
}
