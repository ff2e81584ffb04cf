// Streaming ConvBlock convolution for the HBM-bound depth-1 layers: 32 -> 32 channels, kernel 3, stride 1, 'same'
// (ResnetItem ConvBlocks at channels = 32 and their data gradients; /root/reference/audio_diffusion_pytorch/
// components.py:89, SURVEY.md 8a row a13, 8d "HBM-bound" rows).
//
// At [4, 32, 65536] a ConvBlock conv moves 67-100 MB (A_in + A_out (+A_res)) for 1.6 GFLOP: arithmetic intensity
// ~20 flop/B, right at the MI355X ridge for the exact-f32 matrix cores.  The kernel is therefore a stream with the
// matrix work tucked under it, built BARRIER-FREE, one wave per tile:
//   * a wave owns 32 output channels x 64 positions and never talks to another wave: no LDS, no workgroup barrier.
//     Two to three such waves share a SIMD, so one wave's global loads / epilogue stores run under another's MFMAs
//     (the first stream kernel of this repository -- loader and MMA waves around a double-buffered LDS tile, one
//     barrier per tile -- ran its 16.7 us of streaming and its 12.5 us of MFMAs back to back: DESIGN.md section 4);
//   * the contraction is Winograd F(2,3) (two thirds of the MFMAs of the direct form, plain fp32: conv_mm_impl.h WN):
//     column l31 of the tile is an output PAIR, so a lane's MFMA B operand for input channel r is formed from the four
//     inputs x[r][2j-1 .. 2j+2] -- ONE 16-byte global load at a 4-byte aligned address straight into registers
//     (a half-wave reads 264 contiguous bytes of the row) -- and its output pair is ONE 8-byte store per channel;
//   * the 32 x 96 weight matrix lives in registers as the four Winograd planes U = G g of the MFMA A operand
//     (64 VGPRs), loaded once per wave from the L2-resident 12 KB tensor;
//   * GroupNorm + SiLU of the input (prologue) is applied to the loaded quads in registers; zero padding after it;
//   * the residual is loaded while the second half of the MFMAs runs and added in the epilogue, which also writes the
//     GroupNorm partial statistics of the OUTPUT (one (mean, M2, count) entry per 4-channel row quad and tile: the
//     layout of conv_mm's epilogue, so adp_gn_finalize serves both).
// Algorithmic bytes per launch: 4 * B * 32 * L * (2 + has_res) + 12 KB of weights.
#include <stdlib.h>
#include "adp_rt.h"
#include "adp.h"
#include "conv_internal.h"

namespace {

constexpr int ST_C = 32;   // channels in = channels out
constexpr int ST_KT = 3;
constexpr int ST_TN = 64;  // positions per wave tile (32 output pairs)

template <bool TR, int PRO>
__global__ __launch_bounds__(256, 2) void conv_stream32_kernel(adp_conv_desc d, int tiles_per_row, int total_tiles) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int hi = lane >> 5, l31 = lane & 31;
  const int tile = blockIdx.x * 4 + wave;
  if (tile >= total_tiles) return;  // (waves are independent: no barrier follows)
  const int b = tile / tiles_per_row, tr = tile - b * tiles_per_row, n0 = tr * ST_TN;
  const int L = (int)d.Lin;

  // ---- Winograd planes of the weights: u[s][k], s = 4 * g + c <-> input channel r = 8 * g + c + 4 * hi (the MFMA K
  // pair of step s is the two half-waves' channels), output row m = l31.  U = (g0, g0+g1+g2, g0-g1+g2, g2): the halves
  // of G are applied once in the output transform.
  float u[16][4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    float wv[12];  // taps of channels 8g + 4hi .. +3
    if (!TR) {
      const float* wp = d.w + ((int64_t)l31 * ST_C + 8 * g + 4 * hi) * ST_KT;  // 12 contiguous floats, 16-byte aligned
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(wp + 4 * q);
#pragma unroll
        for (int k = 0; k < 4; ++k) wv[4 * q + k] = t[k];
      }
    } else {  // transposed view with flipped taps: g_t = w[r][m][2 - t]
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int t = 0; t < ST_KT; ++t)
          wv[c * ST_KT + t] = d.w[((int64_t)(8 * g + c + 4 * hi) * ST_C + l31) * ST_KT + (ST_KT - 1 - t)];
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float g0 = wv[c * 3], g1 = wv[c * 3 + 1], g2 = wv[c * 3 + 2], gs = g0 + g2;
      u[4 * g + c][0] = g0;
      u[4 * g + c][1] = gs + g1;
      u[4 * g + c][2] = gs - g1;
      u[4 * g + c][3] = g2;
    }
  }

  // ---- input quads d = x[r][n0 + 2j - 1 .. n0 + 2j + 2] of this lane's pair j = l31 for its 16 channels
  const float* xb = d.x + (int64_t)b * ST_C * L;
  const int p0 = n0 + 2 * l31 - 1;
  const bool edge = (n0 == 0) || (n0 + ST_TN >= L);  // wave-uniform: only the row's first / last tile pads
  f32x4 xq[16];
  if (!edge) {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int r = 8 * (s >> 2) + (s & 3) + 4 * hi;
      xq[s] = *reinterpret_cast<const f32x4u*>(xb + (int64_t)r * L + p0);
    }
  } else {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int r = 8 * (s >> 2) + (s & 3) + 4 * hi;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int p = p0 + i;
        const int pc = p < 0 ? 0 : (p >= L ? L - 1 : p);
        xq[s][i] = xb[(int64_t)r * L + pc];
      }
    }
  }
  if (PRO == 1) {  // GroupNorm + SiLU of the input; channel r's group = r / (32 / G)
    const int cpg = ST_C / (int)d.groups;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int r = 8 * (s >> 2) + (s & 3) + 4 * hi;
      const float* st = d.pro_stats + ((int64_t)b * d.groups + r / cpg) * 2;
      const float pa = (d.pro_gamma ? d.pro_gamma[r] : 1.0f) * st[1];
      const float pb = (d.pro_beta ? d.pro_beta[r] : 0.0f) - st[0] * pa;
#pragma unroll
      for (int i = 0; i < 4; ++i) xq[s][i] = adp_silu_fast(fmaf(xq[s][i], pa, pb));
    }
  }
  if (edge) {  // zero padding is applied after the activation, like nn.Conv1d
#pragma unroll
    for (int s = 0; s < 16; ++s)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int p = p0 + i;
        xq[s][i] = (p >= 0 && p < L) ? xq[s][i] : 0.0f;
      }
  }

  // ---- 64 MFMAs: four Winograd planes, V = (d0-d2, d1+d2, d2-d1, d1-d3)
  f32x16 acc[4];
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[k][r] = 0.0f;
  const int n = n0 + 2 * l31;  // this lane's output pair
  const int64_t obase = (int64_t)b * ST_C * L + n;
  const bool has_res = d.res != nullptr;
  f32x2 rv[16];
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    if (s == 8 && has_res) {  // the first half of the input registers is free now: fetch the residual under the rest
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = (r & 3) + 8 * (r >> 2) + 4 * hi;
        rv[r] = *reinterpret_cast<const f32x2*>(d.res + obase + (int64_t)m * L);
      }
    }
    const float d0 = xq[s][0], d1 = xq[s][1], d2 = xq[s][2], d3 = xq[s][3];
    acc[0] = adp_mfma32(u[s][0], d0 - d2, acc[0]);
    acc[1] = adp_mfma32(u[s][1], d1 + d2, acc[1]);
    acc[2] = adp_mfma32(u[s][2], d2 - d1, acc[2]);
    acc[3] = adp_mfma32(u[s][3], d1 - d3, acc[3]);
  }

  // ---- output transform y0 = P0 + (P1+P2)/2, y1 = (P1-P2)/2 - P3; bias, residual, 8-byte stores
  float v0[16], v1[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = (r & 3) + 8 * (r >> 2) + 4 * hi;
    const float bv = d.bias ? d.bias[m] : 0.0f;
    const float p1 = acc[1][r], p2 = acc[2][r];
    v0[r] = fmaf(0.5f, p1 + p2, acc[0][r]) + bv;
    v1[r] = fmaf(0.5f, p1 - p2, -acc[3][r]) + bv;
    if (has_res) {
      v0[r] += rv[r][0];
      v1[r] += rv[r][1];
    }
    *reinterpret_cast<f32x2*>(d.out + obase + (int64_t)m * L) = f32x2{v0[r], v1[r]};
  }

  // ---- GroupNorm partial statistics of the tile: accumulator registers 4q .. 4q+3 of a half-wave are the 4-channel
  // row quad 2q + hi; 4 rows x 64 positions per entry, two passes in registers
  if (d.gn_part != nullptr) {
    const float fcnt = 4.0f * (float)ST_TN;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float sv = 0.0f;
#pragma unroll
      for (int j = 0; j < 4; ++j) sv += v0[4 * q + j] + v1[4 * q + j];
#pragma unroll
      for (int o = 1; o < 32; o <<= 1) sv += __shfl_xor(sv, o, 64);
      const float mean = sv / fcnt;
      float qv = 0.0f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float e0 = v0[4 * q + j] - mean, e1 = v1[4 * q + j] - mean;
        qv = fmaf(e0, e0, fmaf(e1, e1, qv));
      }
#pragma unroll
      for (int o = 1; o < 32; o <<= 1) qv += __shfl_xor(qv, o, 64);
      if (l31 == 0) {
        float* e = d.gn_part + (((int64_t)b * (ST_C / 4) + 2 * q + hi) * tiles_per_row + tr) * 3;
        e[0] = mean;
        e[1] = qv;
        e[2] = fcnt;
      }
    }
  }
}

// ---- persistent, software-pipelined form for launches with several tiles per SIMD ----------------------------------
// One wave per SIMD (up to 512 VGPRs) walks a contiguous range of tiles with THREE input buffers in registers:
//   iteration i:  issue the 16 loads of tile i+2 | issue the residual loads of tile i |
//                 16 x { 4 MFMAs of tile i ; GroupNorm+SiLU of the same 16-byte piece of tile i+1 (loaded an iteration
//                        ago: no wait) } -- the activation VALU work is issued in the shadow of the 64-cycle MFMAs |
//                 output transform + stores of tile i
// so the global loads have a whole MFMA phase (~2 us) to arrive, the stores drain under the next phase, and the
// matrix pipe only idles during the epilogue's VALU.  (With one tile per wave the chip runs its waves in lockstep --
// every wave loads, then every wave multiplies, then every wave stores: 23 us for the plain data gradient at
// [4, 32, 65536], the same as the LDS-tile kernel before it.)
constexpr int ST_PB = 64;  // batch elements whose GroupNorm constants fit the persistent kernel's LDS table

template <bool TR, int PRO>
__global__ __launch_bounds__(256, 1) void conv_stream32p_kernel(adp_conv_desc d, int tiles_per_row, int total_tiles,
                                                                int total_waves) {
  // workgroup-wide constants in LDS (read with ds_read, whose counter is separate from the global loads' -- a global
  // load of a constant inside the pipeline would order itself behind the 16 prefetch loads in flight):
  //   U[s][lane][k]  Winograd planes of the weights as the MFMA A operand of step s (lane = (hi, l31)), 16 KB
  //   pa / pb        per (batch element, channel) scale and shift of the GroupNorm prologue;  bias
  __shared__ __attribute__((aligned(16))) float sU[16 * 64 * 4];
  __shared__ float sPa[PRO == 1 ? ST_PB * ST_C : 1], sPb[PRO == 1 ? ST_PB * ST_C : 1], sBias[ST_C];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hi = lane >> 5, l31 = lane & 31;
  const int L = (int)d.Lin;
  {  // each wave fills a quarter of the steps of U (the table is the same for the four waves)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int s = 4 * wave + q, r = 8 * (s >> 2) + (s & 3) + 4 * hi;
      float g0, g1, g2;
      if (!TR) {
        const float* wp = d.w + ((int64_t)l31 * ST_C + r) * ST_KT;
        g0 = wp[0], g1 = wp[1], g2 = wp[2];
      } else {  // transposed view with flipped taps
        const float* wp = d.w + ((int64_t)r * ST_C + l31) * ST_KT;
        g0 = wp[2], g1 = wp[1], g2 = wp[0];
      }
      const float gs = g0 + g2;
      *reinterpret_cast<f32x4*>(sU + (s * 64 + lane) * 4) = f32x4{g0, gs + g1, gs - g1, g2};
    }
    if (tid < ST_C) sBias[tid] = d.bias ? d.bias[tid] : 0.0f;
    if (PRO == 1) {
      const int cpg = ST_C / (int)d.groups;
      for (int e = tid; e < (int)d.B * ST_C; e += 256) {
        const int bb = e / ST_C, r = e - bb * ST_C;
        const float* st = d.pro_stats + ((int64_t)bb * d.groups + r / cpg) * 2;
        const float pa = (d.pro_gamma ? d.pro_gamma[r] : 1.0f) * st[1];
        sPa[e] = pa;
        sPb[e] = (d.pro_beta ? d.pro_beta[r] : 0.0f) - st[0] * pa;
      }
    }
  }
  __syncthreads();  // the only barrier: from here on the waves are independent
  const int gw = blockIdx.x * 4 + wave;
  const int t_beg = (int)(((int64_t)gw * total_tiles) / total_waves);
  const int t_end = (int)(((int64_t)(gw + 1) * total_tiles) / total_waves);
  if (t_beg >= t_end) return;
  const bool has_res = d.res != nullptr;
  const bool want_gn = d.gn_part != nullptr;

  // the 16 raw loads of a tile (addresses clamped into the row; edge tiles are fixed up in `activate`)
  auto load_tile = [&](f32x4 (&q)[16], int t) {
    const int tt = t < t_end ? t : t_end - 1;  // (tail prefetches re-read the last tile: never consumed)
    const int b = tt / tiles_per_row, n0 = (tt - b * tiles_per_row) * ST_TN;
    int p0 = n0 + 2 * l31 - 1;
    p0 = p0 < 0 ? 0 : (p0 > L - 4 ? L - 4 : p0);
    const float* xb = d.x + (int64_t)b * ST_C * L + p0;
#pragma unroll
    for (int s = 0; s < 16; ++s) q[s] = *reinterpret_cast<const f32x4u*>(xb + (int64_t)(8 * (s >> 2) + (s & 3) + 4 * hi) * L);
  };
  // GroupNorm + SiLU + zero padding of piece s of a loaded tile
  auto activate = [&](f32x4& v, int s, int b, int n0) {
    if (PRO == 1) {
      const int r = 8 * (s >> 2) + (s & 3) + 4 * hi;
      const float pa = sPa[b * ST_C + r], pb = sPb[b * ST_C + r];
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = adp_silu_fast(fmaf(v[i], pa, pb));
    }
    if (n0 == 0 || n0 + ST_TN >= L) {  // wave-uniform: the clamped load of an edge lane is shifted by one position
      const int p0 = n0 + 2 * l31 - 1;
      const f32x4 w = v;
      if (p0 < 0) {
        v[0] = 0.0f, v[1] = w[0], v[2] = w[1], v[3] = w[2];
      }
      if (p0 > L - 4) {
        v[0] = w[1], v[1] = w[2], v[2] = w[3], v[3] = 0.0f;
      }
    }
  };

  // one pipelined iteration: tile `t` in `cur` (activated), tile t+1 in `nxt` (raw), tile t+2 goes to `fut`
  auto iteration = [&](f32x4 (&cur)[16], f32x4 (&nxt)[16], f32x4 (&fut)[16], int t) {
    load_tile(fut, t + 2);
    const int b = t / tiles_per_row, tr = t - b * tiles_per_row, n0 = tr * ST_TN;
    const int64_t obase = (int64_t)b * ST_C * L + n0 + 2 * l31;
    f32x2 rv[16];
    if (has_res) {
#pragma unroll
      for (int r = 0; r < 16; ++r) rv[r] = *reinterpret_cast<const f32x2*>(d.res + obase + (int64_t)((r & 3) + 8 * (r >> 2) + 4 * hi) * L);
    }
    const bool more = t + 1 < t_end;
    const int tn = more ? t + 1 : t;
    const int bn = tn / tiles_per_row, n0n = (tn - bn * tiles_per_row) * ST_TN;
    f32x16 acc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[k][r] = 0.0f;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const f32x4 uu = *reinterpret_cast<const f32x4*>(sU + (s * 64 + lane) * 4);
      const float d0 = cur[s][0], d1 = cur[s][1], d2 = cur[s][2], d3 = cur[s][3];
      acc[0] = adp_mfma32(uu[0], d0 - d2, acc[0]);
      acc[1] = adp_mfma32(uu[1], d1 + d2, acc[1]);
      acc[2] = adp_mfma32(uu[2], d2 - d1, acc[2]);
      acc[3] = adp_mfma32(uu[3], d1 - d3, acc[3]);
      if (more) activate(nxt[s], s, bn, n0n);  // VALU in the shadow of the MFMAs above
    }
    float v0[16], v1[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = (r & 3) + 8 * (r >> 2) + 4 * hi;
      const float p1 = acc[1][r], p2 = acc[2][r], bv = sBias[m];
      v0[r] = fmaf(0.5f, p1 + p2, acc[0][r]) + bv;
      v1[r] = fmaf(0.5f, p1 - p2, -acc[3][r]) + bv;
      if (has_res) {
        v0[r] += rv[r][0];
        v1[r] += rv[r][1];
      }
      *reinterpret_cast<f32x2*>(d.out + obase + (int64_t)m * L) = f32x2{v0[r], v1[r]};
    }
    if (want_gn) {
      const float fcnt = 4.0f * (float)ST_TN;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float sv = 0.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j) sv += v0[4 * q + j] + v1[4 * q + j];
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) sv += __shfl_xor(sv, o, 64);
        const float mean = sv / fcnt;
        float qv = 0.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float e0 = v0[4 * q + j] - mean, e1 = v1[4 * q + j] - mean;
          qv = fmaf(e0, e0, fmaf(e1, e1, qv));
        }
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) qv += __shfl_xor(qv, o, 64);
        if (l31 == 0) {
          float* e = d.gn_part + (((int64_t)b * (ST_C / 4) + 2 * q + hi) * tiles_per_row + tr) * 3;
          e[0] = mean;
          e[1] = qv;
          e[2] = fcnt;
        }
      }
    }
    // Everything older than this tile's (at most 16 + 12) stores has landed: in particular the loads of tile t+2,
    // issued a whole MFMA phase ago.  Saying so here keeps the compiler from guarding the next iteration's use of that
    // buffer with a wait that also covers the loads the next iteration has just issued (vmcnt is in issue order, and
    // across the loop back-edge the compiler assumes the worst: measured, the pipeline then ran load -> wait -> MFMA).
    adp_wait_vmcnt<28>();
  };

  f32x4 xa[16], xb2[16], xc[16];
  load_tile(xa, t_beg);
  load_tile(xb2, t_beg + 1);
  adp_wait_vmcnt<0>();
  {
    const int b0 = t_beg / tiles_per_row, n00 = (t_beg - b0 * tiles_per_row) * ST_TN;
#pragma unroll
    for (int s = 0; s < 16; ++s) activate(xa[s], s, b0, n00);  // the first tile's activation is not hidden
  }
  for (int t = t_beg; t < t_end; t += 3) {
    iteration(xa, xb2, xc, t);
    if (t + 1 < t_end) iteration(xb2, xc, xa, t + 1);
    if (t + 2 < t_end) iteration(xc, xa, xb2, t + 2);
  }
}

}  // namespace

bool adp_conv_stream_eligible(const adp_conv_desc& d) {
  if (d.R != ST_C || d.R1 != d.R || d.M != ST_C || d.KT != ST_KT) return false;
  if (d.stride != 1 || d.dil != 1 || d.pad != 1 || d.up != 1 || d.store != 0) return false;
  if (d.out_pre || d.e_scale || d.x2) return false;
  if (d.prologue != 0 && d.prologue != 1) return false;
  if (d.prologue == 1 && (d.groups < 1 || ST_C % d.groups != 0)) return false;
  if (d.N != d.Lin || d.N % ST_TN != 0) return false;
  if ((reinterpret_cast<uintptr_t>(d.x) | reinterpret_cast<uintptr_t>(d.w)) & 15) return false;
  if ((reinterpret_cast<uintptr_t>(d.out) | reinterpret_cast<uintptr_t>(d.res)) & 7) return false;
  if (d.B * (d.N / ST_TN) >= (int64_t)1 << 30 || d.B * ST_C * d.Lin >= (int64_t)1 << 40) return false;
  return true;
}

// one GroupNorm partial entry per 64-position tile (the layout conv_mm's epilogue writes)
int64_t adp_conv_stream_gn_entries(const adp_conv_desc& d) { return d.N / ST_TN; }

int adp_conv_stream(const adp_conv_desc& d, void* stream) {
  const int tiles_per_row = (int)(d.N / ST_TN);
  const int total = (int)d.B * tiles_per_row;
  // several tiles per SIMD (256 CUs x 4): the persistent pipelined form, one wave per SIMD.  ADP_STREAM_PERSIST=0 keeps
  // the one-tile-per-wave form (A/B); ADP_STREAM_PERSIST_MIN moves the threshold (tests run small cases through it).
  static const bool persist_off = getenv("ADP_STREAM_PERSIST") && getenv("ADP_STREAM_PERSIST")[0] == '0';
  const int min_tiles = getenv("ADP_STREAM_PERSIST_MIN") ? atoi(getenv("ADP_STREAM_PERSIST_MIN")) : 3 * 1024;
  if (!persist_off && total >= min_tiles && (d.prologue == 0 || d.B <= ST_PB) && d.Lin >= 4) {
    const int waves = 1024;
    const dim3 pgrid(256);
    if (d.transposed) {
      if (d.prologue == 1)
        ADP_LAUNCH((conv_stream32p_kernel<true, 1>), pgrid, dim3(256), stream, d, tiles_per_row, total, waves);
      else
        ADP_LAUNCH((conv_stream32p_kernel<true, 0>), pgrid, dim3(256), stream, d, tiles_per_row, total, waves);
    } else {
      if (d.prologue == 1)
        ADP_LAUNCH((conv_stream32p_kernel<false, 1>), pgrid, dim3(256), stream, d, tiles_per_row, total, waves);
      else
        ADP_LAUNCH((conv_stream32p_kernel<false, 0>), pgrid, dim3(256), stream, d, tiles_per_row, total, waves);
    }
    return ADP_LAUNCH_OK();
  }
  const dim3 grid((unsigned)adp_cdiv(total, 4));
  if (d.transposed) {
    if (d.prologue == 1)
      ADP_LAUNCH((conv_stream32_kernel<true, 1>), grid, dim3(256), stream, d, tiles_per_row, total);
    else
      ADP_LAUNCH((conv_stream32_kernel<true, 0>), grid, dim3(256), stream, d, tiles_per_row, total);
  } else {
    if (d.prologue == 1)
      ADP_LAUNCH((conv_stream32_kernel<false, 1>), grid, dim3(256), stream, d, tiles_per_row, total);
    else
      ADP_LAUNCH((conv_stream32_kernel<false, 0>), grid, dim3(256), stream, d, tiles_per_row, total);
  }
  return ADP_LAUNCH_OK();
}
