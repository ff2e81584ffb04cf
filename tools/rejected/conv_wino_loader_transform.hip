// conv_wino: the MFMA-bound kernel-3 convolutions (forward and data gradient of the wide layers) as Winograd F(2,3)
// on the exact-f32 matrix cores.
//
// A kernel-3 stride-1 convolution spends 6 multiplies on every pair of outputs; F(2,3) spends 4:
//     [y0 y1] = A^T [ (G g) * (B^T d) ],   d = x[2j-1 .. 2j+2],  g = the three taps
//     B^T d = (d0-d2, d1+d2, d2-d1, d1-d3)      G g = (g0, (g0+g1+g2)/2, (g0-g1+g2)/2, g2)
//     y0 = m0+m1+m2     y1 = m1-m2-m3
// Summed over input channels the four products m_k are four independent GEMMs  M_k[m][j] = sum_r U_k[m][r] V_k[r][j]
// over output PAIRS j: 4 * M * R * N/2 multiply-adds instead of 3 * M * R * N -- two thirds of the MFMA work of the
// direct form, in plain fp32 (the transforms are a handful of adds; differences to the direct form are rounding, 1e-6).
// The deep convs of the U-Net are MFMA-bound (conv_mm: 49 of 64 us at depth 7 are the MFMA stream), which is what
// this trades on.
//
// Structure (same skeleton as conv_mm): 32 output channels x 64 positions (32 pairs) per workgroup; 4 loader waves
// + 4 MMA waves, MMA wave k owns plane k (one 32 x 32 accumulator, 16 MFMAs per 32-channel chunk); the loaders apply
// G to the weights and B^T to the activations between the global load and the LDS store (two chunks ahead in
// registers), two LDS stages, one barrier per chunk; at the end the four planes meet in LDS, every wave applies A^T
// to a quarter of the rows and runs the usual epilogue (bias / out_pre / e_scale / residual, 8-byte stores,
// GroupNorm row-quad partial statistics).  Small grids take the cross-workgroup K split of conv_mm (the output
// transform is linear, so partial tiles are transformed before they are parked).
#include <stdlib.h>
#include "adp_rt.h"
#include "conv_internal.h"

namespace {

constexpr int WN_BM = 32, WN_BN = 64, WN_BK = 32;
constexpr int WN_NMMA = 4, WN_NLD = 4;
constexpr int WN_AS = 36;                       // weight plane row stride (floats): 4 mod 8 dwords, b128 reads conflict-free
constexpr int WN_VS = 32;                       // activation plane row stride: one row = the 32 pairs of a channel
constexpr int WN_A = 4 * 32 * WN_AS;            // four planes [32][WN_AS]
constexpr int WN_V = 4 * WN_BK * WN_VS;         // four planes [channel][pair]
constexpr int WN_STAGE = WN_A + WN_V;           // 8704 floats; two stages = 69.6 KB (two workgroups per CU)

// TR = false: out[m][n] = sum_{r,t} w[m][r][t] x[r][n + t - 1]     (w: [M][R][3])
// TR = true : out[m][n] = sum_{r,t} w[r][m][t] x[r][n + 1 - t]     (w: [R][M][3]: the same form with flipped taps)
template <bool TR>
__global__ __launch_bounds__((WN_NMMA + WN_NLD) * 64, 4) void conv_wino_kernel(adp_conv_desc d, int KS) {
  __shared__ __attribute__((aligned(16))) float smem[2 * WN_STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hi = lane >> 5, l31 = lane & 31;
  const int M = (int)d.M, R = (int)d.R, L = (int)d.Lin, N = (int)d.N;

  int id = blockIdx.x;
  const int total = gridDim.x;
  if ((total & 7) == 0) id = (id & 7) * (total >> 3) + (id >> 3);  // an XCD's L2 sees a contiguous range of weight rows
  const int ntn = (N + WN_BN - 1) / WN_BN, per_m = ntn * (int)d.B;
  const int mt = id / per_m, rem = id - mt * per_m;
  const int b = rem / ntn, nt = rem - b * ntn;
  const int m0 = mt * WN_BM, n0 = nt * WN_BN;
  const int ks = blockIdx.y;
  const int nch = R / WN_BK;
  const int c_lo = (int)((int64_t)nch * ks / KS), c_hi = (int)((int64_t)nch * (ks + 1) / KS);

  if (wave >= WN_NMMA) {
    // ------------------------------------------------------------------ loader waves
    const int lt = tid - WN_NMMA * 64;  // 0 .. 255
    // weights: one task per thread = 4 channels x 3 taps of one row (forward) / 4 rows x 3 taps of one channel (TR)
    const int aq = lt & 7, ar = lt >> 3;
    const float* wsrc = TR ? d.w + ((int64_t)ar * M + m0 + 4 * aq) * 3 : d.w + ((int64_t)(m0 + ar) * R + 4 * aq) * 3;
    const int64_t wstep = TR ? (int64_t)WN_BK * M * 3 : (int64_t)WN_BK * 3;  // per chunk
    const int adst = ar * WN_AS + 4 * aq;
    // activations: two tasks per thread = 4 positions (2 pairs) of one channel each.  The 16 lanes of a channel hold
    // its 64 positions, so the neighbours x[u-1] / x[u+4] of a quad come from the adjacent lanes; only the tile's two
    // outer neighbours are loaded (lanes 0-7 of a channel fetch the left one, 8-15 the right one: one load per lane).
    const float* xb = d.x + (int64_t)b * R * L;
    const int q = lt & 15;
    const bool x_ok = n0 + 4 * q < L;  // L % 4 == 0: a quad is entirely inside or outside the row
    const int epos = q < 8 ? n0 - 1 : n0 + WN_BN;
    const bool e_ok = epos >= 0 && epos < L;
    int xoff[2], eoff[2], vdst[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = (lt + 256 * i) >> 4;
      xoff[i] = c * L + (x_ok ? n0 + 4 * q : 0);
      eoff[i] = c * L + (e_ok ? epos : 0);
      vdst[i] = WN_A + c * WN_VS + 2 * q;
    }
    f32x4 ra[3][3], rx[3][2];
    float re[3][2];
    auto issue = [&](int c, int st) {
      const float* wp = wsrc + (int64_t)c * wstep;
#pragma unroll
      for (int q = 0; q < 3; ++q) ra[st][q] = *reinterpret_cast<const f32x4*>(wp + 4 * q);
      const float* xp = xb + (int64_t)c * WN_BK * L;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        rx[st][i] = *reinterpret_cast<const f32x4*>(xp + xoff[i]);
        re[st][i] = xp[eoff[i]];
      }
    };
    auto stage = [&](int st, int buf) {
      float* S = smem + buf * WN_STAGE;
      // G g for four (row, channel) pairs; TR reads the taps flipped
      float g[12];
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int k = 0; k < 4; ++k) g[4 * q + k] = ra[st][q][k];
      f32x4 u0, u1, u2, u3;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float g0 = TR ? g[3 * j + 2] : g[3 * j], g1 = g[3 * j + 1], g2 = TR ? g[3 * j] : g[3 * j + 2];
        const float s = g0 + g2;
        u0[j] = g0;
        u1[j] = 0.5f * (s + g1);
        u2[j] = 0.5f * (s - g1);
        u3[j] = g2;
      }
      float* ad = S + adst;
      *reinterpret_cast<f32x4*>(ad) = u0;
      *reinterpret_cast<f32x4*>(ad + 32 * WN_AS) = u1;
      *reinterpret_cast<f32x4*>(ad + 2 * 32 * WN_AS) = u2;
      *reinterpret_cast<f32x4*>(ad + 3 * 32 * WN_AS) = u3;
      // B^T d for the two pairs of each activation quad (zero padding = zeros outside the row)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        f32x4 v = rx[st][i];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = x_ok ? v[k] : 0.0f;
        const float edge = e_ok ? re[st][i] : 0.0f;
        const float pl = __shfl(v[3], lane - 1, 64), pr = __shfl(v[0], lane + 1, 64);
        const float xl = q == 0 ? edge : pl, xr = q == 15 ? edge : pr;
        float* vd = S + vdst[i];
        *reinterpret_cast<f32x2*>(vd) = f32x2{xl - v[1], v[1] - v[3]};
        *reinterpret_cast<f32x2*>(vd + WN_BK * WN_VS) = f32x2{v[0] + v[1], v[2] + v[3]};
        *reinterpret_cast<f32x2*>(vd + 2 * WN_BK * WN_VS) = f32x2{v[1] - v[0], v[3] - v[2]};
        *reinterpret_cast<f32x2*>(vd + 3 * WN_BK * WN_VS) = f32x2{v[0] - v[2], v[2] - xr};
      }
    };
    // chunk i of this workgroup: register stage i % 3 (two chunks in flight beside the one being staged), LDS stage
    // i % 2; constant indices, six chunks per trip.  The store of chunk i goes to the LDS stage the MMA waves left
    // at barrier B_{i-1}.
    auto step = [&](int c, int rs, int buf) {
      if (c + 2 < c_hi) issue(c + 2, (rs + 2) % 3);
      stage(rs, buf);
      __syncthreads();  // B_c
    };
    if (c_lo < c_hi) issue(c_lo, 0);
    if (c_lo + 1 < c_hi) issue(c_lo + 1, 1);
    for (int c = c_lo; c < c_hi; c += 6) {
      step(c, 0, 0);
      if (c + 1 < c_hi) step(c + 1, 1, 1);
      if (c + 2 < c_hi) step(c + 2, 2, 0);
      if (c + 3 < c_hi) step(c + 3, 0, 1);
      if (c + 4 < c_hi) step(c + 4, 1, 0);
      if (c + 5 < c_hi) step(c + 5, 2, 1);
    }
    __syncthreads();  // (pairs with "staging buffers free" below)
    __syncthreads();  // (pairs with the plane exchange barrier)
    return;
  }

  // -------------------------------------------------------------------- MMA waves: wave k multiplies plane k
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  const int aoff = wave * 32 * WN_AS + (TR ? l31 + 4 * hi * WN_AS : l31 * WN_AS + 4 * hi);
  const int voff = WN_A + wave * WN_BK * WN_VS + 4 * hi * WN_VS + l31;
  for (int c = c_lo; c < c_hi; ++c) {
    __syncthreads();  // B_c: chunk c is in LDS stage (c - c_lo) & 1
    const float* S = smem + ((c - c_lo) & 1) * WN_STAGE;
#pragma unroll
    for (int s = 0; s < WN_BK / 8; ++s) {
      float a[4];
      if (!TR) {
        const f32x4 q = *reinterpret_cast<const f32x4*>(S + aoff + 8 * s);
#pragma unroll
        for (int k = 0; k < 4; ++k) a[k] = q[k];
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) a[k] = S[aoff + (8 * s + k) * WN_AS];
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) acc = adp_mfma32(a[k], S[voff + (8 * s + k) * WN_VS], acc);
    }
  }
  __syncthreads();  // the staging buffers are free

  // ---- plane exchange: wave k parks M_k; then wave w finishes accumulator rows 4w .. 4w+3 of all four planes
#pragma unroll
  for (int r = 0; r < 16; ++r) smem[(wave * 16 + r) * 64 + lane] = acc[r];
  __syncthreads();

  const int64_t ebs = d.e_bstride ? d.e_bstride : M;
  const int n = n0 + 2 * l31;
  const bool nok = n < N;  // N is even: a pair is inside or outside
  float vfin[4][2];
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int r = 4 * wave + rr;
    const float p0 = smem[(0 * 16 + r) * 64 + lane], p1 = smem[(1 * 16 + r) * 64 + lane];
    const float p2 = smem[(2 * 16 + r) * 64 + lane], p3 = smem[(3 * 16 + r) * 64 + lane];
    float y0 = (p0 + p1) + p2, y1 = (p1 - p2) - p3;
    const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
    vfin[rr][0] = vfin[rr][1] = 0.0f;
    if (!nok) continue;
    if (KS > 1) {  // raw partial tile; the epilogue runs in the reduce kernel
      *reinterpret_cast<f32x2*>(d.ws + (((int64_t)ks * d.B + b) * M + m) * N + n) = f32x2{y0, y1};
      continue;
    }
    const int64_t o = ((int64_t)b * M + m) * N + n;
    if (d.bias) {
      const float bv = d.bias[m];
      y0 += bv;
      y1 += bv;
    }
    if (d.out_pre) *reinterpret_cast<f32x2*>(d.out_pre + o) = f32x2{y0, y1};
    if (d.e_scale) {
      const float sc = d.e_scale[b * ebs + m];
      y0 *= sc;
      y1 *= sc;
    }
    if (d.res) {
      const f32x2 rv = *reinterpret_cast<const f32x2*>(d.res + o);
      y0 += rv[0];
      y1 += rv[1];
    }
    *reinterpret_cast<f32x2*>(d.out + o) = f32x2{y0, y1};
    vfin[rr][0] = y0;
    vfin[rr][1] = y1;
  }
  // ---- GroupNorm partial statistics of the tile (layout of conv_mm: one (mean, M2, count) entry per 4-channel row
  // quad and 64-position tile): the 4 accumulator rows of this wave are one quad per half-wave
  if (d.gn_part != nullptr && KS == 1) {
    const int cntv = (N - n0) < WN_BN ? (N - n0) : WN_BN;
    const float fcnt = 4.0f * (float)cntv;
    float sv = 0.0f;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) sv += vfin[rr][0] + vfin[rr][1];
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) sv += __shfl_xor(sv, o, 64);
    const float mean = sv / fcnt;
    float qv = 0.0f;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const float d0 = nok ? vfin[rr][0] - mean : 0.0f, d1 = nok ? vfin[rr][1] - mean : 0.0f;
      qv = fmaf(d0, d0, fmaf(d1, d1, qv));
    }
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) qv += __shfl_xor(qv, o, 64);
    if (l31 == 0) {
      const int m = m0 + 8 * wave + 4 * hi;  // first channel of the quad
      float* e = d.gn_part + (((int64_t)b * (M / 4) + (m >> 2)) * ntn + nt) * 3;
      e[0] = mean;
      e[1] = qv;
      e[2] = fcnt;
    }
  }
}

}  // namespace

// First-generation Winograd kernel (transforms in the loader waves): only on request, ADP_CONV_WINO=L, for A/B against
// conv_mm's WN variant, which replaced it (conv_mm.hip: adp_conv_wino_env).
int adp_conv_wino_env();
bool adp_conv_wino_enabled() { return adp_conv_wino_env() == 'L'; }

bool adp_conv_wino_eligible(const adp_conv_desc& d) {
  if (d.KT != 3 || d.stride != 1 || d.up != 1 || d.dil != 1 || d.pad != 1 || d.prologue != 0 || d.store != 0) return false;
  if (d.R1 != d.R || d.N != d.Lin || d.Lin % 4 != 0) return false;
  if (d.R % WN_BK != 0 || d.M % WN_BM != 0 || d.R < 256) return false;
  if ((reinterpret_cast<uintptr_t>(d.x) | reinterpret_cast<uintptr_t>(d.w) | reinterpret_cast<uintptr_t>(d.out) |
       reinterpret_cast<uintptr_t>(d.res) | reinterpret_cast<uintptr_t>(d.out_pre) | reinterpret_cast<uintptr_t>(d.ws)) & 15)
    return false;
  if (d.B * d.R * d.Lin >= (int64_t)1 << 31 || d.M * d.R * 3 >= (int64_t)1 << 31) return false;
  return true;
}

int64_t adp_conv_wino_ksplit(const adp_conv_desc& d) {
  const int64_t blocks = (d.M / WN_BM) * adp_cdiv(d.N, WN_BN) * d.B;
  const int64_t nch = d.R / WN_BK;
  int64_t ks = 1;
  while (ks < 8 && blocks * ks < 200 && nch / (ks * 2) >= 4) ks *= 2;
  return ks;
}

int adp_conv_wino(const adp_conv_desc& d, void* stream) {
  const int64_t blocks = (d.M / WN_BM) * adp_cdiv(d.N, WN_BN) * d.B;
  const int KS = d.ws ? (int)adp_conv_wino_ksplit(d) : 1;
  if (d.transposed)
    ADP_LAUNCH((conv_wino_kernel<true>), dim3((unsigned)blocks, (unsigned)KS), dim3((WN_NMMA + WN_NLD) * 64), stream, d, KS);
  else
    ADP_LAUNCH((conv_wino_kernel<false>), dim3((unsigned)blocks, (unsigned)KS), dim3((WN_NMMA + WN_NLD) * 64), stream, d, KS);
  if (ADP_LAUNCH_OK() != ADP_OK) return ADP_ERR_LAUNCH;
  if (KS > 1) return adp_conv_splitk_reduce(d, KS, stream);
  return ADP_OK;
}
