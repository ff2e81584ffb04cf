// Weight gradient of the HBM-bound depth-1 ConvBlock convs (32 -> 32 channels, kernel 3, stride 1, 'same') straight from
// registers: dW[m][r][t] = sum_{b,n} dy[b][m][n] * a[b][r][n + t - 1], a = SiLU(GroupNorm(x)) recomputed on load
// (ResnetItem ConvBlocks at channels = 32; /root/reference/audio_diffusion_pytorch/components.py:89).
//
// At [4, 32, 65536] the gradient reads 67 MB for 1.6 GFLOP and writes 12 KB: a stream with a tiny output.  The contraction
// runs over POSITIONS, and both MFMA operands of v_mfma_f32_32x32x2_f32 want "row = lane": A[m][k] in lane m, B[k][r] in lane
// r, k = lane >> 5.  Pairing k-step s of a 32-position tile with the positions (n0 + s, n0 + 16 + s) makes lane (row, hi) own
// the 16 CONSECUTIVE positions n0 + 16 hi .. + 15 of its row: four 16-byte loads per operand and tile, every loaded byte used
// exactly once, and no LDS, no transposition, no barrier between a load and its MFMA -- the three taps are three REGISTER
// INDEX shifts of the activated window (a[s + t], 18 registers incl. the two halo values).  The wgrad_mm kernel this replaces
// for these layers staged both operands through LDS for 64 x 64 chunk tiles it could not fill (35.5 us, 0.24 of the HBM
// peak); this one runs 29-34 us -- the 384 MFMAs per SIMD (12-14 us at the sustained clock) and the 67 MB of loads still do not
// overlap the way the instruction streams would allow (stage staggering as in conv_tile.hip changed nothing here).
//   * one wave = 32 x 96 accumulators (3 taps x 16 registers) over its slice of the positions; 16 waves per workgroup;
//   * the workgroup's waves are summed in a fixed tree through LDS (deterministic), one 12 KB partial tile + 32 bias partials
//     per workgroup go to ws, and adp_wgrad_reduce sums the <= 256 partials in its fixed order.
// Algorithmic bytes per launch: 4 * B * 32 * L * 2 (+ 12 KB).
#include <stdlib.h>
#include "adp_rt.h"
#include "adp.h"
#include "conv_internal.h"

namespace {

constexpr int GT_C = 32, GT_KT = 3, GT_TN = 32;  // channels, taps, positions per wave tile
constexpr int GT_NW = 16;                        // waves per workgroup
constexpr int GT_ACC = GT_KT * 16;               // accumulator registers per lane

__device__ __forceinline__ f32x4 gt_silu4(f32x4 v, float pa, float pb) {
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] = adp_silu_fast(fmaf(v[j], pa, pb));
  return v;
}

template <int PRO>
__global__ __launch_bounds__(64 * GT_NW) void wgrad_tile32_kernel(adp_wgrad_desc d, int tiles_per_b, int ntiles, int tpw,
                                                                  int gap) {
  __shared__ float red[(GT_NW / 2) * GT_ACC * 64];
  __shared__ float bred[GT_NW * 64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = adp_uniform(tid >> 6);
  const int hi = lane >> 5, l31 = lane & 31;
  const int L = (int)d.Lin;
  const int gw = (int)blockIdx.x * GT_NW + wave;
  const int t_beg = gw * tpw, t_end = (t_beg + tpw < ntiles) ? t_beg + tpw : ntiles;

  f32x16 acc[GT_KT];
#pragma unroll
  for (int t = 0; t < GT_KT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
  float bsum = 0.0f;

  // raw loads of one tile: this lane's 16 positions of row l31 of dy and of x, and the two halo values of x
  f32x4 dq[4], xq[4];
  float xl = 0.0f, xr = 0.0f;
  bool okl = false, okr = false;
  int cur_b = -1;
  float pa = 1.0f, pb = 0.0f;
  auto load = [&](int t) {
    const int b = t / tiles_per_b, p0 = (t - b * tiles_per_b) * GT_TN + 16 * hi;
    const int64_t row = ((int64_t)b * GT_C + l31) * L + p0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      dq[i] = *reinterpret_cast<const f32x4*>(d.dy + row + 4 * i);
      xq[i] = *reinterpret_cast<const f32x4*>(d.x + row + 4 * i);
    }
    okl = p0 > 0;
    okr = p0 + 16 < L;
    xl = d.x[row - (okl ? 1 : 0)];
    xr = d.x[row + (okr ? 16 : 15)];
  };
  // Waves w, w + 4, w + 8, w + 12 share a SIMD (conv_tile.hip): stage w >> 2 asks for its first tile `gap` ticks (10 ns) later
  // than the stage before it, so that a SIMD's four waves do not all wait for memory first and then all queue for the matrix
  // pipe (in lockstep the launch was the sum of its load and MFMA phases)
  if (gap > 0 && (wave >> 2) > 0) adp_wait_until(adp_clock() + (long long)(wave >> 2) * gap);
  if (t_beg < t_end) load(t_beg);
  for (int t = t_beg; t < t_end; ++t) {
    // ---- this tile's operands out of the raw registers
    float dyv[16], a[18];
    if (PRO == 1) {
      const int b = t / tiles_per_b;
      if (b != cur_b) {  // GroupNorm constants of (b, row l31): h = x * pa + pb
        cur_b = b;
        const int64_t sg = ((int64_t)b * d.groups + l31 / (GT_C / (int)d.groups)) * 2;
        pa = (d.pro_gamma ? d.pro_gamma[l31] : 1.0f) * d.pro_stats[sg + 1];
        pb = (d.pro_beta ? d.pro_beta[l31] : 0.0f) - d.pro_stats[sg] * pa;
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x4 av = PRO == 1 ? gt_silu4(xq[i], pa, pb) : xq[i];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        dyv[4 * i + j] = dq[i][j];
        a[1 + 4 * i + j] = av[j];
      }
    }
    a[0] = okl ? (PRO == 1 ? adp_silu_fast(fmaf(xl, pa, pb)) : xl) : 0.0f;   // zero padding is applied after the activation
    a[17] = okr ? (PRO == 1 ? adp_silu_fast(fmaf(xr, pa, pb)) : xr) : 0.0f;
    // ---- the next tile's loads go out before this tile's MFMAs
    if (t + 1 < t_end) load(t + 1);
    adp_sched_fence();
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      bsum += dyv[s];
#pragma unroll
      for (int tt = 0; tt < GT_KT; ++tt) acc[tt] = adp_mfma32(dyv[s], a[s + tt], acc[tt]);
    }
  }

  // ---- the workgroup's waves, summed in a fixed tree through LDS: accumulator register i of tap t <-> m = (i & 3) +
  // 8 (i >> 2) + 4 hi, r = l31
  bred[wave * 64 + lane] = bsum;
#pragma unroll
  for (int half = GT_NW / 2; half >= 1; half >>= 1) {
    if (wave >= half && wave < 2 * half) {
      float* o = red + (wave - half) * GT_ACC * 64 + lane;
#pragma unroll
      for (int t = 0; t < GT_KT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[(t * 16 + i) * 64] = acc[t][i];
    }
    __syncthreads();
    if (wave < half) {
      const float* o = red + wave * GT_ACC * 64 + lane;
#pragma unroll
      for (int t = 0; t < GT_KT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] += o[(t * 16 + i) * 64];
    }
    __syncthreads();
  }
  if (wave == 0) {
    float* wsw = d.ws + (int64_t)blockIdx.x * (GT_C * GT_C * GT_KT);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int m = (i & 3) + 8 * (i >> 2) + 4 * hi;
      float* o = wsw + (m * GT_C + l31) * GT_KT;
#pragma unroll
      for (int t = 0; t < GT_KT; ++t) o[t] = acc[t][i];
    }
    if (d.dbias) {  // bias gradient of row l31 = the sum over the NW waves and both position halves (fixed order)
      float s = 0.0f;
#pragma unroll
      for (int w = 0; w < GT_NW; ++w) s += bred[w * 64 + l31] + bred[w * 64 + 32 + l31];
      if (hi == 0) d.ws[(int64_t)gridDim.x * (GT_C * GT_C * GT_KT) + (int64_t)blockIdx.x * GT_C + l31] = s;
    }
  }
}

struct GtPlan {
  int tiles_per_b, ntiles, tpw, grid;
};
static GtPlan gt_plan(const adp_wgrad_desc& d) {
  GtPlan p;
  p.tiles_per_b = (int)(d.N / GT_TN);
  p.ntiles = (int)(d.B * p.tiles_per_b);
  // one resident generation of 256 x 16 waves when the problem has that many tiles, fewer workgroups below
  int waves = 256 * GT_NW;
  if (const char* e = getenv("ADP_WGRAD_TILE_WAVES")) waves = atoi(e);  // tests: several tiles per wave on small problems
  if (waves < GT_NW) waves = GT_NW;
  if (waves > p.ntiles) waves = p.ntiles;
  p.tpw = (p.ntiles + waves - 1) / waves;
  p.grid = (int)adp_cdiv(adp_cdiv(p.ntiles, p.tpw), GT_NW);
  return p;
}

}  // namespace

bool adp_wgrad_tile_eligible(const adp_wgrad_desc& d) {
  const char* e = getenv("ADP_WGRAD_TILE");  // "0": never (A/B against wgrad_mm), "1": also for small problems (tests)
  if (e && e[0] == '0') return false;
  if (d.R != GT_C || d.R1 != d.R || d.M != GT_C || d.KT != GT_KT) return false;
  if (d.stride != 1 || d.dil != 1 || d.pad != 1 || d.up != 1 || d.x2) return false;
  if (d.prologue != 0 && d.prologue != 1) return false;
  if (d.prologue == 1 && (d.groups < 1 || GT_C % d.groups != 0)) return false;
  if (d.N != d.Lin || d.N % GT_TN != 0) return false;
  if ((reinterpret_cast<uintptr_t>(d.x) | reinterpret_cast<uintptr_t>(d.dy)) & 15) return false;
  if (d.B * d.N >= (int64_t)1 << 31 || d.B * GT_C * d.Lin >= (int64_t)1 << 40) return false;
  // Two tiles per wave of a full generation (256 workgroups x 16 waves) is where this form wins: [4, 32, 65536] 34.0 ->
  // 28.8 us without / 38.3 -> 34.4 us with the GroupNorm+SiLU prologue (hipGraph of 20 launches, reduce included); with one
  // tile per wave or less there is nothing to overlap a wave's loads with ([1, 32, 65536]: 12.6 -> 17.3 us) and wgrad_mm stays
  if (!(e && e[0] == '1') && d.B * (d.N / GT_TN) < 2 * 256 * GT_NW) return false;
  return true;
}

int64_t adp_wgrad_tile_ws_floats(const adp_wgrad_desc& d) {
  return (int64_t)gt_plan(d).grid * (GT_C * GT_C * GT_KT + GT_C);
}

int adp_wgrad_tile(const adp_wgrad_desc& d, void* stream) {
  const GtPlan p = gt_plan(d);
  int gap = 120;
  if (const char* e = getenv("ADP_WGRAD_TILE_GAP")) gap = atoi(e);  // kernel work
  if (d.prologue == 1)
    ADP_LAUNCH((wgrad_tile32_kernel<1>), dim3((unsigned)p.grid), dim3(64 * GT_NW), stream, d, p.tiles_per_b, p.ntiles, p.tpw,
               gap);
  else
    ADP_LAUNCH((wgrad_tile32_kernel<0>), dim3((unsigned)p.grid), dim3(64 * GT_NW), stream, d, p.tiles_per_b, p.ntiles, p.tpw,
               gap);
  return adp_wgrad_reduce(d.ws, p.grid, GT_C * GT_C * GT_KT, GT_C, d.dw, d.dbias, (int)d.accumulate, stream);
}
