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
#include <type_traits>
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

// ---- kernel 3, stride 1, at most 8 x 8 channels (the depth-0 ConvBlocks): the weight gradient as a register streaming
// reduction -- no LDS staging (the generic kernel above is bound by its LDS fragment reads at these shapes: four 16-byte reads per
// 12 FMAs).  Lane = (input row r, quad q): the 8 lanes of a row group cover 32 consecutive positions, the 8 groups of the wave
// the 8 input rows; a lane loads ITS row's quad (+ the halo from its neighbour lanes: DPP row shifts of the activated values) and
// the dy quads of ALL output rows (the eight row groups read the same addresses: one fetch per wave instruction), and keeps the
// sums dw[m][r][0..2], m = 0..7 in registers over its share of the positions (packed FMAs: the pair (t0, t1) against the window
// pairs, t2 as two half sums).  One 8-lane butterfly + one LDS round per workgroup at the end; partials in the layout of
// adp_wgrad_reduce like the generic kernel.
constexpr int WD8_MAXBLOCKS = 1024;

template <bool PRO>
__global__ __launch_bounds__(256) void wgrad_direct8_kernel(adp_wgrad_desc d, int bpb, int spans_b) {
  __shared__ float red[4][8][32];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = adp_uniform(tid >> 6);  // scalar: the span loop and its full / tail branch are wave-uniform
  const int r = lane >> 3, q = lane & 7;
  const int M = (int)d.M, R = (int)d.R, R1 = (int)d.R1, N = (int)d.N, G = (int)d.groups;
  const int b = blockIdx.y;
  const int rc = r < R ? r : R - 1;  // (row groups beyond R re-read row R - 1; their sums are never stored)
  const float* xrow = (rc < R1) ? d.x + ((int64_t)b * R1 + rc) * N : d.x2 + ((int64_t)b * (R - R1) + (rc - R1)) * N;
  const float* dyb = d.dy + (int64_t)b * M * N;
  float pa = 1.0f, pb = 0.0f;
  if (PRO) {
    const int g = rc / (R / G);
    const float mean = d.pro_stats[((int64_t)b * G + g) * 2];
    pa = (d.pro_gamma ? d.pro_gamma[rc] : 1.0f) * d.pro_stats[((int64_t)b * G + g) * 2 + 1];
    pb = (d.pro_beta ? d.pro_beta[rc] : 0.0f) - mean * pa;
  }
  f32x2 acc01[8], acc2[8], bs[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) acc01[m] = acc2[m] = bs[m] = f32x2{0.0f, 0.0f};

  struct Span {  // one lane's share of a 32-position span, as loaded: its input row's quad and the dy quad of output row m = r
    f32x4 xq, dyq;
    float ah;
    bool hok, last;
  };
  // the dy quads travel between the wave's eight row groups through a wave-private LDS slot (no barrier: a wave's LDS
  // instructions execute in order).  Every lane fetching all eight rows itself kept the texture path busy with 8x redundant
  // requests: 21 us per launch against 67 MB)
  __shared__ __attribute__((aligned(16))) float dysh[4][2][8][8][4];
  auto load = [&](int s, Span& sp, auto tailc) {
    constexpr bool TAIL = decltype(tailc)::value;  // spans that may be partly filled (N % 32 != 0) take the masked form
    const int n0 = s * 32 + 4 * q;
    const bool valid = !TAIL || n0 < N;
    const int n0c = valid ? n0 : N - 4;
    sp.hok = valid && ((q == 0 && n0 > 0) || (q == 7 && n0 + 4 < N));
    sp.last = TAIL && n0 + 4 >= N;
    const int hoffc = sp.hok ? (q == 0 ? n0 - 1 : n0 + 4) : n0c;
    sp.xq = *reinterpret_cast<const f32x4*>(xrow + n0c);
    sp.ah = xrow[hoffc];
    const int mc = r < M ? r : M - 1;  // (rows beyond M: duplicates, never stored)
    sp.dyq = *reinterpret_cast<const f32x4*>(dyb + (int64_t)mc * N + n0c);
    if (TAIL && !valid) sp.dyq = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  };
  auto compute = [&](const Span& sp, int slot) {
    *reinterpret_cast<f32x4*>(&dysh[wave][slot][r][q][0]) = sp.dyq;
    adp_wave_sync();
    f32x4 dq[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) dq[m] = *reinterpret_cast<const f32x4*>(&dysh[wave][slot][m][q][0]);
    f32x2 a01 = f32x2{sp.xq[0], sp.xq[1]}, a23 = f32x2{sp.xq[2], sp.xq[3]};
    float ah = sp.ah;
    if (PRO) {
      a01 = adp_silu2(a01 * pa + pb);
      a23 = adp_silu2(a23 * pa + pb);
      ah = adp_silu_fast(fmaf(ah, pa, pb));
    }
    ah = sp.hok ? ah : 0.0f;  // zero padding is applied after the activation
    float lft = adp_row_prev(a23[1]), rgt = adp_row_next(a01[0]);
    lft = q == 0 ? ah : lft;
    rgt = (q == 7 || sp.last) ? (q == 7 ? ah : 0.0f) : rgt;
    const f32x2 e01 = f32x2{lft, a01[0]}, a12 = f32x2{a01[1], a23[0]}, e45 = f32x2{a23[1], rgt};
    // dw[t] += dy[p] * win[p + t] over the window win = (lft, a0, a1, a2, a3, rgt): (t0, t1) against the pair (win[p], win[p+1])
#pragma unroll
    for (int m = 0; m < 8; ++m) acc01[m] = e01 * dq[m][0] + acc01[m];
#pragma unroll
    for (int m = 0; m < 8; ++m) acc2[m] = f32x2{dq[m][0], dq[m][1]} * a12 + acc2[m];
#pragma unroll
    for (int m = 0; m < 8; ++m) acc01[m] = a01 * dq[m][1] + acc01[m];
#pragma unroll
    for (int m = 0; m < 8; ++m) acc2[m] = f32x2{dq[m][2], dq[m][3]} * e45 + acc2[m];
#pragma unroll
    for (int m = 0; m < 8; ++m) acc01[m] = a12 * dq[m][2] + acc01[m];
#pragma unroll
    for (int m = 0; m < 8; ++m) bs[m] = bs[m] + (f32x2{dq[m][0], dq[m][1]} + f32x2{dq[m][2], dq[m][3]});
#pragma unroll
    for (int m = 0; m < 8; ++m) acc01[m] = a23 * dq[m][3] + acc01[m];
  };
  // TWO spans per trip, both requested before the first is multiplied: the kernel is a latency-bound stream (one span per trip:
  // 24 KB in flight per CU against the ~60 KB that 8 TB/s x 2 us asks for -- 21 us)
  const int full = N / 32, stride = bpb * 4;
  int s = blockIdx.x * 4 + wave;
  for (; s + stride < full; s += 2 * stride) {
    Span A, B;
    load(s, A, std::false_type{});
    load(s + stride, B, std::false_type{});
    compute(A, 0);
    compute(B, 1);
  }
  for (; s < spans_b; s += stride) {
    Span A;
    load(s, A, std::true_type{});
    compute(A, 0);
    adp_wave_sync();  // (the slot is rewritten by the next trip)
  }
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    adp_pin(acc01[m]);
    adp_pin(acc2[m]);
    adp_pin(bs[m]);
  }
  // ---- the row group's 8 lanes, then the workgroup's 4 waves (fixed order), one partial per workgroup
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    const float t0 = adp_oct_sum(acc01[m][0]), t1 = adp_oct_sum(acc01[m][1]), t2 = adp_oct_sum(acc2[m][0] + acc2[m][1]);
    const float bb = adp_oct_sum(bs[m][0] + bs[m][1]);
    if (q == 0) {
      red[wave][r][3 * m] = t0;
      red[wave][r][3 * m + 1] = t1;
      red[wave][r][3 * m + 2] = t2;
      red[wave][r][24 + m] = bb;
    }
  }
  __syncthreads();
  const int rr = tid >> 5, k = tid & 31;
  const float sum = (red[0][rr][k] + red[1][rr][k]) + (red[2][rr][k] + red[3][rr][k]);
  const int64_t cnt = (int64_t)M * R * 3, P = (int64_t)gridDim.x * gridDim.y, pidx = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
  if (k < 24) {
    const int m = k / 3, t = k - 3 * m;
    if (rr < R && m < M) d.ws[pidx * cnt + ((int64_t)m * R + rr) * 3 + t] = sum;
  } else if (rr == 0 && k - 24 < M && d.dbias) {
    d.ws[P * cnt + pidx * M + (k - 24)] = sum;
  }
}

static bool wd8_ok(const adp_wgrad_desc& d) {
  return d.KT == 3 && d.stride == 1 && d.up == 1 && d.R <= 8 && d.M <= 8 && d.N == d.Lin && d.N >= 4;
}
// workgroups per batch element: four waves x >= 4 spans of 32 positions each, at most WD8_MAXBLOCKS partials in all
static int wd8_bpb(const adp_wgrad_desc& d) {
  const int64_t spans = adp_cdiv(d.N, 32);
  int64_t bpb = adp_cdiv(spans, 16);
  const int64_t cap = WD8_MAXBLOCKS / d.B > 0 ? WD8_MAXBLOCKS / d.B : 1;
  return (int)(bpb < cap ? bpb : cap);
}
static int launch_wd8(const adp_wgrad_desc& d, void* stream) {
  const int bpb = wd8_bpb(d), spans = (int)adp_cdiv(d.N, 32);
  dim3 grid((unsigned)bpb, (unsigned)d.B);
  if (d.prologue == 1) ADP_LAUNCH((wgrad_direct8_kernel<true>), grid, dim3(256), stream, d, bpb, spans);
  else ADP_LAUNCH((wgrad_direct8_kernel<false>), grid, dim3(256), stream, d, bpb, spans);
  if (ADP_LAUNCH_OK() != ADP_OK) return ADP_ERR_LAUNCH;
  return adp_wgrad_reduce(d.ws, (int64_t)bpb * d.B, d.M * d.R * d.KT, d.M, d.dw, d.dbias, (int)(d.accumulate & 1), stream);
}

// ---- weight gradient of the x4 UpsampleItem conv of the narrow end (<= 32 -> <= 8 channels; conv_direct.hip: conv_up4_kernel)
// in its phase form.  With D0..D3 = dy[m][4j .. 4j+3] of one input position j:
//     dw[m][r][0] += D0 x[j-1] + (D1+D2+D3) x[j]     dw[m][r][1] += (D0+D1+D2+D3) x[j]     dw[m][r][2] += (D0+D1+D2) x[j] + D3 x[j+1]
// -- the four dy values are summed once per (m, j) and shared by all input rows: three packed / scalar FMAs per (m, r, j) where the
// gather form (wgrad_direct_kernel<3,1,4>) spends twelve behind LDS reads.  Register streaming reduction like wgrad_direct8:
// lane = (group g, position q): the 8 lanes of a group cover 8 consecutive input positions, group g owns the input rows g, g+8,
// ... and fetches the dy quad of output row m = g, which the groups exchange through a wave-private LDS slot.
template <int RPL>
__global__ __launch_bounds__(256) void wgrad_up4_kernel(adp_wgrad_desc d, int bpb, int spans_b) {
  constexpr int NE = 8 * RPL * 24 + 8;  // sums of one workgroup: [row][m][tap] + bias[m]
  __shared__ __attribute__((aligned(16))) float dysh[4][2][8][8][4];
  __shared__ float red[4][NE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = adp_uniform(tid >> 6);
  const int g = lane >> 3, q = lane & 7;
  const int M = (int)d.M, R = (int)d.R, L = (int)d.Lin, N = (int)d.N;
  const int b = blockIdx.y;
  const float* xrow[RPL];
#pragma unroll
  for (int k = 0; k < RPL; ++k) {
    const int r = g + 8 * k;
    xrow[k] = d.x + ((int64_t)b * R + (r < R ? r : R - 1)) * L;  // (rows beyond R: duplicates, never stored)
  }
  const float* dyrow = d.dy + ((int64_t)b * M + (g < M ? g : M - 1)) * N;
  f32x2 a02[RPL][8];  // (dw[..][0], dw[..][2])
  float a1[RPL][8], bsum = 0.0f;
#pragma unroll
  for (int k = 0; k < RPL; ++k)
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      a02[k][m] = f32x2{0.0f, 0.0f};
      a1[k][m] = 0.0f;
    }
  struct Span {
    float x[RPL], h[RPL];
    f32x4 dyq;
    bool hok, last;
  };
  auto load = [&](int s, Span& sp, auto tailc) {
    constexpr bool TAIL = decltype(tailc)::value;  // the last, partly filled span of a row (L % 8 != 0)
    const int j = s * 8 + q;
    const bool valid = !TAIL || j < L;
    const int jc = valid ? j : L - 1;
    sp.hok = valid && ((q == 0 && j > 0) || (q == 7 && j + 1 < L));
    sp.last = TAIL && j + 1 >= L;
    const int hc = sp.hok ? (q == 0 ? j - 1 : j + 1) : jc;
#pragma unroll
    for (int k = 0; k < RPL; ++k) {
      sp.x[k] = xrow[k][jc];
      sp.h[k] = xrow[k][hc];
    }
    sp.dyq = *reinterpret_cast<const f32x4*>(dyrow + 4 * (int64_t)jc);
    if (TAIL && !valid) sp.dyq = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  };
  auto compute = [&](const Span& sp, int slot) {
    *reinterpret_cast<f32x4*>(&dysh[wave][slot][g][q][0]) = sp.dyq;
    bsum += (sp.dyq[0] + sp.dyq[1]) + (sp.dyq[2] + sp.dyq[3]);
    adp_wave_sync();
    f32x2 p13[8], p03[8];  // (D1+D2+D3, D0+D1+D2) and (D0, D3)
    float ssum[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const f32x4 D = *reinterpret_cast<const f32x4*>(&dysh[wave][slot][m][q][0]);
      const float t = D[1] + D[2];
      p13[m] = f32x2{t + D[3], t + D[0]};
      p03[m] = f32x2{D[0], D[3]};
      ssum[m] = (t + D[3]) + D[0];
    }
#pragma unroll
    for (int k = 0; k < RPL; ++k) {
      const float x = sp.x[k], hv = sp.hok ? sp.h[k] : 0.0f;
      float xl = adp_row_prev(x), xr = adp_row_next(x);
      xl = q == 0 ? hv : xl;
      xr = (q == 7 || sp.last) ? (q == 7 ? hv : 0.0f) : xr;
      const f32x2 lr = f32x2{xl, xr};
#pragma unroll
      for (int m = 0; m < 8; ++m) a02[k][m] = p13[m] * x + a02[k][m];
#pragma unroll
      for (int m = 0; m < 8; ++m) a1[k][m] = fmaf(ssum[m], x, a1[k][m]);
#pragma unroll
      for (int m = 0; m < 8; ++m) a02[k][m] = p03[m] * lr + a02[k][m];
    }
  };
  const int full = L / 8, stride = bpb * 4;
  int s = blockIdx.x * 4 + wave;
  for (; s + stride < full; s += 2 * stride) {  // two spans requested per trip (see wgrad_direct8_kernel)
    Span A, B;
    load(s, A, std::false_type{});
    load(s + stride, B, std::false_type{});
    compute(A, 0);
    compute(B, 1);
  }
  for (; s < spans_b; s += stride) {
    Span A;
    load(s, A, std::true_type{});
    compute(A, 0);
    adp_wave_sync();
  }
#pragma unroll
  for (int k = 0; k < RPL; ++k)
#pragma unroll
    for (int m = 0; m < 8; ++m) adp_pin(a02[k][m]);
  // ---- the group's 8 lanes, then the workgroup's 4 waves (fixed order), one partial per workgroup
#pragma unroll
  for (int k = 0; k < RPL; ++k)
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const float t0 = adp_oct_sum(a02[k][m][0]), t1 = adp_oct_sum(a1[k][m]), t2 = adp_oct_sum(a02[k][m][1]);
      if (q == 0) {
        float* e = &red[wave][((g + 8 * k) * 8 + m) * 3];
        e[0] = t0;
        e[1] = t1;
        e[2] = t2;
      }
    }
  const float bb = adp_oct_sum(bsum);
  if (q == 0) red[wave][8 * RPL * 24 + g] = bb;
  __syncthreads();
  const int64_t cnt = (int64_t)M * R * 3, P = (int64_t)gridDim.x * gridDim.y, pidx = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
  for (int e = tid; e < NE; e += 256) {
    const float sum = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
    if (e < 8 * RPL * 24) {
      const int r = e / 24, k = e - r * 24, m = k / 3, t = k - 3 * m;
      if (r < R && m < M) d.ws[pidx * cnt + ((int64_t)m * R + r) * 3 + t] = sum;
    } else if (e - 8 * RPL * 24 < M && d.dbias) {
      d.ws[P * cnt + pidx * M + (e - 8 * RPL * 24)] = sum;
    }
  }
}

static bool wdu4_ok(const adp_wgrad_desc& d) {
  return d.KT == 3 && d.stride == 1 && d.up == 4 && d.prologue == 0 && d.R <= 32 && d.M <= 8 && d.R1 == d.R &&
         d.N == 4 * d.Lin && d.B <= 65535;
}
static int wdu4_bpb(const adp_wgrad_desc& d) {
  const int64_t spans = adp_cdiv(d.Lin, 8);
  int64_t bpb = adp_cdiv(spans, 16);
  const int64_t cap = WD8_MAXBLOCKS / d.B > 0 ? WD8_MAXBLOCKS / d.B : 1;
  return (int)(bpb < cap ? bpb : cap);
}
static int launch_wdu4(const adp_wgrad_desc& d, void* stream) {
  const int bpb = wdu4_bpb(d), spans = (int)adp_cdiv(d.Lin, 8);
  dim3 grid((unsigned)bpb, (unsigned)d.B);
  if (d.R <= 8) ADP_LAUNCH((wgrad_up4_kernel<1>), grid, dim3(256), stream, d, bpb, spans);
  else if (d.R <= 16) ADP_LAUNCH((wgrad_up4_kernel<2>), grid, dim3(256), stream, d, bpb, spans);
  else ADP_LAUNCH((wgrad_up4_kernel<4>), grid, dim3(256), stream, d, bpb, spans);
  if (ADP_LAUNCH_OK() != ADP_OK) return ADP_ERR_LAUNCH;
  return adp_wgrad_reduce(d.ws, (int64_t)bpb * d.B, d.M * d.R * d.KT, d.M, d.dw, d.dbias, (int)(d.accumulate & 1), stream);
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
  return adp_wgrad_reduce(d.ws, p.blocks, d.M * d.R * d.KT, d.M, d.dw, d.dbias, (int)(d.accumulate & 1), stream);
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
  if (wd8_ok(d) && d.B <= 65535) return (int64_t)wd8_bpb(d) * d.B * (d.M * d.R * d.KT + d.M);
  if (wdu4_ok(d)) return (int64_t)wdu4_bpb(d) * d.B * (d.M * d.R * d.KT + d.M);
  return (int64_t)wd_plan(d).blocks * (d.M * d.R * d.KT + d.M);
}

int adp_wgrad_direct(const adp_wgrad_desc& d, void* stream) {
  if (wd8_ok(d) && d.B <= 65535) return launch_wd8(d, stream);
  if (wdu4_ok(d)) return launch_wdu4(d, stream);
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
