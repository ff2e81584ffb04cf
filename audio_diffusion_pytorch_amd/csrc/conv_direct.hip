// Direct (VALU) Conv1d for the narrow ends of the U-Net: depth-0/1 layers with 2-8 channels on one side
// (ResnetItem at channels = 8, the first DownsampleItem, the last UpsampleItem + SkipModulate merge and their data
// gradients; /root/reference/audio_diffusion_pytorch/components.py:84-99).
//
// These layers are pure streaming: 8 x 8 x 3 weights against 2**18 positions.  A 32x32 matrix-core tile would be
// 75-94 % padding there, so the contraction runs on the vector ALU instead and the kernel is organised around HBM:
//   * one thread owns FOUR consecutive output positions (16-byte loads / stores along L, the contiguous axis) and
//     MB = 8 output channels; a workgroup covers 1024 positions, grid.y tiles the output channels;
//   * per input channel a thread loads its window once (one 16-byte load + the halo scalars, which hit the lines
//     its neighbours fetch), applies the GroupNorm+SiLU prologue in registers, and reuses it for all MB outputs;
//     the nearest-upsample gather (source index = u / UP, integer, exact) and the kernel = stride windows of
//     DownsampleItem are just different window loaders;
//   * the MB x R x KT weights of the workgroup live in LDS in consumption order (the transposed view of the data
//     gradient is resolved while staging them) and are read with wave-uniform (broadcast) 16-byte LDS loads;
//   * epilogue = adp_conv1d's: bias, pre-merge copy, e_scale[b,m], residual, plain or pooled (sum of sp) store.
#include "adp_rt.h"
#include "adp.h"
#include "conv_internal.h"

namespace {

constexpr int DC_MB = 8;       // output channels per thread
constexpr int DC_NCH = 8;      // input-channel chunks whose weights fit the LDS table

template <int KT, int S>
struct DcCfg {
  static constexpr int RCH = (S == 1) ? 8 : (S == 2 ? 4 : 2);  // input channels per register chunk
  static constexpr int W = 3 * S + KT;                          // window: positions touched by 4 outputs
  static constexpr int WBLK = ((RCH * KT + 3) / 4) * 4;         // weights of one (m, chunk), padded to 16 bytes
};

// window of virtual positions ustart .. ustart+W-1 of one input row (upsampled by UP, zero outside [0, Lv)),
// n0s = first position whose 4-aligned quad can be fetched with one wide load (S == 1: the four outputs' centre)
template <int KT, int S, int UP>
__device__ __forceinline__ void dc_load_window(const float* row, int ustart, int Lv, float* win) {
  constexpr int W = DcCfg<KT, S>::W;
  if (UP == 1 && S == 1) {
    // ustart = n0 - PAD with n0 % 4 == 0 and PAD = (KT-1)/2: the aligned quad sits PAD elements into the window
    constexpr int PAD = (KT - 1) / 2;
    const int n0 = ustart + PAD;
    if (n0 + 3 < Lv) {
      const f32x4 q = *reinterpret_cast<const f32x4*>(row + n0);
#pragma unroll
      for (int j = 0; j < 4; ++j) win[PAD + j] = q[j];
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) win[PAD + j] = (n0 + j < Lv) ? row[n0 + j] : 0.0f;
    }
#pragma unroll
    for (int i = 0; i < PAD; ++i) {
      const int ul = ustart + i, ur = n0 + 4 + i;
      win[i] = (ul >= 0) ? row[ul] : 0.0f;
      win[PAD + 4 + i] = (ur < Lv) ? row[ur] : 0.0f;
    }
  } else if (UP == 1) {
    // kernel = stride: 4*S contiguous, 16-byte aligned inputs
#pragma unroll
    for (int q4 = 0; q4 < W / 4; ++q4) {
      const int u = ustart + 4 * q4;
      f32x4 q;
      if (u >= 0 && u + 3 < Lv) {
        q = *reinterpret_cast<const f32x4*>(row + u);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) q[j] = (u + j >= 0 && u + j < Lv) ? row[u + j] : 0.0f;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) win[4 * q4 + j] = q[j];
    }
  } else {
#pragma unroll
    for (int i = 0; i < W; ++i) {
      const int u = ustart + i;
      win[i] = (u >= 0 && u < Lv) ? row[u / UP] : 0.0f;  // consecutive i share a source element: L1 hits
    }
  }
}

template <int KT, int S, int UP>
__global__ __launch_bounds__(256) void conv_direct_kernel(adp_conv_desc d) {
  using C = DcCfg<KT, S>;
  constexpr int MB = DC_MB, RCH = C::RCH, W = C::W, WBLK = C::WBLK;
  __shared__ __attribute__((aligned(16))) float Ws[MB * DC_NCH * WBLK];

  const int tid = threadIdx.x;
  const int M = (int)d.M, R = (int)d.R, R1 = (int)d.R1, L = (int)d.Lin, N = (int)d.N;
  const int pad = (int)d.pad, Lv = L * UP;
  const int b = blockIdx.z, m0 = blockIdx.y * MB;
  const int n0 = (blockIdx.x * 256 + tid) * 4;
  const int nch = (R + RCH - 1) / RCH;

  // ---- weights of this workgroup's MB output channels, in consumption order [m][chunk][r_local][t]
  for (int e = tid; e < MB * nch * WBLK; e += 256) {
    const int m = e / (nch * WBLK), rem = e - m * (nch * WBLK);
    const int ch = rem / WBLK, k = rem - ch * WBLK;
    const int rl = k / KT, t = k - rl * KT, r = ch * RCH + rl;
    float v = 0.0f;
    if (k < RCH * KT && r < R && m0 + m < M)
      v = d.transposed ? d.w[((int64_t)r * M + m0 + m) * KT + (KT - 1 - t)] : d.w[((int64_t)(m0 + m) * R + r) * KT + t];
    Ws[e] = v;
  }
  __syncthreads();
  if (n0 >= N) return;

  float acc[MB][4];
#pragma unroll
  for (int m = 0; m < MB; ++m)
#pragma unroll
    for (int p = 0; p < 4; ++p) acc[m][p] = 0.0f;

  const int ustart = n0 * S - pad;
  const int cpg = (d.prologue == 1) ? R / (int)d.groups : 1;
  for (int ch = 0; ch < nch; ++ch) {
    float win[RCH][W];
#pragma unroll
    for (int rl = 0; rl < RCH; ++rl) {
      const int r = ch * RCH + rl;
      if (r < R) {  // block-uniform
        const float* row = (r < R1) ? d.x + ((int64_t)b * R1 + r) * L : d.x2 + ((int64_t)b * (R - R1) + (r - R1)) * L;
        dc_load_window<KT, S, UP>(row, ustart, Lv, win[rl]);
        if (d.prologue == 1) {
          const int g = r / cpg;
          const float mean = d.pro_stats[((int64_t)b * d.groups + g) * 2];
          const float pa = (d.pro_gamma ? d.pro_gamma[r] : 1.0f) * d.pro_stats[((int64_t)b * d.groups + g) * 2 + 1];
          const float pb = (d.pro_beta ? d.pro_beta[r] : 0.0f) - mean * pa;
#pragma unroll
          for (int i = 0; i < W; ++i) {
            const int u = ustart + i;
            const float a = adp_silu_fast(fmaf(win[rl][i], pa, pb));
            win[rl][i] = (u >= 0 && u < Lv) ? a : 0.0f;  // zero padding is applied after the activation
          }
        }
      } else {
#pragma unroll
        for (int i = 0; i < W; ++i) win[rl][i] = 0.0f;
      }
    }
#pragma unroll
    for (int m = 0; m < MB; ++m) {
      const float* wp = Ws + (m * nch + ch) * WBLK;
      float wv[WBLK];
#pragma unroll
      for (int q = 0; q < WBLK / 4; ++q) {
        const f32x4 t4 = *reinterpret_cast<const f32x4*>(wp + 4 * q);
#pragma unroll
        for (int j = 0; j < 4; ++j) wv[4 * q + j] = t4[j];
      }
#pragma unroll
      for (int rl = 0; rl < RCH; ++rl)
#pragma unroll
        for (int t = 0; t < KT; ++t)
#pragma unroll
          for (int p = 0; p < 4; ++p) acc[m][p] = fmaf(wv[rl * KT + t], win[rl][p * S + t], acc[m][p]);
    }
  }

  // ---- epilogue
  const int sp = (int)d.sp;
  const int64_t ebs = d.e_bstride ? d.e_bstride : M;
  const bool full = (n0 + 3 < N);
#pragma unroll
  for (int m = 0; m < MB; ++m) {
    const int mm = m0 + m;
    if (mm >= M) break;
    float v[4];
    const float bias = d.bias ? d.bias[mm] : 0.0f;
#pragma unroll
    for (int p = 0; p < 4; ++p) v[p] = acc[m][p] + bias;
    const int64_t o = ((int64_t)b * M + mm) * N + n0;
    if (d.out_pre) {
      if (full) {
        f32x4 q;
#pragma unroll
        for (int p = 0; p < 4; ++p) q[p] = v[p];
        *reinterpret_cast<f32x4*>(d.out_pre + o) = q;
      } else {
        for (int p = 0; p < 4 && n0 + p < N; ++p) d.out_pre[o + p] = v[p];
      }
    }
    if (d.e_scale) {
      const float es = d.e_scale[b * ebs + mm];
#pragma unroll
      for (int p = 0; p < 4; ++p) v[p] *= es;
    }
    if (d.store == 0) {
      if (full) {
        f32x4 q;
#pragma unroll
        for (int p = 0; p < 4; ++p) q[p] = v[p];
        if (d.res) {
          const f32x4 rr = *reinterpret_cast<const f32x4*>(d.res + o);
#pragma unroll
          for (int p = 0; p < 4; ++p) q[p] += rr[p];
        }
        *reinterpret_cast<f32x4*>(d.out + o) = q;
      } else {
        for (int p = 0; p < 4 && n0 + p < N; ++p) d.out[o + p] = v[p] + (d.res ? d.res[o + p] : 0.0f);
      }
    } else {
      // pooled store (gradient of the nearest upsample): N % sp == 0 and sp in {2, 4}, so a thread's four
      // positions cover whole groups
      const int64_t op = ((int64_t)b * M + mm) * (N / sp) + n0 / sp;
      if (sp == 4) {
        float s = (v[0] + v[1]) + (v[2] + v[3]);
        if (d.res) s += d.res[op];
        d.out[op] = s;
      } else {
        float s0 = v[0] + v[1], s1 = v[2] + v[3];
        if (d.res) {
          s0 += d.res[op];
          s1 += d.res[op + 1];
        }
        d.out[op] = s0;
        d.out[op + 1] = s1;
      }
    }
  }
}

// ---- kernel 3, stride 1, at most 8 input channels (the depth-0 ConvBlocks and their data gradients: 8 x 8 x 3 weights against
// 2**18 positions).  Same thread <-> data mapping as conv_direct_kernel (four consecutive positions, 8 output channels), on a
// diet (alu_probe: the generic kernel spent ~1500 VALU issue slots per thread on a 67 MB launch):
//   * the two halo elements of a thread's window come from the neighbouring LANES (DPP wave shifts of the ACTIVATED values), so a
//     row costs one 16-byte load + one masked 4-byte load for the wave's two outer lanes, and five activations instead of six;
//   * the contraction runs on packed FMAs over position pairs: the window is kept as the five overlapping pairs
//     (L,a0) (a0,a1) (a1,a2) (a2,a3) (a3,R); SiLU in packed form around its two transcendentals;
//   * every global load of the thread is issued before the first activation; input rows are streamed against acc[8][2 pairs].
template <bool PRO, int RR>
__global__ __launch_bounds__(256) void conv_direct8_kernel(adp_conv_desc d) {
  __shared__ __attribute__((aligned(16))) float Ws[RR * 24];  // [r][t][m], zero beyond R / M
  const int tid = threadIdx.x, lane = tid & 63;
  const int M = (int)d.M, R = (int)d.R, R1 = (int)d.R1, N = (int)d.N;
  const int b = blockIdx.z, m0 = blockIdx.y * 8;
  const int n0 = (blockIdx.x * 256 + tid) * 4;
  if (tid < RR * 24) {
    const int r = tid / 24, k = tid - r * 24, t = k >> 3, m = k & 7;
    float v = 0.0f;
    if (r < R && m0 + m < M)
      v = d.transposed ? d.w[((int64_t)r * M + m0 + m) * 3 + (2 - t)] : d.w[((int64_t)(m0 + m) * R + r) * 3 + t];
    Ws[tid] = v;
  }
  const bool valid = n0 < N;
  // the wave's outer lanes fetch the element beyond the wave's 256 positions (lane 0: n0 - 1, lane 63: n0 + 4)
  const int hoff = lane == 0 ? n0 - 1 : n0 + 4;
  const bool hok = valid && ((lane == 0 && n0 > 0) || (lane == 63 && n0 + 4 < N));
  // every load is unconditional (clamped addresses: rows beyond R re-read row R - 1 against zero weights, threads beyond N
  // re-read the last quad and store nothing) -- conditional loads cost a register copy of the whole window set per branch
  const int n0c = valid ? n0 : N - 4, hoffc = hok ? hoff : n0c;
  f32x4 xq[RR];
  float hv[RR];
#pragma unroll
  for (int r = 0; r < RR; ++r) {
    const int rc = r < R ? r : R - 1;
    const float* row = (rc < R1) ? d.x + ((int64_t)b * R1 + rc) * N : d.x2 + ((int64_t)b * (R - R1) + (rc - R1)) * N;
    xq[r] = *reinterpret_cast<const f32x4*>(row + n0c);
    hv[r] = row[hoffc];
  }
  float pa[RR], pb[RR];
  if (PRO) {
    const int cpg = R / (int)d.groups;
#pragma unroll
    for (int r = 0; r < RR; ++r) {
      const int rc = r < R ? r : R - 1, g = rc / cpg;
      const float mean = d.pro_stats[((int64_t)b * d.groups + g) * 2];
      pa[r] = (d.pro_gamma ? d.pro_gamma[rc] : 1.0f) * d.pro_stats[((int64_t)b * d.groups + g) * 2 + 1];
      pb[r] = (d.pro_beta ? d.pro_beta[rc] : 0.0f) - mean * pa[r];
    }
  }
  __syncthreads();
  f32x2 acc[8][2];
#pragma unroll
  for (int m = 0; m < 8; ++m) acc[m][0] = acc[m][1] = f32x2{0.0f, 0.0f};
#pragma unroll
  for (int r = 0; r < RR; ++r) {  // (rows beyond R carry zero weights)
    f32x2 a01 = f32x2{xq[r][0], xq[r][1]}, a23 = f32x2{xq[r][2], xq[r][3]};
    float ah = hv[r];
    if (PRO) {
      a01 = adp_silu2(a01 * pa[r] + pb[r]);
      a23 = adp_silu2(a23 * pa[r] + pb[r]);
      ah = adp_silu_fast(fmaf(ah, pa[r], pb[r]));
    }
    ah = hok ? ah : 0.0f;  // zero padding (applied after the activation); the row's first / last quad has no neighbour
    float lft = adp_lane_prev(ah, a23[1]), rgt = adp_lane_next(ah, a01[0]);
    lft = n0 > 0 ? lft : 0.0f;
    rgt = n0 + 4 < N ? rgt : 0.0f;
    // the window (lft, a0, a1, a2, a3, rgt) as its five overlapping pairs: e01 a01 a12 a23 e45
    const f32x2 e01 = f32x2{lft, a01[0]}, a12 = f32x2{a01[1], a23[0]}, e45 = f32x2{a23[1], rgt};
    const float* wp = Ws + r * 24;
    float wv[24];
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      const f32x4 t4 = *reinterpret_cast<const f32x4*>(wp + 4 * q);
#pragma unroll
      for (int j = 0; j < 4; ++j) wv[4 * q + j] = t4[j];
    }
    // outputs (0,1) see the pairs e01 / a01 / a12 under taps 0 / 1 / 2, outputs (2,3) a12 / a23 / e45; eight independent
    // accumulators per line, so no packed FMA waits for its predecessor
#pragma unroll
    for (int m = 0; m < 8; ++m) acc[m][0] = e01 * wv[m] + acc[m][0];
#pragma unroll
    for (int m = 0; m < 8; ++m) acc[m][1] = a12 * wv[m] + acc[m][1];
#pragma unroll
    for (int m = 0; m < 8; ++m) acc[m][0] = a01 * wv[8 + m] + acc[m][0];
#pragma unroll
    for (int m = 0; m < 8; ++m) acc[m][1] = a23 * wv[8 + m] + acc[m][1];
#pragma unroll
    for (int m = 0; m < 8; ++m) acc[m][0] = a12 * wv[16 + m] + acc[m][0];
#pragma unroll
    for (int m = 0; m < 8; ++m) acc[m][1] = e45 * wv[16 + m] + acc[m][1];
  }
  // (without this the optimiser sinks the whole contraction into the per-channel blocks of the epilogue below -- all 192
  //  weights live at once, 256 registers + spills)
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    adp_pin(acc[m][0]);
    adp_pin(acc[m][1]);
  }
  if (!valid) return;
  const int64_t ebs = d.e_bstride ? d.e_bstride : M;
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    const int mm = m0 + m;
    if (mm >= M) break;
    const float bias = d.bias ? d.bias[mm] : 0.0f;
    f32x4 v = f32x4{acc[m][0][0] + bias, acc[m][0][1] + bias, acc[m][1][0] + bias, acc[m][1][1] + bias};
    const int64_t o = ((int64_t)b * M + mm) * N + n0;
    if (d.out_pre) *reinterpret_cast<f32x4*>(d.out_pre + o) = v;
    if (d.e_scale) v = v * d.e_scale[b * ebs + mm];
    if (d.res) v = v + *reinterpret_cast<const f32x4*>(d.res + o);
    *reinterpret_cast<f32x4*>(d.out + o) = v;
  }
}

template <bool PRO, int RR>
int launch_dc8(const adp_conv_desc& d, void* stream) {
  dim3 grid((unsigned)adp_cdiv(d.N, 1024), (unsigned)adp_cdiv(d.M, DC_MB), (unsigned)d.B);
  ADP_LAUNCH((conv_direct8_kernel<PRO, RR>), grid, dim3(256), stream, d);
  return ADP_LAUNCH_OK();
}

// ---- nearest x4 upsample + kernel 3 (the UpsampleItem that ends the U-Net: 32 -> 8 channels at 2**18 positions) as FOUR PHASE
// convolutions on the low-resolution input.  Output 4j + s reads the upsampled positions 4j + s - 1 .. 4j + s + 1, i.e. the inputs
//     s = 0: x[j-1], x[j], x[j]      s = 1, 2: x[j], x[j], x[j]      s = 3: x[j], x[j], x[j+1]
// so with the taps summed once per workgroup (A = w0, B = w1 + w2, S = w0 + w1 + w2, C = w0 + w1, D = w2)
//     y[4j] = A x[j-1] + B x[j]      y[4j+1] = y[4j+2] = S x[j]      y[4j+3] = C x[j] + D x[j+1]
// five multiplies per input channel for four outputs where the gather form (conv_direct_kernel<3,1,4>) spends twelve.
// A thread owns two input positions (eight outputs, 8 output channels): packed FMAs over the position pair, halo through DPP lane
// shifts, every load up front, branch-free -- the structure of conv_direct8_kernel.
constexpr int UP4_RMAX = 32;

template <int RR>
__global__ __launch_bounds__(256) void conv_up4_kernel(adp_conv_desc d) {
  __shared__ __attribute__((aligned(16))) float Ws[RR * 40];  // [r][A B S C D][m], zero beyond R / M
  const int tid = threadIdx.x, lane = tid & 63;
  const int M = (int)d.M, R = (int)d.R, L = (int)d.Lin, N = (int)d.N;
  const int b = blockIdx.z, m0 = blockIdx.y * 8;
  const int j0 = (blockIdx.x * 256 + tid) * 2;
  for (int e = tid; e < RR * 40; e += 256) {
    const int r = e / 40, k = e - r * 40, tap = k >> 3, m = k & 7;
    float v = 0.0f;
    if (r < R && m0 + m < M) {
      const float* wp = d.w + ((int64_t)(m0 + m) * R + r) * 3;
      const float w0 = wp[0], w1 = wp[1], w2 = wp[2];
      v = tap == 0 ? w0 : tap == 1 ? w1 + w2 : tap == 2 ? (w0 + w1) + w2 : tap == 3 ? w0 + w1 : w2;
    }
    Ws[e] = v;
  }
  const bool valid = j0 < L;
  const int j0c = valid ? j0 : L - 2;
  const bool hok = valid && ((lane == 0 && j0 > 0) || (lane == 63 && j0 + 2 < L));
  const int hoffc = hok ? (lane == 0 ? j0 - 1 : j0 + 2) : j0c;
  f32x2 xv[RR];
  float hv[RR];
#pragma unroll
  for (int r = 0; r < RR; ++r) {
    const int rc = r < R ? r : R - 1;  // (rows beyond R: re-read against zero weights)
    const float* row = d.x + ((int64_t)b * R + rc) * L;
    xv[r] = *reinterpret_cast<const f32x2*>(row + j0c);
    hv[r] = row[hoffc];
  }
  __syncthreads();
  f32x2 y0[8], y1[8], y3[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) y0[m] = y1[m] = y3[m] = f32x2{0.0f, 0.0f};
#pragma unroll
  for (int r = 0; r < RR; ++r) {
    const float ah = hok ? hv[r] : 0.0f;
    float lft = adp_lane_prev(ah, xv[r][1]), rgt = adp_lane_next(ah, xv[r][0]);
    lft = j0 > 0 ? lft : 0.0f;
    rgt = j0 + 2 < L ? rgt : 0.0f;
    const f32x2 pa = f32x2{lft, xv[r][0]}, pb = xv[r], pc = f32x2{xv[r][1], rgt};
    const float* wp = Ws + r * 40;
    float wv[40];
#pragma unroll
    for (int q = 0; q < 10; ++q) {
      const f32x4 t4 = *reinterpret_cast<const f32x4*>(wp + 4 * q);
#pragma unroll
      for (int i = 0; i < 4; ++i) wv[4 * q + i] = t4[i];
    }
#pragma unroll
    for (int m = 0; m < 8; ++m) y0[m] = pa * wv[m] + y0[m];
#pragma unroll
    for (int m = 0; m < 8; ++m) y1[m] = pb * wv[16 + m] + y1[m];
#pragma unroll
    for (int m = 0; m < 8; ++m) y3[m] = pb * wv[24 + m] + y3[m];
#pragma unroll
    for (int m = 0; m < 8; ++m) y0[m] = pb * wv[8 + m] + y0[m];
#pragma unroll
    for (int m = 0; m < 8; ++m) y3[m] = pc * wv[32 + m] + y3[m];
  }
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    adp_pin(y0[m]);
    adp_pin(y1[m]);
    adp_pin(y3[m]);
  }
  if (!valid) return;
  const int64_t ebs = d.e_bstride ? d.e_bstride : M;
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    const int mm = m0 + m;
    if (mm >= M) break;
    const float bias = d.bias ? d.bias[mm] : 0.0f;
    f32x4 v[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) v[h] = f32x4{y0[m][h] + bias, y1[m][h] + bias, y1[m][h] + bias, y3[m][h] + bias};
    const int64_t o = ((int64_t)b * M + mm) * N + 4 * (int64_t)j0;
    const float es = d.e_scale ? d.e_scale[b * ebs + mm] : 1.0f;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (d.out_pre) *reinterpret_cast<f32x4*>(d.out_pre + o + 4 * h) = v[h];
      f32x4 u = v[h] * es;
      if (d.res) u = u + *reinterpret_cast<const f32x4*>(d.res + o + 4 * h);
      *reinterpret_cast<f32x4*>(d.out + o + 4 * h) = u;
    }
  }
}

template <int RR>
int launch_up4(const adp_conv_desc& d, void* stream) {
  dim3 grid((unsigned)adp_cdiv(d.Lin, 512), (unsigned)adp_cdiv(d.M, DC_MB), (unsigned)d.B);
  ADP_LAUNCH((conv_up4_kernel<RR>), grid, dim3(256), stream, d);
  return ADP_LAUNCH_OK();
}

// ---- data gradient of the same layer (transposed conv + pooled store over the 4 replicas = gradient of the nearest upsample) in
// the phase form: with dys[j] = dy[m][4j + s] and the summed taps A B S C D of conv_up4_kernel (weights w[m][r][t] of the forward),
//     dx[r][j] = sum_m  A dy0[j+1] + B dy0[j] + S (dy1[j] + dy2[j]) + C dy3[j] + D dy3[j-1]
// five multiplies per (m, r, j) on the LOW-resolution grid, no pooling pass.  A thread owns two positions j (two dy quads per
// input row, halo = the neighbour lanes' first / last element) and RR output rows.
template <int RR>
__global__ __launch_bounds__(256) void conv_up4_dgrad_kernel(adp_conv_desc d) {
  __shared__ __attribute__((aligned(16))) float Ws[8 * 5 * RR];  // [m][A B S C D][r], zero beyond the channel counts
  const int tid = threadIdx.x, lane = tid & 63;
  const int Mi = (int)d.R, Ro = (int)d.M, N = (int)d.N, L = N / 4;  // Mi dy rows in, Ro dx rows out
  const int b = blockIdx.z;
  const int j0 = (blockIdx.x * 256 + tid) * 2;
  for (int e = tid; e < 8 * 5 * RR; e += 256) {
    const int m = e / (5 * RR), k = e - m * 5 * RR, tap = k / RR, r = k - tap * RR;
    float v = 0.0f;
    if (m < Mi && r < Ro) {
      const float* wp = d.w + ((int64_t)m * Ro + r) * 3;  // the forward's w[m][r][0..2]
      const float w0 = wp[0], w1 = wp[1], w2 = wp[2];
      v = tap == 0 ? w0 : tap == 1 ? w1 + w2 : tap == 2 ? (w0 + w1) + w2 : tap == 3 ? w0 + w1 : w2;
    }
    Ws[e] = v;
  }
  const bool valid = j0 < L;
  const int j0c = valid ? j0 : L - 2;
  const bool hok = valid && ((lane == 0 && j0 > 0) || (lane == 63 && j0 + 2 < L));
  const int hoffc = hok ? (lane == 0 ? 4 * j0 - 1 : 4 * (j0 + 2)) : 4 * j0c;
  f32x4 q0[8], q1[8];
  float hv[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    const float* row = d.x + ((int64_t)b * Mi + (m < Mi ? m : Mi - 1)) * N;  // (rows beyond Mi: re-read against zero weights)
    q0[m] = *reinterpret_cast<const f32x4*>(row + 4 * (int64_t)j0c);
    q1[m] = *reinterpret_cast<const f32x4*>(row + 4 * (int64_t)j0c + 4);
    hv[m] = row[hoffc];
  }
  __syncthreads();
  f32x2 acc[RR];
#pragma unroll
  for (int r = 0; r < RR; ++r) acc[r] = f32x2{0.0f, 0.0f};
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    const float ah = hok ? hv[m] : 0.0f;
    float prv = adp_lane_prev(ah, q1[m][3]), nxt = adp_lane_next(ah, q0[m][0]);  // dy3[j0-1], dy0[j0+2]
    prv = j0 > 0 ? prv : 0.0f;
    nxt = j0 + 2 < L ? nxt : 0.0f;
    const f32x2 d0 = f32x2{q0[m][0], q1[m][0]}, d0n = f32x2{q1[m][0], nxt};
    const f32x2 d12 = f32x2{q0[m][1] + q0[m][2], q1[m][1] + q1[m][2]};
    const f32x2 d3 = f32x2{q0[m][3], q1[m][3]}, d3p = f32x2{prv, q0[m][3]};
    const float* wp = Ws + m * 5 * RR;
#pragma unroll
    for (int c = 0; c < RR / 4; ++c) {  // four output rows at a time (the weights of a tap are contiguous over r)
      f32x4 wa = *reinterpret_cast<const f32x4*>(wp + 4 * c), wb = *reinterpret_cast<const f32x4*>(wp + RR + 4 * c);
      f32x4 wS = *reinterpret_cast<const f32x4*>(wp + 2 * RR + 4 * c), wc = *reinterpret_cast<const f32x4*>(wp + 3 * RR + 4 * c);
      f32x4 wd = *reinterpret_cast<const f32x4*>(wp + 4 * RR + 4 * c);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[4 * c + i] = d0n * wa[i] + acc[4 * c + i];
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[4 * c + i] = d0 * wb[i] + acc[4 * c + i];
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[4 * c + i] = d12 * wS[i] + acc[4 * c + i];
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[4 * c + i] = d3 * wc[i] + acc[4 * c + i];
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[4 * c + i] = d3p * wd[i] + acc[4 * c + i];
    }
  }
#pragma unroll
  for (int r = 0; r < RR; ++r) adp_pin(acc[r]);
  if (!valid) return;
#pragma unroll
  for (int r = 0; r < RR; ++r) {
    if (r >= Ro) break;
    const int64_t o = ((int64_t)b * Ro + r) * L + j0;
    f32x2 v = acc[r];
    if (d.res) v = v + *reinterpret_cast<const f32x2*>(d.res + o);
    *reinterpret_cast<f32x2*>(d.out + o) = v;
  }
}

template <int RR>
int launch_up4_dgrad(const adp_conv_desc& d, void* stream) {
  dim3 grid((unsigned)adp_cdiv(d.N / 4, 512), 1, (unsigned)d.B);
  ADP_LAUNCH((conv_up4_dgrad_kernel<RR>), grid, dim3(256), stream, d);
  return ADP_LAUNCH_OK();
}

template <int KT, int S, int UP>
int launch_dc(const adp_conv_desc& d, void* stream) {
  dim3 grid((unsigned)adp_cdiv(d.N, 1024), (unsigned)adp_cdiv(d.M, DC_MB), (unsigned)d.B);
  ADP_LAUNCH((conv_direct_kernel<KT, S, UP>), grid, dim3(256), stream, d);
  return ADP_LAUNCH_OK();
}

}  // namespace

bool adp_conv_direct_eligible(const adp_conv_desc& d) {
  if (d.dil != 1 || (d.prologue != 0 && d.prologue != 1) || (d.store != 0 && d.store != 2)) return false;
  const bool s1 = d.stride == 1 && (d.KT == 1 || d.KT == 3) && (d.up == 1 || d.up == 2 || d.up == 4);
  const bool down = (d.stride == 2 || d.stride == 4) && d.KT == d.stride && d.up == 1 && d.pad == 0;
  if (!s1 && !down) return false;
  if (s1 && d.pad != (d.KT - 1) / 2) return false;
  // narrow layers only: the contraction is on the VALU
  const int64_t rch = d.stride == 1 ? 8 : (d.stride == 2 ? 4 : 2);
  if (d.R > rch * DC_NCH) return false;
  if (d.R > 8 && d.M > 8) return false;
  if (d.N % 4 != 0 || (d.Lin * d.up) % 4 != 0) return false;
  if (d.store == 2 && ((d.sp != 2 && d.sp != 4) || d.N % d.sp != 0 || d.out_pre)) return false;
  if ((reinterpret_cast<uintptr_t>(d.x) | reinterpret_cast<uintptr_t>(d.out) | reinterpret_cast<uintptr_t>(d.res) |
       reinterpret_cast<uintptr_t>(d.out_pre) | reinterpret_cast<uintptr_t>(d.x2)) & 15)
    return false;
  if (d.B > 65535 || adp_cdiv(d.M, DC_MB) > 65535 || d.Lin * d.up >= (int64_t)1 << 30) return false;
  return true;
}

int adp_conv_direct(const adp_conv_desc& d, void* stream) {
  if (d.stride == 2) return launch_dc<2, 2, 1>(d, stream);
  if (d.stride == 4) return launch_dc<4, 4, 1>(d, stream);
  if (d.KT == 3) {
    if (d.up == 1 && d.transposed && d.store == 2 && d.sp == 4 && d.R <= 8 && d.M <= UP4_RMAX && d.prologue == 0 && !d.bias &&
        !d.e_scale && d.R1 == d.R && d.N == d.Lin && d.N % 8 == 0 &&
        ((reinterpret_cast<uintptr_t>(d.out) | reinterpret_cast<uintptr_t>(d.res)) & 7) == 0) {
      if (d.M <= 8) return launch_up4_dgrad<8>(d, stream);
      if (d.M <= 16) return launch_up4_dgrad<16>(d, stream);
      return launch_up4_dgrad<32>(d, stream);
    }
    if (d.up == 1 && d.R <= 8 && d.store == 0 && d.N == d.Lin) {
      if (d.R <= 2) return d.prologue == 1 ? launch_dc8<true, 2>(d, stream) : launch_dc8<false, 2>(d, stream);
      return d.prologue == 1 ? launch_dc8<true, 8>(d, stream) : launch_dc8<false, 8>(d, stream);
    }
    if (d.up == 2) return launch_dc<3, 1, 2>(d, stream);
    if (d.up == 4) {
      if (d.R <= UP4_RMAX && d.store == 0 && d.prologue == 0 && d.transposed == 0 && d.R1 == d.R && d.Lin % 2 == 0 &&
          d.N == 4 * d.Lin) {
        if (d.R <= 8) return launch_up4<8>(d, stream);
        if (d.R <= 16) return launch_up4<16>(d, stream);
        return launch_up4<32>(d, stream);
      }
      return launch_dc<3, 1, 4>(d, stream);
    }
    return launch_dc<3, 1, 1>(d, stream);
  }
  if (d.up == 2) return launch_dc<1, 1, 2>(d, stream);
  if (d.up == 4) return launch_dc<1, 1, 4>(d, stream);
  return launch_dc<1, 1, 1>(d, stream);
}
