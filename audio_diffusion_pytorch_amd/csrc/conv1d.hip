// Fused implicit-GEMM Conv1d family for gfx950 on the exact-f32 matrix cores
// (v_mfma_f32_32x32x2_f32).  One kernel template serves:
//   * ConvBlock forward  GroupNorm+SiLU (prologue, applied while staging) -> Conv1d k3 -> +bias (+residual)
//   * Downsample (kernel = stride = f), Upsample (nearest gather folded into the loader) + Conv1d k3
//     with the SkipModulate merge  out = skip + scale[b,c] * conv  as epilogue
//   * 1x1 projections (attention q/kv/out with a LayerNorm prologue, skip adapters)
//   * every data gradient (transposed weight view; pixel-shuffle / pooled stores)
// and a second template computes the weight + bias gradients with the same loader.
//
// Stands in for a_unet's ConvBlock / Downsample / UpsampleInterpolate / Linear modules as composed by
// /root/reference/audio_diffusion_pytorch/components.py:84-99 (semantics: SURVEY.md section 8a rows a11-a16).
//
// Data layout: activations [B, C, L], L fastest.  The GEMM view is  out[m, n] = sum_{(r,t)} A[m,(r,t)] X[(r,t), n]
// with m = output channel, n = output position, r = input channel, t = tap.  Each MFMA consumes a K pair
// (r, r+1) at one tap: lane l supplies A[m = l&31][k = l>>5] and B[k = l>>5][n = l&31], so both LDS fragment
// reads are unit-stride across the 32 lanes of a half-wave (conflict-free ds_read_b32).
#include "adp_rt.h"
#include "adp.h"
#include "conv_internal.h"

namespace {

constexpr int BK = 16;      // input channels staged per K-chunk
constexpr int DILMAX = 4;   // largest supported dilation

struct XSrc {
  const float* x;
  const float* x2;
  const float* stats;
  const float* gamma;
  const float* beta;
  int64_t R, R1, Lin, up;
  int prologue, groups;
};

// Stage ROWS channels [r0, r0+ROWS) x XS virtual positions [ustart, ustart+XS) of batch element b into
// Xs (row stride XSP), applying the prologue; zero outside the tensor (conv zero padding is post-activation).
template <int NW, int ROWS>
__device__ __forceinline__ void stage_x(const XSrc& s, int64_t b, int64_t r0, int64_t ustart, int XS, int XSP,
                                        float* Xs, int wave, int lane) {
  const int64_t Lv = s.Lin * s.up;
  for (int rl = wave; rl < ROWS; rl += NW) {
    const int64_t r = r0 + rl;
    float* row = Xs + rl * XSP;
    if (r >= s.R) {
      for (int p = lane; p < XS; p += 64) row[p] = 0.0f;
      continue;
    }
    const float* src = (r < s.R1) ? s.x + (b * s.R1 + r) * s.Lin : s.x2 + (b * (s.R - s.R1) + (r - s.R1)) * s.Lin;
    float ga = 1.0f, be = 0.0f, mean = 0.0f;
    if (s.prologue != 0) {
      ga = s.gamma ? s.gamma[r] : 1.0f;
      be = s.beta ? s.beta[r] : 0.0f;
    }
    if (s.prologue == 1) {
      const int64_t g = r / (s.R / s.groups);
      mean = s.stats[(b * s.groups + g) * 2];
      ga *= s.stats[(b * s.groups + g) * 2 + 1];
    }
    for (int p = lane; p < XS; p += 64) {
      const int64_t u = ustart + p;
      float v = 0.0f;
      if (u >= 0 && u < Lv) {
        const int64_t l = (s.up == 1) ? u : u / s.up;
        v = src[l];
        if (s.prologue == 1) {
          v = adp_silu(fmaf(v - mean, ga, be));
        } else if (s.prologue == 2) {
          const float mu = s.stats[(b * s.Lin + l) * 2], rs = s.stats[(b * s.Lin + l) * 2 + 1];
          v = fmaf((v - mu) * rs, ga, be);
        }
      }
      row[p] = v;
    }
  }
}

template <int BM, int BN, int WM, int WN, int KT, int S>
__global__ __launch_bounds__((BM / WM) * (BN / WN) * 64) void conv_kernel(adp_conv_desc d) {
  constexpr int NWN = BN / WN, NW = (BM / WM) * NWN, NT = NW * 64;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int XSP = ((BN - 1) * S + (KT - 1) * DILMAX + 1) | 1;
  constexpr int BMP = BM + 1;
  constexpr int QK = BK * KT;
  __shared__ float As[QK * BMP];
  __shared__ float Xs[BK * XSP];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hi = lane >> 5, l31 = lane & 31;
  const int wm0 = (wave / NWN) * WM, wn0 = (wave % NWN) * WN;
  const int64_t b = blockIdx.z, m0 = (int64_t)blockIdx.y * BM, n0 = (int64_t)blockIdx.x * BN;
  const int dil = (int)d.dil;
  const int XS = (BN - 1) * S + (KT - 1) * dil + 1;
  const int64_t ustart = n0 * S - d.pad;
  const int64_t M = d.M, R = d.R;

  XSrc xs{d.x, d.x2, d.pro_stats, d.pro_gamma, d.pro_beta, d.R, d.R1, d.Lin, d.up, (int)d.prologue, (int)d.groups};

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  for (int64_t r0 = 0; r0 < R; r0 += BK) {
    // ---- stage A (weights) into As[(r_local*KT + t)][m]
    if (!d.transposed) {
      for (int e = tid; e < BM * QK; e += NT) {
        const int m = e / QK, q = e - m * QK;
        const int rl = q / KT;
        float v = 0.0f;
        if (m0 + m < M && r0 + rl < R) v = d.w[((m0 + m) * R + r0) * KT + q];
        As[q * BMP + m] = v;
      }
    } else {
      for (int e = tid; e < BK * BM * KT; e += NT) {
        const int rl = e / (BM * KT), rem = e - rl * (BM * KT);
        const int m = rem / KT, tp = rem - m * KT;
        float v = 0.0f;
        if (m0 + m < M && r0 + rl < R) v = d.w[((r0 + rl) * M + m0) * KT + rem];
        As[(rl * KT + (KT - 1 - tp)) * BMP + m] = v;
      }
    }
    // ---- stage X with the fused prologue
    stage_x<NW, BK>(xs, b, r0, ustart, XS, XSP, Xs, wave, lane);
    __syncthreads();

    const int kp = (int)((R - r0) < BK ? (R - r0) : BK);
    const int kpairs = (kp + 1) >> 1;
#pragma unroll
    for (int t = 0; t < KT; ++t) {
      for (int rp = 0; rp < kpairs; ++rp) {
        const int row = 2 * rp + hi;
        float a[TM], bb[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = As[(row * KT + t) * BMP + wm0 + 32 * i + l31];
#pragma unroll
        for (int j = 0; j < TN; ++j) bb[j] = Xs[row * XSP + (wn0 + 32 * j + l31) * S + t * dil];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = adp_mfma32(a[i], bb[j], acc[i][j]);
      }
    }
    __syncthreads();
  }

  // ---- epilogue
  const int64_t N = d.N;
  const int sp = (int)d.sp;
  const int64_t ebs = d.e_bstride ? d.e_bstride : M;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int64_t n = n0 + wn0 + 32 * j + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t m = m0 + wm0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * hi;
        const bool ok = (m < M) && (n < N);
        float v = acc[i][j][r];
        if (ok) {
          if (d.bias) v += d.bias[m];
          if (d.out_pre) d.out_pre[(b * M + m) * N + n] = v;
          if (d.e_scale) v *= d.e_scale[b * ebs + m];
        } else {
          v = 0.0f;
        }
        if (d.store == 0) {
          if (ok) {
            const int64_t o = (b * M + m) * N + n;
            if (d.res) v += d.res[o];
            d.out[o] = v;
          }
        } else if (d.store == 1) {
          if (ok) {
            const int64_t o = (b * (M / sp) + m / sp) * (N * sp) + n * sp + (m % sp);
            if (d.res) v += d.res[o];
            d.out[o] = v;
          }
        } else {
          v += __shfl_xor(v, 1, 64);
          if (sp == 4) v += __shfl_xor(v, 2, 64);
          if (ok && (l31 % sp) == 0) {
            const int64_t o = (b * M + m) * (N / sp) + n / sp;
            if (d.res) v += d.res[o];
            d.out[o] = v;
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Pipelined stride-1 variant (KT = 1 or 3): the workhorse of every ResNet ConvBlock, upsample conv, 1x1
// projection and data gradient.  Software pipeline per K-chunk of BKT channels:
//     issue global loads of chunk c+1 into registers -> MFMA over chunk c from LDS buffer c&1 ->
//     apply the prologue and write chunk c+1 into LDS buffer (c+1)&1 -> one barrier
// so HBM/L2 latency hides under the matrix-core work, with two LDS buffers and a single barrier per chunk.
// Weight tile in LDS: forward  As[q = r*KT+t][m]   (4 scalar loads of 4 rows -> one ds_write_b128 along m),
//                     gradient As[r][m*KT + t']     (straight 16-byte copies of the contiguous [M][KT] run;
//                                                    fragment reads stride KT=3 floats: conflict-free).
// GroupNorm (mean, rstd*gamma, beta) per input channel and LayerNorm (mean, rstd) per staged position sit in
// LDS for the whole kernel, so the prologue never waits on global memory.
// ------------------------------------------------------------------------------------------------
constexpr int PRO_RMAX = 1024;  // channels whose prologue constants fit the LDS tables

template <int BM, int BN, int WM, int WN, int KT, int BKT>
__global__ __launch_bounds__((BM / WM) * (BN / WN) * 64) void conv_s1_kernel(adp_conv_desc d) {
  constexpr int NWN = BN / WN, NW = (BM / WM) * NWN, NT = NW * 64;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int XSP = ((BN - 1) + (KT - 1) * DILMAX + 1) | 1;
  constexpr int BMP = BM + 4;            // forward layout row stride (floats), multiple of 4
  constexpr int RP = BM * KT + 4;        // gradient layout row stride
  constexpr int QK = BKT * KT;
  constexpr int A_ELEMS = QK * BMP;      // >= BKT * RP
  constexpr int NA4 = (BM * QK / 4 + NT - 1) / NT;   // float4 items per thread per chunk
  constexpr int NX = (BKT * XSP + NT - 1) / NT;      // x elements per thread per chunk
  __shared__ __attribute__((aligned(16))) float As[2][A_ELEMS];
  __shared__ float Xs[2][BKT * XSP];
  __shared__ float Pa[PRO_RMAX], Pb[PRO_RMAX];  // GroupNorm: silu(x*Pa + Pb), Pb = beta - mean*Pa
  __shared__ float Lmu[XSP], Lrs[XSP];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hi = lane >> 5, l31 = lane & 31;
  const int wm0 = (wave / NWN) * WM, wn0 = (wave % NWN) * WN;
  const int64_t b = blockIdx.z, m0 = (int64_t)blockIdx.y * BM, n0 = (int64_t)blockIdx.x * BN;
  const int dil = (int)d.dil;
  const int XS = (BN - 1) + (KT - 1) * dil + 1;
  const int64_t ustart = n0 - d.pad;
  const int64_t M = d.M, R = d.R, R1 = d.R1, Lin = d.Lin;
  const int64_t Lv = Lin * d.up;
  const int ush = (d.up == 4) ? 2 : (d.up == 2 ? 1 : 0);
  const int prologue = (int)d.prologue;
  const bool tr = d.transposed != 0;
  // 16-byte weight loads are legal for the gradient layout when rows stay aligned and the tile is interior
  const bool vecA = tr && ((M * KT) % 4 == 0) && (m0 + BM <= M);

  // ---- one-time prologue tables
  if (prologue != 0) {
    for (int64_t r = tid; r < R; r += NT) {
      float ga = d.pro_gamma ? d.pro_gamma[r] : 1.0f, mean = 0.0f;
      if (prologue == 1) {
        const int64_t g = r / (R / d.groups);
        mean = d.pro_stats[(b * d.groups + g) * 2];
        ga *= d.pro_stats[(b * d.groups + g) * 2 + 1];
      }
      Pa[r] = ga;
      Pb[r] = (d.pro_beta ? d.pro_beta[r] : 0.0f) - mean * ga;
    }
    if (prologue == 2) {
      for (int p = tid; p < XSP; p += NT) {
        const int64_t u = ustart + p;
        float mu = 0.0f, rs = 0.0f;
        if (u >= 0 && u < Lv) {
          const int64_t l = u >> ush;
          mu = d.pro_stats[(b * Lin + l) * 2];
          rs = d.pro_stats[(b * Lin + l) * 2 + 1];
        }
        Lmu[p] = mu;
        Lrs[p] = rs;
      }
    }
  }

  float4 ra[NA4];
  float rx[NX];

  auto load_chunk = [&](int64_t r0) {
    if (!tr) {
#pragma unroll
      for (int i = 0; i < NA4; ++i) {
        const int e = tid + i * NT;
        const int q = e % QK, m4 = e / QK;
        const int rl = q / KT;
        float v[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (e < BM * QK / 4 && r0 + rl < R) {
          const float* wp = d.w + ((m0 + 4 * m4) * R + r0) * KT + q;
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (m0 + 4 * m4 + j < M) v[j] = wp[(int64_t)j * R * KT];
        }
        ra[i] = make_float4(v[0], v[1], v[2], v[3]);
      }
    } else {
      constexpr int C4 = BM * KT / 4;  // float4 items per gradient-layout row
#pragma unroll
      for (int i = 0; i < NA4; ++i) {
        const int e = tid + i * NT;
        const int rl = e / C4, c4 = e % C4;
        float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (e < BKT * C4 && r0 + rl < R) {
          const float* wp = d.w + ((r0 + rl) * M + m0) * KT + 4 * c4;
          if (vecA) {
            v = *reinterpret_cast<const float4*>(wp);
          } else {
            const int64_t lim = (M - m0) * KT;  // valid floats in this row of the tile
            float t4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) t4[j] = (4 * c4 + j < lim) ? wp[j] : 0.0f;
            v = make_float4(t4[0], t4[1], t4[2], t4[3]);
          }
        }
        ra[i] = v;
      }
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int e = tid + i * NT;
      const int rl = e / XSP, p = e % XSP;
      const int64_t r = r0 + rl, u = ustart + p;
      float v = 0.0f;
      if (rl < BKT && p < XS && r < R && u >= 0 && u < Lv) {
        const float* src = (r < R1) ? d.x + (b * R1 + r) * Lin : d.x2 + (b * (R - R1) + (r - R1)) * Lin;
        v = src[u >> ush];
      }
      rx[i] = v;
    }
  };

  auto store_chunk = [&](int buf, int64_t r0) {
    float* Ab = As[buf];
    float* Xb = Xs[buf];
    if (!tr) {
#pragma unroll
      for (int i = 0; i < NA4; ++i) {
        const int e = tid + i * NT;
        const int q = e % QK, m4 = e / QK;
        if (e < BM * QK / 4) *reinterpret_cast<float4*>(Ab + q * BMP + 4 * m4) = ra[i];
      }
    } else {
      constexpr int C4 = BM * KT / 4;
#pragma unroll
      for (int i = 0; i < NA4; ++i) {
        const int e = tid + i * NT;
        const int rl = e / C4, c4 = e % C4;
        if (e < BKT * C4) *reinterpret_cast<float4*>(Ab + rl * RP + 4 * c4) = ra[i];
      }
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int e = tid + i * NT;
      const int rl = e / XSP, p = e % XSP;
      if (rl < BKT) {
        const int64_t r = r0 + rl, u = ustart + p;
        float v = rx[i];
        if (prologue != 0 && p < XS && r < R && u >= 0 && u < Lv) {
          const int rr = (int)r;
          if (prologue == 1) v = adp_silu(fmaf(v, Pa[rr], Pb[rr]));
          else v = fmaf((v - Lmu[p]) * Lrs[p], Pa[rr], Pb[rr]);
        }
        Xb[rl * XSP + p] = v;
      }
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  if (prologue != 0) __syncthreads();
  load_chunk(0);
  store_chunk(0, 0);
  __syncthreads();

  int buf = 0;
  for (int64_t r0 = 0; r0 < R; r0 += BKT, buf ^= 1) {
    const bool more = r0 + BKT < R;
    if (more) load_chunk(r0 + BKT);
    const float* Ab = As[buf];
    const float* Xb = Xs[buf];
    const int kp = (int)((R - r0) < BKT ? (R - r0) : BKT);
    const int kpairs = (kp + 1) >> 1;
#pragma unroll
    for (int t = 0; t < KT; ++t) {
      if (kpairs == BKT / 2) {
#pragma unroll
        for (int rp = 0; rp < BKT / 2; ++rp) {
          const int row = 2 * rp + hi;
          float a[TM], bb[TN];
#pragma unroll
          for (int i = 0; i < TM; ++i)
            a[i] = tr ? Ab[row * RP + (wm0 + 32 * i + l31) * KT + (KT - 1 - t)]
                      : Ab[(row * KT + t) * BMP + wm0 + 32 * i + l31];
#pragma unroll
          for (int j = 0; j < TN; ++j) bb[j] = Xb[row * XSP + wn0 + 32 * j + l31 + t * dil];
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = adp_mfma32(a[i], bb[j], acc[i][j]);
        }
      } else {
        for (int rp = 0; rp < kpairs; ++rp) {
          const int row = 2 * rp + hi;
          float a[TM], bb[TN];
#pragma unroll
          for (int i = 0; i < TM; ++i)
            a[i] = tr ? Ab[row * RP + (wm0 + 32 * i + l31) * KT + (KT - 1 - t)]
                      : Ab[(row * KT + t) * BMP + wm0 + 32 * i + l31];
#pragma unroll
          for (int j = 0; j < TN; ++j) bb[j] = Xb[row * XSP + wn0 + 32 * j + l31 + t * dil];
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = adp_mfma32(a[i], bb[j], acc[i][j]);
        }
      }
    }
    if (more) store_chunk(buf ^ 1, r0 + BKT);
    __syncthreads();
  }

  // ---- epilogue (identical contract to conv_kernel)
  const int64_t N = d.N;
  const int sp = (int)d.sp;
  const int64_t ebs = d.e_bstride ? d.e_bstride : M;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int64_t n = n0 + wn0 + 32 * j + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t m = m0 + wm0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * hi;
        const bool ok = (m < M) && (n < N);
        float v = acc[i][j][r];
        if (ok) {
          if (d.bias) v += d.bias[m];
          if (d.out_pre) d.out_pre[(b * M + m) * N + n] = v;
          if (d.e_scale) v *= d.e_scale[b * ebs + m];
        } else {
          v = 0.0f;
        }
        if (d.store == 0) {
          if (ok) {
            const int64_t o = (b * M + m) * N + n;
            if (d.res) v += d.res[o];
            d.out[o] = v;
          }
        } else if (d.store == 1) {
          if (ok) {
            const int64_t o = (b * (M / sp) + m / sp) * (N * sp) + n * sp + (m % sp);
            if (d.res) v += d.res[o];
            d.out[o] = v;
          }
        } else {
          v += __shfl_xor(v, 1, 64);
          if (sp == 4) v += __shfl_xor(v, 2, 64);
          if (ok && (l31 % sp) == 0) {
            const int64_t o = (b * M + m) * (N / sp) + n / sp;
            if (d.res) v += d.res[o];
            d.out[o] = v;
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Weight gradient: block tile = 32 (m) x 32 (r) x KT taps; the 4 waves split the position range of
// each staged chunk and are summed through LDS in a fixed order (deterministic); partial tiles of the
// nsplit position ranges go to ws and are summed by wgrad_reduce_kernel.
// ------------------------------------------------------------------------------------------------
template <int KT, int S>
struct WgradCfg {
  static constexpr int BKN = (S == 4) ? 64 : 128;  // positions staged per chunk
  static constexpr int XSP = ((BKN - 1) * S + (KT - 1) * DILMAX + 1) | 1;
  static constexpr int DP = BKN + 1;
};

template <int KT, int S>
__global__ __launch_bounds__(256) void wgrad_kernel(adp_wgrad_desc d, int64_t PS, int64_t SPB) {
  using C = WgradCfg<KT, S>;
  constexpr int BKN = C::BKN, XSP = C::XSP, DP = C::DP, NW = 4;
  __shared__ float Dys[32 * DP];
  __shared__ float Xs[32 * XSP];
  __shared__ float Red[KT * 1024 + 32];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hi = lane >> 5, l31 = lane & 31;
  const int64_t split = blockIdx.x;
  const int64_t b = split / SPB, pbeg = (split % SPB) * PS;
  const int64_t pend = (pbeg + PS < d.N) ? pbeg + PS : d.N;
  const int64_t m0 = (int64_t)blockIdx.y * 32, r0 = (int64_t)blockIdx.z * 32;
  const int dil = (int)d.dil;
  const int XS = (BKN - 1) * S + (KT - 1) * dil + 1;
  const bool do_bias = (d.dbias != nullptr) && (blockIdx.z == 0);

  XSrc xs{d.x, d.x2, d.pro_stats, d.pro_gamma, d.pro_beta, d.R, d.R1, d.Lin, d.up, (int)d.prologue, (int)d.groups};

  f32x16 acc[KT];
  f32x16 accb;
#pragma unroll
  for (int r = 0; r < 16; ++r) accb[r] = 0.0f;
#pragma unroll
  for (int t = 0; t < KT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

  for (int64_t p0 = pbeg; p0 < pend; p0 += BKN) {
    // stage dy tile [32 m][BKN n]
    for (int ml = wave; ml < 32; ml += NW) {
      const int64_t m = m0 + ml;
      for (int k = lane; k < BKN; k += 64) {
        const int64_t n = p0 + k;
        float v = 0.0f;
        if (m < d.M && n < pend) v = d.dy[(b * d.M + m) * d.N + n];
        Dys[ml * DP + k] = v;
      }
    }
    stage_x<NW, 32>(xs, b, r0, p0 * S - d.pad, XS, XSP, Xs, wave, lane);
    __syncthreads();
    const int kbeg = wave * (BKN / NW);
#pragma unroll 4
    for (int kk = 0; kk < BKN / NW; kk += 2) {
      const int k = kbeg + kk + hi;
      const float a = Dys[l31 * DP + k];
#pragma unroll
      for (int t = 0; t < KT; ++t) {
        const float bv = Xs[l31 * XSP + k * S + t * dil];
        acc[t] = adp_mfma32(a, bv, acc[t]);
      }
      if (do_bias) accb = adp_mfma32(a, 1.0f, accb);
    }
    __syncthreads();
  }

  // deterministic cross-wave sum through LDS: Red[t][row][col], bias in Red[KT*1024 + row]
  for (int w = 0; w < NW; ++w) {
    if (wave == w) {
#pragma unroll
      for (int t = 0; t < KT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
          float* p = &Red[t * 1024 + row * 32 + l31];
          *p = (w == 0) ? acc[t][r] : (*p + acc[t][r]);
        }
      if (do_bias && l31 == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
          float* p = &Red[KT * 1024 + row];
          *p = (w == 0) ? accb[r] : (*p + accb[r]);
        }
      }
    }
    __syncthreads();
  }
  // write partial tile: ws[split][m][r][t]
  float* wsw = d.ws + split * (d.M * d.R * KT);
  for (int e = tid; e < 32 * 32 * KT; e += 256) {
    const int ml = e / (32 * KT), rem = e - ml * (32 * KT);
    const int rl = rem / KT, t = rem - rl * KT;
    const int64_t m = m0 + ml, r = r0 + rl;
    if (m < d.M && r < d.R) wsw[(m * d.R + r) * KT + t] = Red[t * 1024 + ml * 32 + rl];
  }
  if (do_bias && tid < 32 && m0 + tid < d.M) {
    float* wsb = d.ws + (int64_t)gridDim.x * (d.M * d.R * KT) + split * d.M;
    wsb[m0 + tid] = Red[KT * 1024 + tid];
  }
}

// ------------------------------------------------------------------------------------------------
// Pipelined stride-1 weight gradient for wide layers: block tile 64 (m) x 64 (r) x KT, the 4 waves form a
// 2 x 2 grid of 32 x 32 x KT accumulators (no cross-wave reduction), position chunks of 64 are double-buffered
// in LDS with register prefetch exactly like conv_s1_kernel.  A block walks a contiguous range of
// (batch, chunk) pairs; with nsplit == 1 it writes dw/dbias directly, otherwise a partial to ws.
// ------------------------------------------------------------------------------------------------
template <int KT>
__global__ __launch_bounds__(256) void wgrad_s1_kernel(adp_wgrad_desc d, int64_t CPS, int64_t CPB, int64_t nsplit) {
  constexpr int BKN = 64, NT = 256;
  constexpr int DP = BKN + 1;
  constexpr int XSP = (BKN + (KT - 1) * DILMAX + 1) | 1;
  constexpr int NX = (64 * XSP + NT - 1) / NT;
  __shared__ float Dys[2][64 * DP];
  __shared__ float Xs[2][64 * XSP];
  __shared__ float Pa[64], Pb[64];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hi = lane >> 5, l31 = lane & 31;
  const int wm0 = (wave >> 1) * 32, wr0 = (wave & 1) * 32;
  const int64_t split = blockIdx.x;
  const int64_t m0 = (int64_t)blockIdx.y * 64, r0 = (int64_t)blockIdx.z * 64;
  const int dil = (int)d.dil;
  const int XS = BKN + (KT - 1) * dil;
  const int64_t M = d.M, R = d.R, R1 = d.R1, N = d.N, Lin = d.Lin;
  const int64_t Lv = Lin * d.up;
  const int ush = (d.up == 4) ? 2 : (d.up == 2 ? 1 : 0);
  const int prologue = (int)d.prologue;
  const bool do_bias = (d.dbias != nullptr) && (blockIdx.z == 0) && (wr0 == 0);
  const int64_t total = d.B * CPB;
  const int64_t cbeg = split * CPS, cend = (cbeg + CPS < total) ? cbeg + CPS : total;

  float rdy[16];
  float rx[NX];
  int64_t cur_b = -1;

  auto load_chunk = [&](int64_t c) {
    const int64_t b = c / CPB, p0 = (c % CPB) * BKN;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int64_t m = m0 + wave + 4 * i, n = p0 + lane;
      rdy[i] = (m < M && n < N) ? d.dy[(b * M + m) * N + n] : 0.0f;
    }
    const int64_t ustart = p0 - d.pad;
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int e = tid + i * NT;
      const int rl = e / XSP, p = e % XSP;
      const int64_t r = r0 + rl, u = ustart + p;
      float v = 0.0f;
      if (rl < 64 && p < XS && r < R && u >= 0 && u < Lv) {
        const float* src = (r < R1) ? d.x + (b * R1 + r) * Lin : d.x2 + (b * (R - R1) + (r - R1)) * Lin;
        v = src[u >> ush];
      }
      rx[i] = v;
    }
  };
  // GroupNorm constants depend on the batch element: refreshed (by all threads, between barriers) when b changes
  auto set_batch = [&](int64_t b) {
    if (prologue == 1 && tid < 64) {
      const int64_t r = r0 + tid;
      float ga = 1.0f, be = 0.0f;
      if (r < R) {
        const int64_t g = r / (R / d.groups);
        const float mean = d.pro_stats[(b * d.groups + g) * 2];
        ga = (d.pro_gamma ? d.pro_gamma[r] : 1.0f) * d.pro_stats[(b * d.groups + g) * 2 + 1];
        be = (d.pro_beta ? d.pro_beta[r] : 0.0f) - mean * ga;
      }
      Pa[tid] = ga;
      Pb[tid] = be;
    }
  };
  auto store_chunk = [&](int buf, int64_t c) {
    const int64_t p0 = (c % CPB) * BKN;
    float* Db = Dys[buf];
    float* Xb = Xs[buf];
#pragma unroll
    for (int i = 0; i < 16; ++i) Db[(wave + 4 * i) * DP + lane] = rdy[i];
    const int64_t ustart = p0 - d.pad;
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int e = tid + i * NT;
      const int rl = e / XSP, p = e % XSP;
      if (rl < 64) {
        const int64_t r = r0 + rl, u = ustart + p;
        float v = rx[i];
        if (prologue == 1 && p < XS && r < R && u >= 0 && u < Lv) v = adp_silu(fmaf(v, Pa[rl], Pb[rl]));
        Xb[rl * XSP + p] = v;
      }
    }
  };

  f32x16 acc[KT];
  f32x16 accb;
#pragma unroll
  for (int r = 0; r < 16; ++r) accb[r] = 0.0f;
#pragma unroll
  for (int t = 0; t < KT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

  if (cbeg < cend) {
    cur_b = cbeg / CPB;
    set_batch(cur_b);
    __syncthreads();
    load_chunk(cbeg);
    store_chunk(0, cbeg);
    __syncthreads();
  }
  int buf = 0;
  for (int64_t c = cbeg; c < cend; ++c, buf ^= 1) {
    const bool more = c + 1 < cend;
    if (more) load_chunk(c + 1);
    const float* Db = Dys[buf];
    const float* Xb = Xs[buf];
#pragma unroll 8
    for (int kk = 0; kk < BKN; kk += 2) {
      const int k = kk + hi;
      const float a = Db[(wm0 + l31) * DP + k];
#pragma unroll
      for (int t = 0; t < KT; ++t) acc[t] = adp_mfma32(a, Xb[(wr0 + l31) * XSP + k + t * dil], acc[t]);
      if (do_bias) accb = adp_mfma32(a, 1.0f, accb);
    }
    if (more) {
      const int64_t nb = (c + 1) / CPB;
      if (nb != cur_b) {  // block-uniform: the next chunk starts a new batch element -> new GroupNorm constants
        __syncthreads();
        set_batch(nb);
        cur_b = nb;
        __syncthreads();
      }
      store_chunk(buf ^ 1, c + 1);
    }
    __syncthreads();
  }

  const bool direct = (nsplit == 1);
  float* base = direct ? d.dw : d.ws + split * (M * R * KT);
#pragma unroll
  for (int t = 0; t < KT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t m = m0 + wm0 + (r & 3) + 8 * (r >> 2) + 4 * hi, rr = r0 + wr0 + l31;
      if (m < M && rr < R) {
        float* o = base + (m * R + rr) * KT + t;
        *o = (direct && (d.accumulate & 1)) ? *o + acc[t][r] : acc[t][r];
      }
    }
  if (do_bias && l31 == 0) {
    float* bb = direct ? d.dbias : d.ws + nsplit * (M * R * KT) + split * M;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t m = m0 + wm0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      if (m < M) bb[m] = (direct && (d.accumulate & 1)) ? bb[m] + accb[r] : accb[r];
    }
  }
}

template <int BM, int BN, int WM, int WN, int KT, int S>
int launch_conv(const adp_conv_desc& d, void* stream) {
  dim3 grid((unsigned)adp_cdiv(d.N, BN), (unsigned)adp_cdiv(d.M, BM), (unsigned)d.B);
  dim3 block((BM / WM) * (BN / WN) * 64);
  ADP_LAUNCH((conv_kernel<BM, BN, WM, WN, KT, S>), grid, block, stream, d);
  return ADP_LAUNCH_OK();
}

template <int BM, int BN, int WM, int WN, int KT, int BKT>
int launch_conv_s1(const adp_conv_desc& d, void* stream) {
  dim3 grid((unsigned)adp_cdiv(d.N, BN), (unsigned)adp_cdiv(d.M, BM), (unsigned)d.B);
  dim3 block((BM / WM) * (BN / WN) * 64);
  ADP_LAUNCH((conv_s1_kernel<BM, BN, WM, WN, KT, BKT>), grid, block, stream, d);
  return ADP_LAUNCH_OK();
}

// tile choice (shared with adp_conv1d_tile): 32-row tiles for narrow layers, 128x128 when that still fills
// the chip with >= 1.5 workgroups per CU, else 64x64
int64_t pick_tile(const adp_conv_desc& d) {
  if (d.M <= 32) return 32 * 1000 + 128;
  if (d.stride != 4) {
    const int64_t big = adp_cdiv(d.M, 128) * adp_cdiv(d.N, 128) * d.B;
    if (d.M >= 128 && big >= 384) return 128 * 1000 + 128;
  }
  return 64 * 1000 + 64;
}

bool s1_eligible(const adp_conv_desc& d) {
  return d.stride == 1 && (d.KT == 1 || d.KT == 3) && (d.up == 1 || d.up == 2 || d.up == 4) &&
         (d.prologue == 0 || d.R <= PRO_RMAX);
}

template <int KT, int S>
int dispatch_conv(const adp_conv_desc& d, void* stream) {
  const int64_t tile = pick_tile(d);
  if constexpr (S == 1) {
    if (s1_eligible(d)) {
      if (tile == 32128) return launch_conv_s1<32, 128, 32, 32, KT, 32>(d, stream);
      if (tile == 128128) return launch_conv_s1<128, 128, 64, 64, KT, 16>(d, stream);
      return launch_conv_s1<64, 64, 32, 32, KT, 32>(d, stream);
    }
  }
  if (tile == 32128) return launch_conv<32, 128, 32, 32, KT, S>(d, stream);
  if (tile == 128128) return launch_conv<128, 128, 64, 64, KT, S>(d, stream);
  return launch_conv<64, 64, 32, 32, KT, S>(d, stream);
}

void wgrad_split(const adp_wgrad_desc& d, int BKN, int64_t* PS, int64_t* SPB) {
  const int64_t tiles = adp_cdiv(d.M, 32) * adp_cdiv(d.R, 32);
  int64_t want = adp_cdiv(1024, tiles);                       // target ~1024 workgroups
  int64_t spb = adp_cdiv(want, d.B);
  const int64_t maxspb = adp_cdiv(d.N, BKN);
  if (spb > maxspb) spb = maxspb;
  if (spb < 1) spb = 1;
  int64_t ps = adp_cdiv(adp_cdiv(d.N, spb), BKN) * BKN;
  *PS = ps;
  *SPB = adp_cdiv(d.N, ps);
}

int wgrad_bkn(int64_t S) { return S == 4 ? 64 : 128; }

// wide stride-1 layers take the pipelined 64x64 kernel
bool wgrad_s1_eligible(const adp_wgrad_desc& d) {
  return d.stride == 1 && (d.KT == 1 || d.KT == 3) && (d.up == 1 || d.up == 2 || d.up == 4) && d.prologue != 2 &&
         (d.M > 32 || d.R > 32);
}
// chunks per batch element, chunks per split, number of splits (~2 workgroups per CU)
void wgrad_s1_split(const adp_wgrad_desc& d, int64_t* CPB, int64_t* CPS, int64_t* nsplit) {
  const int64_t tiles = adp_cdiv(d.M, 64) * adp_cdiv(d.R, 64);
  const int64_t cpb = adp_cdiv(d.N, 64), total = d.B * cpb;
  int64_t ns = adp_cdiv(512, tiles);
  if (ns > total) ns = total;
  if (ns < 1) ns = 1;
  const int64_t cps = adp_cdiv(total, ns);
  *CPB = cpb;
  *CPS = cps;
  *nsplit = adp_cdiv(total, cps);
}

template <int KT>
int launch_wgrad_s1(const adp_wgrad_desc& d, void* stream) {
  int64_t CPB, CPS, nsplit;
  wgrad_s1_split(d, &CPB, &CPS, &nsplit);
  dim3 grid((unsigned)nsplit, (unsigned)adp_cdiv(d.M, 64), (unsigned)adp_cdiv(d.R, 64));
  ADP_LAUNCH((wgrad_s1_kernel<KT>), grid, dim3(256), stream, d, CPS, CPB, nsplit);
  if (nsplit > 1) {
    return adp_wgrad_reduce(d.ws, nsplit, d.M * d.R * KT, d.M, d.dw, d.dbias, (int)(d.accumulate & 1), stream);
  }
  return ADP_LAUNCH_OK();
}

template <int KT, int S>
int launch_wgrad(const adp_wgrad_desc& d, void* stream) {
  if constexpr (S == 1) {
    if (wgrad_s1_eligible(d)) return launch_wgrad_s1<KT>(d, stream);
  }
  int64_t PS, SPB;
  wgrad_split(d, WgradCfg<KT, S>::BKN, &PS, &SPB);
  const int64_t nsplit = d.B * SPB;
  dim3 grid((unsigned)nsplit, (unsigned)adp_cdiv(d.M, 32), (unsigned)adp_cdiv(d.R, 32));
  ADP_LAUNCH((wgrad_kernel<KT, S>), grid, dim3(256), stream, d, PS, SPB);
  return adp_wgrad_reduce(d.ws, nsplit, d.M * d.R * KT, d.M, d.dw, d.dbias, (int)(d.accumulate & 1), stream);
}

bool ks_supported(int64_t KT, int64_t S) {
  return (KT == 1 && S == 1) || (KT == 2 && S == 2) || (KT == 3 && S == 1) || (KT == 4 && S == 4);
}

}  // namespace

extern "C" int64_t adp_conv1d_gnb_entries(const adp_conv_desc* dp);

extern "C" int adp_conv1d(const adp_conv_desc* dp, void* stream) {
  if (!dp) return ADP_ERR_NULL;
  const adp_conv_desc& d = *dp;
  if (!d.x || !d.w || !d.out) return ADP_ERR_NULL;
  if (d.B <= 0 || d.R <= 0 || d.M <= 0 || d.N <= 0 || d.Lin <= 0 || d.up < 1 || d.R1 < 0 || d.R1 > d.R)
    return ADP_ERR_SHAPE;
  if (d.R1 < d.R && !d.x2) return ADP_ERR_NULL;
  if (d.dil < 1 || d.dil > DILMAX) return ADP_ERR_UNSUPPORTED;
  if (!ks_supported(d.KT, d.stride)) return ADP_ERR_UNSUPPORTED;
  if (d.prologue < 0 || d.prologue > 2 || (d.prologue != 0 && !d.pro_stats)) return ADP_ERR_NULL;
  if (d.prologue == 1 && (d.groups < 1 || d.R % d.groups != 0)) return ADP_ERR_SHAPE;
  if (d.store < 0 || d.store > 2 || (d.out_pre && d.store != 0)) return ADP_ERR_UNSUPPORTED;
  if (d.store == 1 && (d.sp < 1 || d.M % d.sp != 0)) return ADP_ERR_SHAPE;
  if (d.store == 2 && ((d.sp != 2 && d.sp != 4) || d.N % d.sp != 0 || d.bias)) return ADP_ERR_UNSUPPORTED;
  if (d.B > 65535 || adp_cdiv(d.M, 32) > 65535) return ADP_ERR_SHAPE;
  if (d.gnb_ab) {  // (the caller asks adp_conv1d_gnb_entries first; a launch that cannot fill gnb_ab must not pretend to)
    if (!d.gnb_x || !d.gnb_stats || !d.gnb_gamma || !d.gnb_beta) return ADP_ERR_NULL;
    if (d.gnb_groups < 1 || d.M % d.gnb_groups != 0) return ADP_ERR_SHAPE;
    if (adp_conv1d_gnb_entries(dp) <= 0) return ADP_ERR_UNSUPPORTED;
  }
  if (adp_conv_tile_eligible(d)) return adp_conv_tile(d, stream);
  if (adp_conv_tilek_eligible(d)) return adp_conv_tilek(d, stream);
  if (adp_conv_mm4_eligible(d)) return adp_conv_mm4(d, stream);
  if (adp_conv_tilek1_eligible(d)) return adp_conv_tilek1(d, stream);
  if (adp_conv_mm_eligible(d)) return adp_conv_mm(d, stream);
  if (adp_conv_direct_eligible(d)) return adp_conv_direct(d, stream);
  if (d.KT == 1) return dispatch_conv<1, 1>(d, stream);
  if (d.KT == 2) return dispatch_conv<2, 2>(d, stream);
  if (d.KT == 3) return dispatch_conv<3, 1>(d, stream);
  return dispatch_conv<4, 4>(d, stream);
}

extern "C" int64_t adp_conv1d_ws_bytes(const adp_conv_desc* dp) {
  if (!dp) return ADP_ERR_NULL;
  const adp_conv_desc& d = *dp;
  if (d.B <= 0 || d.R <= 0 || d.M <= 0 || d.N <= 0 || d.Lin <= 0) return ADP_ERR_SHAPE;
  if (adp_conv_tile_eligible(d) || adp_conv_tilek_eligible(d)) return 0;  // (tilek: the K split stays inside the workgroup)
  if (adp_conv_mm4_eligible(d)) {
    const int64_t ks4 = adp_conv_mm4_ksplit(d);
    return ks4 > 1 ? ks4 * d.B * d.M * d.N * (int64_t)sizeof(float) : 0;
  }
  if (adp_conv_tilek1_eligible(d)) return 0;  // (its K split stays inside the workgroup)
  if (!adp_conv_mm_eligible(d)) return 0;
  const int64_t ks = adp_conv_mm_ksplit(d);
  return ks > 1 ? ks * d.B * d.M * d.N * (int64_t)sizeof(float) : 0;
}

extern "C" int64_t adp_conv1d_gn_entries(const adp_conv_desc* dp) {
  if (!dp) return ADP_ERR_NULL;
  const adp_conv_desc& d = *dp;
  if (d.B <= 0 || d.R <= 0 || d.M <= 0 || d.N <= 0 || d.Lin <= 0) return ADP_ERR_SHAPE;
  if (d.store != 0 || d.M % 4 != 0) return 0;
  if (adp_conv_tile_eligible(d)) return adp_conv_tile_gn_entries(d);
  if (adp_conv_tilek_eligible(d)) return adp_conv_tilek_gn_entries(d);
  if (adp_conv_mm4_eligible(d)) return adp_conv_mm4_gn_entries(d);
  if (adp_conv_tilek1_eligible(d)) return adp_conv_tilek1_gn_entries(d);
  // (the K split only happens when the caller passed its scratch: set d.ws before asking)
  if (adp_conv_mm_eligible(d))  // one slice per 64-position tile, or the K-split reduce kernel's slices
    return d.ws && adp_conv_mm_ksplit(d) > 1 ? adp_conv_splitk_gn_entries(d) : adp_cdiv(d.N, 64);
  return 0;
}

extern "C" int64_t adp_conv1d_gnb_entries(const adp_conv_desc* dp) {
  if (!dp) return ADP_ERR_NULL;
  const adp_conv_desc& d = *dp;
  if (d.B <= 0 || d.R <= 0 || d.M <= 0 || d.N <= 0 || d.Lin <= 0) return ADP_ERR_SHAPE;
  if (d.store != 0) return 0;
  if (adp_conv_tile_eligible(d)) return adp_conv_tile_gnb_entries(d);
  if (adp_conv_tilek_eligible(d)) return adp_conv_tilek_gnb_entries(d);
  if (adp_conv_mm4_eligible(d)) return adp_conv_mm4_gnb_entries(d);
  if (adp_conv_tilek1_eligible(d)) return 0;
  if (adp_conv_mm_eligible(d)) return d.gn_part ? 0 : adp_conv_mm_gnb_entries(d);
  return 0;
}

// which tile the dispatcher picks for this problem: BM * 1000 + BN (introspection for profiling / roofline reports)
extern "C" int64_t adp_conv1d_tile(const adp_conv_desc* dp) {
  if (!dp) return ADP_ERR_NULL;
  if (adp_conv_tile_eligible(*dp)) return 32 * 1000 + 64;   // wave-tile 32-channel kernel: 32 outputs x 64 positions per wave
  if (adp_conv_tilek_eligible(*dp)) return 48000000 + 64;   // deep-layer wave tiles: 16 / 32 rows x 64 positions, 8 K slices per workgroup
  if (adp_conv_mm4_eligible(*dp)) return 64000000 + 32 * 1000 + 128;
  if (adp_conv_tilek1_eligible(*dp)) return 47000000 + 64;  // 1x1 wave tiles: 16 rows x 64 positions, 8 K slices per workgroup  // F(4,3) block: 6 planes x 4 K groups, 32 rows x 128 positions
  if (adp_conv_mm_eligible(*dp)) return adp_conv_mm_tile(*dp);
  if (adp_conv_direct_eligible(*dp)) return 8 * 1000 + 999;  // direct VALU kernel: 8 output channels x 1024 positions
  return pick_tile(*dp);
}

extern "C" int64_t adp_conv1d_wgrad_ws_bytes(const adp_wgrad_desc* dp) {
  if (!dp || !ks_supported(dp->KT, dp->stride) || dp->B <= 0 || dp->N <= 0) return ADP_ERR_UNSUPPORTED;
  int64_t nsplit;
  if (adp_wgrad_mm_eligible(*dp)) return adp_wgrad_mm_ws_floats(*dp) * (int64_t)sizeof(float);
  if (adp_wgrad_direct_eligible(*dp)) return adp_wgrad_direct_ws_floats(*dp) * (int64_t)sizeof(float);
  if (wgrad_s1_eligible(*dp)) {
    int64_t CPB, CPS;
    wgrad_s1_split(*dp, &CPB, &CPS, &nsplit);
  } else {
    int64_t PS, SPB;
    wgrad_split(*dp, wgrad_bkn(dp->stride), &PS, &SPB);
    nsplit = dp->B * SPB;
  }
  return nsplit * (dp->M * dp->R * dp->KT + dp->M) * (int64_t)sizeof(float);
}

extern "C" int64_t adp_conv1d_wgrad_partials(const adp_wgrad_desc* dp) {
  if (!dp) return ADP_ERR_NULL;
  if (dp->B <= 0 || dp->R <= 0 || dp->M <= 0 || dp->N <= 0 || dp->Lin <= 0 || dp->up < 1) return ADP_ERR_SHAPE;
  return adp_wgrad_mm_eligible(*dp) ? adp_wgrad_mm_nsplit(*dp) : 1;  // (only the matrix-core family has a parked form)
}

extern "C" int adp_wgrad_reduce_batch(const float* const* ws, float* const* dw, float* const* dbias, int64_t n, int64_t nsplit,
                                      int64_t cnt, int64_t M, int64_t accumulate, void* stream) {
  if (!ws || !dw) return ADP_ERR_NULL;
  if (n <= 0 || nsplit < 1 || cnt <= 0 || M <= 0) return ADP_ERR_SHAPE;
  for (int64_t i = 0; i < n; ++i)
    if (!ws[i] || !dw[i] || (dbias && !dbias[i])) return ADP_ERR_NULL;
  for (int64_t i = 0; i < n; i += ADP_WGR_BATCH) {
    const int k = (int)(n - i < ADP_WGR_BATCH ? n - i : ADP_WGR_BATCH);
    const int rc = adp_wgrad_reduce_n(ws + i, dw + i, dbias ? dbias + i : nullptr, k, nsplit, cnt, M, (int)(accumulate & 1), stream);
    if (rc != ADP_OK) return rc;
  }
  return ADP_OK;
}

extern "C" int adp_conv1d_wgrad(const adp_wgrad_desc* dp, void* stream);

extern "C" int adp_conv1d_wgrad_batch(const adp_wgrad_desc* ds, int64_t n, void* stream) {
  if (!ds) return ADP_ERR_NULL;
  if (n <= 0) return ADP_ERR_SHAPE;
  // one shape: every integer field equal, the same optional pointers present
  bool same = true, mm = true;
  for (int64_t i = 0; i < n; ++i) {
    const adp_wgrad_desc &a = ds[0], &b = ds[i];
    same = same && a.B == b.B && a.R == b.R && a.R1 == b.R1 && a.Lin == b.Lin && a.M == b.M && a.N == b.N && a.KT == b.KT &&
           a.stride == b.stride && a.dil == b.dil && a.pad == b.pad && a.up == b.up && a.prologue == b.prologue &&
           a.groups == b.groups && a.accumulate == b.accumulate && !a.pro_stats == !b.pro_stats &&
           !a.pro_gamma == !b.pro_gamma && !a.pro_beta == !b.pro_beta && !a.dbias == !b.dbias && !a.x2 == !b.x2;
    if (!b.x || !b.dy || !b.dw || !b.ws) return ADP_ERR_NULL;
    if (b.B <= 0 || b.R <= 0 || b.M <= 0 || b.N <= 0 || b.Lin <= 0 || b.up < 1 || b.R1 < 0 || b.R1 > b.R) return ADP_ERR_SHAPE;
    mm = mm && adp_wgrad_mm_eligible(b) && !(b.prologue != 0 && !b.pro_stats) &&
         !(b.prologue == 1 && (b.groups < 1 || b.R % b.groups != 0));
  }
  if (!same) return ADP_ERR_SHAPE;
  if (!mm || n == 1) {  // no batched form for this family: one call per item
    for (int64_t i = 0; i < n; ++i) {
      const int rc = adp_conv1d_wgrad(ds + i, stream);
      if (rc != ADP_OK) return rc;
    }
    return ADP_OK;
  }
  for (int64_t i = 0; i < n; i += ADP_WGR_BATCH) {
    const int k = (int)(n - i < ADP_WGR_BATCH ? n - i : ADP_WGR_BATCH);
    const int rc = adp_wgrad_mm_n(ds + i, k, stream);
    if (rc != ADP_OK) return rc;
  }
  return ADP_OK;
}

extern "C" int adp_conv1d_wgrad(const adp_wgrad_desc* dp, void* stream) {
  if (!dp) return ADP_ERR_NULL;
  const adp_wgrad_desc& d = *dp;
  if (!d.x || !d.dy || !d.dw || !d.ws) return ADP_ERR_NULL;
  if (d.B <= 0 || d.R <= 0 || d.M <= 0 || d.N <= 0 || d.Lin <= 0 || d.up < 1 || d.R1 < 0 || d.R1 > d.R)
    return ADP_ERR_SHAPE;
  if (d.R1 < d.R && !d.x2) return ADP_ERR_NULL;
  if (d.dil < 1 || d.dil > DILMAX) return ADP_ERR_UNSUPPORTED;
  if (!ks_supported(d.KT, d.stride)) return ADP_ERR_UNSUPPORTED;
  if (d.prologue < 0 || d.prologue > 2 || (d.prologue != 0 && !d.pro_stats)) return ADP_ERR_NULL;
  if (d.prologue == 1 && (d.groups < 1 || d.R % d.groups != 0)) return ADP_ERR_SHAPE;
  if (adp_cdiv(d.M, 32) > 65535 || adp_cdiv(d.R, 32) > 65535) return ADP_ERR_SHAPE;
  if (adp_wgrad_mm_eligible(d)) return adp_wgrad_mm(d, stream);
  if (adp_wgrad_direct_eligible(d)) return adp_wgrad_direct(d, stream);
  if (d.KT == 1) return launch_wgrad<1, 1>(d, stream);
  if (d.KT == 2) return launch_wgrad<2, 2>(d, stream);
  if (d.KT == 3) return launch_wgrad<3, 1>(d, stream);
  return launch_wgrad<4, 4>(d, stream);
}
