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

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* ws, int64_t nsplit, int64_t cnt, int64_t M,
                                                           float* dw, float* dbias, int accumulate) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < cnt) {
    float s = 0.0f;
    for (int64_t k = 0; k < nsplit; ++k) s += ws[k * cnt + i];
    dw[i] = accumulate ? dw[i] + s : s;
  } else if (dbias && i < cnt + M) {
    const int64_t m = i - cnt;
    const float* wsb = ws + nsplit * cnt;
    float s = 0.0f;
    for (int64_t k = 0; k < nsplit; ++k) s += wsb[k * M + m];
    dbias[m] = accumulate ? dbias[m] + s : s;
  }
}

template <int BM, int BN, int WM, int WN, int KT, int S>
int launch_conv(const adp_conv_desc& d, void* stream) {
  dim3 grid((unsigned)adp_cdiv(d.N, BN), (unsigned)adp_cdiv(d.M, BM), (unsigned)d.B);
  dim3 block((BM / WM) * (BN / WN) * 64);
  ADP_LAUNCH((conv_kernel<BM, BN, WM, WN, KT, S>), grid, block, stream, d);
  return ADP_LAUNCH_OK();
}

// tile choice: fill >= ~2 workgroups per CU when the problem allows; 32-row tiles for narrow layers
template <int KT, int S>
int dispatch_conv(const adp_conv_desc& d, void* stream) {
  const int64_t M = d.M, N = d.N, B = d.B;
  if (M <= 32) return launch_conv<32, 128, 32, 32, KT, S>(d, stream);
  if (S != 4) {
    const int64_t big = adp_cdiv(M, 128) * adp_cdiv(N, 128) * B;
    if (M >= 128 && big >= 384) return launch_conv<128, 128, 64, 64, KT, S>(d, stream);
  }
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

template <int KT, int S>
int launch_wgrad(const adp_wgrad_desc& d, void* stream) {
  int64_t PS, SPB;
  wgrad_split(d, WgradCfg<KT, S>::BKN, &PS, &SPB);
  const int64_t nsplit = d.B * SPB;
  dim3 grid((unsigned)nsplit, (unsigned)adp_cdiv(d.M, 32), (unsigned)adp_cdiv(d.R, 32));
  ADP_LAUNCH((wgrad_kernel<KT, S>), grid, dim3(256), stream, d, PS, SPB);
  const int64_t cnt = d.M * d.R * KT;
  const int64_t tot = cnt + (d.dbias ? d.M : 0);
  ADP_LAUNCH(wgrad_reduce_kernel, dim3((unsigned)adp_cdiv(tot, 256)), dim3(256), stream, (const float*)d.ws, nsplit,
             cnt, d.M, d.dw, d.dbias, (int)d.accumulate);
  return ADP_LAUNCH_OK();
}

bool ks_supported(int64_t KT, int64_t S) {
  return (KT == 1 && S == 1) || (KT == 2 && S == 2) || (KT == 3 && S == 1) || (KT == 4 && S == 4);
}

}  // namespace

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
  if (d.KT == 1) return dispatch_conv<1, 1>(d, stream);
  if (d.KT == 2) return dispatch_conv<2, 2>(d, stream);
  if (d.KT == 3) return dispatch_conv<3, 1>(d, stream);
  return dispatch_conv<4, 4>(d, stream);
}

// which tile the dispatcher picks for this problem: BM * 1000 + BN (introspection for profiling / roofline reports)
extern "C" int64_t adp_conv1d_tile(const adp_conv_desc* dp) {
  if (!dp) return ADP_ERR_NULL;
  const adp_conv_desc& d = *dp;
  if (d.M <= 32) return 32 * 1000 + 128;
  if (d.stride != 4) {
    const int64_t big = adp_cdiv(d.M, 128) * adp_cdiv(d.N, 128) * d.B;
    if (d.M >= 128 && big >= 384) return 128 * 1000 + 128;
  }
  return 64 * 1000 + 64;
}

extern "C" int64_t adp_conv1d_wgrad_ws_bytes(const adp_wgrad_desc* dp) {
  if (!dp || !ks_supported(dp->KT, dp->stride) || dp->B <= 0 || dp->N <= 0) return ADP_ERR_UNSUPPORTED;
  int64_t PS, SPB;
  wgrad_split(*dp, wgrad_bkn(dp->stride), &PS, &SPB);
  const int64_t nsplit = dp->B * SPB;
  return nsplit * (dp->M * dp->R * dp->KT + dp->M) * (int64_t)sizeof(float);
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
  if (d.KT == 1) return launch_wgrad<1, 1>(d, stream);
  if (d.KT == 2) return launch_wgrad<2, 2>(d, stream);
  if (d.KT == 3) return launch_wgrad<3, 1>(d, stream);
  return launch_wgrad<4, 4>(d, stream);
}
