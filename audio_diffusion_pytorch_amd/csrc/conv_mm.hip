// Dispatch of the implicit-GEMM Conv1d family (kernel: conv_mm_impl.h; one translation unit per block tile:
// conv_mm_m64.hip, conv_mm_m32.hip).
#include <stdlib.h>
#include "adp_rt.h"
#include "adp.h"
#include "conv_internal.h"

int adp_conv_mm_m64(const adp_conv_desc& d, void* stream);
int adp_conv_mm_m32(const adp_conv_desc& d, void* stream);

namespace {

constexpr int MM_BKT = 32;        // channels per staged chunk (16 for the stride-4 variant); R must divide by it
constexpr int MM_PRO_RMAX = 1024; // channels whose GroupNorm constants fit the LDS table

// 64 x 64 block tiles (8 MMA waves + 4 loaders) unless that leaves most of the 256 CUs without a block
// workgroups a launch should have before the larger tile is chosen (ADP_MM_MIN_BLOCKS: tests reach the large-tile
// variants with small problems through it)
static int64_t mm_min_blocks() {
  const char* e = getenv("ADP_MM_MIN_BLOCKS");
  return e ? atoll(e) : 200;
}

bool mm_use64(const adp_conv_desc& d) {
  if (d.M % 64 != 0) return false;
  // plain (no prologue) kernel-3 convs of the wide layers: two co-resident 32-row blocks per CU overlap one block's
  // first-load latency / K-group exchange / epilogue with the other's MFMAs (microbench at batch 4, 32- vs 64-row
  // tiles: C=256 42.7 vs 46.7 us, C=512 L=1024 68.9 vs 74.2, dgrad C=1024 L=256 61.1 vs 63.4); with the GroupNorm+SiLU
  // prologue the doubled activation recompute loses instead
  // (direct form only: the Winograd variant's blocks stage the same bytes for two thirds of the MFMAs, and the 64-row
  // block -- half the weight staging per flop -- is the faster one there: 14.37 -> 14.31 ms per step interleaved)
  if (!adp_conv_mm_winograd(d) && d.prologue == 0 && d.KT == 3 && d.stride == 1 && d.up == 1 && d.R >= 256 &&
      (d.M / 32) * adp_cdiv(d.N, 64) * d.B >= 512)
    return false;
  return (d.M / 64) * adp_cdiv(d.N, 64) * d.B >= mm_min_blocks();
}

// sum of the KS split-K partial tiles (fixed order) + the conv epilogue of store mode 0:
//   v = bias[m] + sum_ks ws[ks][b][m][n] ; out_pre = v ; out = e_scale[b,m] * v + res
__global__ __launch_bounds__(256) void conv_splitk_reduce_kernel(adp_conv_desc d, int KS) {
  const int64_t total = d.B * d.M * d.N;
  const int64_t ebs = d.e_bstride ? d.e_bstride : d.M;
  const bool vec = (d.N & 3) == 0 &&
                   ((reinterpret_cast<uintptr_t>(d.ws) | reinterpret_cast<uintptr_t>(d.out) |
                     reinterpret_cast<uintptr_t>(d.res) | reinterpret_cast<uintptr_t>(d.out_pre)) & 15) == 0;
  const int64_t step = vec ? 4 : 1;
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * step; i < total; i += (int64_t)gridDim.x * 256 * step) {
    const int64_t row = i / d.N, m = row % d.M, b = row / d.M;
    const float bias = d.bias ? d.bias[m] : 0.0f;
    const float sc = d.e_scale ? d.e_scale[b * ebs + m] : 1.0f;
    if (vec) {
      f32x4 v = *reinterpret_cast<const f32x4*>(d.ws + i);
      for (int k = 1; k < KS; ++k) {
        const f32x4 p = *reinterpret_cast<const f32x4*>(d.ws + k * total + i);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += p[e];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += bias;
      if (d.out_pre) *reinterpret_cast<f32x4*>(d.out_pre + i) = v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] *= sc;
      if (d.res) {
        const f32x4 r = *reinterpret_cast<const f32x4*>(d.res + i);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += r[e];
      }
      *reinterpret_cast<f32x4*>(d.out + i) = v;
    } else {
      float v = d.ws[i];
      for (int k = 1; k < KS; ++k) v += d.ws[k * total + i];
      v += bias;
      if (d.out_pre) d.out_pre[i] = v;
      v *= sc;
      if (d.res) v += d.res[i];
      d.out[i] = v;
    }
  }
}

// The same sum + epilogue for an output that feeds a GroupNorm: one workgroup per (batch element, 4-channel row quad,
// slice of SPLITK_GN_SLICE positions) keeps its <= 16 values per thread in registers and writes the quad's
// (mean, M2, count) entry of d.gn_part -- the layout the conv epilogues write, so adp_gn_finalize[_act] serves both
// and the consumer's statistics pass over the tensor disappears (batch 1: one launch less per deep ConvBlock).
constexpr int SPLITK_GN_SLICE = 1024;
__global__ __launch_bounds__(256) void conv_splitk_reduce_gn_kernel(adp_conv_desc d, int KS, int E) {
  __shared__ float sh[4];
  const int64_t M = d.M, N = d.N;
  int id = blockIdx.x;
  const int e = id % E;
  id /= E;
  const int64_t q = id % (M / 4), b = id / (M / 4);
  const int64_t n0 = (int64_t)e * SPLITK_GN_SLICE;
  const int cnt = (int)((N - n0) < SPLITK_GN_SLICE ? (N - n0) : SPLITK_GN_SLICE);
  const int64_t total = d.B * M * N;
  const int64_t ebs = d.e_bstride ? d.e_bstride : M;
  float v[16];
  float s = 0.0f;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int idx = (int)threadIdx.x + 256 * k;
    v[k] = 0.0f;
    if (idx < 4 * cnt) {
      const int64_t m = 4 * q + idx / cnt, i = (b * M + m) * N + n0 + idx % cnt;
      float a = d.ws[i];
      for (int j = 1; j < KS; ++j) a += d.ws[j * total + i];
      if (d.bias) a += d.bias[m];
      if (d.out_pre) d.out_pre[i] = a;
      if (d.e_scale) a *= d.e_scale[b * ebs + m];
      if (d.res) a += d.res[i];
      d.out[i] = a;
      v[k] = a;
      s += a;
    }
  }
  const float fcnt = 4.0f * (float)cnt;
  const float mean = adp_block_sum<4>(s, sh) / fcnt;
  float m2 = 0.0f;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const float dv = ((int)threadIdx.x + 256 * k < 4 * cnt) ? v[k] - mean : 0.0f;
    m2 = fmaf(dv, dv, m2);
  }
  m2 = adp_block_sum<4>(m2, sh);
  if (threadIdx.x == 0) {
    float* o = d.gn_part + (((int64_t)b * (M / 4) + q) * E + e) * 3;
    o[0] = mean;
    o[1] = m2;
    o[2] = fcnt;
  }
}

// The sum + epilogue for a data gradient that feeds the backward of SiLU(GroupNorm(gnb_x)) (adp_conv_desc.gnb_ab): one WAVE per
// (row, slice of SPLITK_GN_SLICE positions), 16 bytes per lane and trip; the wave leaves the row's (sum ds * xhat, sum ds) over
// the slice -- the first stage of that backward, which then needs no pass of its own (N % 4 == 0, 16-byte aligned tensors).
__global__ __launch_bounds__(256) void conv_splitk_reduce_gnb_kernel(adp_conv_desc d, int KS, int E) {
  const int64_t M = d.M, N = d.N;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int64_t id = (int64_t)blockIdx.x * 4 + wave;  // (b, m, e)
  if (id >= d.B * M * E) return;
  const int e = (int)(id % E);
  id /= E;
  const int64_t m = id % M, b = id / M;
  const int64_t n0 = (int64_t)e * SPLITK_GN_SLICE;
  const int cnt = (int)((N - n0) < SPLITK_GN_SLICE ? (N - n0) : SPLITK_GN_SLICE);
  const int64_t total = d.B * M * N, row = (b * M + m) * N + n0;
  const int64_t ebs = d.e_bstride ? d.e_bstride : M;
  const float bias = d.bias ? d.bias[m] : 0.0f, sc = d.e_scale ? d.e_scale[b * ebs + m] : 1.0f;
  const float* st = d.gnb_stats + (b * d.gnb_groups + m / (M / d.gnb_groups)) * 2;
  const float mu = st[0], rs = st[1];
  const float ga = d.gnb_gamma[m] * rs, be = d.gnb_beta[m] - mu * ga;
  float sa = 0.0f, sb = 0.0f;
  for (int p = 4 * lane; p < cnt; p += 256) {
    const int64_t i = row + p;
    f32x4 v = *reinterpret_cast<const f32x4*>(d.ws + i);
    const f32x4 xv = *reinterpret_cast<const f32x4*>(d.gnb_x + i);
    for (int k = 1; k < KS; ++k) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(d.ws + k * total + i);
#pragma unroll
      for (int c = 0; c < 4; ++c) v[c] += t[c];
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] += bias;
    if (d.out_pre) *reinterpret_cast<f32x4*>(d.out_pre + i) = v;
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] *= sc;
    if (d.res) {
      const f32x4 r = *reinterpret_cast<const f32x4*>(d.res + i);
#pragma unroll
      for (int c = 0; c < 4; ++c) v[c] += r[c];
    }
    *reinterpret_cast<f32x4*>(d.out + i) = v;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float xh = (xv[c] - mu) * rs;
      const float ds = v[c] * adp_dsilu_fast(fmaf(xv[c], ga, be));
      sa = fmaf(ds, xh, sa);
      sb += ds;
    }
  }
  sa = adp_wave_sum(sa), sb = adp_wave_sum(sb);
  if (lane == 0) *reinterpret_cast<f32x2*>(d.gnb_ab + ((b * M + m) * E + e) * 2) = f32x2{sa, sb};
}

}  // namespace

int64_t adp_conv_splitk_gn_entries(const adp_conv_desc& d) { return adp_cdiv(d.N, SPLITK_GN_SLICE); }

// slices per row of gnb_ab a conv_mm launch leaves: its 64-position tiles, or the split-K reduce kernel's slices
int64_t adp_conv_mm_gnb_entries(const adp_conv_desc& d) {
  // (the instantiations that exist: plain Winograd data gradients)
  if (d.store != 0 || !d.transposed || d.KT != 3 || d.prologue != 0 || d.up != 1 || d.stride != 1 || !adp_conv_mm_winograd(d)) return 0;
  if (!adp_gnb_family_on(d.ws && adp_conv_mm_ksplit(d) > 1 ? 32 : 8)) return 0;
  if (d.ws && adp_conv_mm_ksplit(d) > 1) {
    const bool vec = (d.N & 3) == 0 && ((reinterpret_cast<uintptr_t>(d.ws) | reinterpret_cast<uintptr_t>(d.out) |
                                         reinterpret_cast<uintptr_t>(d.res) | reinterpret_cast<uintptr_t>(d.out_pre) |
                                         reinterpret_cast<uintptr_t>(d.gnb_x)) & 15) == 0;
    return vec ? adp_cdiv(d.N, SPLITK_GN_SLICE) : 0;
  }
  return adp_cdiv(d.N, 64);  // (conv_mm_impl.h: MM_BN)
}

// Cross-workgroup K split: when the output tiles alone leave most of the 256 CUs idle (batch-1 deep layers: depth 8
// has 64 tiles of 32 x 64) the reduction over input channels is cut into 2 / 4 / 8 slices run by separate
// workgroups (>= 4 chunks of 32 channels each), combined by conv_splitk_reduce_kernel (deterministic order).
static int64_t env_or(const char* name, int64_t dflt) {
  const char* e = getenv(name);
  return e ? atoll(e) : dflt;
}

int64_t adp_conv_mm_ksplit(const adp_conv_desc& d) {
  static const bool off = getenv("ADP_MM_NO_KSPLIT") != nullptr;  // A/B switch for kernel work
  if (off || d.store != 0) return 1;  // pixel-shuffle / pooled stores keep their in-kernel epilogue
  const int64_t bm = mm_use64(d) ? 64 : 32;
  const int64_t blocks = (d.M / bm) * adp_cdiv(d.N, 64 * adp_conv_mm_nsp(d)) * d.B;
  const int64_t nchunks = d.R / (d.stride == 4 ? 16 : MM_BKT);
  int64_t ks = 1;
  // (the 200-workgroup target re-measured in round 3 with the Winograd variants, batch-1 step / sampler step in ms:
  //  130-200 -> 7.34 / 2.47, 300-400 -> 7.68 / 2.75, 520 -> 7.97 / 2.93, 100 -> 7.85 / 2.68, no split -> 8.16 / 2.82)
  // (batch 1, larger tiles + deeper splits instead, tools/b1_micro.py, forward us incl. the reduce launch: C=1024 L=256
  //  32-row x ks 2 23.5 | 64-row x ks 4 24.1 | 64 x 128 positions x ks 8 29.9 | 64 x 256 x ks 16 38.6; C=512 L=1024 18.2 | 29.0 |
  //  39.1 | 58.9: at batch 1 the small tile with the shallow split wins everywhere -- the knobs below are for that tool)
  const int64_t target = env_or("ADP_MM_KS_TARGET", 200), ksmax = env_or("ADP_MM_KS_MAX", 8);
  const int64_t minch = env_or("ADP_MM_KS_MINCH", 4);
  while (ks < ksmax && blocks * ks < target && nchunks / (ks * 2) >= minch) ks *= 2;
  // the kernel gives slice i the chunks [i * ceil(n / ks), ...): every slice must own at least one (n = 33, ks = 8 would
  // leave the last two slices empty -- they would launch, restage a ghost chunk and park an all-zero partial tile)
  while (ks > 1 && (ks - 1) * adp_cdiv(nchunks, ks) >= nchunks) ks /= 2;
  return ks;
}

// Winograd F(2,3) variant of conv_mm (WN, conv_mm_impl.h): kernel-3 'same' convs whose K loop is long enough to be
// matrix bound (also the UpsampleItem convs -- the LDS tile holds virtual upsampled positions -- and the pooled-store
// data gradients of those).  ADP_CONV_WINO (read per call): unset / "1" = this variant for every eligible conv with at
// least ADP_WINO_MIN_R (default 32) input channels; "0" = direct form everywhere (A/B and the parity tests).
bool adp_winograd_enabled() {
  const char* e = getenv("ADP_CONV_WINO");
  return e == nullptr || e[0] != '0';
}

bool adp_conv_mm_winograd(const adp_conv_desc& d) {
  if (!adp_winograd_enabled()) return false;
  if (d.KT != 3 || d.stride != 1 || d.dil != 1 || d.pad != 1 || d.R1 != d.R) return false;
  if (d.up != 1 && d.up != 2 && d.up != 4) return false;
  if (d.store != 0 && !(d.store == 2 && (d.sp == 2 || d.sp == 4))) return false;  // plain or pooled store
  if (d.N != d.Lin * d.up || d.N % 4 != 0) return false;
  const char* mr = getenv("ADP_WINO_MIN_R");
  const int64_t min_r = mr ? atoll(mr) : 32;  // (round 4: 64 -> 32 is worth 0.03 ms per step at batch 4, 0.015 at batch 1)
  if (d.R < min_r) return false;
  if ((reinterpret_cast<uintptr_t>(d.out) | reinterpret_cast<uintptr_t>(d.res) | reinterpret_cast<uintptr_t>(d.out_pre) |
       reinterpret_cast<uintptr_t>(d.ws)) & 7)
    return false;  // 8-byte accesses to the output pair
  return true;
}

// Wide-N blocks of the Winograd variant and of the 1x1 convs (NSP, conv_mm_impl.h): 64 rows x 256 positions, 128 when that leaves fewer than 200
// workgroups, else the 64-position block.  ADP_MM_NSP caps it (1 = the 64-position block everywhere: A/B and tests).
// Isolated launches at batch 4, 64 -> 128 / 256 positions, us (tools/nsp_micro.py): C=64 L=16384 conv1 (GroupNorm+SiLU
// prologue) 33.7 -> 27.2 / 26.7, data gradient 27.6 -> 21.7 / 19.3; C=128 L=4096 19.9 -> 16.4; C=256 L=2048 30.1 -> 26.5;
// C=512 L=1024 50.5 -> 45.5 (142 TF in direct-form flops).  Batch 1 keeps the 64-position block (grid too small).
int adp_conv_mm_nsp(const adp_conv_desc& d) {
  if (!(adp_conv_mm_winograd(d) || (d.KT == 1 && d.up == 1)) || !mm_use64(d) || d.stride != 1) return 1;
  const char* e = getenv("ADP_MM_NSP");
  int want = e ? atoi(e) : 4;
  // the 1x1 convs are short of work per byte, not of weight reuse (tools/nsp_micro2.py, batch 4, 64 / 128 / 256 positions:
  // 64 -> 128 channels at L = 16384 40.4 / 36.4 / 49.9 us, 256 -> 256 at L = 2048 20.0 / 22.3 / 22.2): 128 positions up to
  // 128 input channels, the 64-position block above; the upsample convs gain like the plain ones (512 -> 256 x2: 49.8 -> 43.7)
  if (d.KT == 1 && want > 1) want = d.R <= 128 ? 2 : 1;
  const int64_t nsp_min = env_or("ADP_MM_NSP_MIN_BLOCKS", mm_min_blocks());
  while (want > 1 && (d.M / 64) * adp_cdiv(d.N, 64 * want) * d.B < nsp_min) want /= 2;
  return want < 1 ? 1 : want;
}

bool adp_conv_mm_eligible(const adp_conv_desc& d) {
  if (d.R1 != d.R) return false;
  const bool plain = d.stride == 1 && (d.KT == 1 || d.KT == 3) && d.up == 1;                    // ConvBlock family
  const bool upc = d.stride == 1 && d.KT == 3 && (d.up == 2 || d.up == 4) && !d.transposed && d.prologue == 0;
  const bool down = (d.stride == 2 || d.stride == 4) && d.KT == d.stride && d.up == 1 && d.pad == 0 && d.dil == 1 &&
                    !d.transposed && d.prologue == 0;
  if (!plain && !upc && !down) return false;
  if (d.prologue != 0 && d.prologue != 1) return false;
  if (d.prologue == 1 && d.R > MM_PRO_RMAX) return false;
  if (d.R % MM_BKT != 0 || d.M % 32 != 0 || (d.Lin * d.up) % 4 != 0) return false;
  if (d.pad > 4 || (d.KT - 1) * d.dil - d.pad > 4) return false;  // halo of 4 positions on both sides of the X tile
  if ((d.N - 1) * d.stride + (d.KT - 1) * d.dil - d.pad >= d.Lin * d.up + 4) return false;
  if ((reinterpret_cast<uintptr_t>(d.x) | reinterpret_cast<uintptr_t>(d.w)) & 15) return false;
  if (d.B * d.R * d.Lin >= (int64_t)1 << 31 || d.M * d.R * d.KT >= (int64_t)1 << 31) return false;
  return true;
}

// NKG * 1e6 + BM * 1e3 + BN of the tile the dispatcher picks
int64_t adp_conv_mm_tile(const adp_conv_desc& d) {
  const int64_t nkg = d.stride == 4 ? 2 : 4;
  const int64_t nsp = adp_conv_mm_nsp(d);
  return (adp_conv_mm_winograd(d) ? 40000000 : 0) + (nkg / nsp) * 1000000 + (mm_use64(d) ? 64000 : 32000) + 64 * nsp;
}

int adp_conv_splitk_reduce(const adp_conv_desc& d, int64_t ks, void* stream) {
  if (d.gnb_ab != nullptr) {  // (never together with gn_part: a data gradient feeds no GroupNorm forward)
    const int64_t E = adp_cdiv(d.N, SPLITK_GN_SLICE);
    ADP_LAUNCH(conv_splitk_reduce_gnb_kernel, dim3((unsigned)adp_cdiv(d.B * d.M * E, 4)), dim3(256), stream, d, (int)ks, (int)E);
    return ADP_LAUNCH_OK();
  }
  if (d.gn_part != nullptr && d.M % 4 == 0) {
    const int64_t E = adp_conv_splitk_gn_entries(d);
    ADP_LAUNCH(conv_splitk_reduce_gn_kernel, dim3((unsigned)(d.B * (d.M / 4) * E)), dim3(256), stream, d, (int)ks, (int)E);
    return ADP_LAUNCH_OK();
  }
  const int64_t total = d.B * d.M * d.N;
  int64_t g = adp_cdiv(total, 1024);
  if (g > 2048) g = 2048;
  ADP_LAUNCH(conv_splitk_reduce_kernel, dim3((unsigned)g), dim3(256), stream, d, (int)ks);
  return ADP_LAUNCH_OK();
}

int adp_conv_mm(const adp_conv_desc& d, void* stream) {
  const int rc = mm_use64(d) ? adp_conv_mm_m64(d, stream) : adp_conv_mm_m32(d, stream);
  const int64_t ks = d.ws ? adp_conv_mm_ksplit(d) : 1;
  if (rc == ADP_OK && ks > 1) return adp_conv_splitk_reduce(d, ks, stream);
  return rc;
}
