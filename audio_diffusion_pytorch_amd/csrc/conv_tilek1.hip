// Wave-tile 1x1 convolution with the K split INSIDE the workgroup, for the attention items' projections (to_q / to_out and their
// data gradients: kernel 1, 512-1024 channels over 128-512 positions at batch 1; /root/reference/audio_diffusion_pytorch/
// components.py:92-93 via a_unet's AttentionItem / CrossAttentionItem, BASELINE configs[3]) -- the 1x1 sibling of conv_tilek.hip.
//
// conv_mm gives such a projection 32-128 blocks and fills the chip with a cross-workgroup K split: partial tiles through HBM and a
// second launch that sums them (80 reduce launches per config-4 step).  Here a workgroup owns one 16 x 64 output tile and its
// eight waves split the input channels (R / 8 each), every wave on its own 16-channel chunks:
//   * the x tile [16][64] is fetched with coalesced 16-byte loads one chunk pair ahead and parked in a wave-PRIVATE LDS region;
//   * K index kq of step ks is channel 4 kq + ks of the chunk (any bijection works when both operands use it): the lane's four
//     weights of a chunk are then ONE 16-byte load of row j (forward view; the transposed view takes four 4-byte loads of
//     64 contiguous bytes per K index), and its B operands of a step are one 16-byte LDS read -- position 4 j + t for tile t, so the
//     four accumulator tiles hold an output quad per lane and row;
//   * v_mfma_f32_16x16x4_f32, 16 per chunk; no workgroup barrier in the K loop;
//   * the eight partial tiles meet in LDS once, summed in wave order by the four waves that also add bias / residual, store
//     16 bytes per lane and leave the GroupNorm partial statistics of the output (conv_tilek's entry format).
// fp32 throughout.  Algorithmic bytes per launch: 4 * (B * R * L + B * M * L (+ residual) + M * R).
#include <stdlib.h>
#include "adp_rt.h"
#include "adp.h"
#include "conv_internal.h"

namespace {

constexpr int T1_TN = 64;          // positions per tile (16 output quads)
constexpr int T1_CH = 16;          // channels per chunk (four K steps of v_mfma_f32_16x16x4_f32)
constexpr int T1_NKW = 8;          // waves per workgroup = K slices
constexpr int T1_XF = T1_CH * T1_TN;  // floats of a wave's region: the x tile, later its partial tile [4][64] float4

// SH: pixel-shuffle store (store mode 1: the DownsampleItem data gradient as a 1x1 conv over the space-to-depth view) --
//     out[b][m / sp][n * sp + m % sp] (+ residual there); its own instantiation
template <bool TR, int NKW = T1_NKW, bool SH = false>
__global__ __launch_bounds__(64 * NKW) void conv_tilek1_kernel(adp_conv_desc d, int ntn) {
  const bool RES = d.res != nullptr, GN = d.gn_part != nullptr;  // (workgroup-uniform)
  __shared__ __attribute__((aligned(16))) float lds[NKW * T1_XF + 64];
  float* const gsh = lds + NKW * T1_XF;  // statistics scratch [wave < 4][kq][2]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = adp_uniform(tid >> 6);
  const int j = lane & 15, kq = lane >> 4;  // MFMA 16x16x4: lane = (row / column j, K index kq)
  const int M = (int)d.M, R = (int)d.R, L = (int)d.Lin;

  // ---- XCD-aware decode of the 1-D grid: an XCD gets a contiguous range of row tiles (its L2 keeps their weight slabs)
  int id = blockIdx.x;
  const int total = gridDim.x;
  if ((total & 7) == 0) id = (id & 7) * (total >> 3) + (id >> 3);
  const int per_m = ntn * (int)d.B;
  const int mt = id / per_m, rem = id - mt * per_m;
  const int b = rem / ntn, nt = rem - b * ntn;
  const int m0 = mt * 16, n0 = nt * T1_TN;

  const int kc = R / NKW;  // this wave's channels [c_lo, c_lo + kc)
  const int c_lo = wave * kc, nchunks = kc / T1_CH;
  float* const X = lds + wave * T1_XF;

  // ---- chunk loads (registers, one chunk pair ahead): x tile 16 rows x 16 quads = 4 float4 per lane; weights: 4 floats
  const float* xb = d.x + ((int64_t)b * R + c_lo) * L + n0;
  const int xoff = (lane >> 4) * L + 4 * (lane & 15);  // + 4 i rows
  // forward view w[m][r]: row m0 + j, channels c0 + 4 kq .. + 3 (one float4); transposed view w[r][m]: channels c0 + 4 kq + ks
  const float* wb = TR ? d.w + (int64_t)(c_lo + 4 * kq) * M + m0 + j : d.w + (int64_t)(m0 + j) * R + c_lo + 4 * kq;
  struct ChunkRegs {
    f32x4 x[4];
    f32x4 w;
  };
  auto load_chunk = [&](ChunkRegs& c, int chunk) {
    const float* xp = xb + (int64_t)chunk * T1_CH * L;
#pragma unroll
    for (int i = 0; i < 4; ++i) c.x[i] = *reinterpret_cast<const f32x4*>(xp + xoff + 4 * i * L);
    if (TR) {
      const float* wp = wb + (int64_t)chunk * T1_CH * M;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) c.w[ks] = wp[ks * M];
    } else {
      c.w = *reinterpret_cast<const f32x4*>(wb + chunk * T1_CH);
    }
  };

  // epilogue operands of the finishing waves: requested before the K loop
  const bool fin = wave < 4;
  const int fr = wave & 3;
  const int fch = m0 + 4 * kq + fr;  // this lane's output channel when its wave finishes
  const int sp = SH ? (int)d.sp : 1;
  // SH: element k of the lane's quad goes to foff + k * sp (row fch / sp of a [B, M / sp, L * sp] tensor, phase fch % sp)
  const int64_t foff = SH ? ((int64_t)b * (M / sp) + fch / sp) * ((int64_t)L * sp) + (int64_t)(n0 + 4 * j) * sp + fch % sp
                          : ((int64_t)b * M + fch) * L + n0 + 4 * j;
  f32x4 rv = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  float bv = 0.0f;
  if (fin) {
    if (RES) {
      if (SH) {
#pragma unroll
        for (int k = 0; k < 4; ++k) rv[k] = d.res[foff + k * sp];
      } else {
        rv = *reinterpret_cast<const f32x4*>(d.res + foff);
      }
    }
    if (d.bias) bv = d.bias[fch];
  }

  f32x4 acc[4];  // tile t: rows 4 kq + r, column j <-> position n0 + 4 j + t
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

  const float* Xr = X + 4 * kq * T1_TN + 4 * j;  // + ks rows: channel 4 kq + ks of the chunk

  auto run_chunk = [&](ChunkRegs& cr, int next) {
    {  // park the x tile
      float* o = X + (lane >> 4) * T1_TN + 4 * (lane & 15);
#pragma unroll
      for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(o + 4 * i * T1_TN) = cr.x[i];
    }
    const f32x4 wv = cr.w;
    adp_wave_sync();
    // this set's next chunk, in flight under two chunks of MFMAs.  UNCONDITIONAL (past the end the last chunk is fetched again and
    // never used): a branch around the loads makes the compiler wait for what it has just requested (conv_tilek.hip)
    load_chunk(cr, next < nchunks ? next : nchunks - 1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const f32x4 bq = *reinterpret_cast<const f32x4*>(Xr + ks * T1_TN);
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = adp_mfma16(wv[ks], bq[t], acc[t]);
    }
    adp_wave_sync();  // the next chunk overwrites this wave's tile
  };
  ChunkRegs cs[2];
  load_chunk(cs[0], 0);
  load_chunk(cs[1], nchunks > 1 ? 1 : 0);
  for (int chunk = 0; chunk < nchunks; chunk += 2) {  // (nchunks is even: eligibility)
    run_chunk(cs[0], chunk + 2);
    run_chunk(cs[1], chunk + 3);
  }

  // ---- the wave's partial tile goes to the head of its own region: [r][lane] float4 = the lane's quad of output row 4 kq + r
#pragma unroll
  for (int r = 0; r < 4; ++r)
    *reinterpret_cast<f32x4*>(X + (r * 64 + lane) * 4) = f32x4{acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
  __syncthreads();
  float gmean = 0.0f, gm2 = 0.0f;
  if (fin) {
    f32x4 y = *reinterpret_cast<const f32x4*>(lds + (fr * 64 + lane) * 4);
#pragma unroll
    for (int w = 1; w < NKW; ++w) {  // wave order: deterministic
      const f32x4 t = *reinterpret_cast<const f32x4*>(lds + w * T1_XF + (fr * 64 + lane) * 4);
#pragma unroll
      for (int k = 0; k < 4; ++k) y[k] += t[k];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) y[k] += bv;
    if (RES) {
#pragma unroll
      for (int k = 0; k < 4; ++k) y[k] += rv[k];
    }
    if (SH) {
#pragma unroll
      for (int k = 0; k < 4; ++k) d.out[foff + k * sp] = y[k];
    } else {
      *reinterpret_cast<f32x4*>(d.out + foff) = y;
    }
    if (GN && !SH) {
      // (mean, M2) of this wave's 64 positions of channel fch, shifted by the row's first value
      const float gk = __shfl(y[0], lane & 48, 64);
      float s = 0.0f, q = 0.0f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float e = y[k] - gk;
        s += e;
        q = fmaf(e, e, q);
      }
      const float sv = adp_row16_sum(s), qv = adp_row16_sum(q);
      gmean = gk + sv / (float)T1_TN;
      gm2 = fmaxf(qv - sv * (sv / (float)T1_TN), 0.0f);
      if (j == 0) {
        gsh[(wave * 4 + kq) * 2] = gmean;
        gsh[(wave * 4 + kq) * 2 + 1] = gm2;
      }
    }
  }
  if (GN && !SH) {
    __syncthreads();
    if (tid < 4) {
      // row quad tid = kq: its four channels were finished by waves r = 0..3 (Chan's pairwise update)
      const int qk = tid;
      constexpr float cnt = (float)T1_TN;
      float mean = gsh[qk * 2], m2 = gsh[qk * 2 + 1], n = cnt;
      for (int r = 1; r < 4; ++r) {
        const float mw = gsh[(r * 4 + qk) * 2], dl = mw - mean, nn = n + cnt;
        mean += dl * (cnt / nn);
        m2 += gsh[(r * 4 + qk) * 2 + 1] + dl * dl * (n * cnt / nn);
        n = nn;
      }
      float* e = d.gn_part + (((int64_t)b * (M / 4) + m0 / 4 + tid) * ntn + nt) * 3;
      e[0] = mean;
      e[1] = m2;
      e[2] = n;
    }
  }
}

int64_t tilek1_tiles(const adp_conv_desc& d) { return (d.M / 16) * d.B * (d.N / T1_TN); }

}  // namespace

// Taken where 16-row tiles fill the chip at most about twice (conv_mm splits K across workgroups below ~200 of its blocks): ADP_CONV_TILEK1=0 switches it off (A/B, tests), ADP_TILEK1_MIN_R / _MAX_TILES / _MIN_TILES move the window.
bool adp_conv_tilek1_eligible(const adp_conv_desc& d) {
  const char* e = getenv("ADP_CONV_TILEK1");
  if (e && e[0] == '0') return false;
  if (d.KT != 1 || d.stride != 1 || d.dil != 1 || d.pad != 0 || d.up != 1) return false;
  if (d.store != 0 && !(d.store == 1 && d.transposed && !d.gn_part && d.sp >= 1 && d.M % d.sp == 0)) return false;  // (SH instantiation)
  if (d.prologue != 0 || d.x2 || d.R1 != d.R || d.out_pre || d.e_scale || d.gnb_ab) return false;
  const char* mr = getenv("ADP_TILEK1_MIN_R");
  const char* xr = getenv("ADP_TILEK1_MAX_R");  // (2048 channels = 16 serial chunks per wave: conv_mm's split wins, 15.2 vs 18.3 us)
  if (d.R < (mr ? atoll(mr) : 256) || d.R > (xr ? atoll(xr) : 1024) || d.R % (2 * T1_NKW * T1_CH) != 0 || d.M % 16 != 0) return false;
  if (d.gn_part && d.M % 4 != 0) return false;
  if (d.N != d.Lin || d.N % T1_TN != 0) return false;
  const char* mt = getenv("ADP_TILEK1_MAX_TILES");
  const char* mn = getenv("ADP_TILEK1_MIN_TILES");
  const int64_t tiles = tilek1_tiles(d);
  if (tiles > (mt ? atoll(mt) : 512) || tiles < (mn ? atoll(mn) : 32)) return false;
  // hipGraph microbench (tools/tilek1_micro.py), us per launch, conv_mm (incl. its reduce launch) -> this kernel, forward + residual /
  // data gradient: [1,512->512,512] 11.1 / 10.7 -> 6.3 / 6.3; [1,1024->512,128] 10.6 / 9.4 -> 8.8 / 6.3; [1,512->1024,256] 10.9 /
  // 11.0 -> 6.1 / 9.2; [1,512->512,1024] (no K split in conv_mm) 14.9 / 12.2 -> 10.4 / 10.1; [4,1024->512,128] 14.9 / 11.8 -> 9.3 / 9.9;
  // beyond 512 tiles: [4,512->512,512] 20.0 / 17.2 -> 19.2 / 19.7, [1,256->512,2048] 14.2 / 12.3 -> 12.3 / 12.1: left to conv_mm
  const char* mb = getenv("ADP_TILEK1_MM_BLOCKS");  // (A/B: only where conv_mm has fewer blocks than this)
  if (mb && (d.M / 32) * (d.N / 64) * d.B >= atoll(mb)) return false;
  if ((reinterpret_cast<uintptr_t>(d.x) | reinterpret_cast<uintptr_t>(d.w) | reinterpret_cast<uintptr_t>(d.out) |
       reinterpret_cast<uintptr_t>(d.res)) & 15)
    return false;
  if (!d.transposed && (d.R & 3)) return false;
  if (d.B * d.R * d.Lin >= (int64_t)1 << 31 || d.M * d.R >= (int64_t)1 << 31 || d.B * d.M * d.N >= (int64_t)1 << 40) return false;
  return true;
}

int64_t adp_conv_tilek1_gn_entries(const adp_conv_desc& d) { return d.N / T1_TN; }

int adp_conv_tilek1(const adp_conv_desc& d, void* stream) {
  const int ntn = (int)(d.N / T1_TN);
  const unsigned grid = (unsigned)((d.M / 16) * d.B * ntn);
  // ADP_TILEK1_NKW=16 (A/B, tests): sixteen K slices, four waves per SIMD.  Measured 0.5-0.8 us SLOWER per launch on every shape
  // ([1,1024->512,128] 8.8 -> 9.6 us, [1,512->512,512] 6.3 -> 6.8): the K loop is not what these launches wait for.  Default: eight.
  const char* e = getenv("ADP_TILEK1_NKW");
  const bool w16 = e && atoi(e) == 16 && d.R % (2 * 16 * T1_CH) == 0 && d.store == 0;
  if (w16) {
    if (d.transposed) ADP_LAUNCH((conv_tilek1_kernel<true, 16>), dim3(grid), dim3(64 * 16), stream, d, ntn);
    else ADP_LAUNCH((conv_tilek1_kernel<false, 16>), dim3(grid), dim3(64 * 16), stream, d, ntn);
    return ADP_LAUNCH_OK();
  }
  if (d.store == 1) ADP_LAUNCH((conv_tilek1_kernel<true, T1_NKW, true>), dim3(grid), dim3(64 * T1_NKW), stream, d, ntn);
  else if (d.transposed) ADP_LAUNCH((conv_tilek1_kernel<true>), dim3(grid), dim3(64 * T1_NKW), stream, d, ntn);
  else ADP_LAUNCH((conv_tilek1_kernel<false>), dim3(grid), dim3(64 * T1_NKW), stream, d, ntn);
  return ADP_LAUNCH_OK();
}
