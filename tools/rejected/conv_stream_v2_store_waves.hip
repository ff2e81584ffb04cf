// Streaming ConvBlock convolution for the HBM-bound depth-1 layers: 32 -> 32 channels, kernel 3, stride 1, 'same'
// (ResnetItem ConvBlocks at channels = 32 and their data gradients; /root/reference/audio_diffusion_pytorch/
// components.py:89, SURVEY.md 8a row a13, 8d "HBM-bound" rows).
//
// At [4, 32, 65536] a ConvBlock conv moves 67-100 MB (A_in + A_out (+A_res)) for 1.6 GFLOP: arithmetic intensity
// ~20 flop/B, right at the MI355X ridge, so the kernel is built like a stream with the matrix cores fed from it:
//   * the whole 32 x 96 weight matrix lives in REGISTERS as MFMA A-operands (48 VGPRs per lane) for the lifetime of
//     a persistent workgroup -- there is no K loop, no K-group exchange, no weight traffic after start-up;
//   * one workgroup per CU walks a contiguous range of 256-position tiles with THREE wave roles on two wave sets:
//       4 MMA waves (one per SIMD, 64 positions each, two accumulator tiles): per tile 96 MFMAs over the staged input,
//         then the finished tile goes to LDS -- they never touch global memory inside the loop;
//       4 load/store waves: stage tile it+1 (16-byte global loads two tiles ahead, GroupNorm+SiLU in registers, 16-byte
//         LDS stores), and DRAIN tile it-1 from LDS to global with 16-byte stores, adding the residual they prefetched.
//     Round 1 let the MMA waves store their own accumulators (64 4-byte store instructions per wave and tile): the
//     wave sat in the store ISSUE queue for as long as the tile's write bandwidth takes and could not start the next
//     tile's MFMAs -- elimination builds showed 16.7 us of streaming and 12.5 us of MFMA adding up to 29.2 us instead
//     of overlapping (tools/rejected/README.md).  With the stores on the other wave set the two overlap.
//   * one workgroup barrier per tile hands the double-buffered input and output tiles over; the barrier orders LDS
//     traffic only (adp_barrier_lds), global loads and stores stay in flight across it.
// Algorithmic bytes per launch: 4 * B * 32 * L * (2 + has_res) + 12 KB of weights.
#include "adp_rt.h"
#include "adp.h"
#include "conv_internal.h"

namespace {

constexpr int ST_C = 32;           // channels in = channels out
constexpr int ST_KT = 3;
constexpr int ST_TN = 256;         // positions per workgroup iteration (4 MMA waves x 64)
constexpr int ST_XS = ST_TN + 8;   // LDS row stride of the input tile: positions n0-4 .. n0+TN+3
constexpr int ST_XQ = ST_XS / 4;
constexpr int ST_NX4 = (ST_C * ST_XQ + 255) / 256;  // staging quads per load/store thread
constexpr int ST_YS = ST_TN + 4;   // LDS row stride of the output tile (16-byte aligned rows)
constexpr int ST_IN = 2 * ST_C * ST_XS, ST_OUT = 2 * ST_C * ST_YS;

template <bool TR, int PRO>
__global__ __launch_bounds__(512) void conv_stream32_kernel(adp_conv_desc d, int tiles_per_b, int wpb) {
  __shared__ __attribute__((aligned(16))) float smem[ST_IN + ST_OUT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hi = lane >> 5, l31 = lane & 31;
  const int L = (int)d.Lin;
  // contiguous tile range of this workgroup, inside ONE batch element (wpb workgroups per batch element): the
  // GroupNorm partial statistics of the output are then one entry per workgroup and row quad of that element
  const int wb = blockIdx.x / wpb, wi = blockIdx.x - wb * wpb;
  const int t_beg = wb * tiles_per_b + (int)(((int64_t)wi * tiles_per_b) / wpb);
  const int t_end = wb * tiles_per_b + (int)(((int64_t)(wi + 1) * tiles_per_b) / wpb);
  const int niter = t_end - t_beg;
  const int nrounds = (niter + 1) & ~1;  // two tiles per loop trip (static register stages); a ghost pads odd counts
  float* Yall = smem + ST_IN;

  if (wave >= 4) {
    // =========================== load / store waves ===========================
    const int lt = tid - 256, lw = wave - 4;
    int x_dst[ST_NX4], x_off[ST_NX4], x_pos[ST_NX4];
    float x_ga[ST_NX4], x_be[ST_NX4];
    int x_st[ST_NX4];
#pragma unroll
    for (int i = 0; i < ST_NX4; ++i) {
      const int e = (lt + i * 256) % (ST_C * ST_XQ);
      const int row = e / ST_XQ, pq = e - row * ST_XQ;
      x_dst[i] = row * ST_XS + 4 * pq;
      x_off[i] = row * L;
      x_pos[i] = 4 * pq - 4;
      if (PRO == 1) {
        x_st[i] = (row / (ST_C / (int)d.groups)) * 2;
        x_ga[i] = d.pro_gamma ? d.pro_gamma[row] : 1.0f;
        x_be[i] = d.pro_beta ? d.pro_beta[row] : 0.0f;
      }
    }
    // two register stages: tile it+2 is requested while tile it+1 is being staged
    f32x4 rxs[2][ST_NX4];
    float rms[2][ST_NX4], rrs[2][ST_NX4];
    bool oks[2][ST_NX4];
    auto load_tile = [&](f32x4 (&rx)[ST_NX4], float (&rm)[ST_NX4], float (&rr)[ST_NX4], bool (&ok)[ST_NX4], int it) {
      const int t = t_beg + (it < niter ? it : niter - 1);
      const int n0 = (t - wb * tiles_per_b) * ST_TN;
      const float* xb = d.x + (int64_t)wb * ST_C * L;
#pragma unroll
      for (int i = 0; i < ST_NX4; ++i) {
        const int u = n0 + x_pos[i];
        ok[i] = (u >= 0 && u < L);  // L % 4 == 0: a quad is entirely inside or outside the row
        rx[i] = *reinterpret_cast<const f32x4*>(xb + x_off[i] + (ok[i] ? u : 0));
        if (PRO == 1) {
          rm[i] = d.pro_stats[(int64_t)wb * d.groups * 2 + x_st[i]];
          rr[i] = d.pro_stats[(int64_t)wb * d.groups * 2 + x_st[i] + 1];
        }
      }
    };
    auto store_tile = [&](const f32x4 (&rx)[ST_NX4], const float (&rm)[ST_NX4], const float (&rr)[ST_NX4],
                          const bool (&ok)[ST_NX4], int it) {
      float* Xb = smem + (it & 1) * (ST_C * ST_XS);
#pragma unroll
      for (int i = 0; i < ST_NX4; ++i) {
        f32x4 v = rx[i];
        if (PRO == 1) {
          const float pa = x_ga[i] * rr[i], pb = x_be[i] - rm[i] * pa;
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = adp_silu_fast(fmaf(v[j], pa, pb));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = ok[i] ? v[j] : 0.0f;  // zero padding is applied after the activation
        *reinterpret_cast<f32x4*>(Xb + x_dst[i]) = v;
      }
    };
    // output side: this wave drains rows 8 lw .. 8 lw + 7 (two row quads) of a tile, lane = one 4-position quad
    const bool has_res = d.res != nullptr;
    const bool want_gn = d.gn_part != nullptr;
    f32x4 rres[8];
    float gs[2] = {0.0f, 0.0f}, gq[2] = {0.0f, 0.0f};  // running sum / sum of squares of the two row quads
    auto out_off = [&](int it, int k) {
      const int t = t_beg + it;
      return ((int64_t)wb * ST_C + 8 * lw + k) * L + (int64_t)(t - wb * tiles_per_b) * ST_TN + 4 * lane;
    };
    auto prefetch_res = [&](int it) {
      if (!has_res || it >= niter) return;
#pragma unroll
      for (int k = 0; k < 8; ++k) rres[k] = *reinterpret_cast<const f32x4*>(d.res + out_off(it, k));
    };
    auto drain = [&](int it) {  // tile `it` sits in the output buffer it & 1 (written by the MMA waves before B_{it+1})
      const float* Yb = Yall + (it & 1) * (ST_C * ST_YS);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        f32x4 v = *reinterpret_cast<const f32x4*>(Yb + (8 * lw + k) * ST_YS + 4 * lane);
        if (has_res) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] += rres[k][j];
        }
        *reinterpret_cast<f32x4*>(d.out + out_off(it, k)) = v;
        if (want_gn) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            gs[k >> 2] += v[j];
            gq[k >> 2] = fmaf(v[j], v[j], gq[k >> 2]);
          }
        }
      }
    };
    // Timeline: the MMA waves compute tile it between B_it and B_{it+1}.  Between B_{it-1} and B_it this wave
    //   drains tile it-2 (complete since B_{it-1}; its output buffer is rewritten for tile it only after B_it),
    //   stages the input of tile it (buffer it & 1, last read by the MFMAs of tile it-2), requests tile it+2, and
    //   prefetches the residual of tile it-1 (consumed by the next iteration's drain).
    load_tile(rxs[0], rms[0], rrs[0], oks[0], 0);
    load_tile(rxs[1], rms[1], rrs[1], oks[1], 1);
    for (int it0 = 0; it0 < nrounds; it0 += 2) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int it = it0 + s;
        if (it >= 2 && it - 2 < niter) drain(it - 2);
        store_tile(rxs[s], rms[s], rrs[s], oks[s], it);          // ghost tile (odd niter): restages the last one
        load_tile(rxs[s], rms[s], rrs[s], oks[s], it + 2);       // unconditional, clamped: never consumed
        if (it >= 1) prefetch_res(it - 1);
        adp_barrier_lds();                                       // B_it
      }
    }
    if (nrounds - 2 < niter) drain(nrounds - 2);                 // complete since B_{nrounds-1}
    adp_barrier_lds();                                           // B_nrounds: the last tile is in its output buffer
    if (nrounds - 1 < niter) {
      prefetch_res(nrounds - 1);
      drain(nrounds - 1);
    }
    if (want_gn) {
      // one (mean, M2, count) entry per row quad for the 4 x 256 x niter values this wave wrote
      const float cnt = 4.0f * (float)ST_TN * (float)niter;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const float sv = adp_wave_sum(gs[q]), qv = adp_wave_sum(gq[q]);
        if (lane == 0) {
          const float mean = sv / cnt;
          float* e = d.gn_part + (((int64_t)wb * (ST_C / 4) + 2 * lw + q) * wpb + wi) * 3;
          e[0] = mean;
          e[1] = fmaxf(qv - sv * mean, 0.0f);
          e[2] = cnt;
        }
      }
    }
    return;
  }

  // =========================== MMA waves ===========================
  // A operands: av[g][c*KT + t] = A(m = l31, channel 8g + c + 4hi, tap t)
  float av[4][4 * ST_KT];
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int t = 0; t < ST_KT; ++t) {
        const int r = 8 * g + c + 4 * hi;
        av[g][c * ST_KT + t] = TR ? d.w[((int64_t)r * ST_C + l31) * ST_KT + (ST_KT - 1 - t)]
                                  : d.w[((int64_t)l31 * ST_C + r) * ST_KT + t];
      }
  float bias[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) bias[r] = d.bias ? d.bias[(r & 3) + 8 * (r >> 2) + 4 * hi] : 0.0f;
  const int xfrag = 4 * hi * ST_XS + 64 * wave + l31 + 4 - 1;  // + ni*32 + (8g + c) * XS + t
  const int yfrag = 4 * hi * ST_YS + 64 * wave + l31;          // + ((r & 3) + 8 (r >> 2)) * YS + ni*32
#ifndef ADP_EMULATE
  // Make the weight / bias loads complete HERE (otherwise the compiler covers their first use inside the loop with a
  // vmcnt(0) on every iteration).
#pragma unroll
  for (int g = 0; g < 4; ++g)
    asm volatile("" ::"v"(av[g][0]), "v"(av[g][1]), "v"(av[g][2]), "v"(av[g][3]), "v"(av[g][4]), "v"(av[g][5]),
                 "v"(av[g][6]), "v"(av[g][7]), "v"(av[g][8]), "v"(av[g][9]), "v"(av[g][10]), "v"(av[g][11]));
  asm volatile("" ::"v"(bias[0]), "v"(bias[1]), "v"(bias[2]), "v"(bias[3]), "v"(bias[4]), "v"(bias[5]), "v"(bias[6]),
               "v"(bias[7]), "v"(bias[8]), "v"(bias[9]), "v"(bias[10]), "v"(bias[11]), "v"(bias[12]), "v"(bias[13]),
               "v"(bias[14]), "v"(bias[15]));
#endif

  for (int it = 0; it < nrounds; ++it) {
    adp_barrier_lds();  // B_it: tile it is in the input buffer it & 1
    if (it >= niter) continue;  // ghost iteration: only the barrier
    const float* Xb = smem + (it & 1) * (ST_C * ST_XS);
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      acc0[r] = bias[r];
      acc1[r] = bias[r];
    }
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int tt = 0; tt < ST_KT; ++tt) {
          const float x0 = Xb[xfrag + (8 * g + c) * ST_XS + tt];
          const float x1 = Xb[xfrag + 32 + (8 * g + c) * ST_XS + tt];
          acc0 = adp_mfma32(av[g][c * ST_KT + tt], x0, acc0);
          acc1 = adp_mfma32(av[g][c * ST_KT + tt], x1, acc1);
        }
    float* Yb = Yall + (it & 1) * (ST_C * ST_YS);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int mo = ((r & 3) + 8 * (r >> 2)) * ST_YS;
      Yb[yfrag + mo] = acc0[r];
      Yb[yfrag + mo + 32] = acc1[r];
    }
  }
  adp_barrier_lds();  // B_nrounds: hands the last output tile to the load/store waves
}

}  // namespace

bool adp_conv_stream_eligible(const adp_conv_desc& d) {
  if (d.R != ST_C || d.R1 != d.R || d.M != ST_C || d.KT != ST_KT) return false;
  if (d.stride != 1 || d.dil != 1 || d.pad != 1 || d.up != 1 || d.store != 0) return false;
  if (d.out_pre || d.e_scale || d.x2) return false;
  if (d.prologue != 0 && d.prologue != 1) return false;
  if (d.prologue == 1 && (d.groups < 1 || ST_C % d.groups != 0)) return false;
  if (d.N != d.Lin || d.N % ST_TN != 0) return false;
  if ((reinterpret_cast<uintptr_t>(d.x) | reinterpret_cast<uintptr_t>(d.out) | reinterpret_cast<uintptr_t>(d.res)) & 15)
    return false;  // 16-byte loads of the input, 16-byte stores of the output, 16-byte loads of the residual
  if (d.B * (d.N / ST_TN) >= (int64_t)1 << 30 || d.B * ST_C * d.Lin >= (int64_t)1 << 40) return false;
  return true;
}

// workgroups per batch element: about one persistent workgroup per CU in total, each inside one batch element
static int stream_wpb(const adp_conv_desc& d) {
  const int tiles_per_b = (int)(d.N / ST_TN);
  int wpb = (int)(256 / d.B);
  if (wpb < 1) wpb = 1;
  if (wpb > tiles_per_b) wpb = tiles_per_b;
  return wpb;
}

int64_t adp_conv_stream_gn_entries(const adp_conv_desc& d) { return (int64_t)stream_wpb(d); }

int adp_conv_stream(const adp_conv_desc& d, void* stream) {
  const int tiles_per_b = (int)(d.N / ST_TN);
  const int wpb = stream_wpb(d);
  const int total = wpb;                       // (kernel argument: workgroups per batch element)
  const int grid = (int)d.B * wpb;
  if (d.transposed) {
    if (d.prologue == 1)
      ADP_LAUNCH((conv_stream32_kernel<true, 1>), dim3((unsigned)grid), dim3(512), stream, d, tiles_per_b, total);
    else
      ADP_LAUNCH((conv_stream32_kernel<true, 0>), dim3((unsigned)grid), dim3(512), stream, d, tiles_per_b, total);
  } else {
    if (d.prologue == 1)
      ADP_LAUNCH((conv_stream32_kernel<false, 1>), dim3((unsigned)grid), dim3(512), stream, d, tiles_per_b, total);
    else
      ADP_LAUNCH((conv_stream32_kernel<false, 0>), dim3((unsigned)grid), dim3(512), stream, d, tiles_per_b, total);
  }
  return ADP_LAUNCH_OK();
}
