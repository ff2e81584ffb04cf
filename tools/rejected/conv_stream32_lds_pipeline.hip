// Streaming ConvBlock convolution for the HBM-bound depth-1 layers: 32 -> 32 channels, kernel 3, stride 1, 'same'
// (ResnetItem ConvBlocks at channels = 32 and their data gradients; /root/reference/audio_diffusion_pytorch/
// components.py:89, SURVEY.md 8a row a13, 8d "HBM-bound" rows).
//
// At [4, 32, 65536] a ConvBlock conv moves 67-100 MB (A_in + A_out (+A_res)) for 1.6 GFLOP: arithmetic intensity
// ~20 flop/B, right at the MI355X ridge, so the kernel is built like a stream with the matrix cores fed from it:
//   * the whole 32 x 96 weight matrix lives in REGISTERS as MFMA A-operands (48 VGPRs per lane) for the lifetime of
//     a persistent workgroup -- there is no K loop, no K-group exchange, no weight traffic after start-up;
//   * one workgroup per CU walks a contiguous range of 256-position tiles.  Four loader waves stage the next tile
//     (32 channels x (256 + 8 halo) positions: 16-byte global loads, GroupNorm+SiLU in registers, 16-byte LDS
//     stores) while the MMA waves run 98 MFMAs (96 + the bias step) per 64 positions over the current one; one workgroup
//     barrier per tile hands the double-buffered LDS tile over (same protocol as conv_mm_impl.h);
//   * EIGHT MMA waves in two groups of four (one wave of each group per SIMD) take the tiles alternately, half a period
//     apart: while one group multiplies tile i, the other fetches the residual of, stores and accumulates the GroupNorm
//     statistics of tile i-1 straight from its accumulators (fire and forget), so the epilogue's HBM latency lies under the
//     other group's MFMAs.  Round-3 elimination builds at [4, 32, 65536] (conv2, us): full 29.8 | no MFMA 21.8 | no x loads
//     26.7 | no residual, no stores 22.9 | skeleton 8.9 -- with four tiles per workgroup the launch is ramp (first tile's
//     load latency), four MFMA-bound intervals and drain (last tile's stores); one group of four waves, where MFMAs and
//     epilogue of a tile were serial in one wave, took 31.6.
// Algorithmic bytes per launch: 4 * B * 32 * L * (2 + has_res) + 12 KB of weights.
#include <stdlib.h>
#include "adp_rt.h"
#include "adp.h"
#include "conv_internal.h"

namespace {

constexpr int ST_C = 32;           // channels in = channels out
constexpr int ST_KT = 3;
constexpr int ST_TN = 256;         // positions per workgroup iteration (4 MMA waves x 64)
constexpr int ST_XS = ST_TN + 8;   // LDS row stride: positions n0-4 .. n0+TN+3
constexpr int ST_XQ = ST_XS / 4;
constexpr int ST_NX4 = (ST_C * ST_XQ + 255) / 256;  // staging quads per loader thread

template <bool TR, int PRO>
__global__ __launch_bounds__(768) void conv_stream32_kernel(adp_conv_desc d, int tiles_per_b, int wpb) {
  __shared__ __attribute__((aligned(16))) float smem[2 * ST_C * ST_XS];
  // the 32 x 32 x 3 weights, staged ONCE per workgroup with coalesced 16-byte loads (rows of 96 floats at a stride of 97:
  // the fragment reads below are conflict-free).  Each of the eight MMA waves used to gather its 48 A operands straight
  // from global memory -- 48 load instructions with a 384-byte lane stride, 1536 cache-line requests per wave: most of the
  // 7-9 us launch skeleton the round-3 elimination builds showed.
  constexpr int ST_WS = ST_C * ST_KT + 1;
  __shared__ float wsm[ST_C * ST_WS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  f32x4 wq = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  const bool wvec = (reinterpret_cast<uintptr_t>(d.w) & 15) == 0;
  if (wvec) {  // 768 threads x 4 floats = the whole matrix
    wq = *reinterpret_cast<const f32x4*>(d.w + 4 * tid);
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) wq[k] = d.w[4 * tid + k];
  }
  const int hi = lane >> 5, l31 = lane & 31;
  const int L = (int)d.Lin;
  // contiguous tile range of this workgroup, inside ONE batch element (wpb workgroups per batch element): the
  // GroupNorm partial statistics of the output are then one entry per (workgroup, MMA wave) of that element
  const int wb = blockIdx.x / wpb, wi = blockIdx.x - wb * wpb;
  const int t_beg = wb * tiles_per_b + (int)(((int64_t)wi * tiles_per_b) / wpb);
  const int t_end = wb * tiles_per_b + (int)(((int64_t)(wi + 1) * tiles_per_b) / wpb);
  const int niter = t_end - t_beg;
  const int nrounds = (niter + 1) & ~1;  // the loaders run two tiles per loop trip; a ghost iteration pads odd counts

  if (wave >= 8) {
    // =========================== loader waves ===========================
    const int lt = tid - 512;
    int x_dst[ST_NX4], x_off[ST_NX4], x_pos[ST_NX4];
    float x_pa[ST_NX4], x_pb[ST_NX4];  // h = x * pa + pb (the workgroup stays inside one batch element: constants)
#pragma unroll
    for (int i = 0; i < ST_NX4; ++i) {
      const int e = (lt + i * 256) % (ST_C * ST_XQ);
      const int row = e / ST_XQ, pq = e - row * ST_XQ;
      x_dst[i] = row * ST_XS + 4 * pq;
      x_off[i] = row * L;
      x_pos[i] = 4 * pq - 4;
    }
    // two register stages: tile it+2 is requested while tile it+1 is being staged, so a tile's loads have a full
    // MFMA phase + a staging phase to arrive instead of one phase
    f32x4 rxs[2][ST_NX4];
    bool oks[2][ST_NX4];
    const float* xb = d.x + (int64_t)wb * ST_C * L;
    auto load_tile = [&](f32x4 (&rx)[ST_NX4], bool (&ok)[ST_NX4], int it) {
      const int t = t_beg + (it < niter ? it : niter - 1);
      const int n0 = (t - wb * tiles_per_b) * ST_TN;
#pragma unroll
      for (int i = 0; i < ST_NX4; ++i) {
        const int u = n0 + x_pos[i];
        ok[i] = (u >= 0 && u < L);  // L % 4 == 0: a quad is entirely inside or outside the row
        rx[i] = *reinterpret_cast<const f32x4*>(xb + x_off[i] + (ok[i] ? u : 0));
      }
    };
    auto store_tile = [&](const f32x4 (&rx)[ST_NX4], const bool (&ok)[ST_NX4], int it) {
      float* Xb = smem + (it & 1) * (ST_C * ST_XS);
#pragma unroll
      for (int i = 0; i < ST_NX4; ++i) {
        f32x4 v = rx[i];
        if (PRO == 1) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = adp_silu_fast(fmaf(v[j], x_pa[i], x_pb[i]));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = ok[i] ? v[j] : 0.0f;  // zero padding is applied after the activation
        *reinterpret_cast<f32x4*>(Xb + x_dst[i]) = v;
      }
    };
    // the store of tile it goes to LDS[it & 1], last read by the MFMAs of tile it-2; the MMA waves that ran those
    // arrived at barrier B_{it-1} after issuing them, and this wave passed B_{it-1} before starting iteration it
    load_tile(rxs[0], oks[0], 0);
    load_tile(rxs[1], oks[1], 1);
#pragma unroll
    for (int k = 0; k < 4; ++k) wsm[((4 * tid + k) / (ST_C * ST_KT)) * ST_WS + (4 * tid + k) % (ST_C * ST_KT)] = wq[k];
    __syncthreads();  // weights staged (pairs with the MMA waves' barrier below)
    if (PRO == 1) {  // (after the first tiles' loads are on their way: these constants come from memory too)
#pragma unroll
      for (int i = 0; i < ST_NX4; ++i) {
        const int row = ((lt + i * 256) % (ST_C * ST_XQ)) / ST_XQ;
        const int64_t sg = ((int64_t)wb * d.groups + row / (ST_C / (int)d.groups)) * 2;
        x_pa[i] = (d.pro_gamma ? d.pro_gamma[row] : 1.0f) * d.pro_stats[sg + 1];
        x_pb[i] = (d.pro_beta ? d.pro_beta[row] : 0.0f) - d.pro_stats[sg] * x_pa[i];
      }
    }
    for (int it0 = 0; it0 < nrounds; it0 += 2) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        store_tile(rxs[s], oks[s], it0 + s);   // ghost tile (odd niter): restages the last one
        load_tile(rxs[s], oks[s], it0 + s + 2);  // unconditional, clamped: never consumed
        __syncthreads();                         // B_it
      }
    }
    return;
  }

  // =========================== MMA waves ===========================
  // Two groups of four (one wave of each group per SIMD) take the tiles alternately and run half a period apart: in the
  // interval after barrier B_it, group it & 1 multiplies tile it while the other group adds the residual to, stores and
  // accumulates the statistics of the tile it multiplied one interval earlier -- so every SIMD has one wave on the
  // matrix cores and one on the memory pipes at any time (elimination build of the one-group form: 10.5 of 31.8 us were
  // MFMAs nothing else ran under).
  const int grp = wave >> 2, wqt = wave & 3;
#pragma unroll
  for (int k = 0; k < 4; ++k) wsm[((4 * tid + k) / (ST_C * ST_KT)) * ST_WS + (4 * tid + k) % (ST_C * ST_KT)] = wq[k];
  __syncthreads();  // weights staged
  // A operands: av[g][c*KT + t] = A(m = l31, channel 8g + c + 4hi, tap t), from the staged copy (w[M][R][KT], or w[R][M][KT]
  // with the taps flipped for the data gradient)
  float av[4][4 * ST_KT];
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int t = 0; t < ST_KT; ++t) {
        const int r = 8 * g + c + 4 * hi;
        av[g][c * ST_KT + t] = TR ? wsm[r * ST_WS + l31 * ST_KT + (ST_KT - 1 - t)] : wsm[l31 * ST_WS + r * ST_KT + t];
      }
  // the bias enters through one extra MFMA step per accumulator tile (A = bias[m] in the k = 0 half, B = 1): one register
  // instead of the 16 a per-row add in the epilogue would hold for the lifetime of the workgroup
  const float bias_a = (d.bias && hi == 0) ? d.bias[l31] : 0.0f;
  const int xfrag = 4 * hi * ST_XS + 64 * wqt + l31 + 4 - 1;  // + ni*32 + (8g + c) * XS + t
  const bool has_res = d.res != nullptr;
  const bool want_gn = d.gn_part != nullptr;
  float gs[4], gq[4];  // running sum / sum of squares of this lane's share of each output ROW QUAD (GroupNorm partials)
#pragma unroll
  for (int q = 0; q < 4; ++q) gs[q] = gq[q] = 0.0f;
#ifndef ADP_EMULATE
  // Make the weight / bias loads complete HERE.  Otherwise the first MFMA of the loop body is the first use of
  // registers that are pending on the loop-entry path only, and the compiler covers it with s_waitcnt vmcnt(0) on
  // EVERY iteration -- which also waits for the residual loads just issued and the previous tile's stores, i.e.
  // serialises HBM latency with the matrix cores.
#pragma unroll
  for (int g = 0; g < 4; ++g)
    asm volatile("" ::"v"(av[g][0]), "v"(av[g][1]), "v"(av[g][2]), "v"(av[g][3]), "v"(av[g][4]), "v"(av[g][5]),
                 "v"(av[g][6]), "v"(av[g][7]), "v"(av[g][8]), "v"(av[g][9]), "v"(av[g][10]), "v"(av[g][11]));
  asm volatile("" ::"v"(bias_a));
#endif

  f32x16 acc0, acc1;
  int64_t pbase = 0;     // output offset (wave-uniform part) of the tile whose accumulators wait for their epilogue
  const int lo = 4 * hi * L + l31;  // lane part: row 4 * hi of the tile, column l31 (rows add a uniform multiple of L)
  bool pending = false;
  auto finish_tile = [&]() {
    // (the residual is fetched here, not before the MFMAs: this wave has the whole interval -- the other group's MFMA
    // phase -- for it, and 32 registers less are live while its own MFMAs run)
    float rv[2][16];
    if (has_res) {
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float* rrow = d.res + pbase + (int64_t)((r & 3) + 8 * (r >> 2)) * L + 32 * ni;  // uniform
          rv[ni][r] = rrow[lo];
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v0 = acc0[r], v1 = acc1[r];
      if (has_res) {
        v0 += rv[0][r];
        v1 += rv[1][r];
      }
      float* orow = d.out + pbase + (int64_t)((r & 3) + 8 * (r >> 2)) * L;  // uniform
      orow[lo] = v0;
      orow[lo + 32] = v1;
      if (want_gn) {
        gs[r >> 2] += v0 + v1;
        gq[r >> 2] = fmaf(v0, v0, fmaf(v1, v1, gq[r >> 2]));
      }
    }
  };
  for (int it = 0; it < nrounds; ++it) {
    adp_barrier_consume();  // B_it: tile it is in LDS[it & 1] (no wait for this wave's stores in flight: adp_rt.h)
    if (pending) {
      finish_tile();
      pending = false;
    } else if (it < niter && (it & 1) == grp) {
      const int t = t_beg + it;
      const int n0 = (t - wb * tiles_per_b) * ST_TN;
      pbase = (int64_t)wb * ST_C * L + n0 + 64 * wqt;
      const float* Xb = smem + (it & 1) * (ST_C * ST_XS);
#pragma unroll
      for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.0f;
      acc0 = adp_mfma32(bias_a, 1.0f, acc0);
      acc1 = adp_mfma32(bias_a, 1.0f, acc1);
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
      pending = true;
    }
  }
  if (pending) finish_tile();
  if (want_gn) {
    // one (mean, M2, count) entry per ROW QUAD (accumulator registers 4q .. 4q+3 = 4 consecutive channels) for the
    // positions this wave produced: 64 per tile of its group (a group without tiles writes an empty entry)
    const int ntl = (niter + 1 - grp) / 2;
    const float cnt = 4.0f * 64.0f * (float)ntl;
    const int E = wpb * 8;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float sv = gs[q], qv = gq[q];
#pragma unroll
      for (int o = 1; o < 32; o <<= 1) {
        sv += __shfl_xor(sv, o, 64);
        qv += __shfl_xor(qv, o, 64);
      }
      if (l31 == 0) {
        const int m = 8 * q + 4 * hi;  // first channel of the quad
        const float mean = ntl > 0 ? sv / cnt : 0.0f;
        float* e = d.gn_part + (((int64_t)wb * (ST_C / 4) + (m >> 2)) * E + wi * 8 + wave) * 3;
        e[0] = mean;
        e[1] = fmaxf(qv - sv * mean, 0.0f);
        e[2] = cnt;
      }
    }
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
  if (reinterpret_cast<uintptr_t>(d.x) & 15) return false;
  if (d.B * (d.N / ST_TN) >= (int64_t)1 << 30 || d.B * ST_C * d.Lin >= (int64_t)1 << 40) return false;
  return true;
}

// workgroups per batch element: about one persistent workgroup per CU in total, each inside one batch element
static int stream_wpb(const adp_conv_desc& d) {
  const int tiles_per_b = (int)(d.N / ST_TN);
  int wpb = (int)(256 / d.B);
  if (const char* e = getenv("ADP_STREAM_WPB")) wpb = atoi(e);  // tests: several tiles per workgroup on small problems
  if (wpb < 1) wpb = 1;
  if (wpb > tiles_per_b) wpb = tiles_per_b;
  return wpb;
}

int64_t adp_conv_stream_gn_entries(const adp_conv_desc& d) { return (int64_t)stream_wpb(d) * 8; }

int adp_conv_stream(const adp_conv_desc& d, void* stream) {
  const int tiles_per_b = (int)(d.N / ST_TN);
  const int wpb = stream_wpb(d);
  const int total = wpb;                       // (kernel argument: workgroups per batch element)
  const int grid = (int)d.B * wpb;
  if (d.transposed) {
    if (d.prologue == 1)
      ADP_LAUNCH((conv_stream32_kernel<true, 1>), dim3((unsigned)grid), dim3(768), stream, d, tiles_per_b, total);
    else
      ADP_LAUNCH((conv_stream32_kernel<true, 0>), dim3((unsigned)grid), dim3(768), stream, d, tiles_per_b, total);
  } else {
    if (d.prologue == 1)
      ADP_LAUNCH((conv_stream32_kernel<false, 1>), dim3((unsigned)grid), dim3(768), stream, d, tiles_per_b, total);
    else
      ADP_LAUNCH((conv_stream32_kernel<false, 0>), dim3((unsigned)grid), dim3(768), stream, d, tiles_per_b, total);
  }
  return ADP_LAUNCH_OK();
}
