// Wave-tile Winograd F(4,3) convolution for the DEEP ConvBlock layers when their output tiles alone cannot fill the chip:
// kernel 3, stride 1, 'same', 512-1024 channels over 128-1024 positions (ResnetItem ConvBlocks of depths 5-8 at batch 1, depth 8
// at batch 4; /root/reference/audio_diffusion_pytorch/components.py:89; BASELINE configs 1, 3, 4).
//
// conv_mm / conv_mm4 give such a layer 64-128 blocks of 32 rows x 64-128 positions and fill the other CUs with a cross-workgroup
// K split: partial tiles through HBM and a second launch that sums them (45 reduce launches per U-Net forward at batch 1).  Here
// the K split stays INSIDE the workgroup and the tile can be 16 rows:
//   * a workgroup owns one (16 RB) x 64 output tile; its 8 waves split the input channels (R / 8 each) and every wave runs
//     conv_tile.hip's barrier-free wave tile on its own channels: 16-channel chunks fetched with coalesced 16-byte loads one
//     chunk ahead; the x tile [16][66] is parked in a wave-PRIVATE LDS region, the weights never touch LDS: in the MFMA's A
//     layout lane (j, kq) multiplies row j by channel 4 ks + kq, so each lane fetches exactly the 4 RB tap triples it will
//     multiply (12-byte loads) and forms its six planes of U = G g in registers; F(4,3) on v_mfma_f32_16x16x4_f32 with V = B^T d
//     formed in registers -- no workgroup barrier anywhere in the K loop, the SIMD's other waves cover one wave's load phases
//     (first version: raw weights and U through LDS, 28-46 KB of LDS traffic per chunk and wave against 10 KB now -- the LDS, one
//     per CU, was the bound: [1, 1024, 256] 18.3 us, [4, 1024, 128] 28.0 us);
//   * the eight partial tiles meet in LDS once (after A^T: four values per lane and output row instead of six planes), summed in
//     wave order by the 4 RB waves that also add bias / residual, store 16 bytes per lane and form the GroupNorm partial
//     statistics of the output (shifted sums per row, Chan-combined per 4-channel row quad: conv_tile.hip's entry format);
//   * 16-row tiles (RB = 1) double the tile count where 32-row tiles are fewer than ~200: [1, 1024, 256] is 256 workgroups of
//     8 waves instead of 128 blocks x 2 K slices + a reduce launch.
// Matrix work: half of the direct form's, three quarters of conv_mm's F(2,3).  fp32 throughout (error against fp64 ~1e-6 of the
// output's max norm, tests/test_kernels.py).  Algorithmic bytes per launch: 4 * (B * R * L + B * M * L (+ residual) + 3 * M * R).
#include <stdlib.h>
#include "adp_rt.h"
#include "adp.h"
#include "conv_internal.h"

#ifdef ADP_KTRACE
static __device__ unsigned long long* tk_kt_buf = nullptr;
extern "C" int adp_ktrace_set_tilek(void* p) {
  return hipMemcpyToSymbol(HIP_SYMBOL(tk_kt_buf), &p, sizeof(p)) == hipSuccess ? 0 : -1;
}
#define TK_KT_BUF tk_kt_buf
#else
#define TK_KT_BUF nullptr
#endif

namespace {

constexpr int TK_KT = 3;
constexpr int TK_TN = 64;            // positions per tile (16 output quads)
constexpr int TK_RS = TK_TN + 2;     // LDS row of the x tile: index i <-> position n0 - 1 + i
constexpr int TK_CH = 16;            // channels per chunk (four K steps of v_mfma_f32_16x16x4_f32)
constexpr int TK_NKW = 8;            // waves per workgroup = K slices
constexpr int TK_XF = TK_CH * TK_RS;

// NCH: 16-channel chunks per wave, unrolled (4: 512 input channels, 8: 1024) -- in a rolled loop the chunk-ahead registers are
// loop-carried values and the compiler parks them through register copies behind s_waitcnt vmcnt(0) at the back edge, which
// turns the prefetch into a wait for what was just requested; 0 = rolled loop (any other channel count)
// GNB: the launch also leaves the first stage of a GroupNorm backward (adp_conv_desc.gnb_ab): separate instantiations (data
//      gradients, PF = 2), every other launch compiles as it did without it
template <bool TR, int RB, int NCH, int PF = 2, bool GNB = false>
__global__ __launch_bounds__(64 * TK_NKW) void conv_tilek_kernel(adp_conv_desc d, int ntn) {
  const bool RES = d.res != nullptr, GN = d.gn_part != nullptr;  // (workgroup-uniform)
  constexpr int ROWS = 16 * RB;
  constexpr int YF = RB * 4 * 64 * 4;                        // a wave's partial tile after A^T: [rb][r][lane] float4
  constexpr int WF = TK_XF > YF ? TK_XF : YF;                // one wave's region: x tile, later its partial tile
  __shared__ __attribute__((aligned(16))) float lds[TK_NKW * WF + TK_NKW * 8];
  float* const gsh = lds + TK_NKW * WF;            // statistics scratch [wave][kq][2]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = adp_uniform(tid >> 6);
  const int j = lane & 15, kq = lane >> 4;         // MFMA 16x16x4: lane = (row / quad column j, K index kq)
  const int M = (int)d.M, R = (int)d.R, L = (int)d.Lin;
  ADP_KT_DECL(TK_KT_BUF)
  ADP_KT(0);
#ifdef ADP_KTRACE
  int kt_chunk = 0;
#endif

  // ---- XCD-aware decode of the 1-D grid: an XCD gets a contiguous range of row tiles (its L2 keeps their weight slabs)
  int id = blockIdx.x;
  const int total = gridDim.x;
  if ((total & 7) == 0) id = (id & 7) * (total >> 3) + (id >> 3);
  const int per_m = ntn * (int)d.B;
  const int mt = id / per_m, rem = id - mt * per_m;
  const int b = rem / ntn, nt = rem - b * ntn;
  const int m0 = mt * ROWS, n0 = nt * TK_TN;

  const int kc = R / TK_NKW;                       // this wave's channels [c_lo, c_lo + kc)
  const int c_lo = wave * kc, nchunks = NCH > 0 ? NCH : kc / TK_CH;
  float* const X = lds + wave * WF;

  // ---- chunk loads (registers, one chunk ahead): x tile 16 rows x 16 quads = 4 float4 per lane + halo; weights 3 RB float4
  const float* xb = d.x + ((int64_t)b * R + c_lo) * L + n0;
  const int xoff = (lane >> 4) * L + 4 * (lane & 15);          // + 4 i rows
  // halo: lanes 0-31 = (row, side); lanes 32-63 request the SAME addresses (no further cache lines) and do not write
  const int hrow = lane & 15, hside = (lane >> 4) & 1;
  const int hpos = hside ? TK_TN : -1;
  const bool hok = n0 + hpos >= 0 && n0 + hpos < L;
  const int hoff = hrow * L + (hok ? hpos : 0);
  // weights: lane (j, kq) needs w[m0 + 16 rb + j][c0 + 4 ks + kq][0..2] (TR: w[c0 + 4 ks + kq][m0 + 16 rb + j][.]), ks = 0..3
  const float* wb = TR ? d.w + ((int64_t)(c_lo + kq) * M + m0 + j) * TK_KT : d.w + ((int64_t)(m0 + j) * R + c_lo + kq) * TK_KT;
  const int w_ks = TR ? 4 * M * TK_KT : 4 * TK_KT;            // + 4 channels
  const int w_rb = TR ? 16 * TK_KT : 16 * R * TK_KT;          // + 16 rows
  // register sets: a chunk's loads are requested PF chunks ahead (a 16-channel chunk is 24 RB MFMAs = 0.4-0.7 us of matrix
  // work, an L2 / HBM round trip 1-2 us)
  struct ChunkRegs {
    f32x4 x[4];
    float w[RB][4][3];
    float h;
  };
  auto load_chunk = [&](ChunkRegs& c, int chunk) {
    const float* xp = xb + (int64_t)chunk * TK_CH * L;
#pragma unroll
    for (int i = 0; i < 4; ++i) c.x[i] = *reinterpret_cast<const f32x4*>(xp + xoff + 4 * i * L);
    c.h = xp[hoff];
    const float* wp = wb + (TR ? (int64_t)chunk * TK_CH * M * TK_KT : (int64_t)chunk * TK_CH * TK_KT);
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int t = 0; t < 3; ++t) c.w[rb][ks][t] = wp[rb * w_rb + ks * w_ks + t];
  };

  // epilogue operands of the finishing waves: requested before the K loop
  const bool fin = wave < 4 * RB;
  const int frb = wave >> 2, fr = wave & 3;
  const int fch = m0 + 16 * frb + 4 * kq + fr;               // this lane's output channel when its wave finishes
  const int64_t foff = ((int64_t)b * M + fch) * L + n0 + 4 * j;
  f32x4 rv = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  float bv = 0.0f;
  // first stage of the backward of SiLU(GroupNorm(gnb_x)) whose output gradient this launch produces (adp_conv_desc.gnb_ab)
  f32x4 gx = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  float gmu = 0.0f, grs = 0.0f, gga = 0.0f, gbe = 0.0f;
  if (fin) {
    if (RES) rv = *reinterpret_cast<const f32x4*>(d.res + foff);
    if (d.bias) bv = d.bias[fch];
    if (GNB) {
      gx = *reinterpret_cast<const f32x4*>(d.gnb_x + foff);
      const float* st = d.gnb_stats + ((int64_t)b * d.gnb_groups + fch / (M / (int)d.gnb_groups)) * 2;
      gmu = st[0], grs = st[1];
      gga = d.gnb_gamma[fch] * grs;
      gbe = d.gnb_beta[fch] - gmu * gga;
    }
  }

  f32x4 Macc[RB][6];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int p = 0; p < 6; ++p) Macc[rb][p] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

  const float* Xr = X + kq * TK_RS + 4 * j;        // + 4 ks rows

  auto run_chunk = [&](ChunkRegs& cr, int next) {
    // ---- park the chunk: x tile (index i of a row <-> position n0 - 1 + i) and the raw weights
#ifdef ADP_KTRACE
    if (kt_chunk < 8) ADP_KT(1 + 6 * kt_chunk);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PF == 2 ? 9 : 27) : "memory");  // (roughly: this set's loads have landed)
    if (kt_chunk < 8) ADP_KT(2 + 6 * kt_chunk);
#endif
    {
      float* o = X + (lane >> 4) * TK_RS + 4 * (lane & 15) + 1;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        o[4 * i * TK_RS] = cr.x[i][0];
        *reinterpret_cast<f32x2*>(o + 4 * i * TK_RS + 1) = f32x2{cr.x[i][1], cr.x[i][2]};  // even index: 8-byte aligned
        o[4 * i * TK_RS + 3] = cr.x[i][3];
      }
      if (lane < 32) X[hrow * TK_RS + (hside ? TK_RS - 1 : 0)] = hok ? cr.h : 0.0f;  // zero padding
    }
    // ---- U = G g in registers; G = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1]
    float uf[RB][4][6];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const float a = cr.w[rb][ks][0], g1 = cr.w[rb][ks][1], e = cr.w[rb][ks][2];
        const float g0 = TR ? e : a, g2 = TR ? a : e;  // the data gradient runs the taps backwards
        const float s02 = g0 + g2, q = 0.041666666666666664f * g0 + 0.16666666666666666f * g2;
        uf[rb][ks][0] = 0.25f * g0;
        uf[rb][ks][1] = -0.16666666666666666f * (s02 + g1);
        uf[rb][ks][2] = -0.16666666666666666f * (s02 - g1);
        uf[rb][ks][3] = q + 0.08333333333333333f * g1;
        uf[rb][ks][4] = q - 0.08333333333333333f * g1;
        uf[rb][ks][5] = g2;
      }
    adp_wave_sync();
#ifdef ADP_KTRACE
    if (kt_chunk < 8) ADP_KT(3 + 6 * kt_chunk);
#endif
    // this set's next chunk, in flight under two chunks of MFMAs.  UNCONDITIONAL (past the end the last chunk is fetched again
    // and never used): a branch around the loads makes the compiler merge the two paths with register copies behind an
    // s_waitcnt vmcnt(0) -- every iteration then waits for the loads it has just requested
    load_chunk(cr, next < nchunks ? next : nchunks - 1);
#ifdef ADP_KTRACE
    if (kt_chunk < 8) ADP_KT(4 + 6 * kt_chunk);
#endif
    // ---- four K steps: lane (j, kq) reads the six inputs around its quad and its rows' six planes of channel 4 ks + kq
    f32x2 dn[3];
    auto frags = [&](int ks) {
#pragma unroll
      for (int q = 0; q < 3; ++q) dn[q] = *reinterpret_cast<const f32x2*>(Xr + 4 * ks * TK_RS + 2 * q);
    };
    frags(0);
#pragma unroll
    for (int ks = 0; ks < TK_CH / 4; ++ks) {
      f32x2 dc[3];
#pragma unroll
      for (int q = 0; q < 3; ++q) dc[q] = dn[q];
      if (ks + 1 < TK_CH / 4) frags(ks + 1);
      adp_sched_fence();
      // B^T d: (4 d0 - 5 d2 + d4, -4 d1 - 4 d2 + d3 + d4, 4 d1 - 4 d2 - d3 + d4, -2 d1 - d2 + 2 d3 + d4,
      //         2 d1 - d2 - 2 d3 + d4, 4 d1 - 5 d3 + d5)
      const float d0 = dc[0][0], d1 = dc[0][1], d2 = dc[1][0], d3 = dc[1][1], d4 = dc[2][0], d5 = dc[2][1];
      const float a = fmaf(-4.0f, d2, d4), bb = fmaf(-4.0f, d1, d3), c = d4 - d2, e = 2.0f * (d3 - d1);
      float v[6];
      v[0] = fmaf(4.0f, d0, fmaf(-5.0f, d2, d4));
      v[1] = a + bb;
      v[2] = a - bb;
      v[3] = c + e;
      v[4] = c - e;
      v[5] = fmaf(4.0f, d1, fmaf(-5.0f, d3, d5));
#pragma unroll
      for (int p = 0; p < 6; ++p)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) Macc[rb][p] = adp_mfma16(uf[rb][ks][p], v[p], Macc[rb][p]);
      adp_sched_fence();
    }
#ifdef ADP_KTRACE
    if (kt_chunk < 8) ADP_KT(5 + 6 * kt_chunk);
#endif
    adp_wave_sync();  // the next chunk overwrites this wave's tiles
#ifdef ADP_KTRACE
    if (kt_chunk < 8) ADP_KT(6 + 6 * kt_chunk);
    ++kt_chunk;
#endif
  };
  // PF register sets = chunks in flight per wave (29 registers a set with 16-row tiles, 41 with 32-row tiles)
  ChunkRegs cs[PF];
#pragma unroll
  for (int i = 0; i < PF; ++i) load_chunk(cs[i], i < nchunks ? i : nchunks - 1);
  if (NCH > 0) {
#pragma unroll
    for (int chunk = 0; chunk < NCH; chunk += PF)
#pragma unroll
      for (int i = 0; i < PF; ++i) run_chunk(cs[i], chunk + i + PF);
  } else {
    for (int chunk = 0; chunk < nchunks; chunk += 2) {  // (nchunks is even: eligibility)
      run_chunk(cs[0], chunk + 2);
      run_chunk(cs[1], chunk + 3);
    }
  }

  ADP_KT(60);
  // ---- y = A^T m per output row (A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]); the wave's partial tile goes
  // to the head of its own region: [rb][r][lane] float4
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float m0v = Macc[rb][0][r], m1 = Macc[rb][1][r], m2 = Macc[rb][2][r], m3 = Macc[rb][3][r], m4 = Macc[rb][4][r],
                  m5 = Macc[rb][5][r];
      const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
      *reinterpret_cast<f32x4*>(X + ((rb * 4 + r) * 64 + lane) * 4) =
          f32x4{m0v + s12 + s34, fmaf(2.0f, d34, d12), fmaf(4.0f, s34, s12), fmaf(8.0f, d34, d12) + m5};
    }
  __syncthreads();
  ADP_KT(61);
  float gmean = 0.0f, gm2 = 0.0f;
  if (fin) {
    // accumulator register r of row block rb <-> output channel 16 rb + 4 kq + r, column j = output quad: this wave sums row
    // (frb, fr) of all K slices in wave order
    f32x4 y = *reinterpret_cast<const f32x4*>(lds + ((frb * 4 + fr) * 64 + lane) * 4);
#pragma unroll
    for (int w = 1; w < TK_NKW; ++w) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(lds + w * WF + ((frb * 4 + fr) * 64 + lane) * 4);
#pragma unroll
      for (int k = 0; k < 4; ++k) y[k] += t[k];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) y[k] += bv;
    if (RES) {
#pragma unroll
      for (int k = 0; k < 4; ++k) y[k] += rv[k];
    }
    *reinterpret_cast<f32x4*>(d.out + foff) = y;
    if (GNB) {  // (sum ds * xhat, sum ds) of channel fch over the tile's 64 positions, ds = da * silu'(gamma * xhat + beta)
      float sa = 0.0f, sb = 0.0f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float xh = (gx[k] - gmu) * grs;
        const float ds = y[k] * adp_dsilu_fast(fmaf(gx[k], gga, gbe));
        sa = fmaf(ds, xh, sa);
        sb += ds;
      }
      sa = adp_row16_sum(sa), sb = adp_row16_sum(sb);
      if (j == 0) *reinterpret_cast<f32x2*>(d.gnb_ab + (((int64_t)b * M + fch) * ntn + nt) * 2) = f32x2{sa, sb};
    }
    if (GN) {
      // (mean, M2) of this wave's 64 positions of channel fch, shifted by the row's first value (|mean| >> sigma costs no digits)
      const float gk = __shfl(y[0], lane & 48, 64);
      float s = 0.0f, q = 0.0f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float e = y[k] - gk;
        s += e;
        q = fmaf(e, e, q);
      }
      const float sv = adp_row16_sum(s), qv = adp_row16_sum(q);
      gmean = gk + sv / (float)TK_TN;
      gm2 = fmaxf(qv - sv * (sv / (float)TK_TN), 0.0f);
      if (j == 0) {
        gsh[(wave * 4 + kq) * 2] = gmean;
        gsh[(wave * 4 + kq) * 2 + 1] = gm2;
      }
    }
  }
  if (GN) {
    __syncthreads();
    if (tid < 4 * RB) {
      // row quad tid = 4 rb + kq: its four channels were finished by waves 4 rb + r, r = 0..3 (Chan's pairwise update)
      const int rb = tid >> 2, qk = tid & 3;
      constexpr float cnt = (float)TK_TN;
      float mean = gsh[((4 * rb) * 4 + qk) * 2], m2 = gsh[((4 * rb) * 4 + qk) * 2 + 1], n = cnt;
      for (int r = 1; r < 4; ++r) {
        const float mw = gsh[((4 * rb + r) * 4 + qk) * 2], dl = mw - mean, nn = n + cnt;
        mean += dl * (cnt / nn);
        m2 += gsh[((4 * rb + r) * 4 + qk) * 2 + 1] + dl * dl * (n * cnt / nn);
        n = nn;
      }
      float* e = d.gn_part + (((int64_t)b * (M / 4) + m0 / 4 + tid) * ntn + nt) * 3;
      e[0] = mean;
      e[1] = m2;
      e[2] = n;
    }
  }
  ADP_KT(63);
  ADP_KT_DUMP(blockIdx.x);
}

static int64_t tilek_tiles32(const adp_conv_desc& d) { return (d.M / 32) * d.B * (d.N / TK_TN); }

// row blocks per tile: 32-row tiles when they alone give every CU a workgroup, else 16-row tiles (twice the workgroups)
static int tilek_rb(const adp_conv_desc& d) {
  const char* e = getenv("ADP_TILEK_RB");  // tests / A-B: force 1 or 2
  if (e && (e[0] == '1' || e[0] == '2')) return e[0] - '0';
  return tilek_tiles32(d) >= 200 ? 2 : 1;
}

template <bool TR, int RB>
int launch_tilek(const adp_conv_desc& d, void* stream) {
  const int ntn = (int)(d.N / TK_TN);
  const unsigned grid = (unsigned)((d.M / (16 * RB)) * d.B * ntn);
  const int64_t nch = d.R / TK_NKW / TK_CH;
  const char* pf = getenv("ADP_TILEK_PF");
  // chunks in flight per wave: 2.  In-step A/B (hipGraph replay, same box), 2 -> 4: batch-1 step 6.22 -> 6.28 ms, config-4 layout
  // 10.12 -> 10.19 ms -- the CUs are short of L1 fill rate, not of requests in flight (ADP_TILEK_PF=4: the deeper variant, 16-row tiles)
  const bool pf4 = RB == 1 && pf && atoi(pf) == 4;
  if constexpr (TR) {
    if (d.gnb_ab) {
      if (nch == 4) ADP_LAUNCH((conv_tilek_kernel<TR, RB, 4, 2, true>), dim3(grid), dim3(64 * TK_NKW), stream, d, ntn);
      else if (nch == 8) ADP_LAUNCH((conv_tilek_kernel<TR, RB, 8, 2, true>), dim3(grid), dim3(64 * TK_NKW), stream, d, ntn);
      else ADP_LAUNCH((conv_tilek_kernel<TR, RB, 0, 2, true>), dim3(grid), dim3(64 * TK_NKW), stream, d, ntn);
      return ADP_LAUNCH_OK();
    }
  }
  if (nch == 4 && pf4) ADP_LAUNCH((conv_tilek_kernel<TR, RB, 4, (RB == 1 ? 4 : 2)>), dim3(grid), dim3(64 * TK_NKW), stream, d, ntn);
  else if (nch == 8 && pf4) ADP_LAUNCH((conv_tilek_kernel<TR, RB, 8, (RB == 1 ? 4 : 2)>), dim3(grid), dim3(64 * TK_NKW), stream, d, ntn);
  else if (nch == 4) ADP_LAUNCH((conv_tilek_kernel<TR, RB, 4>), dim3(grid), dim3(64 * TK_NKW), stream, d, ntn);
  else if (nch == 8) ADP_LAUNCH((conv_tilek_kernel<TR, RB, 8>), dim3(grid), dim3(64 * TK_NKW), stream, d, ntn);
  else ADP_LAUNCH((conv_tilek_kernel<TR, RB, 0>), dim3(grid), dim3(64 * TK_NKW), stream, d, ntn);
  return ADP_LAUNCH_OK();
}

}  // namespace

// Taken where the output tiles alone leave the chip short of work (ADP_TILEK_MIN_TILES .. ADP_TILEK_MAX_TILES 32-row tiles,
// default 100 .. 320: depths 5-7 at batch 1, depth 8 at batch 2-4) and the K loop is long enough to split eight ways in 16-channel chunks (ADP_TILEK_MIN_R, 512).
// ADP_CONV_TILEK=0: those layers stay on conv_mm / conv_mm4 with their cross-workgroup K split (A/B).
bool adp_conv_tilek_eligible(const adp_conv_desc& d) {
  if (!adp_winograd_enabled()) return false;
  const char* e = getenv("ADP_CONV_TILEK");
  if (e && e[0] == '0') return false;
  if (d.KT != TK_KT || d.stride != 1 || d.dil != 1 || d.pad != 1 || d.up != 1 || d.store != 0) return false;
  if (d.prologue != 0 || d.x2 || d.R1 != d.R || d.out_pre || d.e_scale) return false;
  const char* mr = getenv("ADP_TILEK_MIN_R");
  if (d.R < (mr ? atoll(mr) : 512) || d.R % (2 * TK_NKW * TK_CH) != 0 || d.M % 32 != 0) return false;  // (chunk pairs)
  if (d.N != d.Lin || d.N % TK_TN != 0) return false;
  // hipGraph microbench, us per launch (conv2 + residual; mm = conv_mm / conv_mm4 incl. its reduce launch -> this kernel):
  //   [1,512,1024] 17.4 -> 16.4 (32-row)   [1,512,512] 12.7 -> 10.0   [1,1024,256] 19.3 -> 16.5   [2,1024,128] 18.8 -> 16.1 (16-row)
  //   [4,1024,128] 28.4 -> 25.9 (32-row; data gradient 28.0 -> 22.1)    [1,1024,128] 13.6 -> 15.7 and [8,1024,128] 40 -> 38: left out
  const char* mt = getenv("ADP_TILEK_MAX_TILES");
  const char* mn = getenv("ADP_TILEK_MIN_TILES");
  if (tilek_tiles32(d) > (mt ? atoll(mt) : 320) || tilek_tiles32(d) < (mn ? atoll(mn) : 100)) return false;
  if ((reinterpret_cast<uintptr_t>(d.x) | reinterpret_cast<uintptr_t>(d.w) | reinterpret_cast<uintptr_t>(d.out) |
       reinterpret_cast<uintptr_t>(d.res)) & 15)
    return false;
  if (d.B * d.R * d.Lin >= (int64_t)1 << 31 || d.M * d.R * d.KT >= (int64_t)1 << 31 || d.B * d.M * d.N >= (int64_t)1 << 40)
    return false;
  return true;
}

int64_t adp_conv_tilek_gn_entries(const adp_conv_desc& d) { return d.N / TK_TN; }
// one slice per row and 64-position tile (data gradients: the instantiations that exist)
int64_t adp_conv_tilek_gnb_entries(const adp_conv_desc& d) { return d.transposed && adp_gnb_family_on(4) ? d.N / TK_TN : 0; }

int adp_conv_tilek(const adp_conv_desc& d, void* stream) {
  if (tilek_rb(d) == 2) return d.transposed ? launch_tilek<true, 2>(d, stream) : launch_tilek<false, 2>(d, stream);
  return d.transposed ? launch_tilek<true, 1>(d, stream) : launch_tilek<false, 1>(d, stream);
}
