// Barrier-free ConvBlock convolution for the HBM-bound depth-1 layers: 32 -> 32 channels, kernel 3, stride 1, 'same'
// (ResnetItem ConvBlocks at channels = 32 and their data gradients; /root/reference/audio_diffusion_pytorch/
// components.py:89, SURVEY.md 8a row a13, 8d "HBM-bound" rows).
//
// At [4, 32, 65536] a ConvBlock conv moves 67-100 MB (A_in + A_out (+A_res)) for 1.6 GFLOP -- arithmetic intensity ~20
// flop/B, the MI355X ridge.  The persistent LDS-pipeline kernel of rounds 1-3 (tools/rejected/conv_stream32_lds_pipeline.hip)
// ran it at 0.37 of the HBM peak: four 256-position tiles per workgroup = ramp + four barrier-paced intervals + drain, every
// interval paced by the slower of its loader and MMA waves.  This kernel has no pipeline to fill and no workgroup barrier
// in the data path:
//   * ONE WAVE owns one 32-channel x 64-position output tile from its first load to its last store.  It fetches the 32 x 64
//     input tile with eight coalesced 16-byte loads per lane (every row piece is two whole cache lines) plus one 4-byte halo
//     load, applies GroupNorm+SiLU in registers, parks the tile in a wave-PRIVATE LDS region (the LDS runs one wave's
//     instructions in order: no barrier, adp_wave_sync is a compiler fence), reads it back in the matrix cores' fragment
//     shape, and stores straight from the accumulators.
//   * 16 waves per CU (four per SIMD, <= 128 registers) cover the whole [4, 32, 65536] problem in ONE resident generation of
//     4096 waves: the chip's own wave scheduler overlaps one wave's HBM latency with another's MFMAs and a third's stores;
//     nothing is synchronised, so nothing waits for the slowest participant.
//   * Winograd F(4,3) on v_mfma_f32_16x16x4_f32 in the wave's registers: column j of the MFMA tile is an output QUAD; per K step
//     (four input channels) the lane reads the six inputs around its quad (three 8-byte LDS reads), forms V = B^T d with 13 VALU
//     ops and issues 2 x 6 MFMAs on the six Winograd planes -- 96 MFMAs of 32 cycles per tile where F(2,3) needs 65 of 64 cycles
//     and the direct form 144.  The transformed weights U = G g are built once per workgroup into 24 KB of LDS (the one
//     workgroup barrier, reached while the first stage's tile loads are already in flight).
//   * y = A^T m per quad: 16-byte stores, 256 contiguous bytes per row; bias and residual as 16-byte operands.  GroupNorm partial
//     statistics of the output: shifted sums per 4-channel row quad (shift = a sample of the quad, so |mean| >> sigma costs no
//     digits), Chan-combined over the workgroup's waves.
//   * the four waves that share a SIMD start their tiles `gap` apart (wall-clock stagger, no barrier): tools/probe/tile_probe.
// Algorithmic bytes per launch: 4 * B * 32 * L * (2 + has_res) + 12 KB of weights.
#include <stdlib.h>
#include "adp_rt.h"
#include "adp.h"
#include "conv_internal.h"

// tools/probe/tile_probe.hip builds this file with -DADP_TILE_TRACE: every wave stamps its phase boundaries (100 MHz
// wall clock) and its hardware placement into d.ws; the product build compiles the stamps away.
#ifdef ADP_TILE_TRACE
#define WT_STAMP(k)                                                                                  \
  do {                                                                                               \
    if (lane == 0) reinterpret_cast<long long*>(d.ws)[((int64_t)blockIdx.x * NW + wave) * 8 + (k)] = \
        (k) == 7 ? (long long)__builtin_amdgcn_s_getreg(63492) : (long long)wall_clock64();          \
  } while (0)
#else
#define WT_STAMP(k)
#endif

namespace {

constexpr int WT_C = 32;            // channels in = channels out
constexpr int WT_KT = 3;
constexpr int WT_TN = 64;           // positions per wave tile (32 output pairs)
constexpr int WT_RS = WT_TN + 2;    // LDS row: index i <-> position n0 - 1 + i, i = 0 .. 65
constexpr int WT_XT = WT_C * WT_RS;  // floats of one wave's tile

// SiLU of two values: packed multiplies / adds around the two transcendental pairs (v_exp_f32, v_rcp_f32)
__device__ __forceinline__ f32x2 wt_silu2(f32x2 h) {
  const f32x2 t = h * -1.4426950408889634f;
  const f32x2 den = f32x2{adp_exp2(t[0]), adp_exp2(t[1])} + 1.0f;
  return h * f32x2{adp_rcp(den[0]), adp_rcp(den[1])};
}

constexpr int WT_US = WT_C * 2 * 16 * 6;  // floats of the transformed weights U[c][rb][m16][6]

// GNB: this launch is a data gradient whose output feeds the backward of SiLU(GroupNorm(gnb_x)); the epilogue also leaves that
//      backward's first stage (adp_conv_desc.gnb_ab): per channel and workgroup (NW tiles = 64 NW positions)
template <bool TR, int PRO, int NW, bool RES, bool GN, bool GNB = false>
__global__ __launch_bounds__(64 * NW) void conv_tile32_kernel(adp_conv_desc d, int tiles_per_b, int cfg) {
  // One LDS block, U first: its fragment reads (and the tile's) then take their K-step offsets as 16-bit immediates.
  //   U[c][rb][m][6] | (pa, pb) per input channel | statistics scratch [NW][8][2] | NW wave tiles [32][66]
  __shared__ __attribute__((aligned(16))) float lds[WT_US + 2 * WT_C + NW * 16 + NW * WT_XT];
  float* const us = lds;
  float* const pab = lds + WT_US;
  float* const gsh = pab + 2 * WT_C;
  float* const xs = gsh + NW * 16;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = adp_uniform(tid >> 6);  // scalar: everything addressed per tile sits on SGPR bases + 32-bit lane offsets
  const int j = lane & 15, kq = lane >> 4;  // MFMA 16x16x4: lane = (output row / quad column j, K index kq)
  const int L = (int)d.Lin;
  // the workgroup's NW tiles are consecutive and lie in one batch element (tiles_per_b % NW == 0)
  const int t = (int)blockIdx.x * NW + wave;
  const int b = t / tiles_per_b;
  const int n0 = (t - b * tiles_per_b) * WT_TN;
  const int64_t tbase = (int64_t)b * WT_C * L + n0;  // element (channel 0, position n0) of this tile
  const unsigned Lb = 4u * (unsigned)L;              // row pitch in bytes (32 rows < 2^32 bytes: eligibility)

  // ---- tile loads: lane -> quads q = lane + 64 i of the 32 x 16 body (row = q >> 4), one halo scalar.
  // Waves w, w + 4, w + 8, w + 12 of a workgroup share a SIMD (cyclic placement; tools/probe/tile_probe prints it).  On this
  // chip an exact-f32 MFMA stream leaves the other waves of its SIMD about five issue slots per 64-cycle MFMA
  // (tools/probe/alu_probe: a VALU op beside it takes 13 cycles instead of 5), so the kernel is bound by the instructions a
  // SIMD issues: everything around the MFMAs of a tile is written for instruction count (scalar bases, packed f32 ops, DPP
  // reductions).  Stage s = w >> 2 asks for its tile s * `gap` after the launch.
  const char* xt = reinterpret_cast<const char*>(d.x + tbase);
  const int stage = wave >> 2;
  const long long t_start = adp_clock();
  WT_STAMP(0);
  WT_STAMP(7);
  f32x4 rx[8];
  const int hi = lane >> 5, l31 = lane & 31;
  const int hrel = hi ? WT_TN : -1;  // right / left halo of row l31
  const bool hok = n0 + hrel >= 0 && n0 + hrel < L;
  float hx = 0.0f;
  const unsigned ldb = (unsigned)(lane >> 4) * Lb + 16u * (unsigned)(lane & 15);
  const unsigned hb = (unsigned)l31 * Lb + 4u * (unsigned)(hok ? hrel + 1 : 1);  // relative to position n0 - 1 (offsets are unsigned)
  auto load_tile = [&]() {
#pragma unroll
    for (int i = 0; i < 8; ++i) rx[i] = *reinterpret_cast<const f32x4*>(xt + (ldb + 4u * (unsigned)i * Lb));
    hx = *reinterpret_cast<const float*>(xt - 4 + hb);
  };
#ifdef ADP_TILE_TRACE
  const int elim = cfg >> 16;  // probe builds only: elimination runs (1: no MFMAs, 2: no stores / residual, 4: no tile loads)
#else
  constexpr int elim = 0;
#endif
  const int gap = cfg & 0xffff;  // stagger between the stages in 10 ns ticks
  if (elim & 4) {
#pragma unroll
    for (int i = 0; i < 8; ++i) rx[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  }
  // Who builds U and the GroupNorm constants: with staged waves (NW > 4) stages 1.. -- they have nothing else to do yet, and
  // stage 0 reaches the barrier with nothing but its tile loads in flight (a use of weight registers behind those loads would
  // make it wait for the tile first: vmcnt counts in order).  All weight loads of a thread are issued before the first use.
  constexpr bool STAGED = NW > 4;
  constexpr int NST = STAGED ? 64 * NW - 256 : 64 * NW, NIT = (WT_C * WT_C + NST - 1) / NST;
  const int sid = STAGED ? tid - 256 : tid;
  if (STAGED && stage == 0 && !(elim & 4)) load_tile();
  if (!STAGED || stage > 0) {
    float wv[NIT][3];
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int p = sid + k * NST;
#pragma unroll
      for (int q = 0; q < 3; ++q) wv[k][q] = p < WT_C * WT_C ? d.w[3 * p + q] : 0.0f;
    }
    float pg = 1.0f, pbt = 0.0f, pmean = 0.0f, prstd = 1.0f;
    if (PRO == 1 && sid < WT_C) {
      const int64_t sg = ((int64_t)b * d.groups + sid / (WT_C / (int)d.groups)) * 2;
      if (d.pro_gamma) pg = d.pro_gamma[sid];
      if (d.pro_beta) pbt = d.pro_beta[sid];
      pmean = d.pro_stats[sg];
      prstd = d.pro_stats[sg + 1];
    }
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int p = sid + k * NST;
      if (p < WT_C * WT_C) {
        // p = m * 32 + c reads w[m][c][0..2]; the data gradient (w[c][m][.], taps flipped) takes p = c * 32 + m.
        // Winograd F(4,3) weight transform G g, G = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1]
        const float a = wv[k][0], g1 = wv[k][1], e = wv[k][2];
        const float g0 = TR ? e : a, g2 = TR ? a : e;
        const int m = TR ? (p & 31) : (p >> 5), c = TR ? (p >> 5) : (p & 31);
        const float s02 = g0 + g2, q = 0.041666666666666664f * g0 + 0.16666666666666666f * g2;
        float* u = us + ((c * 2 + (m >> 4)) * 16 + (m & 15)) * 6;
        *reinterpret_cast<f32x2*>(u) = f32x2{0.25f * g0, -0.16666666666666666f * (s02 + g1)};
        *reinterpret_cast<f32x2*>(u + 2) = f32x2{-0.16666666666666666f * (s02 - g1), q + 0.08333333333333333f * g1};
        *reinterpret_cast<f32x2*>(u + 4) = f32x2{q - 0.08333333333333333f * g1, g2};
      }
    }
    if (PRO == 1 && sid < WT_C) {
      pab[2 * sid] = pg * prstd;
      pab[2 * sid + 1] = pbt - pmean * pg * prstd;
    }
  }
  WT_STAMP(1);
  adp_barrier_lds();  // (not __syncthreads(): its vmcnt(0) would hold every wave until stage 0's tiles have landed)
  if (!STAGED || stage > 0) {
    if (STAGED) adp_wait_until(t_start + (long long)stage * gap);
    if (!(elim & 4)) load_tile();
  }
  WT_STAMP(2);

  // ---- activate and park the tile in this wave's LDS region: index i of a row <-> position n0 - 1 + i
  // (Issue priority was tried both ways -- the multiplying stage first, and the activation above the MFMA loops: the latter
  // does bring "tile in LDS" forward from 7.7 / 10.3 / 12.8 us to 6.7 / 7.7 / 9.0 us for stages 1-3, but the four tiles of a
  // SIMD still end at the same time (its VALU + MFMA instruction total is what bounds it) and conv2 got 2 us slower in the
  // hipGraph microbench; no s_setprio is left in the kernel.)
  float* X = xs + wave * WT_XT;
  {
    float* o = X + (lane >> 4) * WT_RS + 4 * (lane & 15) + 1;  // + 4 i rows: immediates
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      f32x2 lo = f32x2{rx[i][0], rx[i][1]}, hp = f32x2{rx[i][2], rx[i][3]};
      if (PRO == 1) {
        const f32x2 ab = *reinterpret_cast<const f32x2*>(pab + 2 * ((lane >> 4) + 4 * i));
        lo = wt_silu2(lo * ab[0] + ab[1]);
        hp = wt_silu2(hp * ab[0] + ab[1]);
      }
      o[4 * i * WT_RS] = lo[0];
      *reinterpret_cast<f32x2*>(o + 4 * i * WT_RS + 1) = f32x2{lo[1], hp[0]};  // even index: 8-byte aligned
      o[4 * i * WT_RS + 3] = hp[1];
    }
    if (PRO == 1) {
      const f32x2 ab = *reinterpret_cast<const f32x2*>(pab + 2 * l31);
      const float h = fmaf(hx, ab[0], ab[1]);
      hx = h * adp_rcp(1.0f + adp_exp2(-1.4426950408889634f * h));
    }
    X[l31 * WT_RS + (hi ? WT_RS - 1 : 0)] = hok ? hx : 0.0f;  // zero padding is applied after the activation
  }
  adp_wave_sync();
  WT_STAMP(3);

  // ---- Winograd F(4,3) on v_mfma_f32_16x16x4_f32: column j of the MFMA tile is an output QUAD (positions n0 + 4j .. + 3),
  // the two 16-row blocks rb cover the 32 output channels, a K step is four input channels (lane kq holds channel 4 ks + kq).
  // Per K step a lane reads the six inputs around its quad (three 8-byte LDS reads), forms V = B^T d with 13 VALU ops and
  // issues 2 x 6 MFMAs on the six Winograd planes: 96 MFMAs of 32 cycles per tile against 65 of 64 cycles for F(2,3)
  // (144 of 64 cycles in the direct form).  fp32 error against fp64 1.2e-6 of the output's max norm (direct form 3.7e-7).
  f32x4 M[2][6];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int p = 0; p < 6; ++p) M[rb][p] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  const float* Xr = X + kq * WT_RS + 4 * j;    // + 4 ks rows
  const float* Ur = us + (kq * 32 + j) * 6;    // + 4 ks channels (192 floats each), + rb * 96
  if (!(elim & 1)) {
    f32x2 dn[3], un[2][3];
    auto frags = [&](int ks) {
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        dn[q] = *reinterpret_cast<const f32x2*>(Xr + 4 * ks * WT_RS + 2 * q);
        un[0][q] = *reinterpret_cast<const f32x2*>(Ur + 4 * ks * 192 + 2 * q);
        un[1][q] = *reinterpret_cast<const f32x2*>(Ur + 4 * ks * 192 + 96 + 2 * q);
      }
    };
    frags(0);
#pragma unroll
    for (int ks = 0; ks < WT_C / 4; ++ks) {
      f32x2 dc[3], uc[2][3];
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        dc[q] = dn[q];
        uc[0][q] = un[0][q];
        uc[1][q] = un[1][q];
      }
      if (ks + 1 < WT_C / 4) frags(ks + 1);  // requested before this step's MFMAs are issued
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
      for (int p = 0; p < 6; ++p) {
        M[0][p] = adp_mfma16(uc[0][p >> 1][p & 1], v[p], M[0][p]);
        M[1][p] = adp_mfma16(uc[1][p >> 1][p & 1], v[p], M[1][p]);
      }
      adp_sched_fence();
    }
  }

  WT_STAMP(4);
  // ---- epilogue: accumulator register r of block rb <-> output channel 16 rb + 4 kq + r, column j = output quad;
  // y = A^T m, A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]; 16-byte stores, 256 contiguous bytes per row
  char* ot = reinterpret_cast<char*>(d.out + tbase);
  const char* rt = reinterpret_cast<const char*>(RES ? d.res + tbase : d.out + tbase);
  const unsigned ob = 4u * (unsigned)kq * Lb + 16u * (unsigned)j;  // lane part; rows add scalar multiples of the pitch
  const bool has_res = RES && !(elim & 2);
  constexpr bool want_gn = GN;
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
    f32x4 rv[4], bv = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    if (has_res) {
#pragma unroll
      for (int r = 0; r < 4; ++r) rv[r] = *reinterpret_cast<const f32x4*>(rt + (ob + (unsigned)(16 * rb + r) * Lb));
    }
    if (d.bias) bv = *reinterpret_cast<const f32x4*>(d.bias + 16 * rb + 4 * kq);
    f32x4 gxq[4], gga = bv, gbe = bv;
    if (GNB) {
      const char* gxt = reinterpret_cast<const char*>(d.gnb_x + tbase);
#pragma unroll
      for (int r = 0; r < 4; ++r) gxq[r] = *reinterpret_cast<const f32x4*>(gxt + (ob + (unsigned)(16 * rb + r) * Lb));
      gga = *reinterpret_cast<const f32x4*>(d.gnb_gamma + 16 * rb + 4 * kq);
      gbe = *reinterpret_cast<const f32x4*>(d.gnb_beta + 16 * rb + 4 * kq);
    }
    float gk = 0.0f;
    f32x2 gs = f32x2{0.0f, 0.0f}, gq = f32x2{0.0f, 0.0f};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float m0 = M[rb][0][r], m1 = M[rb][1][r], m2 = M[rb][2][r], m3 = M[rb][3][r], m4 = M[rb][4][r], m5 = M[rb][5][r];
      const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
      f32x2 ya = f32x2{m0 + s12 + s34, fmaf(2.0f, d34, d12)} + bv[r];
      f32x2 yb = f32x2{fmaf(4.0f, s34, s12), fmaf(8.0f, d34, d12) + m5} + bv[r];
      if (has_res) {
        ya = ya + f32x2{rv[r][0], rv[r][1]};
        yb = yb + f32x2{rv[r][2], rv[r][3]};
      }
      if (!(elim & 2))
        *reinterpret_cast<f32x4*>(ot + (ob + (unsigned)(16 * rb + r) * Lb)) = f32x4{ya[0], ya[1], yb[0], yb[1]};
      if (GNB) {  // (sum ds * xhat, sum ds) of channel 16 rb + 4 kq + r over the wave's 64 positions -> the wave's (dead) tile region
        const int c = 16 * rb + 4 * kq + r;
        const float* st = d.gnb_stats + ((int64_t)b * d.gnb_groups + c / (WT_C / (int)d.gnb_groups)) * 2;
        const float mu = st[0], rs = st[1];
        const float ga = gga[r] * rs, be = gbe[r] - mu * ga;
        const float yv[4] = {ya[0], ya[1], yb[0], yb[1]};
        float sa = 0.0f, sb = 0.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float xh = (gxq[r][k] - mu) * rs;
          const float ds = yv[k] * adp_dsilu_fast(fmaf(gxq[r][k], ga, be));
          sa = fmaf(ds, xh, sa);
          sb += ds;
        }
        sa = adp_row16_sum(sa), sb = adp_row16_sum(sb);
        if (j == 0) *reinterpret_cast<f32x2*>(xs + wave * WT_XT + 2 * c) = f32x2{sa, sb};
      }
      if (want_gn) {
        if (r == 0) gk = __shfl(ya[0], lane & 48, 64);  // shift of this row quad: its first value, lane j = 0 of the kq row
        const f32x2 ea = ya - gk, eb = yb - gk;
        gs = gs + ea + eb;
        gq = ea * ea + (eb * eb + gq);
      }
    }
    if (want_gn) {
      // one (mean, M2) per (wave, row quad 4 rb + kq): the quad's 4 channels are this lane's registers, its 64 positions the
      // 16 lanes of the kq row
      constexpr float cnt = 4.0f * WT_TN;
      const float sv = adp_row16_sum(gs[0] + gs[1]), qv = adp_row16_sum(gq[0] + gq[1]);
      if (j == 0) {
        float* e = gsh + (wave * 8 + 4 * rb + kq) * 2;
        e[0] = gk + sv / cnt;
        e[1] = fmaxf(qv - sv * (sv / cnt), 0.0f);
      }
    }
  }
  WT_STAMP(5);
  if (GNB) {  // the workgroup's NW waves in wave order -> one entry per channel and workgroup
    __syncthreads();
    if (tid < WT_C) {
      float sa = 0.0f, sb = 0.0f;
      for (int w = 0; w < NW; ++w) {
        const f32x2 p = *reinterpret_cast<const f32x2*>(xs + w * WT_XT + 2 * tid);
        sa += p[0], sb += p[1];
      }
      const int E = tiles_per_b / NW;
      *reinterpret_cast<f32x2*>(d.gnb_ab + (((int64_t)b * WT_C + tid) * E + ((int)blockIdx.x - b * E)) * 2) = f32x2{sa, sb};
    }
  }
  if (want_gn) {
    // Chan-combined over the workgroup's NW waves -> one gn_part entry per row quad
    constexpr float cnt = 4.0f * WT_TN;
    __syncthreads();
    if (tid < 8) {
      float mean = gsh[tid * 2], m2 = gsh[tid * 2 + 1], n = cnt;
      for (int w = 1; w < NW; ++w) {
        const float mw = gsh[(w * 8 + tid) * 2], dl = mw - mean, nn = n + cnt;
        mean += dl * (cnt / nn);
        m2 += gsh[(w * 8 + tid) * 2 + 1] + dl * dl * (n * cnt / nn);
        n = nn;
      }
      const int E = tiles_per_b / NW;
      float* e = d.gn_part + (((int64_t)b * (WT_C / 4) + tid) * E + ((int)blockIdx.x - b * E)) * 3;
      e[0] = mean;
      e[1] = m2;
      e[2] = n;
    }
  }
}

// waves per workgroup: 16 (one resident generation of 4096 waves at [4, 32, 65536]) down to 1 for short rows / small
// grids; the workgroup's tiles must lie in one batch element
static int tile_nw(const adp_conv_desc& d) {
  const int64_t tiles_per_b = d.N / WT_TN, tiles = tiles_per_b * d.B;
  const char* e = getenv("ADP_TILE_NW");  // tests: every variant on small problems
  const int want = e ? atoi(e) : 16;
  for (int nw : {16, 4})
    if (nw <= want && tiles_per_b % nw == 0 && (e || tiles / nw >= 256)) return nw;
  return 1;
}

template <bool TR, int PRO, bool RES, bool GN>
int launch_tile(const adp_conv_desc& d, void* stream) {
  const int tiles_per_b = (int)(d.N / WT_TN);
  const int nw = tile_nw(d);
  const unsigned grid = (unsigned)(d.B * tiles_per_b / nw);
  int cfg = 120;  // stagger between the wave stages: 1.2 us
  if (const char* e = getenv("ADP_TILE_CFG")) cfg = atoi(e);  // kernel work: gap | elimination bits << 16 (probe builds)
  switch (nw) {
    case 16: ADP_LAUNCH((conv_tile32_kernel<TR, PRO, 16, RES, GN>), dim3(grid), dim3(1024), stream, d, tiles_per_b, cfg); break;
    case 4: ADP_LAUNCH((conv_tile32_kernel<TR, PRO, 4, RES, GN>), dim3(grid), dim3(256), stream, d, tiles_per_b, cfg); break;
    default: ADP_LAUNCH((conv_tile32_kernel<TR, PRO, 1, RES, GN>), dim3(grid), dim3(64), stream, d, tiles_per_b, cfg); break;
  }
  return ADP_LAUNCH_OK();
}

static bool tile_gnb_ok(const adp_conv_desc& d) {
  return d.transposed && d.prologue == 0 && !d.res && !d.gn_part && !d.bias && WT_C % (d.gnb_groups > 0 ? d.gnb_groups : 1) == 0 &&
         ((reinterpret_cast<uintptr_t>(d.gnb_x) | reinterpret_cast<uintptr_t>(d.gnb_gamma) | reinterpret_cast<uintptr_t>(d.gnb_beta)) & 15) == 0;
}

int launch_tile_gnb(const adp_conv_desc& d, void* stream) {
  const int tiles_per_b = (int)(d.N / WT_TN);
  const int nw = tile_nw(d);
  const unsigned grid = (unsigned)(d.B * tiles_per_b / nw);
  const int cfg = 120;
  switch (nw) {
    case 16: ADP_LAUNCH((conv_tile32_kernel<true, 0, 16, false, false, true>), dim3(grid), dim3(1024), stream, d, tiles_per_b, cfg); break;
    case 4: ADP_LAUNCH((conv_tile32_kernel<true, 0, 4, false, false, true>), dim3(grid), dim3(256), stream, d, tiles_per_b, cfg); break;
    default: ADP_LAUNCH((conv_tile32_kernel<true, 0, 1, false, false, true>), dim3(grid), dim3(64), stream, d, tiles_per_b, cfg); break;
  }
  return ADP_LAUNCH_OK();
}

template <bool TR, int PRO>
int launch_tile2(const adp_conv_desc& d, void* stream) {
  if (d.res) return d.gn_part ? launch_tile<TR, PRO, true, true>(d, stream) : launch_tile<TR, PRO, true, false>(d, stream);
  return d.gn_part ? launch_tile<TR, PRO, false, true>(d, stream) : launch_tile<TR, PRO, false, false>(d, stream);
}

}  // namespace

bool adp_conv_tile_eligible(const adp_conv_desc& d) {
  if (d.R != WT_C || d.R1 != d.R || d.M != WT_C || d.KT != WT_KT) return false;
  if (d.stride != 1 || d.dil != 1 || d.pad != 1 || d.up != 1 || d.store != 0) return false;
  if (d.out_pre || d.e_scale || d.x2) return false;
  if (d.prologue != 0 && d.prologue != 1) return false;
  if (d.prologue == 1 && (d.groups < 1 || WT_C % d.groups != 0)) return false;
  if (d.N != d.Lin || d.N % WT_TN != 0) return false;
  if (reinterpret_cast<uintptr_t>(d.x) & 15) return false;
  if ((reinterpret_cast<uintptr_t>(d.out) | reinterpret_cast<uintptr_t>(d.res) | reinterpret_cast<uintptr_t>(d.bias)) & 15)
    return false;  // 16-byte epilogue accesses
  if (d.B * (d.N / WT_TN) >= (int64_t)1 << 30 || d.B * WT_C * d.Lin >= (int64_t)1 << 40) return false;
  if ((WT_C + 1) * d.Lin * 4 >= (int64_t)1 << 32) return false;  // a tile's rows are addressed by 32-bit byte offsets
  return true;
}

int64_t adp_conv_tile_gn_entries(const adp_conv_desc& d) { return d.N / WT_TN / tile_nw(d); }
// slices per row of gnb_ab: one per workgroup (plain data gradients only; gnb_x / gamma / beta 16-byte aligned)
int64_t adp_conv_tile_gnb_entries(const adp_conv_desc& d) { return tile_gnb_ok(d) && adp_gnb_family_on(16) ? d.N / WT_TN / tile_nw(d) : 0; }

int adp_conv_tile(const adp_conv_desc& d, void* stream) {
  if (d.gnb_ab) return launch_tile_gnb(d, stream);  // (adp_conv1d has checked adp_conv_tile_gnb_entries)
  if (d.transposed) return d.prologue == 1 ? launch_tile2<true, 1>(d, stream) : launch_tile2<true, 0>(d, stream);
  return d.prologue == 1 ? launch_tile2<false, 1>(d, stream) : launch_tile2<false, 0>(d, stream);
}
