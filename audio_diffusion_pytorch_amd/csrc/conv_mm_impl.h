// Implicit-GEMM Conv1d kernel for gfx950: kernel 1 / 3 at stride 1 (plus the kernel = stride = 2 / 4 DownsampleItem
// and the nearest-upsample UpsampleItem variants), channel counts that are multiples of 32.
// This is the kernel the MFMA-bound half of the U-Net lives in (ResnetItem ConvBlocks and their data
// gradients at depths 1-8; /root/reference/audio_diffusion_pytorch/components.py:89).
//
// Shape of the machine it is written for:
//   * v_mfma_f32_32x32x2_f32 retires one 32x32x2 tile per 64 cycles per SIMD, so operand traffic is tiny and the
//     only thing that matters is that every SIMD always has an MFMA to issue.  Measured on MI355X
//     (tools/probe/mfma_probe.hip): a loop of "LDS fragment read + MFMA, one barrier per 12-24 MFMAs" sustains
//     128-138 TF of the 157 TF peak, but the same loop with the operand staging (global load -> GroupNorm+SiLU ->
//     ds_write) in the SAME waves ran at 80 TF: the staging phase and the MFMA phase of a chunk did not overlap
//     (staging alone 0.84 us per chunk, MFMAs alone 1.7 us, together 2.3 us).
//   * so the block is WAVE-SPECIALISED: NLD = 4 loader waves (one per SIMD) own the global -> register ->
//     (prologue) -> LDS path and never touch the matrix cores; the MMA waves only read fragments and issue MFMAs.
//     While the MMA waves are inside chunk c, the loaders write chunk c+1 into the other LDS buffer; one
//     workgroup barrier per chunk hands the buffers over:
//         loader:  store chunk c -> LDS[c&1] ; issue global loads of chunk c+PD ; barrier B_c
//         MMA   :  barrier B_c ; MFMAs over LDS[c&1]
//     (LDS[(c+1)&1] is rewritten by the loaders during iteration c+1, after every MMA wave has passed B_{c+1},
//     i.e. finished chunk c-1's sibling buffer... see the hazard note at the loops).
//   * an MMA wave owns a 32 x 64 output tile (two accumulator tiles sharing the A fragment).  Deep layers have few
//     output tiles (depth 7 at batch 4: 256 tiles of 64x64 for 256 CUs), so the MMA waves of a block split K:
//     NKG = BKT/8 wave groups each take 8 of the BKT channels of every staged chunk; the partial tiles meet in LDS
//     at the end and every MMA wave sums + stores a quarter of the rows (fixed order: deterministic).
//   * weights are copied as they lie in memory:
//       forward   As[m][r*KT+t]  (row = BKT channels x KT taps, contiguous in w[M][R][KT]); the A fragment
//                 of lane (m, hi) is 4 channels x KT taps = KT ds_read_b128 (row stride = 4 mod 8 dwords: no
//                 bank conflicts);
//       gradient  As[k][m*KT+t]  (row = BM outputs x KT taps, contiguous in w[R][M][KT]); fragment reads are
//                 stride-KT ds_read_b32 (conflict-free for KT = 1, 3).
//     The MFMA K pair is (channel c + 4*hi), which both layouts share.
//   * the GroupNorm+SiLU prologue is applied by the loaders between the global load and the LDS store, from
//     per-(b,channel) constants kept in LDS; zero padding is applied after the activation, like nn.Conv1d.
//   * the 1-D grid is decoded XCD-aware: workgroup id -> XCD id%8 (round-robin dispatch), so ids are permuted
//     to give each XCD a contiguous range of weight row tiles; its 4 MiB L2 then holds 1/8 of the weights.
#pragma once
#include <stdlib.h>
#include "adp_rt.h"
#include "adp.h"
#include "conv_internal.h"

namespace {

constexpr int MM_PRO_RMAX = 1024; // channels whose GroupNorm constants fit the LDS table
constexpr int MM_BN = 64;         // output positions per block
// loader waves per block: the GroupNorm+SiLU prologue is VALU work (two transcendentals per element) on the
// loaders' critical path, so those variants get twice the loaders -- except the one-chunk 32-row blocks of the
// HBM-bound shallow layers, where more resident blocks per CU matter more (measured: depth 1, 54 vs 63 us)
// (the wide-N blocks, NSP > 1, keep 4: their 8 MMA waves need the 168-register budget of a 12-wave block)
constexpr int mm_nld(int PRO, int BM, int PD, int NSP = 1) { return (PRO == 1 && NSP == 1 && !(BM == 32 && PD == 1)) ? 8 : 4; }
// K groups (MMA wave groups that split the channels of a staged chunk): 8 channels each up to BKT = 32; a 64-channel
// chunk (half the barriers per K) keeps 4 groups of 16 channels
constexpr int mm_nkg(int BKT) { return BKT >= 32 ? 4 : BKT / 8; }

// four consecutive virtual positions u0..u0+3 (u0 % 4 == 0) of a row upsampled by UP: 4 / 2 / 1 source floats
template <int UP>
__device__ __forceinline__ f32x4 load_xquad(const float* p) {
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

template <int A, int B>
struct cmax {
  static constexpr int v = A > B ? A : B;
};

// BM: output channels per block (32 / 64); S: conv stride (1, or kernel = stride = 2 / 4 for DownsampleItem);
// UP: nearest-upsample factor folded into the X loader (UpsampleItem: the [B, C, L*UP] intermediate is never
// materialised); BKT: channels per staged chunk; PD: loader prefetch distance in chunks (register stages).
//
// WN = true (kernel 3, stride 1, pad 1, dil 1; store modes 0 and 2): the same block, loaders and LDS tiles, but the MMA waves
// multiply in the Winograd F(2,3) domain.  Column l31 of a wave's tile is an output PAIR (positions n0 + 2*l31, +1);
// per input channel the lane reads the three taps g and the four inputs d = x[2j-1 .. 2j+2] around its pair, forms
//     U = (g0, g0+g1+g2, g0-g1+g2, g2)        V = (d0-d2, d1+d2, d2-d1, d1-d3)
// in registers (7 VALU adds, issued in the shadow of the 64-cycle MFMAs) and issues FOUR MFMAs -- one per Winograd
// plane, four accumulator tiles -- where the direct form issues six.  After the K loop
//     y0 = P0 + (P1 + P2) / 2      y1 = (P1 - P2) / 2 - P3
// turns the planes into the two output tiles (the halves of G are applied here, once, instead of per weight).
// Two thirds of the matrix work of the direct form in plain fp32 arithmetic; the loaders stay pure copies, so the
// overlap of staging and MFMAs that conv_mm measures is kept (the first Winograd kernel of this repository
// transformed in the loader waves and lost its MFMA saving to exactly that: DESIGN.md section 4).
//
// NSP > 1 (2 / 4; short K, long N: the mid-depth layers): the block covers NSP adjacent 64-position tiles and the MMA
// waves trade K groups for positions (NKG = 4 / NSP), so a block stages the same weight chunk once for NSP times the
// outputs, half / none of the K-group exchange remains, and a wave issues NSP times the MFMAs per barrier.
// GNB: the launch also leaves the first stage of a GroupNorm backward (adp_conv_desc.gnb_ab): its own instantiations (plain
//      Winograd data gradients without the K split); every other launch compiles as it did without it
template <int BM, int KT, int S, int UP, bool TR, int PRO, int BKT, int PD, bool WN = false, int NSP = 1, bool PF = false,
          bool GNB = false>
__global__ __launch_bounds__(((BM / 32) * mm_nkg(BKT) + mm_nld(PRO, BM, PD, NSP)) * 64) void conv_mm_kernel(
    adp_conv_desc d, int KS) {
  static_assert(!WN || (KT == 3 && S == 1), "Winograd F(2,3): kernel 3, stride 1 (any upsample factor: the LDS tile "
                                            "holds virtual positions)");
  static_assert(NSP == 1 || (mm_nkg(BKT) % NSP == 0 && S == 1), "wide-N blocks: stride-1 convs");
  constexpr int MM_NLD = mm_nld(PRO, BM, PD, NSP);
  constexpr int BN = MM_BN * NSP, NKG = mm_nkg(BKT) / NSP, NQM = BM / 32;
  constexpr int CPK = BKT / NKG;                    // channels of a chunk one K group multiplies (8, 16 or 32)
  constexpr int NMMA = NQM * NKG * NSP;             // MMA waves
  constexpr int NLT = MM_NLD * 64;                  // loader threads
  constexpr int QK = BKT * KT;
  constexpr int AS = TR ? (BM * KT + 4) : (QK + 4);  // A row stride in floats
  constexpr int AROWS = TR ? BKT : BM;
  constexpr int AQ = (TR ? BM * KT : QK) / 4;        // float4 per A row
  constexpr int XSP = BN * S + 8, XQ = XSP / 4;      // X row: (virtual) positions n0*S-4 .. n0*S+BN*S+3
  constexpr int A_ELEMS = AROWS * AS, X_ELEMS = BKT * XSP;
  constexpr int NA4 = (AROWS * AQ + NLT - 1) / NLT, NX4 = (BKT * XQ + NLT - 1) / NLT;
  constexpr int RED = NMMA * 2048;                   // every MMA wave parks its two accumulator tiles
  constexpr int SM = cmax<2 * (A_ELEMS + X_ELEMS), RED>::v;
  static_assert(BKT % 8 == 0 && NKG >= 1 && NKG <= 4 && CPK % 8 == 0, "8 or 16 channels per K group");
  __shared__ __attribute__((aligned(16))) float smem[SM];
  __shared__ float Pa[PRO == 1 ? MM_PRO_RMAX : 1], Pb[PRO == 1 ? MM_PRO_RMAX : 1];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hi = lane >> 5, l31 = lane & 31;

  const int M = (int)d.M, R = (int)d.R, L = (int)d.Lin, N = (int)d.N;
  const int dil = (int)d.dil, pad = (int)d.pad;
  const int Lv = L * UP;  // length of the (virtual) upsampled row

  // ---- XCD-aware decode of the 1-D grid
  int id = blockIdx.x;
  const int total = gridDim.x;
  if ((total & 7) == 0) id = (id & 7) * (total >> 3) + (id >> 3);
  const int ntn = (N + BN - 1) / BN, per_m = ntn * (int)d.B;
  const int mt = id / per_m, rem = id - mt * per_m;
  const int b = rem / ntn, nt = rem - b * ntn;
  const int m0 = mt * BM, n0 = nt * BN;
  // Cross-workgroup K split (grid.y = KS > 1: small grids, i.e. the deep layers at batch 1): this block reduces
  // channel chunks [c_lo, c_lo + nchunks) only and parks its raw partial tile in d.ws[ks]; the epilogue (bias,
  // e_scale, residual) then runs in conv_splitk_reduce_kernel, which sums the KS partials in a fixed order.
  const int ks = blockIdx.y;
  const int cps = (R / BKT + KS - 1) / KS;
  const int c_lo = ks * cps;
  const int nchunks = (R / BKT - c_lo) < cps ? (R / BKT - c_lo) : cps;
  const int nrounds = ((nchunks + PD - 1) / PD) * PD;  // ghost iterations (barrier only) pad the loop to PD

  if (PRO == 1) {
    const int cpg = R / (int)d.groups;
    for (int r = tid; r < R; r += (NMMA + MM_NLD) * 64) {
      const int g = r / cpg;
      const float mean = d.pro_stats[((int64_t)b * d.groups + g) * 2];
      const float ga = (d.pro_gamma ? d.pro_gamma[r] : 1.0f) * d.pro_stats[((int64_t)b * d.groups + g) * 2 + 1];
      Pa[r] = ga;
      Pb[r] = (d.pro_beta ? d.pro_beta[r] : 0.0f) - mean * ga;
    }
  }

  if (wave >= NMMA) {
    // =========================== loader waves ===========================
    const int lt = tid - NMMA * 64;
    const float* xb = d.x + (int64_t)b * R * L;
    const float* wbase = TR ? d.w + (int64_t)m0 * KT : d.w + (int64_t)m0 * R * KT;
    // per-thread staging slots (chunk independent parts).  Slot indices wrap around instead of being guarded: a
    // few threads then stage the same 16 bytes twice, and the loop body stays branch-free.
    int a_src[NA4], a_dst[NA4];
#pragma unroll
    for (int i = 0; i < NA4; ++i) {
      const int e = (lt + i * NLT) % (AROWS * AQ);
      const int row = e / AQ, qq = e - row * AQ;
      a_dst[i] = row * AS + 4 * qq;
      a_src[i] = TR ? row * M * KT + 4 * qq : row * R * KT + 4 * qq;
    }
    int x_src[NX4], x_dst[NX4], x_row[NX4];
    bool x_ok[NX4];
#pragma unroll
    for (int i = 0; i < NX4; ++i) {
      const int e = (lt + i * NLT) % (BKT * XQ);
      const int rl = e / XQ, pq = e - rl * XQ;
      const int u = n0 * S - 4 + 4 * pq;
      x_dst[i] = rl * XSP + 4 * pq;
      x_ok[i] = (u >= 0 && u < Lv);  // Lv % 4 == 0: a quad is entirely inside or outside the row
      x_src[i] = rl * L + (x_ok[i] ? u / UP : 0);  // nearest upsample: source index = floor(u / UP), exact
      x_row[i] = rl;
    }
    // Chunk k travels  global -> register stage k % PD -> LDS[k & 1].  Loads are unconditional (the tail re-reads
    // the last chunk, which is never consumed) so that the number of loads in flight is a compile-time constant
    // and the compiler waits with vmcnt(loads of the younger stages) instead of vmcnt(0).
    f32x4 ra[PD][NA4], rx[PD][NX4];
    auto load_chunk = [&](f32x4 (&a)[NA4], f32x4 (&x)[NX4], int chunk) {
      const int rn = (c_lo + (chunk < nchunks ? chunk : nchunks - 1)) * BKT;
      const float* wp = TR ? wbase + (int64_t)rn * M * KT : wbase + rn * KT;
#pragma unroll
      for (int i = 0; i < NA4; ++i) a[i] = *reinterpret_cast<const f32x4*>(wp + a_src[i]);
      const float* xp = xb + (int64_t)rn * L;
#pragma unroll
      for (int i = 0; i < NX4; ++i) x[i] = load_xquad<UP>(xp + x_src[i]);
    };
    auto store_chunk = [&](const f32x4 (&a)[NA4], const f32x4 (&x)[NX4], int chunk) {
      const int r0 = (c_lo + (chunk < nchunks ? chunk : nchunks - 1)) * BKT;
      float* Ab = smem + (chunk & 1) * (A_ELEMS + X_ELEMS);
      float* Xb = Ab + A_ELEMS;
#pragma unroll
      for (int i = 0; i < NA4; ++i) *reinterpret_cast<f32x4*>(Ab + a_dst[i]) = a[i];
#pragma unroll
      for (int i = 0; i < NX4; ++i) {
        f32x4 v = x[i];
        if (PRO == 1) {
          const float pa = Pa[r0 + x_row[i]], pb = Pb[r0 + x_row[i]];
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = adp_silu_fast(fmaf(v[j], pa, pb));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = x_ok[i] ? v[j] : 0.0f;  // zero padding is applied after the activation
        *reinterpret_cast<f32x4*>(Xb + x_dst[i]) = v;
      }
    };
#pragma unroll
    for (int s = 0; s < PD; ++s) load_chunk(ra[s], rx[s], s);
    if (PRO == 1) __syncthreads();  // Pa / Pb complete
    // Hazard note: the store of chunk c goes to LDS[c & 1], last read by the MFMAs of chunk c-2; every MMA wave
    // finished those before it arrived at barrier B_{c-1}, which this wave passed before starting iteration c.
    for (int c0 = 0; c0 < nrounds; c0 += PD) {
#pragma unroll
      for (int s = 0; s < PD; ++s) {
        store_chunk(ra[s], rx[s], c0 + s);   // ghost chunks (>= nchunks) restage the last chunk: never consumed
        load_chunk(ra[s], rx[s], c0 + s + PD);
        __syncthreads();                     // B_c
      }
    }
    __syncthreads();  // partial tiles parked
    __syncthreads();  // (pairs with the barrier after the K-group exchange below)
    return;
  }

  // =========================== MMA waves ===========================
  const int mq = wave % NQM, kg = (wave / NQM) % NKG, nq = wave / (NQM * NKG);
  const int wm0 = mq * 32;
  const int nw0 = n0 + nq * MM_BN;  // first output position of this wave's 64-position tile
  constexpr int NACC = WN ? 4 : 2;
  f32x16 acc[NACC];
#pragma unroll
  for (int ni = 0; ni < NACC; ++ni)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[ni][r] = 0.0f;

  // WN epilogue operands of the rows this wave finishes (accumulator registers kg * RPW ..): fetched NOW, so that their
  // global-memory latency lies under the K loop instead of on the launch's tail (every block of a launch reaches its
  // epilogue at the same time: elimination build, depth 7 forward with residual: 8.8 of 55 us)
  // (not the NKG = 1 wide-N blocks: all 16 rows of a lane would cost 64 registers; their K loop is two to four chunks)
  constexpr int RPW_ = 16 / NKG;
  constexpr bool PRE = WN && RPW_ <= 8;
  f32x2 pre_res[PRE ? RPW_ : 1];
  float pre_bias[PRE ? RPW_ : 1], pre_scale[PRE ? RPW_ : 1];
  if constexpr (PRE) {
    const int n = nw0 + 2 * l31;
#pragma unroll
    for (int rr = 0; rr < RPW_; ++rr) {
      const int r = kg * RPW_ + rr;
      const int m = m0 + wm0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      const bool ok = (m < M) && (n < N);
      const int mc = m < M ? m : M - 1;
      pre_bias[rr] = d.bias ? d.bias[mc] : 0.0f;
      pre_scale[rr] = d.e_scale ? d.e_scale[b * (d.e_bstride ? d.e_bstride : M) + mc] : 1.0f;
      pre_res[rr] = f32x2{0.0f, 0.0f};
      if (d.res && ok && KS == 1 && d.store == 0) pre_res[rr] = *reinterpret_cast<const f32x2*>(d.res + ((int64_t)b * M + m) * N + n);
    }
  }

  // lane-constant fragment offsets
  const int xfrag = nq * MM_BN * S +
                    (WN ? 4 * hi * XSP + 2 * l31 + 2                  // 8-byte pieces at +0, +2, +4: x[2j-2 .. 2j+3]
                        : 4 * hi * XSP + l31 * S + 4 - pad);          // + ni*32*S + (ci + c) * XSP + t * dil
  const int afrag = TR ? 4 * hi * AS + (wm0 + l31) * KT                // + (ci + c) * AS + (KT - 1 - t)
                       : (wm0 + l31) * AS + 4 * hi * KT;               // + ci * KT + (c * KT + t)
  if (PRO == 1) __syncthreads();
  if constexpr (PF) {
    // ---- fragment reads one CHUNK ahead (8-wave blocks: the register file has room for a second fragment set): after barrier
    // B_c the wave REQUESTS chunk c's fragments and multiplies chunk c-1 from the other register set, so the LDS latency of a
    // chunk lies under the previous chunk's MFMAs instead of between a barrier and the first MFMA (one MMA wave per SIMD here:
    // nothing else hides it; tools/probe wn_probe: 93 -> 62-66 cycles per MFMA).  Same barriers, same loader code: chunk c's
    // reads are complete when the wave arrives at B_{c+1} (the barrier's own lgkmcnt(0)), which is what the loaders' store of
    // chunk c+2 into the same buffer waits for.
    static_assert(WN && CPK == 8 && NSP == 1, "chunk-ahead fragments: one K-group step per chunk");
    auto rd = [&](int c, float (&av)[4 * KT], f32x2 (&px)[4][3]) {
      const float* Ab = smem + (c & 1) * (A_ELEMS + X_ELEMS);
      const float* Xb = Ab + A_ELEMS;
      const int ci = kg * CPK;
      if (!TR) {
        const float* ap = Ab + afrag + ci * KT;
#pragma unroll
        for (int j = 0; j < KT; ++j) {
          const f32x4 q = *reinterpret_cast<const f32x4*>(ap + 4 * j);
#pragma unroll
          for (int k = 0; k < 4; ++k) av[4 * j + k] = q[k];
        }
      } else {
#pragma unroll
        for (int cc = 0; cc < 4; ++cc)
#pragma unroll
          for (int t = 0; t < KT; ++t) av[cc * KT + t] = Ab[afrag + (ci + cc) * AS + (KT - 1 - t)];
      }
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        const float* xp = Xb + xfrag + (ci + cc) * XSP;
#pragma unroll
        for (int q = 0; q < 3; ++q) px[cc][q] = *reinterpret_cast<const f32x2*>(xp + 2 * q);
      }
    };
    auto mm = [&](const float (&av)[4 * KT], const f32x2 (&px)[4][3]) {
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        const float d0 = px[cc][0][1], d1 = px[cc][1][0], d2 = px[cc][1][1], d3 = px[cc][2][0];
        const float g0 = av[cc * KT], g1 = av[cc * KT + 1], g2 = av[cc * KT + 2];
        const float gs = g0 + g2;
        acc[0] = adp_mfma32(g0, d0 - d2, acc[0]);
        acc[1] = adp_mfma32(gs + g1, d1 + d2, acc[1]);
        acc[2] = adp_mfma32(gs - g1, d2 - d1, acc[2]);
        acc[3] = adp_mfma32(g2, d1 - d3, acc[3]);
      }
      // the outer floats of the 24-byte windows are never used: keep their registers occupied until here, or the register
      // allocator hands them out as temporaries while the read that fills them is still in flight (a wait in mid-chunk)
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) adp_keep(px[cc][0][0], px[cc][2][1]);
    };
    float avA[4 * KT], avB[4 * KT];
    f32x2 pxA[4][3], pxB[4][3];
    for (int c = 0; c < nrounds; c += 2) {  // (ghost chunks re-read the last staged chunk: never multiplied)
      __syncthreads();  // B_c
      rd(c, avA, pxA);
      if (c >= 1 && c - 1 < nchunks) mm(avB, pxB);
      if (c + 1 < nrounds) {
        __syncthreads();  // B_{c+1}
        rd(c + 1, avB, pxB);
        if (c < nchunks) mm(avA, pxA);
      }
    }
    if (nrounds - 1 < nchunks) {
      if ((nrounds - 1) & 1) mm(avB, pxB);
      else mm(avA, pxA);
    }
  } else
  for (int c = 0; c < nrounds; ++c) {
    __syncthreads();  // B_c: chunk c is in LDS[c & 1]
    if (c < nchunks) {
      const float* Ab = smem + (c & 1) * (A_ELEMS + X_ELEMS);
      const float* Xb = Ab + A_ELEMS;
#pragma unroll
      for (int sub = 0; sub < CPK / 8; ++sub) {
        const int ci = kg * CPK + sub * 8;
        if constexpr (WN) {
          float av[4 * KT];  // av[cc * 3 + t] = tap t of channel ci + cc + 4 * hi for this lane's output row
          if (!TR) {
            const float* ap = Ab + afrag + ci * KT;
#pragma unroll
            for (int j = 0; j < KT; ++j) {
              const f32x4 q = *reinterpret_cast<const f32x4*>(ap + 4 * j);
#pragma unroll
              for (int k = 0; k < 4; ++k) av[4 * j + k] = q[k];
            }
          } else {
#pragma unroll
            for (int cc = 0; cc < 4; ++cc)
#pragma unroll
              for (int t = 0; t < KT; ++t) av[cc * KT + t] = Ab[afrag + (ci + cc) * AS + (KT - 1 - t)];
          }
          // every fragment read of the K group's four channel pairs is issued before the first MFMA (36 registers): the
          // LDS latency is paid once per chunk instead of once per channel pair (elimination build: the barrier +
          // fragment-read + transform skeleton of a depth-7 launch was 16 of 55 us, not overlapped with the MFMAs)
          f32x2 px[4][3];
#pragma unroll
          for (int cc = 0; cc < 4; ++cc) {
            const float* xp = Xb + xfrag + (ci + cc) * XSP;
#pragma unroll
            for (int q = 0; q < 3; ++q) px[cc][q] = *reinterpret_cast<const f32x2*>(xp + 2 * q);
          }
          // (measured in round 4: a scheduler fence here + adp_keep on the unused window floats gives "barrier, 9 reads, 16 MFMAs"
          //  in the ISA instead of the compiler's interleaving of reads and MFMAs -- and 13.16 -> 13.31 ms per step: with two MMA
          //  waves per SIMD the spread-out reads share the LDS better than eight waves reading everything right after the barrier)
#pragma unroll
          for (int cc = 0; cc < 4; ++cc) {
            const float d0 = px[cc][0][1], d1 = px[cc][1][0], d2 = px[cc][1][1], d3 = px[cc][2][0];
            const float g0 = av[cc * KT], g1 = av[cc * KT + 1], g2 = av[cc * KT + 2];
            const float gs = g0 + g2;
            acc[0] = adp_mfma32(g0, d0 - d2, acc[0]);
            acc[1] = adp_mfma32(gs + g1, d1 + d2, acc[1]);
            acc[2] = adp_mfma32(gs - g1, d2 - d1, acc[2]);
            acc[3] = adp_mfma32(g2, d1 - d3, acc[3]);
          }
        } else if (!TR) {
          float av[4 * KT];
          const float* ap = Ab + afrag + ci * KT;
#pragma unroll
          for (int j = 0; j < KT; ++j) {
            const f32x4 q = *reinterpret_cast<const f32x4*>(ap + 4 * j);
#pragma unroll
            for (int k = 0; k < 4; ++k) av[4 * j + k] = q[k];
          }
#pragma unroll
          for (int cc = 0; cc < 4; ++cc)
#pragma unroll
            for (int t = 0; t < KT; ++t) {
              const float x0 = Xb[xfrag + (ci + cc) * XSP + t * dil];
              const float x1 = Xb[xfrag + 32 * S + (ci + cc) * XSP + t * dil];
              acc[0] = adp_mfma32(av[cc * KT + t], x0, acc[0]);
              acc[1] = adp_mfma32(av[cc * KT + t], x1, acc[1]);
            }
        } else {
#pragma unroll
          for (int cc = 0; cc < 4; ++cc)
#pragma unroll
            for (int t = 0; t < KT; ++t) {
              const float a = Ab[afrag + (ci + cc) * AS + (KT - 1 - t)];
              const float x0 = Xb[xfrag + (ci + cc) * XSP + t * dil];
              const float x1 = Xb[xfrag + 32 * S + (ci + cc) * XSP + t * dil];
              acc[0] = adp_mfma32(a, x0, acc[0]);
              acc[1] = adp_mfma32(a, x1, acc[1]);
            }
        }
      }
    }
  }
  __syncthreads();  // the staging buffers are free
  if constexpr (WN) {  // output transform A^T (linear: applied to this K group's partial planes before the exchange)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p0 = acc[0][r], p1 = acc[1][r], p2 = acc[2][r], p3 = acc[3][r];
      acc[0][r] = fmaf(0.5f, p1 + p2, p0);
      acc[1][r] = fmaf(0.5f, p1 - p2, -p3);
    }
  }

  // ---- K-group exchange through LDS: every MMA wave parks both tiles, then sums + stores accumulator rows
  // [16/NKG * kg, 16/NKG * (kg+1)) of its (mq) tile over the NKG groups in the fixed order 0..NKG-1.
  constexpr int RPW = 16 / NKG;  // accumulator registers (= output rows per half-wave) finished per wave
  if (NKG > 1) {
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      float* rp = smem + (((nq * NKG + kg) * NQM + mq) * 2 + ni) * 1024 + lane;
#pragma unroll
      for (int r = 0; r < 16; ++r) rp[r * 64] = acc[ni][r];
    }
  }
  __syncthreads();

  // ---- epilogue (same contract as adp_conv1d's generic kernel)
  const int sp = (int)d.sp;
  const int64_t ebs = d.e_bstride ? d.e_bstride : M;
  float vfin[2][RPW];  // final output values of this lane (GroupNorm partial statistics below)
  if constexpr (WN) {
    // tiles 0 / 1 hold the even / odd position of the lane's output pair: 8-byte accesses, 256 contiguous bytes per
    // output row and half-wave
    const int n = nw0 + 2 * l31;
    const bool nok = n < N;  // N is even: a pair is inside or outside
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
      const int r = kg * RPW + rr;
      float v0, v1;
      if (NKG > 1) {
        v0 = v1 = 0.0f;
#pragma unroll
        for (int g = 0; g < NKG; ++g) {
          v0 += smem[(((nq * NKG + g) * NQM + mq) * 2 + 0) * 1024 + r * 64 + lane];
          v1 += smem[(((nq * NKG + g) * NQM + mq) * 2 + 1) * 1024 + r * 64 + lane];
        }
      } else {
        v0 = acc[0][rr];
        v1 = acc[1][rr];
      }
      const int m = m0 + wm0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      const bool ok = (m < M) && nok;
      vfin[0][rr] = vfin[1][rr] = 0.0f;
      if (d.store == 2) {  // pooled store (gradient of the nearest upsample): sum of sp adjacent positions, no bias
        float v = ok ? v0 + v1 : 0.0f;
        if (sp == 4) v += __shfl_xor(v, 1, 64);
        if (ok && (sp == 2 || (l31 & 1) == 0)) {
          const int64_t o = ((int64_t)b * M + m) * (N / sp) + n / sp;
          if (d.res) v += d.res[o];
          d.out[o] = v;
        }
        continue;
      }
      if (!ok) continue;
      if (KS > 1) {  // raw partial tile; the epilogue runs in the reduce kernel
        *reinterpret_cast<f32x2*>(d.ws + (((int64_t)ks * d.B + b) * M + m) * N + n) = f32x2{v0, v1};
        continue;
      }
      const int64_t o = ((int64_t)b * M + m) * N + n;
      float e_bias, e_sc;
      f32x2 e_res;
      if constexpr (PRE) {
        e_bias = pre_bias[rr];
        e_sc = pre_scale[rr];
        e_res = pre_res[rr];
      } else {
        e_bias = d.bias ? d.bias[m] : 0.0f;
        e_sc = d.e_scale ? d.e_scale[b * ebs + m] : 1.0f;
        e_res = d.res ? *reinterpret_cast<const f32x2*>(d.res + o) : f32x2{0.0f, 0.0f};
      }
      v0 += e_bias;
      v1 += e_bias;
      if (d.out_pre) *reinterpret_cast<f32x2*>(d.out_pre + o) = f32x2{v0, v1};
      v0 = fmaf(v0, e_sc, e_res[0]);
      v1 = fmaf(v1, e_sc, e_res[1]);
      *reinterpret_cast<f32x2*>(d.out + o) = f32x2{v0, v1};
      vfin[0][rr] = v0;
      vfin[1][rr] = v1;
    }
  } else {
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const int n = nw0 + ni * 32 + l31;
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
      const int r = kg * RPW + rr;
      float v;
      if (NKG > 1) {
        v = 0.0f;
#pragma unroll
        for (int g = 0; g < NKG; ++g) v += smem[(((nq * NKG + g) * NQM + mq) * 2 + ni) * 1024 + r * 64 + lane];
      } else {
        v = acc[ni][rr];
      }
      const int m = m0 + wm0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      const bool ok = (m < M) && (n < N);
      if (KS > 1) {  // raw partial tile; the epilogue runs in the reduce kernel
        if (ok) d.ws[(((int64_t)ks * d.B + b) * M + m) * N + n] = v;
        continue;
      }
      if (ok) {
        if (d.bias) v += d.bias[m];
        if (d.out_pre) d.out_pre[((int64_t)b * M + m) * N + n] = v;
        if (d.e_scale) v *= d.e_scale[b * ebs + m];
      } else {
        v = 0.0f;
      }
      if (d.store == 0) {
        if (ok) {
          const int64_t o = ((int64_t)b * M + m) * N + n;
          if (d.res) v += d.res[o];
          d.out[o] = v;
        }
        vfin[ni][rr] = ok ? v : 0.0f;
      } else if (d.store == 1) {
        if (ok) {
          const int64_t o = ((int64_t)b * (M / sp) + m / sp) * ((int64_t)N * sp) + (int64_t)n * sp + (m % sp);
          if (d.res) v += d.res[o];
          d.out[o] = v;
        }
      } else {
        v += __shfl_xor(v, 1, 64);
        if (sp == 4) v += __shfl_xor(v, 2, 64);
        if (ok && (l31 % sp) == 0) {
          const int64_t o = ((int64_t)b * M + m) * (N / sp) + n / sp;
          if (d.res) v += d.res[o];
          d.out[o] = v;
        }
      }
    }
  }
  }
  // ---- first stage of the backward of SiLU(GroupNorm(gnb_x)) whose output gradient this tile is (adp_conv_desc.gnb_ab; store 0,
  // no K split): (sum ds * xhat, sum ds) per finished row over the tile's <= 64 positions, one entry per row and 64-position tile
  if (GNB && KS == 1 && d.store == 0 && nw0 < N) {
    const int cg = M / (int)d.gnb_groups;
    const int n_a = WN ? nw0 + 2 * l31 : nw0 + l31, n_b = WN ? n_a + 1 : n_a + 32;  // positions of vfin[0] / vfin[1]
    const int E = (N + MM_BN - 1) / MM_BN;
    float xa[RPW], xb[RPW];
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {  // (every row's operands requested before the first is used)
      const int r = kg * RPW + rr;
      const int m = m0 + wm0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      const float* xr = d.gnb_x + ((int64_t)b * M + (m < M ? m : M - 1)) * N;
      xa[rr] = xr[n_a < N ? n_a : N - 1];
      xb[rr] = xr[n_b < N ? n_b : N - 1];
    }
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
      const int r = kg * RPW + rr;
      const int m = m0 + wm0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      const int mc = m < M ? m : M - 1;
      const float* st = d.gnb_stats + ((int64_t)b * d.gnb_groups + mc / cg) * 2;
      const float mu = st[0], rs = st[1];
      const float ga = d.gnb_gamma[mc] * rs, be = d.gnb_beta[mc] - mu * ga;
      const float xh0 = (xa[rr] - mu) * rs, xh1 = (xb[rr] - mu) * rs;
      const float ds0 = vfin[0][rr] * adp_dsilu_fast(fmaf(xa[rr], ga, be));  // (vfin = 0 outside the tensor)
      const float ds1 = vfin[1][rr] * adp_dsilu_fast(fmaf(xb[rr], ga, be));
      const float sa = adp_half_sum(fmaf(ds0, xh0, ds1 * xh1)), sb = adp_half_sum(ds0 + ds1);  // valid in lanes 16-31 / 48-63
      if (l31 == 16 && m < M) *reinterpret_cast<f32x2*>(d.gnb_ab + (((int64_t)b * M + m) * E + nt * NSP + nq) * 2) = f32x2{sa, sb};
    }
  }
  // ---- GroupNorm partial statistics of the tile just stored (store 0, no K split): one (mean, M2, count) entry per
  // ROW QUAD (the 4 consecutive output channels a lane holds in accumulator registers 4q .. 4q+3) over the tile's
  // <= 64 positions: 8 values per lane, then the 32 lanes of the half-wave; two passes in registers.
  if (d.gn_part != nullptr && KS == 1 && d.store == 0 && nw0 < N) {
    const int cntv = (N - nw0) < MM_BN ? (N - nw0) : MM_BN;
    const bool ok0 = WN ? (nw0 + 2 * l31 < N) : (nw0 + l31 < N), ok1 = WN ? ok0 : (nw0 + 32 + l31 < N);
    const float fcnt = 4.0f * (float)cntv;
#pragma unroll
    for (int q = 0; q < RPW / 4; ++q) {
      float sv = 0.0f;
#pragma unroll
      for (int j = 0; j < 4; ++j) sv += vfin[0][4 * q + j] + vfin[1][4 * q + j];
#pragma unroll
      for (int o = 1; o < 32; o <<= 1) sv += __shfl_xor(sv, o, 64);
      const float mean = sv / fcnt;
      float qv = 0.0f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float d0 = ok0 ? vfin[0][4 * q + j] - mean : 0.0f, d1 = ok1 ? vfin[1][4 * q + j] - mean : 0.0f;
        qv = fmaf(d0, d0, fmaf(d1, d1, qv));
      }
#pragma unroll
      for (int o = 1; o < 32; o <<= 1) qv += __shfl_xor(qv, o, 64);
      if (l31 == 0) {
        const int r = kg * RPW + 4 * q;
        const int m = m0 + wm0 + 8 * (r >> 2) + 4 * hi;  // first channel of the quad
        float* e = d.gn_part + (((int64_t)b * (M / 4) + (m >> 2)) * ((N + MM_BN - 1) / MM_BN) + nt * NSP + nq) * 3;
        e[0] = mean;
        e[1] = qv;
        e[2] = fcnt;
      }
    }
  }
}

template <int BM, int KT, int S, int UP, bool TR, int PRO, int BKT, int PD, bool WN = false, int NSP = 1, bool GNB = false>
int launch_mm(const adp_conv_desc& d, void* stream) {
  const int64_t blocks = (d.M / BM) * adp_cdiv(d.N, MM_BN * NSP) * d.B;
  const int KS = d.ws ? (int)adp_conv_mm_ksplit(d) : 1;
  // chunk-ahead fragment reads (PF) for the 8-wave blocks -- one MMA wave per SIMD: step 13.32 -> 13.25 ms, batch 1 7.09 -> 6.99,
  // sampler 2.407 -> 2.383 (ADP_MM_PF=0/1 interleaved on one box).  The 12-wave 64-row blocks (two MMA waves per SIMD hide each
  // other's reads) lose with it: 13.16 -> 13.27 ms.
  if constexpr (WN && BM == 32 && NSP == 1 && BKT == 32 && UP == 1 && mm_nld(PRO, BM, PD, NSP) == 4) {
    const char* e = getenv("ADP_MM_PF");
    if (!e || e[0] != '0') {
      ADP_LAUNCH((conv_mm_kernel<BM, KT, S, UP, TR, PRO, BKT, PD, WN, NSP, true, GNB>), dim3((unsigned)blocks, (unsigned)KS),
                 dim3(((BM / 32) * mm_nkg(BKT) + mm_nld(PRO, BM, PD, NSP)) * 64), stream, d, KS);
      return ADP_LAUNCH_OK();
    }
  }
  ADP_LAUNCH((conv_mm_kernel<BM, KT, S, UP, TR, PRO, BKT, PD, WN, NSP, false, GNB>), dim3((unsigned)blocks, (unsigned)KS),
             dim3(((BM / 32) * mm_nkg(BKT) + mm_nld(PRO, BM, PD, NSP)) * 64), stream, d, KS);
  return ADP_LAUNCH_OK();
}

// short K (one or two chunks: the HBM-bound shallow layers) runs without ghost iterations
template <int BM, int KT, int S, int UP, bool TR, int PRO, int BKT, bool WN = false, bool GNB = false>
int launch_pd(const adp_conv_desc& d, void* stream) {
  const int64_t KS = d.ws ? adp_conv_mm_ksplit(d) : 1;
  if constexpr (BM == 64 && S == 1 && (WN || KT == 1)) {  // wide-N blocks (one register stage: their chunks are long)
    const int nsp = adp_conv_mm_nsp(d);
    if constexpr (WN) {  // (the 1x1 convs stop at 128 positions: adp_conv_mm_nsp)
      if (nsp == 4) return launch_mm<BM, KT, S, UP, TR, PRO, BKT, 1, WN, 4, GNB>(d, stream);
    }
    if (nsp == 2) return launch_mm<BM, KT, S, UP, TR, PRO, BKT, 1, WN, 2, GNB>(d, stream);
  }
  // (a 64-channel chunk already is two 32-channel register stages; a second one does not fit the register file)
  // (a third register stage for the short Winograd chunks was measured: 14.37 -> 15.17 ms per step, rejected)
  // (four register stages for the batch-1 deep layers, 16 chunks per block: batch-1 step 7.97 -> 8.10 ms, sampler 2.75 -> 2.80)
  if (BKT < 64 && d.R / BKT / KS >= 4) return launch_mm<BM, KT, S, UP, TR, PRO, BKT, 2, WN, 1, GNB>(d, stream);
  return launch_mm<BM, KT, S, UP, TR, PRO, BKT, 1, WN, 1, GNB>(d, stream);
}

// every (kernel, stride, upsample, direction, prologue) variant of one block tile
template <int BM>
int run_tile(const adp_conv_desc& d, void* stream) {
  const bool tr = d.transposed != 0;
  if (d.stride == 2) return launch_pd<BM, 2, 2, 1, false, 0, 32>(d, stream);
  if (d.stride == 4) return launch_pd<BM, 4, 4, 1, false, 0, 16>(d, stream);
  if (d.up == 2)
    return adp_conv_mm_winograd(d) ? launch_pd<BM, 3, 1, 2, false, 0, 32, true>(d, stream)
                                   : launch_pd<BM, 3, 1, 2, false, 0, 32>(d, stream);
  if (d.up == 4)
    return adp_conv_mm_winograd(d) ? launch_pd<BM, 3, 1, 4, false, 0, 32, true>(d, stream)
                                   : launch_pd<BM, 3, 1, 4, false, 0, 32>(d, stream);
  if (d.KT == 3) {
    if (adp_conv_mm_winograd(d)) {  // Winograd F(2,3) in the MMA waves' registers (two thirds of the MFMAs)
      if (d.prologue == 1)
        return tr ? launch_pd<BM, 3, 1, 1, true, 1, 32, true>(d, stream)
                  : launch_pd<BM, 3, 1, 1, false, 1, 32, true>(d, stream);
      // (data gradient that also leaves a GroupNorm backward's first stage: unsplit launches -- with the K split the reduce kernel does)
      if (tr && d.gnb_ab && !(d.ws && adp_conv_mm_ksplit(d) > 1)) return launch_pd<BM, 3, 1, 1, true, 0, 32, true, true>(d, stream);
      return tr ? launch_pd<BM, 3, 1, 1, true, 0, 32, true>(d, stream)
                : launch_pd<BM, 3, 1, 1, false, 0, 32, true>(d, stream);
    }
    if (d.prologue == 1)
      return tr ? launch_pd<BM, 3, 1, 1, true, 1, 32>(d, stream) : launch_pd<BM, 3, 1, 1, false, 1, 32>(d, stream);
    return tr ? launch_pd<BM, 3, 1, 1, true, 0, 32>(d, stream) : launch_pd<BM, 3, 1, 1, false, 0, 32>(d, stream);
  }
  if (d.prologue == 1)
    return tr ? launch_pd<BM, 1, 1, 1, true, 1, 32>(d, stream) : launch_pd<BM, 1, 1, 1, false, 1, 32>(d, stream);
  // 1x1 convs (the DownsampleItem data gradients as a 1x1 over the space-to-depth view, attention projections): one tap per
  // channel pair = 8 MFMAs per wave and barrier with 32-channel chunks -- 64-channel chunks halve the barriers per MFMA
  // (ADP_MM_K1_BKT=32: the old chunk; not with the cross-workgroup K split, whose slices are counted in 32-channel chunks)
  {
    const char* e = getenv("ADP_MM_K1_BKT");
    if ((!e || atoi(e) == 64) && d.R % 64 == 0 && d.R >= 256 && (!d.ws || adp_conv_mm_ksplit(d) == 1))
      return tr ? launch_pd<BM, 1, 1, 1, true, 0, 64>(d, stream) : launch_pd<BM, 1, 1, 1, false, 0, 64>(d, stream);
  }
  return tr ? launch_pd<BM, 1, 1, 1, true, 0, 32>(d, stream) : launch_pd<BM, 1, 1, 1, false, 0, 32>(d, stream);
}

}  // namespace
