// Weight (+bias) gradient of the stride-1 'same' convolutions (kernel 3 pad 1, kernel 1 pad 0) on the f32 matrix
// cores -- autograd's conv weight gradient for the ResnetItem ConvBlocks of
// /root/reference/audio_diffusion_pytorch/components.py:89.
//
//   dw[m][r][t] = sum_{b,n} dy[b,m,n] * Xa[b, r, n + t - pad]        dbias[m] = sum_{b,n} dy[b,m,n]
//
// GEMM view: output (m, r) per tap, contraction over positions.  A block owns a BM x BR tile of (m, r) for all
// KT taps (KT accumulator tiles per wave: the dy fragment is shared by the taps); positions are staged in chunks
// of 64, double-buffered in LDS with register prefetch and one barrier per chunk, exactly like conv_mm.hip.  The
// NKG wave groups of a block split every chunk's 64 positions (in-block split-K, summed through LDS in a fixed
// order), and when the (m, r) tile grid is smaller than the chip the position range is additionally split across
// workgroups (partials to the workspace, summed by wgrad_reduce_kernel): deterministic, no atomics.
//   fragments: lane (row, hi) reads 4 consecutive positions 8s+4hi.. of its dy row (one ds_read_b128) and the
//   12 positions around them of its x row (three ds_read_b128, tap t / position j -> register j + 4 - pad + t);
//   row strides are 4 mod 8 dwords, so the 16-byte reads are bank-conflict free.
//   The GroupNorm+SiLU of the conv input is recomputed on the way into LDS (per-slot gamma/beta in registers,
//   the (mean, rstd) pair of the chunk's batch element fetched with the prefetch).
//   dbias is summed by the dy loader threads on the VALU (each staging slot owns one row), not on the MFMA.
#include <stdlib.h>
#include "adp_rt.h"
#include "adp.h"
#include "conv_internal.h"

// the per-item pointers of a batched launch (adp_conv1d_wgrad_batch): everything else of the items' descriptors is equal
struct adp_wg_items {
  const float* x[ADP_WGR_BATCH];
  const float* dy[ADP_WGR_BATCH];
  const float* pro_stats[ADP_WGR_BATCH];
  const float* pro_gamma[ADP_WGR_BATCH];
  const float* pro_beta[ADP_WGR_BATCH];
  float* dw[ADP_WGR_BATCH];
  float* dbias[ADP_WGR_BATCH];
  float* ws[ADP_WGR_BATCH];
};

#ifdef ADP_KTRACE
static __device__ unsigned long long* wg_kt_buf = nullptr;
extern "C" int adp_ktrace_set_wgrad(void* p) {
  return hipMemcpyToSymbol(HIP_SYMBOL(wg_kt_buf), &p, sizeof(p)) == hipSuccess ? 0 : -1;
}
#define WG_KT_BUF wg_kt_buf
#else
#define WG_KT_BUF nullptr
#endif

namespace {

struct __attribute__((packed, aligned(4))) adp_f32x3 {  // 12-byte global access (global_store_dwordx3)
  float a, b, c;
};
template <int K>
struct __attribute__((packed, aligned(4))) adp_taps {  // K adjacent taps of one (m, r) pair at a 4-byte aligned address
  float v[K];
};

constexpr int WG_BKN = 64;  // positions per staged chunk

// S: conv stride (1, or kernel = stride = 2 / 4: DownsampleItem, no halo, lane reads 4*S consecutive floats);
// UP: nearest-upsample factor folded into the x loader (UpsampleItem).
template <int UP>
__device__ __forceinline__ f32x4 wg_load_xquad(const float* p) {
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

constexpr int WG_NLD = 4;  // loader waves per block

// BM = BR: edge of the (m, r) tile (64 / 32); PD: loader prefetch distance in chunks (register stages).
// Wave-specialised like conv_mm_impl.h: WG_NLD loader waves stage dy / x chunks (global -> registers ->
// GroupNorm+SiLU -> LDS) one chunk ahead of the MMA waves, which only read fragments and issue MFMAs; one
// workgroup barrier per chunk hands the double-buffered LDS tiles over.  MMA wave = one 32x32 (m, r) quad x KT taps
// (KT accumulator tiles sharing the dy fragment); NKG wave groups split every chunk's 64 positions.
//
// WN = true (kernel 3, stride 1, pad 1): the same tiles and loaders, the contraction in the Winograd F(2,3) domain --
// the weight gradient of F(2,3) is, per output PAIR n = (p, p+1) with e = dy[p .. p+1] and d = x[p-1 .. p+2],
//     dw[0..2] += G^T [ (A e) * (B^T d) ]      A e = (e0, e0+e1, e0-e1, -e1)      B^T d = (d0-d2, d1+d2, d2-d1, d1-d3)
// i.e. FOUR rank-1 updates (one per Winograd plane: four accumulator tiles) where the direct form spends six.  Both
// transforms are a handful of VALU adds on the fragments the lane has read anyway, issued in the shadow of the
// 64-cycle MFMAs; G^T (with its halves) is applied once to the accumulators at the end:
//     dw0 = P0 + (P1+P2)/2     dw1 = (P1-P2)/2     dw2 = (P1+P2)/2 + P3      (P3 is accumulated with +e1, so: - P3)
//
// W4 = true (with WN): the same in the F(4,3) domain -- per output QUAD n = (p .. p+3) with e = dy[p .. p+3] and d = x[p-1 .. p+4]
//     dw[0..2] += G^T [ (A e) * (B^T d) ]      A e   = (e0, e0+e1+e2+e3, e0-e1+e2-e3, e0+2e1+4e2+8e3, e0-2e1+4e2-8e3, e3)
//                                               B^T d = (4d0-5d2+d4, t1+t2, t1-t2, t3+2t4, t3-2t4, 4d1-5d3+d5)
// (t1 = d4-4d2, t2 = d3-4d1, t3 = d4-d2, t4 = d3-d1): SIX rank-1 updates per four positions where F(2,3) spends eight and the
// direct form twelve; six accumulator tiles; G^T with its constants once at the end:
//     dw0 = P0/4 - (P1+P2)/6 + (P3+P4)/24     dw1 = (P2-P1)/6 + (P3-P4)/12     dw2 = -(P1+P2)/6 + (P3+P4)/6 + P5
// 21 VALU ops per 6 MFMAs on fragments the lane reads anyway (the F(2,3) form: 7 per 8); fp32 error ~1e-6 of the max norm.
// NKGP: in-block K groups (0 = the default: 2 for 64 x 64 tiles, 4 for 32 x 32).  1 = ONE MMA wave per SIMD (round 6): the
// in-kernel timeline showed the two MMA waves of a SIMD leaving every chunk barrier in lock step, the older one winning the
// arbitration for ~3100 cycles and the younger one closing the barrier ~1000 cycles later with the pipe a third idle; a lone MMA
// wave per SIMD owns its pipe, the block shrinks to 8 waves and the K-group exchange through LDS disappears.
template <int BM, int KT, int S, int UP, int PRO, int PD, bool WN = false, bool W4 = false, int NKGP = 0>
__global__ __launch_bounds__(((BM / 32) * (BM / 32) * (NKGP ? NKGP : (BM == 64 ? 2 : 4)) + WG_NLD) * 64, NKGP == 1 ? 4 : 1)
void wgrad_mm_kernel(  // (NKGP = 1: TWO 8-wave blocks per CU -- 72 KB of LDS each -- = 4 waves per SIMD = at most 128 registers)
    adp_wgrad_desc d_in, int CPB, int CPS, int nsplit, adp_wg_items items, int nitems) {
  // nitems > 1: `nitems` weight gradients of ONE shape in this launch (adp_conv1d_wgrad_batch): blockIdx.x = item * nsplit + split,
  // the items differ in their eight pointers only.  The workgroups of consecutive items follow each other on a CU without the
  // drain / launch / ramp of a kernel boundary between them.
  adp_wgrad_desc d = d_in;
  int split_ = blockIdx.x;
  if (nitems > 1) {
    const int item = (int)blockIdx.x / nsplit;
    split_ -= item * nsplit;
    d.x = items.x[item];
    d.dy = items.dy[item];
    d.pro_stats = items.pro_stats[item];
    d.pro_gamma = items.pro_gamma[item];
    d.pro_beta = items.pro_beta[item];
    d.dw = items.dw[item];
    d.dbias = items.dbias[item];
    d.ws = items.ws[item];
  }
  static_assert(!WN || (KT == 3 && S == 1), "Winograd F(2,3): kernel 3, stride 1 (any upsample factor)");
  static_assert(!W4 || WN, "F(4,3) is a variant of the Winograd form");
  constexpr int BR = BM, BKN = WG_BKN, NKG = NKGP ? NKGP : (BM == 64 ? 2 : 4), PPW = BKN / NKG;
  constexpr int NQR = BR / 32, NQ = (BM / 32) * NQR, NMMA = NQ * NKG, NLT = WG_NLD * 64;
  constexpr int PAD = (KT - 1) / 2;
  constexpr int HALO = (S == 1) ? 4 : 0;                       // positions staged on each side of the chunk's x rows
  constexpr int DS = BKN + 4, XS = BKN * S + 2 * HALO + 4;     // row strides (floats), both 4 mod 8
  constexpr int DQ = BKN / 4, XQ = (BKN * S + 2 * HALO) / 4;
  constexpr int D_ELEMS = BM * DS, X_ELEMS = BR * XS;
  constexpr int ND4 = (BM * DQ + NLT - 1) / NLT, NX4 = (BR * XQ + NLT - 1) / NLT;
  constexpr int RED = NQ * KT * 1024;
  constexpr int STAGE = 2 * (D_ELEMS + X_ELEMS);
  constexpr int SM = STAGE > RED ? STAGE : RED;
  static_assert(PPW % 8 == 0, "a wave consumes positions in groups of 8");
  __shared__ __attribute__((aligned(16))) float smem[SM];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hi = lane >> 5, l31 = lane & 31;
  ADP_KT_DECL(WG_KT_BUF)
  ADP_KT(0);
#ifdef ADP_KTRACE
  const int kt_block = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
#endif

  const int M = (int)d.M, R = (int)d.R, L = (int)d.Lin, N = (int)d.N, G = (int)d.groups;
  const int Lv = L * UP;
  const int split = split_;
  const int m0 = blockIdx.y * BM, r0 = blockIdx.z * BR;
  const int total = (int)d.B * CPB;
  const int cbeg = split * CPS, cend = (cbeg + CPS < total) ? cbeg + CPS : total;
  const int nloc = cend - cbeg;                               // chunks of this workgroup (>= 1)
  const int nrounds = ((nloc + PD - 1) / PD) * PD;            // ghost iterations (barrier only) pad the loop to PD
  const bool direct = (nsplit == 1);
  const int64_t cnt = (int64_t)M * R * KT;

  if (wave >= NMMA) {
    // =========================== loader waves ===========================
    ADP_LOADER_PRIO_SET();
    const int lt = tid - NMMA * 64;
    const bool do_bias = (d.dbias != nullptr) && (blockIdx.z == 0);
    if constexpr (S == 1 && UP == 1 && (PRO == 0 || BM == 64)) {  // (32 x 32 tiles with the prologue: measured slower, 92 -> 115 us)
      if (N % BKN == 0) {
        // ---- LEAN loader (round 6; every layer of the README net: plain input, row length a multiple of the chunk).  The in-kernel
        // timeline (tools/ktrace.py) showed the F(4,3) blocks waiting for THIS wave, not for memory: its loads landed within
        // ~100-400 cycles of being asked for, but the ~150 instructions around them (a select per staged element for padding that
        // only the two halo quads of a row can need, 64-bit address arithmetic per slot, per-slot LDS addresses) took 4500-5000
        // cycles of issue slots next to two MFMA waves per SIMD, against 3060 cycles of MFMAs per chunk.  Here a lane's slots are
        // rows row0 + 16 i of ONE quad column: one 32-bit lane offset on a wave-uniform base per tensor (scalar address updates per
        // chunk), LDS addresses = one base + immediates, no select on the 16 interior quads; the two halo quads of a row are staged
        // by the first 2 * BM lanes, the only ones that test for the row's ends.
        constexpr int NS = BM / 16;  // slots per lane and tensor (16 quads of 4 positions per row and chunk)
        const int row0 = lt >> 4, q = lt & 15;
        const unsigned d_off = (unsigned)((m0 + row0) * N + 4 * q), x_off = (unsigned)((r0 + row0) * L + 4 * q);
        const int d_lds = row0 * DS + 4 * q, x_lds = row0 * XS + 4 * q + HALO;
        const bool halo_lane = (KT == 3) && lt < 2 * BM;  // (wave-uniform: 2 * BM is a multiple of 64)
        const int hrow = lt % BM, hside = lt / BM;        // side 0: positions p0-4 .. p0-1, side 1: p0+64 .. p0+67
        const unsigned h_off = (unsigned)((r0 + hrow) * L);
        const int h_rel = hside ? BKN : -HALO, h_lds = hrow * XS + (hside ? BKN + HALO : 0);
        float bsum[NS];
#pragma unroll
        for (int i = 0; i < NS; ++i) bsum[i] = 0.0f;
        // PRO = 1: GroupNorm + SiLU of the conv input on the way into LDS (per-slot gamma / beta in registers, the (mean, rstd)
        // pair of the chunk's batch element fetched with the prefetch)
        float x_ga[NS], x_be[NS], h_ga = 1.0f, h_be = 0.0f;
        int x_st[NS], h_st = 0;
        if constexpr (PRO == 1) {
#pragma unroll
          for (int i = 0; i < NS; ++i) {
            const int r = r0 + row0 + 16 * i;
            x_st[i] = (r / (R / G)) * 2;
            x_ga[i] = d.pro_gamma ? d.pro_gamma[r] : 1.0f;
            x_be[i] = d.pro_beta ? d.pro_beta[r] : 0.0f;
          }
          if (halo_lane) {
            const int r = r0 + hrow;
            h_st = (r / (R / G)) * 2;
            h_ga = d.pro_gamma ? d.pro_gamma[r] : 1.0f;
            h_be = d.pro_beta ? d.pro_beta[r] : 0.0f;
          }
        }
        f32x4 rd[PD][NS], rx[PD][NS], rh[PD];
        float rm[PD][NS + 1], rr[PD][NS + 1];  // (mean, rstd) of the slots' rows, [NS] = the halo lane's row
        bool h_ok[PD];
        auto load_chunk = [&](f32x4 (&qd)[NS], f32x4 (&qx)[NS], f32x4& qh, bool& okh, float (&qm)[NS + 1], float (&qr)[NS + 1],
                              int k) {
          const int c = cbeg + (k < nloc ? k : nloc - 1);
          const int b = c / CPB, p0 = (c - b * CPB) * BKN;
          const float* dyb = d.dy + ((int64_t)b * M * N + p0);   // wave-uniform bases
          const float* xbp = d.x + ((int64_t)b * R * L + p0);
#pragma unroll
          for (int i = 0; i < NS; ++i) qd[i] = *reinterpret_cast<const f32x4*>(dyb + (d_off + (unsigned)(i * 16 * N)));
#pragma unroll
          for (int i = 0; i < NS; ++i) qx[i] = *reinterpret_cast<const f32x4*>(xbp + (x_off + (unsigned)(i * 16 * L)));
          if (halo_lane) {
            okh = (p0 + h_rel >= 0) && (p0 + h_rel < L);
            qh = *reinterpret_cast<const f32x4*>(d.x + ((int64_t)b * R * L + h_off) + (okh ? p0 + h_rel : 0));
          }
          if constexpr (PRO == 1) {
            const float* st = d.pro_stats + (int64_t)b * G * 2;
#pragma unroll
            for (int i = 0; i < NS; ++i) qm[i] = st[x_st[i]], qr[i] = st[x_st[i] + 1];
            if (halo_lane) qm[NS] = st[h_st], qr[NS] = st[h_st + 1];
          }
        };
        auto store_chunk = [&](const f32x4 (&qd)[NS], const f32x4 (&qx)[NS], const f32x4& qh, bool okh,
                               const float (&qm)[NS + 1], const float (&qr)[NS + 1], int k) {
          float* Db = smem + (k & 1) * (D_ELEMS + X_ELEMS);
          float* Xb = Db + D_ELEMS;
          if (k < nloc) {  // (ghost chunks pad the loop to PD: their rows must not count into dbias)
#pragma unroll
            for (int i = 0; i < NS; ++i) bsum[i] += (qd[i][0] + qd[i][1]) + (qd[i][2] + qd[i][3]);
          }
#pragma unroll
          for (int i = 0; i < NS; ++i) *reinterpret_cast<f32x4*>(Db + d_lds + i * 16 * DS) = qd[i];
#pragma unroll
          for (int i = 0; i < NS; ++i) {
            f32x4 v = qx[i];
            if constexpr (PRO == 1) {
              const float pa = x_ga[i] * qr[i], pb = x_be[i] - qm[i] * pa;
#pragma unroll
              for (int j = 0; j < 4; ++j) v[j] = adp_silu_fast(fmaf(v[j], pa, pb));
            }
            *reinterpret_cast<f32x4*>(Xb + x_lds + i * 16 * XS) = v;
          }
          if (halo_lane) {
            f32x4 v = qh;
            if constexpr (PRO == 1) {
              const float pa = h_ga * qr[NS], pb = h_be - qm[NS] * pa;
#pragma unroll
              for (int j = 0; j < 4; ++j) v[j] = adp_silu_fast(fmaf(v[j], pa, pb));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = okh ? v[j] : 0.0f;
            *reinterpret_cast<f32x4*>(Xb + h_lds) = v;
          }
        };
#pragma unroll
        for (int s = 0; s < PD; ++s) load_chunk(rd[s], rx[s], rh[s], h_ok[s], rm[s], rr[s], s);
        for (int k0 = 0; k0 < nrounds; k0 += PD) {
#pragma unroll
          for (int s = 0; s < PD; ++s) {
#ifdef ADP_KTRACE
            if (k0 + s < 16) ADP_KT(1 + 3 * (k0 + s));
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (k0 + s < 16) ADP_KT(2 + 3 * (k0 + s));
#endif
            store_chunk(rd[s], rx[s], rh[s], h_ok[s], rm[s], rr[s], k0 + s);
            load_chunk(rd[s], rx[s], rh[s], h_ok[s], rm[s], rr[s], k0 + s + PD);
#ifdef ADP_KTRACE
            if (k0 + s < 16) ADP_KT(3 + 3 * (k0 + s));
#endif
            __syncthreads();  // B_k
          }
        }
        ADP_KT(60);
        __syncthreads();
#pragma unroll
        for (int g = 1; g < NKG; ++g) {
          __syncthreads();
          __syncthreads();
        }
        if (do_bias) {  // a dy row is staged by 16 consecutive lanes of one slot: summed in a fixed order
          float* bb = direct ? d.dbias : d.ws + (int64_t)nsplit * cnt + (int64_t)split * M;
#pragma unroll
          for (int i = 0; i < NS; ++i) {
            float sv = bsum[i];
            sv += __shfl_xor(sv, 1, 64);
            sv += __shfl_xor(sv, 2, 64);
            sv += __shfl_xor(sv, 4, 64);
            sv += __shfl_xor(sv, 8, 64);
            if (q == 0) {
              const int m = m0 + row0 + 16 * i;
              bb[m] = (direct && (d.accumulate & 1)) ? bb[m] + sv : sv;
            }
          }
        }
#ifdef ADP_KTRACE
        ADP_KT_DUMP(kt_block);
#endif
        return;
      }
    }
    // staging slots (chunk independent parts); slot indices wrap instead of being guarded
    int d_src[ND4], d_dst[ND4], d_pos[ND4];
#pragma unroll
    for (int i = 0; i < ND4; ++i) {
      const int e = (lt + i * NLT) % (BM * DQ);
      const int row = e / DQ, q = e - row * DQ;
      d_dst[i] = row * DS + 4 * q;
      d_src[i] = (m0 + row) * N + 4 * q;
      d_pos[i] = 4 * q;
    }
    int x_src[NX4], x_dst[NX4], x_pos[NX4], x_st[NX4];
    float x_ga[NX4], x_be[NX4];
#pragma unroll
    for (int i = 0; i < NX4; ++i) {
      const int e = (lt + i * NLT) % (BR * XQ);
      const int row = e / XQ, q = e - row * XQ;
      x_dst[i] = row * XS + 4 * q;
      x_src[i] = (r0 + row) * L;
      x_pos[i] = 4 * q - HALO;
      if (PRO == 1) {
        const int r = r0 + row;
        x_st[i] = (r / (R / G)) * 2;
        x_ga[i] = d.pro_gamma ? d.pro_gamma[r] : 1.0f;
        x_be[i] = d.pro_beta ? d.pro_beta[r] : 0.0f;
      }
    }
    float bsum[ND4];
#pragma unroll
    for (int i = 0; i < ND4; ++i) bsum[i] = 0.0f;

    f32x4 rd[PD][ND4], rx[PD][NX4];
    float rmean[PD][NX4], rrstd[PD][NX4];
    bool d_ok[PD][ND4], x_ok[PD][NX4];

    // local chunk k -> batch element b, first position p0.  Loads are unconditional (ghost chunks re-read the last
    // one, never consumed) so that the number of loads in flight is a compile-time constant.
    auto load_chunk = [&](f32x4 (&qd)[ND4], f32x4 (&qx)[NX4], float (&qm)[NX4], float (&qr)[NX4], bool (&okd)[ND4],
                          bool (&okx)[NX4], int k) {
      const int c = cbeg + (k < nloc ? k : nloc - 1);
      const int b = c / CPB, p0 = (c - b * CPB) * BKN;
      const float* dyb = d.dy + (int64_t)b * M * N + p0;
#pragma unroll
      for (int i = 0; i < ND4; ++i) {
        okd[i] = (p0 + d_pos[i] < N);
        qd[i] = *reinterpret_cast<const f32x4*>(dyb + (okd[i] ? d_src[i] : d_src[i] - d_pos[i] - p0));
      }
      const float* xbp = d.x + (int64_t)b * R * L;
#pragma unroll
      for (int i = 0; i < NX4; ++i) {
        const int u = p0 * S + x_pos[i];
        okx[i] = (u >= 0 && u < Lv);
        qx[i] = wg_load_xquad<UP>(xbp + x_src[i] + (okx[i] ? u / UP : 0));
        if (PRO == 1) {
          qm[i] = d.pro_stats[(int64_t)b * G * 2 + x_st[i]];
          qr[i] = d.pro_stats[(int64_t)b * G * 2 + x_st[i] + 1];
        }
      }
    };
    auto store_chunk = [&](const f32x4 (&qd)[ND4], const f32x4 (&qx)[NX4], const float (&qm)[NX4],
                           const float (&qr)[NX4], const bool (&okd)[ND4], const bool (&okx)[NX4], int k) {
      float* Db = smem + (k & 1) * (D_ELEMS + X_ELEMS);
      float* Xb = Db + D_ELEMS;
      const bool real = k < nloc;
#pragma unroll
      for (int i = 0; i < ND4; ++i) {
        f32x4 v = qd[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = okd[i] ? v[j] : 0.0f;
        // dbias: this slot's row is fixed; wrapped duplicate slots (lt + i*NLT >= BM*DQ) and ghost chunks must not
        // count
        if (lt + i * NLT < BM * DQ) bsum[i] += real ? (v[0] + v[1]) + (v[2] + v[3]) : 0.0f;
        *reinterpret_cast<f32x4*>(Db + d_dst[i]) = v;
      }
#pragma unroll
      for (int i = 0; i < NX4; ++i) {
        f32x4 v = qx[i];
        if (PRO == 1) {
          const float pa = x_ga[i] * qr[i], pb = x_be[i] - qm[i] * pa;
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = adp_silu_fast(fmaf(v[j], pa, pb));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = okx[i] ? v[j] : 0.0f;
        *reinterpret_cast<f32x4*>(Xb + x_dst[i]) = v;
      }
    };
#pragma unroll
    for (int s = 0; s < PD; ++s) load_chunk(rd[s], rx[s], rmean[s], rrstd[s], d_ok[s], x_ok[s], s);
    // Hazard note: the store of chunk k goes to LDS[k & 1], last read by the MFMAs of chunk k-2; every MMA wave
    // finished those before it arrived at barrier B_{k-1}, which this wave passed before starting iteration k.
    for (int k0 = 0; k0 < nrounds; k0 += PD) {
#pragma unroll
      for (int s = 0; s < PD; ++s) {
#ifdef ADP_KTRACE
        if (k0 + s < 16) ADP_KT(1 + 3 * (k0 + s));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (k0 + s < 16) ADP_KT(2 + 3 * (k0 + s));
#endif
        store_chunk(rd[s], rx[s], rmean[s], rrstd[s], d_ok[s], x_ok[s], k0 + s);
        load_chunk(rd[s], rx[s], rmean[s], rrstd[s], d_ok[s], x_ok[s], k0 + s + PD);
#ifdef ADP_KTRACE
        if (k0 + s < 16) ADP_KT(3 + 3 * (k0 + s));
#endif
        __syncthreads();  // B_k
      }
    }
    ADP_KT(60);
    __syncthreads();
#pragma unroll
    for (int g = 1; g < NKG; ++g) {
      __syncthreads();
      __syncthreads();
    }
    if (do_bias) {
      // a dy row is staged by DQ = 16 consecutive lanes of one slot: sum them in a fixed order
      float* bb = direct ? d.dbias : d.ws + (int64_t)nsplit * cnt + (int64_t)split * M;
#pragma unroll
      for (int i = 0; i < ND4; ++i) {
        float s = bsum[i];
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        s += __shfl_xor(s, 4, 64);
        s += __shfl_xor(s, 8, 64);
        const int e = lt + i * NLT;
        if (e < BM * DQ && (e % DQ) == 0) {
          const int m = m0 + e / DQ;
          bb[m] = (direct && (d.accumulate & 1)) ? bb[m] + s : s;
        }
      }
    }
#ifdef ADP_KTRACE
    ADP_KT_DUMP(kt_block);
#endif
    return;
  }

  // =========================== MMA waves ===========================
  const int quad = wave % NQ, kg = wave / NQ;
  const int wm0 = (quad / NQR) * 32, wr0 = (quad % NQR) * 32;
  constexpr int NACC = W4 ? 6 : (WN ? 4 : KT);
  f32x16 acc[NACC];
#pragma unroll
  for (int t = 0; t < NACC; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

  for (int k = 0; k < nrounds; ++k) {
#ifdef ADP_KTRACE
    if (k < 16) ADP_KT(1 + 3 * k);
#endif
    __syncthreads();  // B_k: chunk k is in LDS[k & 1]
#ifdef ADP_KTRACE
    if (k < 16) ADP_KT(2 + 3 * k);
#endif
    if (k < nloc) {
      const float* Db = smem + (k & 1) * (D_ELEMS + X_ELEMS);
      const float* Xb = Db + D_ELEMS;
      if constexpr (W4 && S == 1) {
        // F(4,3): of the lane's x row only d0 .. d5 = x[base-1 .. base+4] are read -- one 16-byte quad and two scalars instead of
        // three quads (40 instead of 64 bytes per lane and position group; round 6).
        const float* dp = Db + (wm0 + l31) * DS + kg * PPW + 4 * hi;
        const float* xp = Xb + (wr0 + l31) * XS + kg * PPW + 4 * hi;   // xp[i] = x[base - 4 + i]
#ifdef ADP_WG_PIPE  // (measured round 6: requesting group s + 1 ahead of group s's MFMAs changes nothing -- the pipe's idle time
                    //  is the lock step of the two MMA waves of a SIMD after each barrier, not the LDS round trips; kept for A/B)
        f32x4 dqn = *reinterpret_cast<const f32x4*>(dp), xmn = *reinterpret_cast<const f32x4*>(xp + 4);
        float d0n = xp[3], d5n = xp[8];
#endif
#pragma unroll
        for (int s = 0; s < PPW / 8; ++s) {
#ifdef ADP_WG_PIPE
          const f32x4 dq = dqn, xm = xmn;
          const float d0 = d0n, d5 = d5n;
          if (s + 1 < PPW / 8) {
            dqn = *reinterpret_cast<const f32x4*>(dp + 8 * (s + 1));
            xmn = *reinterpret_cast<const f32x4*>(xp + 8 * (s + 1) + 4);
            d0n = xp[8 * (s + 1) + 3];
            d5n = xp[8 * (s + 1) + 8];
          }
          adp_sched_fence();  // (the requests above stay above the matrix work that hides them)
#else
          const f32x4 dq = *reinterpret_cast<const f32x4*>(dp + 8 * s), xm = *reinterpret_cast<const f32x4*>(xp + 8 * s + 4);
          const float d0 = xp[8 * s + 3], d5 = xp[8 * s + 8];
#endif
          const float e0 = dq[0], e1 = dq[1], e2 = dq[2], e3 = dq[3];
          const float d1 = xm[0], d2 = xm[1], d3 = xm[2], d4 = xm[3];
          const float s02 = e0 + e2, s13 = e1 + e3;
          const float et = fmaf(4.0f, e2, e0), ev = fmaf(4.0f, e3, e1);
          const float t1 = fmaf(-4.0f, d2, d4), t2 = fmaf(-4.0f, d1, d3), t3 = d4 - d2, t4 = d3 - d1;
          acc[0] = adp_mfma32(e0, fmaf(4.0f, d0, fmaf(-5.0f, d2, d4)), acc[0]);
          acc[1] = adp_mfma32(s02 + s13, t1 + t2, acc[1]);
          acc[2] = adp_mfma32(s02 - s13, t1 - t2, acc[2]);
          acc[3] = adp_mfma32(fmaf(2.0f, ev, et), fmaf(2.0f, t4, t3), acc[3]);
          acc[4] = adp_mfma32(fmaf(-2.0f, ev, et), fmaf(-2.0f, t4, t3), acc[4]);
          acc[5] = adp_mfma32(e3, fmaf(4.0f, d1, fmaf(-5.0f, d3, d5)), acc[5]);
        }
      } else {
#pragma unroll
      for (int s = 0; s < PPW / 8; ++s) {
        const int base = kg * PPW + 8 * s + 4 * hi;
        const f32x4 dq = *reinterpret_cast<const f32x4*>(Db + (wm0 + l31) * DS + base);
        float xq[S == 1 ? 12 : 4 * S];
        if (S == 1) {
          const float* xp = Xb + (wr0 + l31) * XS + base;
          if (KT == 3) {
#pragma unroll
            for (int q = 0; q < 3; ++q) {
              const f32x4 v = *reinterpret_cast<const f32x4*>(xp + 4 * q);
#pragma unroll
              for (int j = 0; j < 4; ++j) xq[4 * q + j] = v[j];
            }
          } else {
            const f32x4 v = *reinterpret_cast<const f32x4*>(xp + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) xq[4 + j] = v[j];
          }
          if constexpr (W4) {
            // xq[i] = x[base - 4 + i]: the lane's quad = positions base .. base+3 with d = xq[3..8]; the MFMA K pair is (this
            // half-wave's quad, the other half-wave's quad)
            const float e0 = dq[0], e1 = dq[1], e2 = dq[2], e3 = dq[3];
            const float d0 = xq[3], d1 = xq[4], d2 = xq[5], d3 = xq[6], d4 = xq[7], d5 = xq[8];
            const float s02 = e0 + e2, s13 = e1 + e3;
            const float et = fmaf(4.0f, e2, e0), ev = fmaf(4.0f, e3, e1);
            const float t1 = fmaf(-4.0f, d2, d4), t2 = fmaf(-4.0f, d1, d3), t3 = d4 - d2, t4 = d3 - d1;
            acc[0] = adp_mfma32(e0, fmaf(4.0f, d0, fmaf(-5.0f, d2, d4)), acc[0]);
            acc[1] = adp_mfma32(s02 + s13, t1 + t2, acc[1]);
            acc[2] = adp_mfma32(s02 - s13, t1 - t2, acc[2]);
            acc[3] = adp_mfma32(fmaf(2.0f, ev, et), fmaf(2.0f, t4, t3), acc[3]);
            acc[4] = adp_mfma32(fmaf(-2.0f, ev, et), fmaf(-2.0f, t4, t3), acc[4]);
            acc[5] = adp_mfma32(e3, fmaf(4.0f, d1, fmaf(-5.0f, d3, d5)), acc[5]);
          } else if constexpr (WN) {
            // xq[i] = x[base - 4 + i]: pair 0 = positions (base, base+1) with d = xq[3..6], pair 1 = (base+2, base+3)
            // with d = xq[5..8]; the MFMA K pair is (this half-wave's pair, the other half-wave's pair)
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {
              const float e0 = dq[2 * pp], e1 = dq[2 * pp + 1];
              const float d0 = xq[3 + 2 * pp], d1 = xq[4 + 2 * pp], d2 = xq[5 + 2 * pp], d3 = xq[6 + 2 * pp];
              acc[0] = adp_mfma32(e0, d0 - d2, acc[0]);
              acc[1] = adp_mfma32(e0 + e1, d1 + d2, acc[1]);
              acc[2] = adp_mfma32(e0 - e1, d2 - d1, acc[2]);
              acc[3] = adp_mfma32(e1, d1 - d3, acc[3]);
            }
          } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int t = 0; t < KT; ++t) acc[t] = adp_mfma32(dq[j], xq[j + 4 - PAD + t], acc[t]);
          }
        } else {
          const float* xp = Xb + (wr0 + l31) * XS + base * S;
#pragma unroll
          for (int q = 0; q < S; ++q) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(xp + 4 * q);
#pragma unroll
            for (int j = 0; j < 4; ++j) xq[4 * q + j] = v[j];
          }
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int t = 0; t < KT; ++t) acc[t] = adp_mfma32(dq[j], xq[j * S + t], acc[t]);
        }
      }
      }
    }
#ifdef ADP_KTRACE
    if (k < 16) ADP_KT(3 + 3 * k);
#endif
  }
  ADP_KT(60);
  __syncthreads();
  ADP_KT(61);
  if constexpr (W4) {  // G^T of F(4,3): six planes -> three taps (linear, so applied to this K group's partial sums)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p0 = acc[0][r], p1 = acc[1][r], p2 = acc[2][r], p3 = acc[3][r], p4 = acc[4][r], p5 = acc[5][r];
      const float a12 = (p1 + p2) * (-1.0f / 6.0f), a34 = p3 + p4;
      acc[0][r] = fmaf(0.25f, p0, fmaf(1.0f / 24.0f, a34, a12));
      acc[1][r] = fmaf(1.0f / 6.0f, p2 - p1, (1.0f / 12.0f) * (p3 - p4));
      acc[2][r] = fmaf(1.0f / 6.0f, a34, a12) + p5;
    }
  } else if constexpr (WN) {  // G^T: planes -> taps (linear, so applied to this K group's partial sums)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p0 = acc[0][r], p1 = acc[1][r], p2 = acc[2][r], p3 = acc[3][r];
      const float h = 0.5f * (p1 + p2);
      acc[0][r] = p0 + h;
      acc[1][r] = 0.5f * (p1 - p2);
      acc[2][r] = h - p3;
    }
  }

  // ---- fixed-order sum of the K groups through LDS, one group per round
#pragma unroll
  for (int g = 1; g < NKG; ++g) {
    if (kg == g) {
#pragma unroll
      for (int t = 0; t < KT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) smem[((quad * KT + t) * 16 + r) * 64 + lane] = acc[t][r];
    }
    __syncthreads();
    if (kg == 0) {
#pragma unroll
      for (int t = 0; t < KT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] += smem[((quad * KT + t) * 16 + r) * 64 + lane];
    }
    __syncthreads();
  }

  if (kg == 0) {
    float* base = direct ? d.dw : d.ws + (int64_t)split * cnt;
    const bool accum = direct && (d.accumulate & 1);
    if constexpr (KT == 3) {
      // the three taps of an (m, r) pair are adjacent in dw[M][R][3]: ONE 12-byte store per accumulator row -- the 32
      // lanes of a half-wave then write 384 contiguous bytes (the per-tap 4-byte form scattered them over three
      // instructions at a 12-byte stride: every cache line written three times, partially)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm0 + (r & 3) + 8 * (r >> 2) + 4 * hi, rr = r0 + wr0 + l31;
        adp_f32x3* o = reinterpret_cast<adp_f32x3*>(base + ((int64_t)m * R + rr) * 3);
        adp_f32x3 v{acc[0][r], acc[1][r], acc[2][r]};
        if (accum) {
          const adp_f32x3 old = *o;
          v.a += old.a, v.b += old.b, v.c += old.c;
        }
        *o = v;
      }
    } else if constexpr (KT == 2 || KT == 4) {  // DownsampleItem (kernel = stride): 8- / 16-byte tap groups
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm0 + (r & 3) + 8 * (r >> 2) + 4 * hi, rr = r0 + wr0 + l31;
        adp_taps<KT>* o = reinterpret_cast<adp_taps<KT>*>(base + ((int64_t)m * R + rr) * KT);
        adp_taps<KT> v;
#pragma unroll
        for (int t = 0; t < KT; ++t) v.v[t] = acc[t][r];
        if (accum) {
          const adp_taps<KT> old = *o;
#pragma unroll
          for (int t = 0; t < KT; ++t) v.v[t] += old.v[t];
        }
        *o = v;
      }
    } else {
#pragma unroll
      for (int t = 0; t < KT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + wm0 + (r & 3) + 8 * (r >> 2) + 4 * hi, rr = r0 + wr0 + l31;
          float* o = base + ((int64_t)m * R + rr) * KT + t;
          *o = accum ? *o + acc[t][r] : acc[t][r];
        }
    }
  }
  ADP_KT(63);
#ifdef ADP_KTRACE
  ADP_KT_DUMP(kt_block);
#endif
}

struct WgPlan {
  int bm, nkg;          // tile edge (BM = BR) and in-block K groups
  int64_t cpb, cps, nsplit;
};

// Position split of one (m, r) tile grid: ns workgroups per tile, each >= 4 chunks of 64 positions (start-up amortised;
// the round-1 floor of 8 left the narrow layers on 128 of the 256 CUs at batch 4 and on 32 at batch 1: microbench
// C=64 37.0 -> 28.1 us at batch 4, 34.7 -> 14.3 us at batch 1).
static int64_t wg_split(int64_t tiles, int64_t total, int64_t target) {
  int64_t ns = adp_cdiv(target, tiles);
  if (ns > total / 4) ns = total / 4;
  if (ns > total) ns = total;
  if (ns < 1) ns = 1;
  return ns;
}

// `n` > 1: the plan of a BATCHED launch (adp_conv1d_wgrad_batch: n same-shape items, blockIdx.x = item * nsplit + split).  The
// items fill the chip together, so each needs 1/n of the position split a lone launch takes: fewer (or no) partial tiles written
// and read back, longer K loops per workgroup, and for the 512-channel layers (64 tiles x 4 items) no second stage at all.  The tile
// edge is the lone launch's (the scratch was sized for that plan; a batched plan never needs more).  ADP_WGRAD_BATCH_SPLIT=0: every
// item keeps the lone split (A/B; then batched and lone calls are bit-identical, otherwise they differ in summation order).
WgPlan wg_plan(const adp_wgrad_desc& d, int n = 1, int64_t slots64 = 256) {  // slots64: 64 x 64 workgroups the chip holds at once
  WgPlan p;
  p.nkg = 4;
  p.cpb = adp_cdiv(d.N, WG_BKN);
  const int64_t total = d.B * p.cpb;
  // 12-wave workgroups (64x64 tiles): one per CU fills the SIMDs; 8-wave workgroups (32x32): about three per CU
  const bool can64 = d.M % 64 == 0 && d.R % 64 == 0 && d.stride != 4;  // stride 4: x rows are 4x wider in LDS
  const int64_t t64 = (d.M / 64) * (d.R / 64), t32 = (d.M / 32) * (d.R / 32);
  const int64_t ns64 = can64 ? wg_split(t64, total, slots64) : 0, ns32 = wg_split(t32, total, 768);
  // 64x64 tiles unless they leave most CUs idle while 32x32 tiles (4x as many, half the staging per workgroup) do not
  p.bm = (can64 && (t64 * ns64 >= 200 || t32 * ns32 <= 3 * t64 * ns64)) ? 64 : 32;
  // (32 x 32 tiles WITHOUT the position split were measured for the 512-channel layers: batch 4 +0.13 ms, batch 1 -0.08 ms)
  int64_t ns = p.bm == 64 ? ns64 : ns32;
  if (n > 1) {
    const char* e = getenv("ADP_WGRAD_BATCH_SPLIT");
    if (!e || e[0] != '0') ns = p.bm == 64 ? wg_split(t64 * n, total, slots64) : wg_split(t32 * n, total, 768);
  }
  p.cps = adp_cdiv(total, ns);
  p.nsplit = adp_cdiv(total, p.cps);
  return p;
}

}  // namespace

// Second stage of every split weight gradient: out[i] = sum_k ws[k][i] (dw: i < cnt, dbias: cnt <= i < cnt + M, its
// partials start at ws + nsplit*cnt).  64 outputs x 16 split lanes per workgroup: rows of 256 contiguous bytes,
// 16 independent accumulation chains per output instead of one serial walk over all splits; the 16 lane sums are
// combined in a fixed order (deterministic).
// (up to ADP_WGR_BATCH same-shape weight gradients per launch, blockIdx.y = item: adp_wgrad_reduce_batch)
struct adp_wgr_batch {
  const float* ws[ADP_WGR_BATCH];
  float* dw[ADP_WGR_BATCH];
  float* dbias[ADP_WGR_BATCH];
};

__global__ __launch_bounds__(1024) void adp_wgrad_reduce_kernel(adp_wgr_batch t, int64_t nsplit, int64_t cnt, int64_t M,
                                                                int accumulate) {
  const float* ws = t.ws[blockIdx.y];
  float* dw = t.dw[blockIdx.y];
  float* dbias = t.dbias[blockIdx.y];
  __shared__ float part[16][64];
  const int il = threadIdx.x & 63, ks = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * 64 + il;
  const int64_t tot = cnt + (dbias ? M : 0);
  float s = 0.0f;
  // the partial rows of one output are independent loads: issue them eight at a time (a plain loop waits for each
  // row's latency in turn), and keep the summation order fixed
  const float* col = nullptr;
  int64_t step = 0;
  if (i < cnt) {
    col = ws + i;
    step = cnt;
  } else if (i < tot) {
    col = ws + nsplit * cnt + (i - cnt);
    step = M;
  }
  if (col) {
    int64_t k = ks;
    for (; k + 7 * 16 < nsplit; k += 8 * 16) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = col[(k + 16 * u) * step];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; k < nsplit; k += 16) s += col[k * step];
  }
  part[ks][il] = s;
  __syncthreads();
  if (ks == 0 && i < tot) {
    float sum = 0.0f;
#pragma unroll
    for (int k = 0; k < 16; ++k) sum += part[k][il];
    float* o = (i < cnt) ? dw + i : dbias + (i - cnt);
    *o = accumulate ? *o + sum : sum;
  }
}

// few splits: one thread per output walks them (the 16-lane form would idle most of its lanes)
__global__ __launch_bounds__(256) void adp_wgrad_reduce_small_kernel(adp_wgr_batch t, int64_t nsplit, int64_t cnt,
                                                                     int64_t M, int accumulate) {
  const float* ws = t.ws[blockIdx.y];
  float* dw = t.dw[blockIdx.y];
  float* dbias = t.dbias[blockIdx.y];
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < cnt) {
    float s = 0.0f;
#pragma unroll 8
    for (int64_t k = 0; k < nsplit; ++k) s += ws[k * cnt + i];
    dw[i] = accumulate ? dw[i] + s : s;
  } else if (dbias && i < cnt + M) {
    const int64_t m = i - cnt;
    const float* wsb = ws + nsplit * cnt;
    float s = 0.0f;
#pragma unroll 8
    for (int64_t k = 0; k < nsplit; ++k) s += wsb[k * M + m];
    dbias[m] = accumulate ? dbias[m] + s : s;
  }
}

// n <= ADP_WGR_BATCH weight gradients of ONE shape (nsplit, cnt, M; dbias all set or all NULL) in one launch
int adp_wgrad_reduce_n(const float* const* ws, float* const* dw, float* const* dbias, int n, int64_t nsplit, int64_t cnt,
                       int64_t M, int accumulate, void* stream) {
  adp_wgr_batch t;
  for (int i = 0; i < ADP_WGR_BATCH; ++i) {
    t.ws[i] = ws[i < n ? i : 0];
    t.dw[i] = dw[i < n ? i : 0];
    t.dbias[i] = dbias ? dbias[i < n ? i : 0] : nullptr;
  }
  const int64_t tot = cnt + (t.dbias[0] ? M : 0);
  if (nsplit <= 8) {
    ADP_LAUNCH(adp_wgrad_reduce_small_kernel, dim3((unsigned)adp_cdiv(tot, 256), (unsigned)n), dim3(256), stream, t, nsplit,
               cnt, M, accumulate);
    return ADP_LAUNCH_OK();
  }
  ADP_LAUNCH(adp_wgrad_reduce_kernel, dim3((unsigned)adp_cdiv(tot, 64), (unsigned)n), dim3(1024), stream, t, nsplit, cnt, M,
             accumulate);
  return ADP_LAUNCH_OK();
}

int adp_wgrad_reduce(const float* ws, int64_t nsplit, int64_t cnt, int64_t M, float* dw, float* dbias, int accumulate,
                     void* stream) {
  return adp_wgrad_reduce_n(&ws, &dw, dbias ? &dbias : nullptr, 1, nsplit, cnt, M, accumulate, stream);
}

namespace {

// PD: a 64x64 chunk is 2.6 us of MFMAs (one register stage), a 32x32 chunk 0.64 us (two)
template <int BM, int KT, int S, int UP, int PRO, bool WN = false, bool W4 = false, int PD = (BM == 64 ? 1 : 2), int NKGP = 0>
int launch_wg(const adp_wgrad_desc* ds, int n, const WgPlan& p, void* stream) {
  const adp_wgrad_desc& d = ds[0];
  adp_wg_items it;
  for (int i = 0; i < ADP_WGR_BATCH; ++i) {
    const adp_wgrad_desc& e = ds[i < n ? i : 0];
    it.x[i] = e.x, it.dy[i] = e.dy, it.pro_stats[i] = e.pro_stats, it.pro_gamma[i] = e.pro_gamma, it.pro_beta[i] = e.pro_beta;
    it.dw[i] = e.dw, it.dbias[i] = e.dbias, it.ws[i] = e.ws;
  }
  dim3 grid((unsigned)(p.nsplit * n), (unsigned)(d.M / BM), (unsigned)(d.R / BM));
  constexpr int NTH = ((BM / 32) * (BM / 32) * (NKGP ? NKGP : (BM == 64 ? 2 : 4)) + WG_NLD) * 64;
  ADP_LAUNCH((wgrad_mm_kernel<BM, KT, S, UP, PRO, PD, WN, W4, NKGP>), grid, dim3(NTH), stream, d, (int)p.cpb, (int)p.cps,
             (int)p.nsplit, it, n);
  if (p.nsplit > 1 && !(d.accumulate & 2)) {  // (bit 1 of `accumulate`: the caller parks the second stage, adp.h)
    if (ADP_LAUNCH_OK() != ADP_OK) return ADP_ERR_LAUNCH;
    return adp_wgrad_reduce_n(it.ws, it.dw, d.dbias ? it.dbias : nullptr, n, p.nsplit, d.M * d.R * KT, d.M,
                              (int)(d.accumulate & 1), stream);
  }
  return ADP_LAUNCH_OK();
}

bool wg_winograd4(const adp_wgrad_desc& d);

template <int KT, int S, int UP, int PRO, bool WN = false>
int pick_wg(const adp_wgrad_desc* ds, int n, void* stream) {
  const WgPlan p = wg_plan(ds[0], n);
  if constexpr (WN) {  // the F(4,3) form of the kernel-3 weight gradients (wg_winograd4)
    if (wg_winograd4(ds[0])) {
      if constexpr (S == 1 && UP == 1 && PRO == 0) {
        // One MMA wave per SIMD, TWO 8-wave blocks per CU (independent blocks: no lock step, K loop at 96 % of the matrix pipe),
        // planned for 512 resident workgroups.  Measured per batched launch, 12-wave -> solo: n8 [4,1024,256] 270.9 -> 246.8 us,
        // n8 [4,1024,128] 152.0 -> 132.5 us.  Taken when the 512-slot plan keeps >= 4 chunks per workgroup, fills the slots and
        // needs no more scratch than the lone launch's plan sized (adp_wgrad_mm_ws_floats).  ADP_WG_SOLO=0 / 1 forces either.
        const WgPlan ps = wg_plan(ds[0], n, 512);
        const int64_t t64 = (ds[0].M / 64) * (ds[0].R / 64);
        const char* so = getenv("ADP_WG_SOLO");
        const bool fits = ps.bm == 64 && (ps.nsplit == 1 || ps.nsplit <= wg_plan(ds[0]).nsplit);
        const bool solo = fits && (so ? so[0] == '1' : (t64 * n * ps.nsplit >= 512 && ps.cps >= 4));  // (batch 1, 4 chunks per block: step 6.16 -> 6.11 ms)
        if (solo) return launch_wg<64, KT, S, UP, PRO, true, true, 1, 1>(ds, n, ps, stream);
      }
      if (p.bm == 64) return launch_wg<64, KT, S, UP, PRO, true, true>(ds, n, p, stream);
      return launch_wg<32, KT, S, UP, PRO, true, true>(ds, n, p, stream);
    }
  }
  if constexpr (S != 4) {
    if (p.bm == 64) return launch_wg<64, KT, S, UP, PRO, WN>(ds, n, p, stream);
  }
  return launch_wg<32, KT, S, UP, PRO, WN>(ds, n, p, stream);
}

// Winograd F(2,3) form of the kernel-3 weight gradients (WN): same switch as the forward / data-gradient convs
// (ADP_CONV_WINO, conv_mm.hip), for layers with at least ADP_WINO_WGRAD_MIN_R (default 32) channels
bool wg_winograd(const adp_wgrad_desc& d) {
  if (!adp_winograd_enabled()) return false;
  const char* mr = getenv("ADP_WINO_WGRAD_MIN_R");
  const int64_t min_r = mr ? atoll(mr) : 32;
  return d.KT == 3 && d.stride == 1 && d.pad == 1 && d.R >= min_r;
}

// F(4,3) form (W4) of the same: ADP_WGRAD_WINO4 (read per call; "0" = F(2,3)), layers with at least ADP_WINO4_WGRAD_MIN_R
// (default 32) channels
bool wg_winograd4(const adp_wgrad_desc& d) {
  const char* e = getenv("ADP_WGRAD_WINO4");
  if (e && e[0] == '0') return false;
  const char* mr = getenv("ADP_WINO4_WGRAD_MIN_R");
  const char* u = getenv("ADP_WGRAD_WINO4_UP");  // (A/B: "0" keeps the UpsampleItem convs' gradients on F(2,3))
  if (d.up != 1 && u && u[0] == '0') return false;
  return wg_winograd(d) && d.R >= (mr ? atoll(mr) : 32);  // (round 6: also the 32-channel layers -- 92 -> 85 us per batched launch)
}

}  // namespace

bool adp_wgrad_mm_eligible(const adp_wgrad_desc& d) {
  if (d.R1 != d.R || d.dil != 1) return false;
  const bool plain = d.stride == 1 && d.up == 1 && ((d.KT == 3 && d.pad == 1) || (d.KT == 1 && d.pad == 0));
  const bool upc = d.stride == 1 && (d.up == 2 || d.up == 4) && d.KT == 3 && d.pad == 1 && d.prologue == 0;
  const bool down = (d.stride == 2 || d.stride == 4) && d.KT == d.stride && d.pad == 0 && d.up == 1 && d.prologue == 0;
  if (!plain && !upc && !down) return false;
  if (d.prologue != 0 && d.prologue != 1) return false;
  if (d.R % 32 != 0 || d.M % 32 != 0 || (d.Lin * d.up) % 4 != 0 || d.N % 4 != 0) return false;
  if (d.N * d.stride != d.Lin * d.up) return false;
  if ((reinterpret_cast<uintptr_t>(d.x) | reinterpret_cast<uintptr_t>(d.dy)) & 15) return false;
  if (d.M * d.N >= (int64_t)1 << 31 || d.R * d.Lin >= (int64_t)1 << 31 || d.B * adp_cdiv(d.N, WG_BKN) >= (int64_t)1 << 31)
    return false;
  if (d.M / 32 > 65535 || d.R / 32 > 65535) return false;
  return true;
}

int64_t adp_wgrad_mm_nsplit(const adp_wgrad_desc& d) { return wg_plan(d).nsplit; }

int64_t adp_wgrad_mm_ws_floats(const adp_wgrad_desc& d) {
  const WgPlan p = wg_plan(d);
  return p.nsplit * (d.M * d.R * d.KT + d.M);
}

// n <= ADP_WGR_BATCH weight gradients of one shape (descriptors equal up to their pointers) in one launch
int adp_wgrad_mm_n(const adp_wgrad_desc* ds, int n, void* stream) {
  const adp_wgrad_desc& d = ds[0];
  if (d.stride == 2) return pick_wg<2, 2, 1, 0>(ds, n, stream);
  if (d.stride == 4) return pick_wg<4, 4, 1, 0>(ds, n, stream);
  if (d.up == 2) return wg_winograd(d) ? pick_wg<3, 1, 2, 0, true>(ds, n, stream) : pick_wg<3, 1, 2, 0>(ds, n, stream);
  if (d.up == 4) return wg_winograd(d) ? pick_wg<3, 1, 4, 0, true>(ds, n, stream) : pick_wg<3, 1, 4, 0>(ds, n, stream);
  if (d.KT == 3 && wg_winograd(d))
    return d.prologue == 1 ? pick_wg<3, 1, 1, 1, true>(ds, n, stream) : pick_wg<3, 1, 1, 0, true>(ds, n, stream);
  if (d.KT == 3) return d.prologue == 1 ? pick_wg<3, 1, 1, 1>(ds, n, stream) : pick_wg<3, 1, 1, 0>(ds, n, stream);
  return d.prologue == 1 ? pick_wg<1, 1, 1, 1>(ds, n, stream) : pick_wg<1, 1, 1, 0>(ds, n, stream);
}

int adp_wgrad_mm(const adp_wgrad_desc& d, void* stream) { return adp_wgrad_mm_n(&d, 1, stream); }
