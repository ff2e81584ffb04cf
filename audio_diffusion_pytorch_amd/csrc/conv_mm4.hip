// Winograd F(4,3) implicit-GEMM Conv1d for the wide ResnetItem ConvBlock convs and their data gradients
// (kernel 3, stride 1, 'same' padding, >= 64 channels (round 6; 128 before), SiLU(GroupNorm(.)) already materialised by the statistics' second
// stage; /root/reference/audio_diffusion_pytorch/components.py:89).  Same machine model as
// conv_mm_impl.h -- wave-specialised block, loaders that only copy, exact-f32 v_mfma_f32_32x32x2_f32 -- with one change of
// tile economy:
//   * column l31 of an MMA tile is an output QUAD (positions n0 + 4*l31 ..+3), so a 32 x 32 MFMA tile covers 32 rows x 128
//     positions, and the six Winograd planes  P_p = sum_c U_p[m][c] * V_p[c][quad]  replace the 12 tap x position-tile
//     products of the direct form (F(2,3): 8): 6 MFMAs per channel pair and 128 positions = HALF the direct form's matrix work,
//     three quarters of F(2,3)'s.
//   * the PLANES are what the MMA waves of a block split (next to the channels of a staged chunk): wave = (plane group pg,
//     K group kg); plane group 0 owns planes 0-2, group 1 planes 3-5.  Every wave reads the same untransformed LDS tiles --
//     three taps and a 16-byte input quad plus one halo value (its left / right neighbour through a DPP lane shift, the
//     tile edge through a broadcast read) -- and forms only ITS three planes of
//         U' = (g0, g0+g1+g2, g0-g1+g2 | g0+2g1+4g2, g0-2g1+4g2, g2)           (G without its constants)
//         V  = (4d0-5d2+d4, t1+t2, t1-t2 | t3+2t4, t3-2t4, 4d1-5d3+d5)         t1 = d4-4d2, t2 = d3-4d1, t3 = d4-d2, t4 = d3-d1
//     with 9 VALU ops per 3 MFMAs, in the shadow of the 64-cycle MFMAs (tools/probe/alu_probe: ~5 issue slots per MFMA are free).
//     Three accumulator tiles per wave (48 registers) instead of four.
//   * the planes and K groups meet in LDS once, after the K loop: every wave sums two accumulator rows of all 24 partial tiles
//     in a fixed order, applies the plane constants (1/4, -1/6, -1/6, 1/24, 1/24, 1) and A^T, and stores 16 bytes per lane:
//         y0 = p0+p1+p2+p3+p4   y1 = (p1-p2) + 2(p3-p4)   y2 = (p1+p2) + 4(p3+p4)   y3 = (p1-p2) + 8(p3-p4) + p5
//   * a block is 32 rows x 128 positions x all channels: 8 MMA waves (2 per SIMD) + 4 loader waves, 256 blocks for the
//     [4, 1024, 256] layers of depth 7 -- one resident generation on 256 CUs.
// fp32 throughout; error against fp64 ~1e-6 of the output's max norm (F(4,3)'s constants; tests/test_kernels.py).
#include <stdlib.h>
#include "adp_rt.h"
#include "adp.h"
#include "conv_internal.h"

#ifdef ADP_KTRACE
static __device__ unsigned long long* m4_kt_buf = nullptr;
extern "C" int adp_ktrace_set_mm4(void* p) {
  return hipMemcpyToSymbol(HIP_SYMBOL(m4_kt_buf), &p, sizeof(p)) == hipSuccess ? 0 : -1;
}
#define M4_KT_BUF m4_kt_buf
#else
#define M4_KT_BUF nullptr
#endif

namespace {

constexpr int M4_BM = 32, M4_BN = 128, M4_KT = 3;
constexpr int M4_NPG = 2, M4_NLD = 4;
constexpr bool M4_VPRE_DEFAULT = false;
constexpr int M4_NLT = M4_NLD * 64;
constexpr int M4_XSP = M4_BN + 8, M4_XQ = M4_XSP / 4;         // X row: positions n0-4 .. n0+131

// four consecutive virtual positions u0..u0+3 (u0 % 4 == 0) of a row upsampled by UP: 4 / 2 / 1 source floats (UpsampleItem:
// the [B, C, L*UP] intermediate is never materialised; the LDS tile holds virtual positions, so the MMA waves see a plain conv)
template <int UP>
__device__ __forceinline__ f32x4 m4_load_xquad(const float* p) {
  if (UP == 1) return *reinterpret_cast<const f32x4*>(p);
  if (UP == 2) {
    const f32x2 t = *reinterpret_cast<const f32x2*>(p);
    return f32x4{t[0], t[0], t[1], t[1]};
  }
  const float t = *p;
  return f32x4{t, t, t, t};
}

// BKT: channels per staged chunk (32, or 64: half the barriers per K, 16 channels per K group and chunk)
// NKG: K groups (MMA waves per plane group): 4 = the 12-wave block of the long-K layers (one per CU: 96 KB of partial tiles);
//      2 = an 8-wave block with 60 KB of LDS for the short-K layers -- two of them share a CU, one's loads / exchange / stores
//      under the other's MFMAs (a 4-chunk K loop is all ramp and drain otherwise)
// UP: nearest-upsample factor folded into the X loader (the UpsampleItem convs); store mode 2 (runtime: the pooled store of
//     their data gradients -- sums of sp = 2 / 4 adjacent outputs of the lane's quad, 8- / 4-byte stores, + residual)
// GNB: the launch also leaves the first stage of a GroupNorm backward (adp_conv_desc.gnb_ab) -- its own instantiations (plain
//      transposed launches), so that every other launch keeps the registers and instruction stream it had without it
// NPG: plane groups = MMA waves per K group.  2: three planes per wave (the blocks above).  3 (round 6, 16-wave block: 12 MMA + 4
//      loader waves, THREE MMA waves per SIMD, two planes each -- (0,1) with the left halo, (2,3) with none, (4,5) with the right one):
//      a wave's stream is a third shorter, three of them interleave on the SIMD; the first eight MMA waves run the epilogue
// VPRE (round 6, full 128-position tiles, UP = 1): the LOADER waves stage the TRANSFORMED inputs -- V[quad][plane][channel], the six
//      planes of a quad for four channels per 16-byte store -- instead of raw x rows: the MMA waves then read one float4 per plane
//      and four channels and spend no VALU / DPP on B^T d at all (their stream was ~45 VALU + 20 LDS instructions per 12 MFMAs and
//      closed the barrier; the loaders had the slack).
template <bool TR, int PD, int BKT, int M4_NKG, int UP = 1, bool GNB = false, int NPG = M4_NPG, bool VPRE = false>
// (second launch bound = waves per SIMD the register allocation has to leave room for: the light 8-wave block lives on TWO
//  blocks per CU = 4 waves per SIMD = at most 128 registers; the 12-wave block on one = 3 per SIMD)
__global__ __launch_bounds__((M4_NKG * NPG + M4_NLD) * 64, (M4_NKG == 2 && BKT == 32) ? 4 : 1) void conv_mm4_kernel(adp_conv_desc d) {
  static_assert(NPG == 2 || (NPG == 3 && M4_NKG == 4), "three plane groups: the 16-wave form of the 4-K-group block");
  static_assert(!VPRE || (UP == 1 && NPG == 2), "pre-transformed inputs: plain convs, two plane groups");
  constexpr int M4_QS = 6 * BKT + 4;                            // VPRE: floats per quad row of V (4 mod 64: conflict-free b128 columns)
  constexpr int M4_NMMA = M4_NKG * NPG, PPW = 6 / NPG;          // MMA waves; planes per wave
  constexpr int NEPI = NPG == 3 ? 8 : M4_NMMA, RPW = 16 / NEPI;  // waves that run the epilogue; accumulator rows each finishes (2 or 4)
  constexpr int M4_RED = M4_NKG * 6 * 1024;                     // parked partial tiles
  constexpr int M4_QK = BKT * M4_KT;                            // floats of a forward weight row per chunk
  constexpr int M4_AROWS = TR ? BKT : M4_BM;
  constexpr int M4_AS = (TR ? M4_BM * M4_KT : M4_QK) + 4;       // A row stride (4 mod 8 dwords)
  constexpr int M4_AQ = (TR ? M4_BM * M4_KT : M4_QK) / 4;       // float4 per A row
  constexpr int M4_A_ELEMS = M4_AROWS * M4_AS, M4_X_ELEMS = VPRE ? 32 * M4_QS : BKT * M4_XSP;
  constexpr int M4_NA4 = (M4_AROWS * M4_AQ + M4_NLT - 1) / M4_NLT, M4_NX4 = (BKT * M4_XQ + M4_NLT - 1) / M4_NLT;
  constexpr int M4_STAGE = 2 * (M4_A_ELEMS + M4_X_ELEMS);
  constexpr int M4_SM = M4_RED > M4_STAGE ? M4_RED : M4_STAGE;
  constexpr int CPK = BKT / M4_NKG;                             // channels of a chunk one K group multiplies (8 or 16)
  __shared__ __attribute__((aligned(16))) float smem[M4_SM];
  constexpr int KT = M4_KT, AS = M4_AS, XSP = M4_XSP;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hi = lane >> 5, l31 = lane & 31;
  const int M = (int)d.M, R = (int)d.R, L = (int)d.Lin, N = (int)d.N;
  ADP_KT_DECL(M4_KT_BUF)
  ADP_KT(0);

  // ---- XCD-aware decode of the 1-D grid (each XCD gets a contiguous range of weight row tiles)
  int id = blockIdx.x;
  const int total = gridDim.x;
  if ((total & 7) == 0) id = (id & 7) * (total >> 3) + (id >> 3);
  const int ntn = (N + M4_BN - 1) / M4_BN, per_m = ntn * (int)d.B;
  const int mt = id / per_m, rem = id - mt * per_m;
  const int b = rem / ntn, nt = rem - b * ntn;
  const int m0 = mt * M4_BM, n0 = nt * M4_BN;
  // Cross-workgroup K split (gridDim.y = KS > 1: grids of fewer than ~200 blocks, e.g. the [4, 1024, 128] layers of depth 8):
  // this block reduces channel chunks [c_lo, c_lo + nchunks) only and parks its raw (output-transformed) partial tile in
  // d.ws[ks]; bias / e_scale / residual / GroupNorm partials then run in conv_splitk_reduce[_gn]_kernel (conv_mm.hip), which
  // sums the KS partials in a fixed order.
  const int ks = blockIdx.y, KS = gridDim.y;
  const int cps = (R / BKT + KS - 1) / KS;
  const int c_lo = ks * cps;
  const int nchunks = (R / BKT - c_lo) < cps ? (R / BKT - c_lo) : cps;
  const int nrounds = ((nchunks + PD - 1) / PD) * PD;

  if (wave >= M4_NMMA) {
    // =========================== loader waves (pure copies: global -> registers -> LDS) ===========================
    ADP_LOADER_PRIO_SET();
    const int lt = tid - M4_NMMA * 64;
    const float* xb = d.x + (int64_t)b * R * L;
    const float* wbase = TR ? d.w + (int64_t)m0 * KT : d.w + (int64_t)m0 * R * KT;
    if constexpr (UP == 1) {
      if (N % M4_BN == 0) {
        // ---- LEAN loader (round 6; full tiles: every ConvBlock conv of the README net).  tools/ktrace.py: at [4, 1024, 256] the
        // MMA waves spent 1400-1700 of every chunk's ~5000 cycles in the barrier waiting for the LOADERS -- whose global loads
        // had landed within 100-400 cycles, and whose ~130 instructions per chunk (four selects per staged quad for zero padding
        // that only the two halo quads of a row can need, 64-bit address arithmetic and an LDS address per slot) then took
        // 4000-4700 cycles of issue slots beside two MFMA waves per SIMD.  Here a lane's slots walk rows (X) / quad columns (A) at
        // constant strides: one 32-bit lane offset per tile on wave-uniform bases, LDS addresses = base + immediates, no select on
        // the 32 interior quads of an X row; the first 2 * BKT lanes stage the halo quads and are the only ones to test the row's ends.
        constexpr int LPR = M4_NLT / M4_AROWS;        // lanes per A row (8, or 4 for the 64-row transposed view)
        constexpr int NA = M4_AQ / LPR;               // A slots per lane (quad columns qq0 + LPR * i)
        constexpr int NX = VPRE ? BKT / 32 * 4 : BKT / 8;  // X slots per lane (rows row0 + 8 i of one interior quad column;
                                                      // VPRE: BKT / 32 groups of four consecutive channels of ONE quad)
        static_assert(M4_AQ % LPR == 0 && M4_NLT % M4_AROWS == 0, "A tile: whole quad columns per lane");
        const int arow = lt / LPR, aq0 = lt % LPR;
        const unsigned a_off = (unsigned)((TR ? arow * M * KT : arow * R * KT) + 4 * aq0);
        const int a_lds = arow * AS + 4 * aq0;
        const int xrow0 = lt >> 5, xpq = 1 + (lt & 31);
        const unsigned x_off = VPRE ? (unsigned)(4 * xrow0 * L + (n0 + 4 * (lt & 31))) : (unsigned)(xrow0 * L + (n0 - 4 + 4 * xpq));
        const int x_lds = VPRE ? (lt & 31) * M4_QS + 4 * xrow0 : xrow0 * XSP + 4 * xpq;
        // VPRE: quad q = lt & 31 of channels 4 (xrow0 + 8 g) .. + 3, g < BKT / 32; the tile-edge lanes (q = 0 / 31) fetch the one input
        // beyond the tile themselves, the others take d0 / d5 from their neighbour lanes
        const int vq = lt & 31;
        const bool v_edge = vq == 0 || vq == 31;
        const int v_hu = vq == 0 ? n0 - 1 : n0 + M4_BN;
        const bool v_hok = v_hu >= 0 && v_hu < L;
        const unsigned v_hoff = (unsigned)(4 * xrow0 * L + (v_hok ? v_hu : 0));
        const bool halo_lane = !VPRE && lt < 2 * BKT;  // (wave-uniform: 2 * BKT is a multiple of 64)
        const int hrow = lt % BKT, hside = lt / BKT;  // side 0: positions n0-4 .. n0-1, side 1: n0+128 .. n0+131
        const int hu = hside ? n0 + M4_BN : n0 - 4;
        const bool h_ok = hu >= 0 && hu < L;          // (a property of the tile, not of the chunk)
        const unsigned h_off = (unsigned)(hrow * L + (h_ok ? hu : 0));
        const int h_lds = hrow * XSP + (hside ? M4_BN + 4 : 0);
        f32x4 ra[PD][NA], rx[PD][NX], rh[PD][2];  // (rh: the halo quad of a row; VPRE: d0 / d5 of the edge lanes, one float4 per group)
        auto load_chunk = [&](f32x4 (&a)[NA], f32x4 (&x)[NX], f32x4 (&h)[2], int chunk) {
          const int rn = (c_lo + (chunk < nchunks ? chunk : nchunks - 1)) * BKT;  // (the tail re-reads the last chunk: never consumed)
          const float* wp = TR ? wbase + (int64_t)rn * M * KT : wbase + rn * KT;  // wave-uniform
          const float* xp = xb + (int64_t)rn * L;
#pragma unroll
          for (int i = 0; i < NA; ++i) a[i] = *reinterpret_cast<const f32x4*>(wp + (a_off + (unsigned)(4 * LPR * i)));
          if constexpr (VPRE) {
#pragma unroll
            for (int i = 0; i < NX; ++i)  // slot i = 4 g + k: channel 4 (xrow0 + 8 g) + k
              x[i] = *reinterpret_cast<const f32x4*>(xp + (x_off + (unsigned)((32 * (i >> 2) + (i & 3)) * L)));
            if (v_edge) {
#pragma unroll
              for (int g = 0; g < BKT / 32; ++g)
#pragma unroll
                for (int k = 0; k < 4; ++k) h[g][k] = xp[v_hoff + (unsigned)((32 * g + k) * L)];
            }
          } else {
#pragma unroll
            for (int i = 0; i < NX; ++i) x[i] = *reinterpret_cast<const f32x4*>(xp + (x_off + (unsigned)(8 * i * L)));
            if (halo_lane) h[0] = *reinterpret_cast<const f32x4*>(xp + h_off);
          }
        };
        auto store_chunk = [&](const f32x4 (&a)[NA], const f32x4 (&x)[NX], const f32x4 (&h)[2], int chunk) {
          float* Ab = smem + (chunk & 1) * (M4_A_ELEMS + M4_X_ELEMS);
          float* Xb = Ab + M4_A_ELEMS;
#pragma unroll
          for (int i = 0; i < NA; ++i) *reinterpret_cast<f32x4*>(Ab + a_lds + 4 * LPR * i) = a[i];
          if constexpr (VPRE) {
            // V = B^T d per quad: (4 d0 - 5 d2 + d4, t1 + t2, t1 - t2 | t3 + 2 t4, t3 - 2 t4, 4 d1 - 5 d3 + d5), t1 = d4 - 4 d2, t2 = d3 - 4 d1,
            // t3 = d4 - d2, t4 = d3 - d1; plane p of the group's four channels = one 16-byte store at V[q][p][4 (xrow0 + 8 g)]
#pragma unroll
            for (int g = 0; g < BKT / 32; ++g) {
              f32x4 vp[6];
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                const f32x4 dq = x[4 * g + k];
                const float d1 = dq[0], d2 = dq[1], d3 = dq[2], d4 = dq[3];
                const float pv = adp_lane_prev(0.0f, d4), nx = adp_lane_next(0.0f, d1);
                const float hv = v_hok ? h[g][k] : 0.0f;  // zero padding beyond the row's ends
                const float d0 = vq == 0 ? hv : pv, d5 = vq == 31 ? hv : nx;
                const float t1 = fmaf(-4.0f, d2, d4), t2 = fmaf(-4.0f, d1, d3), t3 = d4 - d2, t4 = d3 - d1;
                vp[0][k] = fmaf(4.0f, d0, fmaf(-5.0f, d2, d4));
                vp[1][k] = t1 + t2;
                vp[2][k] = t1 - t2;
                vp[3][k] = fmaf(2.0f, t4, t3);
                vp[4][k] = fmaf(-2.0f, t4, t3);
                vp[5][k] = fmaf(4.0f, d1, fmaf(-5.0f, d3, d5));
              }
#pragma unroll
              for (int pl = 0; pl < 6; ++pl) *reinterpret_cast<f32x4*>(Xb + x_lds + 32 * g + pl * BKT) = vp[pl];
            }
            return;
          }
#pragma unroll
          for (int i = 0; i < NX; ++i) *reinterpret_cast<f32x4*>(Xb + x_lds + 8 * i * XSP) = x[i];
          if (halo_lane) {
            f32x4 v = h[0];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = h_ok ? v[j] : 0.0f;  // zero padding
            *reinterpret_cast<f32x4*>(Xb + h_lds) = v;
          }
        };
#pragma unroll
        for (int s = 0; s < PD; ++s) load_chunk(ra[s], rx[s], rh[s], s);
        for (int c0 = 0; c0 < nrounds; c0 += PD) {
#pragma unroll
          for (int s = 0; s < PD; ++s) {
#ifdef ADP_KTRACE
            if (c0 + s < 16) ADP_KT(1 + 3 * (c0 + s));
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (c0 + s < 16) ADP_KT(2 + 3 * (c0 + s));
#endif
            store_chunk(ra[s], rx[s], rh[s], c0 + s);
            load_chunk(ra[s], rx[s], rh[s], c0 + s + PD);
#ifdef ADP_KTRACE
            if (c0 + s < 16) ADP_KT(3 + 3 * (c0 + s));
#endif
            __syncthreads();  // B_c
          }
        }
        ADP_KT(60);
        __syncthreads();  // staging buffers free
        __syncthreads();  // partial tiles parked
        ADP_KT_DUMP(blockIdx.x);
        return;
      }
    }
    int a_src[M4_NA4], a_dst[M4_NA4];
#pragma unroll
    for (int i = 0; i < M4_NA4; ++i) {
      const int e = (lt + i * M4_NLT) % (M4_AROWS * M4_AQ);
      const int row = e / M4_AQ, qq = e - row * M4_AQ;
      a_dst[i] = row * AS + 4 * qq;
      a_src[i] = TR ? row * M * KT + 4 * qq : row * R * KT + 4 * qq;
    }
    int x_src[M4_NX4], x_dst[M4_NX4];
    bool x_ok[M4_NX4];
#pragma unroll
    for (int i = 0; i < M4_NX4; ++i) {
      const int e = (lt + i * M4_NLT) % (BKT * M4_XQ);
      const int rl = e / M4_XQ, pq = e - rl * M4_XQ;
      const int u = n0 - 4 + 4 * pq;  // (virtual position on the upsampled row)
      x_dst[i] = rl * XSP + 4 * pq;
      x_ok[i] = (u >= 0 && u < L * UP);  // (L * UP) % 4 == 0: a quad is entirely inside or outside the row
      x_src[i] = rl * L + (x_ok[i] ? u / UP : 0);  // nearest upsample: source index = floor(u / UP), exact
    }
    f32x4 ra[PD][M4_NA4], rx[PD][M4_NX4];
    auto load_chunk = [&](f32x4 (&a)[M4_NA4], f32x4 (&x)[M4_NX4], int chunk) {
      const int rn = (c_lo + (chunk < nchunks ? chunk : nchunks - 1)) * BKT;  // (the tail re-reads the last chunk: never consumed)
      const float* wp = TR ? wbase + (int64_t)rn * M * KT : wbase + rn * KT;
#pragma unroll
      for (int i = 0; i < M4_NA4; ++i) a[i] = *reinterpret_cast<const f32x4*>(wp + a_src[i]);
      const float* xp = xb + (int64_t)rn * L;
#pragma unroll
      for (int i = 0; i < M4_NX4; ++i) x[i] = m4_load_xquad<UP>(xp + x_src[i]);
    };
    auto store_chunk = [&](const f32x4 (&a)[M4_NA4], const f32x4 (&x)[M4_NX4], int chunk) {
      float* Ab = smem + (chunk & 1) * (M4_A_ELEMS + M4_X_ELEMS);
      float* Xb = Ab + M4_A_ELEMS;
#pragma unroll
      for (int i = 0; i < M4_NA4; ++i) *reinterpret_cast<f32x4*>(Ab + a_dst[i]) = a[i];
#pragma unroll
      for (int i = 0; i < M4_NX4; ++i) {
        f32x4 v = x[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = x_ok[i] ? v[j] : 0.0f;  // zero padding
        *reinterpret_cast<f32x4*>(Xb + x_dst[i]) = v;
      }
    };
#pragma unroll
    for (int s = 0; s < PD; ++s) load_chunk(ra[s], rx[s], s);
    // Hazard note (as conv_mm): the store of chunk c goes to LDS[c & 1], last read by the MFMAs of chunk c-2; every MMA wave
    // finished those before it arrived at barrier B_{c-1}, which this wave passed before starting iteration c.
    for (int c0 = 0; c0 < nrounds; c0 += PD) {
#pragma unroll
      for (int s = 0; s < PD; ++s) {
#ifdef ADP_KTRACE
        if (c0 + s < 16) ADP_KT(1 + 3 * (c0 + s));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (c0 + s < 16) ADP_KT(2 + 3 * (c0 + s));
#endif
        store_chunk(ra[s], rx[s], c0 + s);
        load_chunk(ra[s], rx[s], c0 + s + PD);
#ifdef ADP_KTRACE
        if (c0 + s < 16) ADP_KT(3 + 3 * (c0 + s));
#endif
        __syncthreads();  // B_c
      }
    }
    ADP_KT(60);
    __syncthreads();  // staging buffers free
    __syncthreads();  // partial tiles parked
    ADP_KT_DUMP(blockIdx.x);
    return;
  }

  // =========================== MMA waves ===========================
  const int pg = NPG == 3 ? wave % 3 : (wave & 1), kg = NPG == 3 ? wave / 3 : (wave >> 1);
  const int ewave = wave < NEPI ? wave : 0;  // (NPG = 3: waves 8-11 leave after the exchange; their epilogue operands shadow wave 0's)
  f32x16 acc[PPW];
#pragma unroll
  for (int p = 0; p < PPW; ++p)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[p][r] = 0.0f;

  // epilogue operands of the two accumulator rows this wave finishes, fetched NOW (their latency lies under the K loop)
  const int nq = n0 + 4 * l31;
  f32x4 pre_res[RPW];
  float pre_bias[RPW], pre_scale[RPW];
#pragma unroll
  for (int rr = 0; rr < RPW; ++rr) {
    const int r = RPW * ewave + rr;
    const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
    const int mc = m < M ? m : M - 1;
    pre_bias[rr] = d.bias ? d.bias[mc] : 0.0f;
    pre_scale[rr] = d.e_scale ? d.e_scale[b * (d.e_bstride ? d.e_bstride : M) + mc] : 1.0f;
    pre_res[rr] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    if (d.res && m < M && nq < N && KS == 1 && d.store == 0) pre_res[rr] = *reinterpret_cast<const f32x4*>(d.res + ((int64_t)b * M + m) * N + nq);
  }

  // (gnb_x, wanted after the K loop, is touched now -- one dword per lane and row, dropped: the rows' lines wait in L2 by then)
  float gnb_warm = 0.0f;
  if (GNB && KS == 1 && d.store == 0 && nq < N) {
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
      const int r = RPW * ewave + rr;
      const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      if (m < M) gnb_warm += d.gnb_x[((int64_t)b * M + m) * N + nq];
    }
  }

  const int xfrag = 4 * hi * XSP + 4 * l31 + 4;                         // + (ci + cc) * XSP: the lane's input quad d1..d4
  const int hfrag = 4 * hi * XSP + (pg == 0 ? 3 : 4 + M4_BN);           // tile-edge halo (d0 of quad 0 / d5 of quad 31)
  const int afrag = TR ? 4 * hi * AS + l31 * KT : l31 * AS + 4 * hi * KT;
  const bool edge = pg == 0 ? (l31 == 0) : (l31 == 31);                 // (NPG = 3: plane group 1 needs no halo)

  for (int c = 0; c < nrounds; ++c) {
#ifdef ADP_KTRACE
    if (c < 16) ADP_KT(1 + 3 * c);
#endif
    __syncthreads();  // B_c: chunk c is in LDS[c & 1]
#ifdef ADP_KTRACE
    if (c < 16) ADP_KT(2 + 3 * c);
#endif
    if (c < nchunks) {
      const float* Ab = smem + (c & 1) * (M4_A_ELEMS + M4_X_ELEMS);
      const float* Xb = Ab + M4_A_ELEMS;
      // The fragments of channel group sub + 1 are requested BEFORE the twelve MFMAs of group sub (round 6): left to itself the
      // compiler kept ONE register set per group (ds_read, s_waitcnt lgkmcnt(0), transforms, MFMAs, next ds_read ...), so every
      // group exposed an LDS round trip to the matrix pipe (tools/ktrace.py: ~4600 cycles per chunk and SIMD for 3072 of MFMAs).
      auto load_frag = [&](float (&av)[4 * KT], f32x4 (&qx)[4], float (&hx)[4], int sub) {
        const int ci = kg * CPK + sub * 8;
        if constexpr (VPRE) {  // qx[p] = plane 3 pg + p of this lane's quad for channels ci + 4 hi .. + 3 (one float4); hx unused
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) qx[pl] = *reinterpret_cast<const f32x4*>(Xb + l31 * M4_QS + (3 * pg + pl) * BKT + ci + 4 * hi);
          qx[3] = qx[0];
#pragma unroll
          for (int cc = 0; cc < 4; ++cc) hx[cc] = 0.0f;
        }
        // av[cc * 3 + t] = tap t of channel ci + cc + 4 * hi for this lane's output row
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
        if constexpr (!VPRE) {
#pragma unroll
          for (int cc = 0; cc < 4; ++cc) {
            qx[cc] = *reinterpret_cast<const f32x4*>(Xb + xfrag + (ci + cc) * XSP);
            hx[cc] = Xb[hfrag + (ci + cc) * XSP];  // one address per half-wave: a broadcast read
          }
        }
      };
      // (measured round 6 on the 12-wave block: no gain -- the pipe's idle time is the lock step of the two MMA waves of a SIMD
      //  after each barrier, not the LDS round trips; -DADP_MM4_PIPE keeps the variant for A/B.  Never in the light block, which
      //  has to stay within 128 registers to share its CU with a second block.)
#ifdef ADP_MM4_PIPE
      constexpr bool PIPE = (M4_NKG == 4);
#else
      constexpr bool PIPE = false;
#endif
      float avn[4 * KT], hxn[4];
      f32x4 qxn[4];
      if (PIPE) load_frag(avn, qxn, hxn, 0);
#pragma unroll
      for (int sub = 0; sub < CPK / 8; ++sub) {
        float av[4 * KT], hx[4];
        f32x4 qx[4];
        if (PIPE) {
#pragma unroll
          for (int j = 0; j < 4 * KT; ++j) av[j] = avn[j];
#pragma unroll
          for (int j = 0; j < 4; ++j) qx[j] = qxn[j], hx[j] = hxn[j];
          if (sub + 1 < CPK / 8) load_frag(avn, qxn, hxn, sub + 1);
          adp_sched_fence();  // (the requests above stay above the matrix work that hides them)
        } else {
          load_frag(av, qx, hx, sub);
        }
        if constexpr (VPRE) {
          if (pg == 0) {
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
              const float g0 = av[cc * KT], g1 = av[cc * KT + 1], g2 = av[cc * KT + 2];
              const float gs = g0 + g2;
              acc[0] = adp_mfma32(g0, qx[0][cc], acc[0]);
              acc[1] = adp_mfma32(gs + g1, qx[1][cc], acc[1]);
              acc[2] = adp_mfma32(gs - g1, qx[2][cc], acc[2]);
            }
          } else {
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
              const float g0 = av[cc * KT], g1 = av[cc * KT + 1], g2 = av[cc * KT + 2];
              const float gq = fmaf(4.0f, g2, g0);
              acc[0] = adp_mfma32(fmaf(2.0f, g1, gq), qx[0][cc], acc[0]);
              acc[1] = adp_mfma32(fmaf(-2.0f, g1, gq), qx[1][cc], acc[1]);
              acc[2] = adp_mfma32(g2, qx[2][cc], acc[2]);
            }
          }
        } else if constexpr (NPG == 3) {
          if (pg == 0) {         // planes 0, 1
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
              const float d1 = qx[cc][0], d2 = qx[cc][1], d3 = qx[cc][2], d4 = qx[cc][3];
              const float nb = adp_lane_prev(0.0f, d4);
              const float d0 = edge ? hx[cc] : nb;
              const float g0 = av[cc * KT], g1 = av[cc * KT + 1], g2 = av[cc * KT + 2];
              acc[0] = adp_mfma32(g0, fmaf(4.0f, d0, fmaf(-5.0f, d2, d4)), acc[0]);
              acc[1] = adp_mfma32(g0 + g2 + g1, fmaf(-4.0f, d2, d4) + fmaf(-4.0f, d1, d3), acc[1]);
            }
          } else if (pg == 1) {  // planes 2, 3: no halo
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
              const float d1 = qx[cc][0], d2 = qx[cc][1], d3 = qx[cc][2], d4 = qx[cc][3];
              const float g0 = av[cc * KT], g1 = av[cc * KT + 1], g2 = av[cc * KT + 2];
              acc[0] = adp_mfma32(g0 + g2 - g1, fmaf(-4.0f, d2, d4) - fmaf(-4.0f, d1, d3), acc[0]);
              acc[1] = adp_mfma32(fmaf(2.0f, g1, fmaf(4.0f, g2, g0)), fmaf(2.0f, d3 - d1, d4 - d2), acc[1]);
            }
          } else {               // planes 4, 5
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
              const float d1 = qx[cc][0], d2 = qx[cc][1], d3 = qx[cc][2], d4 = qx[cc][3];
              const float nb = adp_lane_next(0.0f, d1);
              const float d5 = edge ? hx[cc] : nb;
              const float g0 = av[cc * KT], g1 = av[cc * KT + 1], g2 = av[cc * KT + 2];
              acc[0] = adp_mfma32(fmaf(-2.0f, g1, fmaf(4.0f, g2, g0)), fmaf(-2.0f, d3 - d1, d4 - d2), acc[0]);
              acc[1] = adp_mfma32(g2, fmaf(4.0f, d1, fmaf(-5.0f, d3, d5)), acc[1]);
            }
          }
        } else if (pg == 0) {
#pragma unroll
          for (int cc = 0; cc < 4; ++cc) {
            const float d1 = qx[cc][0], d2 = qx[cc][1], d3 = qx[cc][2], d4 = qx[cc][3];
            const float nb = adp_lane_prev(0.0f, d4);         // the left neighbour quad's last input
            const float d0 = edge ? hx[cc] : nb;
            const float g0 = av[cc * KT], g1 = av[cc * KT + 1], g2 = av[cc * KT + 2];
            const float t1 = fmaf(-4.0f, d2, d4), t2 = fmaf(-4.0f, d1, d3);
            const float gs = g0 + g2;
            acc[0] = adp_mfma32(g0, fmaf(4.0f, d0, fmaf(-5.0f, d2, d4)), acc[0]);
            acc[1] = adp_mfma32(gs + g1, t1 + t2, acc[1]);
            acc[PPW - 1] = adp_mfma32(gs - g1, t1 - t2, acc[PPW - 1]);
          }
        } else {
#pragma unroll
          for (int cc = 0; cc < 4; ++cc) {
            const float d1 = qx[cc][0], d2 = qx[cc][1], d3 = qx[cc][2], d4 = qx[cc][3];
            const float nb = adp_lane_next(0.0f, d1);         // the right neighbour quad's first input
            const float d5 = edge ? hx[cc] : nb;
            const float g0 = av[cc * KT], g1 = av[cc * KT + 1], g2 = av[cc * KT + 2];
            const float t3 = d4 - d2, t4 = d3 - d1;
            const float gq = fmaf(4.0f, g2, g0);
            acc[0] = adp_mfma32(fmaf(2.0f, g1, gq), fmaf(2.0f, t4, t3), acc[0]);
            acc[1] = adp_mfma32(fmaf(-2.0f, g1, gq), fmaf(-2.0f, t4, t3), acc[1]);
            acc[PPW - 1] = adp_mfma32(g2, fmaf(4.0f, d1, fmaf(-5.0f, d3, d5)), acc[PPW - 1]);
          }
        }
      }
    }
#ifdef ADP_KTRACE
    if (c < 16) ADP_KT(3 + 3 * c);
#endif
  }
  ADP_KT(60);
  // first stage of the backward of SiLU(GroupNorm(gnb_x)) whose output gradient this tile is (adp_conv_desc.gnb_ab): the rows'
  // operands are requested here, AFTER the K loop (no registers held through it), under the plane exchange below
  const bool gnb = GNB && KS == 1 && d.store == 0;
#ifndef ADP_EMULATE
  if (GNB) asm volatile("" ::"v"(gnb_warm));  // (keeps the touch above alive; the value is not used)
#endif
  f32x4 gnb_xq[RPW];
  if (gnb) {  // (the x quads only: the rows' statistics / affine parameters are fetched where they are used, after the stores --
              //  the light block lives on 128 registers)
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
      const int r = RPW * ewave + rr;
      const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      gnb_xq[rr] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      if (m < M && nq < N) gnb_xq[rr] = *reinterpret_cast<const f32x4*>(d.gnb_x + ((int64_t)b * M + m) * N + nq);
    }
  }
  __syncthreads();  // the staging buffers are free
  ADP_KT(61);

  // ---- plane / K-group exchange through LDS: tile (kg, plane P) at smem[(kg * 6 + P) * 1024 + r * 64 + lane]
#pragma unroll
  for (int p = 0; p < PPW; ++p) {
    float* rp = smem + (kg * 6 + pg * PPW + p) * 1024 + lane;
#pragma unroll
    for (int r = 0; r < 16; ++r) rp[r * 64] = acc[p][r];
  }
  __syncthreads();
  if (NPG == 3 && wave >= NEPI) {  // (the last four MMA waves have no rows to finish)
    ADP_KT_DUMP(blockIdx.x);
    return;
  }

  // ---- output transform: this wave finishes accumulator rows RPW * wave .. (for both halves of the wave)
  const bool nok = nq < N;  // N % 4 == 0: a quad is inside or outside
  f32x4 yv[RPW];
#pragma unroll
  for (int rr = 0; rr < RPW; ++rr) {
    const int r = RPW * wave + rr;
    float s[6];
#pragma unroll
    for (int P = 0; P < 6; ++P) {
      float v = 0.0f;
#pragma unroll
      for (int g = 0; g < M4_NKG; ++g) v += smem[(g * 6 + P) * 1024 + r * 64 + lane];  // fixed order: deterministic
      s[P] = v;
    }
    const float p0 = 0.25f * s[0], p1 = s[1] * (-1.0f / 6.0f), p2 = s[2] * (-1.0f / 6.0f);
    const float p3 = s[3] * (1.0f / 24.0f), p4 = s[4] * (1.0f / 24.0f), p5 = s[5];
    const float a12 = p1 + p2, d12 = p1 - p2, a34 = p3 + p4, d34 = p3 - p4;
    yv[rr][0] = p0 + a12 + a34;
    yv[rr][1] = fmaf(2.0f, d34, d12);
    yv[rr][2] = fmaf(4.0f, a34, a12);
    yv[rr][3] = fmaf(8.0f, d34, d12) + p5;
  }
  const bool final_tile = (KS == 1);

  // ---- epilogue
  float vfin[RPW][4];
#pragma unroll
  for (int rr = 0; rr < RPW; ++rr) {
    const int r = RPW * wave + rr;
    f32x4 y = yv[rr];
    const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
    const bool ok = (m < M) && nok;
#pragma unroll
    for (int k = 0; k < 4; ++k) vfin[rr][k] = 0.0f;
    if (!ok) continue;
    if (d.store == 2) {  // pooled store (gradient of the nearest upsample): sums of sp adjacent positions, no bias
      if (d.sp == 2) {
        const int64_t o = ((int64_t)b * M + m) * (N / 2) + nq / 2;
        f32x2 v{y[0] + y[1], y[2] + y[3]};
        if (d.res) {
          const f32x2 r2 = *reinterpret_cast<const f32x2*>(d.res + o);
          v[0] += r2[0], v[1] += r2[1];
        }
        *reinterpret_cast<f32x2*>(d.out + o) = v;
      } else {
        const int64_t o = ((int64_t)b * M + m) * (N / 4) + nq / 4;
        float v = (y[0] + y[1]) + (y[2] + y[3]);
        if (d.res) v += d.res[o];
        d.out[o] = v;
      }
      continue;
    }
    if (!final_tile) {  // raw partial tile; the epilogue runs in the reduce kernel
      *reinterpret_cast<f32x4*>(d.ws + (((int64_t)ks * d.B + b) * M + m) * N + nq) = y;
      continue;
    }
    const int64_t o = ((int64_t)b * M + m) * N + nq;
#pragma unroll
    for (int k = 0; k < 4; ++k) y[k] += pre_bias[rr];
    if (d.out_pre) *reinterpret_cast<f32x4*>(d.out_pre + o) = y;
#pragma unroll
    for (int k = 0; k < 4; ++k) y[k] = fmaf(y[k], pre_scale[rr], pre_res[rr][k]);
    *reinterpret_cast<f32x4*>(d.out + o) = y;
#pragma unroll
    for (int k = 0; k < 4; ++k) vfin[rr][k] = y[k];
  }

  // ---- GroupNorm partial statistics of the tile just stored: one (mean, M2, count) entry per (the RPW rows of a row quad this
  // wave finished: half a quad or all of it) x (128-position tile); the consumer Chan-combines the entries whatever their counts
  if (d.gn_part != nullptr && n0 < N && final_tile && d.store == 0) {
    const int cntv = (N - n0) < M4_BN ? (N - n0) : M4_BN;
    const float fcnt = (float)RPW * (float)cntv;
    float sv = 0.0f;
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr)
#pragma unroll
      for (int k = 0; k < 4; ++k) sv += vfin[rr][k];
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) sv += __shfl_xor(sv, o, 64);
    const float mean = sv / fcnt;
    float qv = 0.0f;
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float dv = nok ? vfin[rr][k] - mean : 0.0f;
        qv = fmaf(dv, dv, qv);
      }
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) qv += __shfl_xor(qv, o, 64);
    const int r = RPW * wave;
    const int m = m0 + 8 * (r >> 2) + 4 * hi;  // first channel of the row quad
    if (l31 == 0 && m < M) {
      constexpr int EPT = 4 / RPW;  // entries per row quad and tile
      float* e = d.gn_part + (((int64_t)b * (M / 4) + (m >> 2)) * (EPT * ntn) + EPT * nt + (EPT == 2 ? (wave & 1) : 0)) * 3;
      e[0] = mean;
      e[1] = qv;
      e[2] = fcnt;
    }
  }
  // ---- (sum ds * xhat, sum ds) of each finished row over the tile's 128 positions, ds = da * silu'(gamma * xhat + beta): what
  // gn_bwd_reduce_vec_kernel (norm.hip) computes from a pass over x and da, taken from the registers that hold da
  if (gnb && n0 < N) {
    const int cg = M / (int)d.gnb_groups;
    float gnb_mean[RPW], gnb_rstd[RPW], gnb_gm[RPW], gnb_bt[RPW];
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {  // (all requested before the first use)
      const int r = RPW * wave + rr;
      const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      const int mc = m < M ? m : M - 1;
      const float* st = d.gnb_stats + ((int64_t)b * d.gnb_groups + mc / cg) * 2;
      gnb_mean[rr] = st[0], gnb_rstd[rr] = st[1];
      gnb_gm[rr] = d.gnb_gamma[mc], gnb_bt[rr] = d.gnb_beta[mc];
    }
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
      const int r = RPW * wave + rr;
      const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      const float ga = gnb_gm[rr] * gnb_rstd[rr], be = gnb_bt[rr] - gnb_mean[rr] * ga;
      float sa = 0.0f, sb = 0.0f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float xh = (gnb_xq[rr][k] - gnb_mean[rr]) * gnb_rstd[rr];
        const float ds = vfin[rr][k] * adp_dsilu_fast(fmaf(gnb_xq[rr][k], ga, be));  // (vfin = 0 outside the tensor)
        sa = fmaf(ds, xh, sa);
        sb += ds;
      }
      sa = adp_half_sum(sa), sb = adp_half_sum(sb);  // (five DPP adds each; the half-wave sums stand in lanes 16-31 / 48-63)
      if (l31 == 16 && m < M) *reinterpret_cast<f32x2*>(d.gnb_ab + (((int64_t)b * M + m) * ntn + nt) * 2) = f32x2{sa, sb};
    }
  }
  ADP_KT(63);
  ADP_KT_DUMP(blockIdx.x);
}

int64_t m4_min_blocks() {
  const char* e = getenv("ADP_MM4_MIN_BLOCKS");  // (the tests reach this block with small problems through it)
  return e ? atoll(e) : 200;
}

}  // namespace

int64_t adp_conv_mm4_ksplit(const adp_conv_desc& d);

// ADP_CONV_WINO4 (read per call): unset / "1" = this kernel for every eligible conv; "0" = conv_mm's F(2,3) variant (A/B, tests)
bool adp_conv_mm4_eligible(const adp_conv_desc& d) {
  if (!adp_winograd_enabled()) return false;
  const char* e = getenv("ADP_CONV_WINO4");
  if (e && e[0] == '0') return false;
  if (d.KT != 3 || d.stride != 1 || d.dil != 1 || d.pad != 1 || d.R1 != d.R || d.x2) return false;
  if (d.up != 1 && !((d.up == 2 || d.up == 4) && !d.transposed && d.store == 0)) return false;   // UpsampleItem conv (forward)
  if (d.prologue != 0) return false;
  if (d.store != 0 && !(d.store == 2 && (d.sp == 2 || d.sp == 4) && d.up == 1 && !d.bias && !d.e_scale && !d.out_pre))
    return false;  // plain store, or the pooled store of the UpsampleItem convs' data gradients
  if (d.N != d.Lin * d.up || d.N % 4 != 0) return false;
  {
    const char* eu = getenv("ADP_MM4_UP");  // (A/B: "0" keeps the UpsampleItem convs and their data gradients on conv_mm)
    if (eu && eu[0] == '0' && (d.up != 1 || d.store != 0)) return false;
  }
  const char* mr = getenv("ADP_WINO4_MIN_R");
  // (from 128 channels: the materialised-activation layers; with the 12-wave block only, the short-K layers of depths 3-4 lost
  //  to conv_mm's wide-N F(2,3) blocks -- 12.22 vs 12.15 ms per step -- the light block wins them back: m4_nkg)
  if (d.R < (mr ? atoll(mr) : 64) || d.R % 32 != 0 || d.M % M4_BM != 0) return false;  // (round 6: from 64 channels)
  if ((reinterpret_cast<uintptr_t>(d.x) | reinterpret_cast<uintptr_t>(d.w) | reinterpret_cast<uintptr_t>(d.out) |
       reinterpret_cast<uintptr_t>(d.res) | reinterpret_cast<uintptr_t>(d.out_pre)) & 15)
    return false;
  if (d.B * d.R * d.Lin >= (int64_t)1 << 31 || d.M * d.R * d.KT >= (int64_t)1 << 31) return false;
  // one block per CU and more (with the K split: adp_conv_mm4_ksplit); below that conv_mm's 64-position blocks fill the chip better
  return (d.M / M4_BM) * adp_cdiv(d.N, M4_BN) * d.B * adp_conv_mm4_ksplit(d) >= m4_min_blocks();
}

// Cross-workgroup K split of the F(4,3) block: 2 / 4 slices when the tiles alone leave the chip half empty -- the [4, 1024, 128]
// layers of depth 8 are 128 tiles.  OFF by default (ADP_MM4_KS_MAX=2 / 4 switches it on): measured at batch 4 it is worth
// 12.16 -> 12.11 ms and 11.90 -> 11.875 ms per step (the reduce launch eats most of what the matrix cores save) for 32 more
// launches per step; depth 8 stays on conv_mm's 64-position F(2,3) blocks.
int64_t adp_conv_mm4_ksplit(const adp_conv_desc& d) {
  const char* e = getenv("ADP_MM4_KS_MAX");
  const int64_t ksmax = e ? atoll(e) : 1;
  const int64_t blocks = (d.M / M4_BM) * adp_cdiv(d.N, M4_BN) * d.B, nchunks = d.R / 64;
  if (d.store != 0 || d.up != 1) return 1;  // (the pooled store and the upsample loader keep their in-kernel epilogue)
  int64_t ks = 1;
  // (every slice keeps >= 8 chunks of 64 channels: at batch 1 the 512-channel layers would qualify with 4 and lose to conv_mm's
  //  64-position blocks -- batch-1 step 6.46 -> 6.53 ms; depth 8 at batch 4: step 12.16 -> 12.11 ms)
  const char* mc = getenv("ADP_MM4_KS_MINCH");
  const int64_t minch = mc ? atoll(mc) : 8;
  const char* tg = getenv("ADP_MM4_KS_TARGET");
  const int64_t target = tg ? atoll(tg) : m4_min_blocks();
  while (ks < ksmax && blocks * ks < target && d.R % 64 == 0 && nchunks / (ks * 2) >= minch) ks *= 2;
  return ks;
}

// K groups of the block: 2 = the light 8-wave block, taken when the grid has at least two blocks per CU (ADP_MM4_LIGHT_MIN_BLOCKS,
// default 400) so that two of them share a CU; else the 12-wave block.  Isolated launches at batch 4, us (tools/mm4_micro.py,
// conv_mm F(2,3) -> 12-wave F(4,3) -> light): C=512 L=1024 45.9 -> 44.9 -> 39.8; C=256 L=2048 29.2 -> 29.0 -> 25.4; C=128 L=4096
// 18.7 -> 21.3 -> 17.4; with one block per CU the light block loses (C=512 L=512 25.6 -> 22.5 -> 24.9).
// The K split the launcher will really take: only with the caller's scratch (adp_conv1d_ws_bytes sizes it from the potential
// split).  Block shape, GroupNorm entry count and the launch all read THIS value, so they cannot disagree.
static int64_t m4_ks_eff(const adp_conv_desc& d) { return d.ws ? adp_conv_mm4_ksplit(d) : 1; }

static bool m4_light(const adp_conv_desc& d) {
  const char* e = getenv("ADP_MM4_LIGHT_MIN_BLOCKS");
  const int64_t blocks = (d.M / M4_BM) * adp_cdiv(d.N, M4_BN) * d.B * m4_ks_eff(d);
  return blocks >= (e ? atoll(e) : 400);
}
static int m4_nkg(const adp_conv_desc& d) { return m4_light(d) ? 2 : 4; }

int64_t adp_conv_mm4_gn_entries(const adp_conv_desc& d) {
  if (d.store != 0) return 0;
  if (m4_ks_eff(d) > 1) return adp_conv_splitk_gn_entries(d);
  return (m4_nkg(d) == 2 ? 1 : 2) * adp_cdiv(d.N, M4_BN);
}

// slices per row of gnb_ab: one per 128-position tile (unsplit launches).  From 128 rows: the 64-channel layer ([4,64,16384]) is
// HBM-bound, the epilogue's read of x costs what the separate first stage's did (measured: conv 24.8 -> 34.8 us for 11.3 saved).
// Per-launch effect at batch 4 (eager event pairs, us): conv +0 .. +4, second stage +0.6 .. +2.3, first stage's 6.7 .. 8.5 gone;
// 48 launches less per step, step time within +-0.04 ms (a small kernel costs ~3 us inside the replayed graph).
int64_t adp_conv_mm4_gnb_entries(const adp_conv_desc& d) {
  if (d.store != 0 || m4_ks_eff(d) > 1 || !d.transposed || d.up != 1) return 0;  // (the instantiations that exist)
  if (d.M < 128 && d.B * d.M * d.N > (4 << 20)) return 0;
  if (!adp_gnb_family_on(m4_nkg(d) == 2 ? 2 : 1)) return 0;  // (the HBM-bound case above; at batch 1 the tensor is 4 MB and cached)
  return adp_cdiv(d.N, M4_BN);
}

int adp_conv_mm4(const adp_conv_desc& d, void* stream) {
  const int64_t blocks = (d.M / M4_BM) * adp_cdiv(d.N, M4_BN) * d.B;
  const int64_t ks = m4_ks_eff(d);  // (without the caller's scratch: the unsplit path, still correct)
  const dim3 grid((unsigned)blocks, (unsigned)ks);
  if (d.up != 1) {  // UpsampleItem convs: the light block for grids of two blocks per CU, else 64-channel chunks (32 when R % 64)
    const bool light = m4_nkg(d) == 2, c64 = d.R % 64 == 0;
    const dim3 block(((light ? 2 : 4) * M4_NPG + M4_NLD) * 64);
    if (d.up == 2) {
      if (light) ADP_LAUNCH((conv_mm4_kernel<false, 1, 32, 2, 2>), grid, block, stream, d);
      else if (c64) ADP_LAUNCH((conv_mm4_kernel<false, 1, 64, 4, 2>), grid, block, stream, d);
      else ADP_LAUNCH((conv_mm4_kernel<false, 1, 32, 4, 2>), grid, block, stream, d);
    } else {
      if (light) ADP_LAUNCH((conv_mm4_kernel<false, 1, 32, 2, 4>), grid, block, stream, d);
      else if (c64) ADP_LAUNCH((conv_mm4_kernel<false, 1, 64, 4, 4>), grid, block, stream, d);
      else ADP_LAUNCH((conv_mm4_kernel<false, 1, 32, 4, 4>), grid, block, stream, d);
    }
    return ADP_LAUNCH_OK();
  }
  // loaders stage the transformed inputs (VPRE) for full 128-position tiles: ADP_MM4_VPRE=0 / 1 (A/B)
  const char* ev = getenv("ADP_MM4_VPRE");
  const bool vpre = d.N % M4_BN == 0 && (ev ? ev[0] != '0' : M4_VPRE_DEFAULT);
  if (m4_nkg(d) == 2 && vpre) {
    const dim3 block((2 * M4_NPG + M4_NLD) * 64);
    if (d.transposed && d.gnb_ab) ADP_LAUNCH((conv_mm4_kernel<true, 1, 32, 2, 1, true, 2, true>), grid, block, stream, d);
    else if (d.transposed) ADP_LAUNCH((conv_mm4_kernel<true, 1, 32, 2, 1, false, 2, true>), grid, block, stream, d);
    else ADP_LAUNCH((conv_mm4_kernel<false, 1, 32, 2, 1, false, 2, true>), grid, block, stream, d);
    if (ADP_LAUNCH_OK() != ADP_OK) return ADP_ERR_LAUNCH;
    return ks > 1 ? adp_conv_splitk_reduce(d, ks, stream) : ADP_OK;
  }
  if (m4_nkg(d) == 2) {  // light block: 32-channel chunks (60 KB of LDS: two blocks per CU)
    const dim3 block((2 * M4_NPG + M4_NLD) * 64);
    if (d.transposed && d.gnb_ab) ADP_LAUNCH((conv_mm4_kernel<true, 1, 32, 2, 1, true>), grid, block, stream, d);
    else if (d.transposed) ADP_LAUNCH((conv_mm4_kernel<true, 1, 32, 2>), grid, block, stream, d);
    else ADP_LAUNCH((conv_mm4_kernel<false, 1, 32, 2>), grid, block, stream, d);
    if (ADP_LAUNCH_OK() != ADP_OK) return ADP_ERR_LAUNCH;
    return ks > 1 ? adp_conv_splitk_reduce(d, ks, stream) : ADP_OK;
  }
  const dim3 block((4 * M4_NPG + M4_NLD) * 64);
  // 64-channel chunks (24 MFMAs per wave and barrier instead of 12) unless ADP_MM4_BKT=32: step 12.15 -> 12.09 ms
  const char* e = getenv("ADP_MM4_BKT");
  const int bkt = (e ? atoi(e) : 64) == 64 && d.R % 64 == 0 ? 64 : 32;
  const char* e3 = getenv("ADP_MM4_NPG");
  if (bkt == 64 && e3 && atoi(e3) == 3) {  // 16-wave block: three plane groups (see NPG)
    const dim3 block16((4 * 3 + M4_NLD) * 64);
    if (d.transposed && d.gnb_ab) ADP_LAUNCH((conv_mm4_kernel<true, 1, 64, 4, 1, true, 3>), grid, block16, stream, d);
    else if (d.transposed) ADP_LAUNCH((conv_mm4_kernel<true, 1, 64, 4, 1, false, 3>), grid, block16, stream, d);
    else ADP_LAUNCH((conv_mm4_kernel<false, 1, 64, 4, 1, false, 3>), grid, block16, stream, d);
  } else if (bkt == 64 && vpre) {
    if (d.transposed && d.gnb_ab) ADP_LAUNCH((conv_mm4_kernel<true, 1, 64, 4, 1, true, 2, true>), grid, block, stream, d);
    else if (d.transposed) ADP_LAUNCH((conv_mm4_kernel<true, 1, 64, 4, 1, false, 2, true>), grid, block, stream, d);
    else ADP_LAUNCH((conv_mm4_kernel<false, 1, 64, 4, 1, false, 2, true>), grid, block, stream, d);
  } else if (bkt == 64) {  // (one register stage: a second one with 64-channel chunks spills)
    if (d.transposed && d.gnb_ab) ADP_LAUNCH((conv_mm4_kernel<true, 1, 64, 4, 1, true>), grid, block, stream, d);
    else if (d.transposed) ADP_LAUNCH((conv_mm4_kernel<true, 1, 64, 4>), grid, block, stream, d);
    else ADP_LAUNCH((conv_mm4_kernel<false, 1, 64, 4>), grid, block, stream, d);
  } else {
    const bool pd2 = d.R / 32 / ks >= 4;
    if (d.transposed && d.gnb_ab) {
      if (pd2) ADP_LAUNCH((conv_mm4_kernel<true, 2, 32, 4, 1, true>), grid, block, stream, d);
      else ADP_LAUNCH((conv_mm4_kernel<true, 1, 32, 4, 1, true>), grid, block, stream, d);
    } else if (d.transposed) {
      if (pd2) ADP_LAUNCH((conv_mm4_kernel<true, 2, 32, 4>), grid, block, stream, d);
      else ADP_LAUNCH((conv_mm4_kernel<true, 1, 32, 4>), grid, block, stream, d);
    } else {
      if (pd2) ADP_LAUNCH((conv_mm4_kernel<false, 2, 32, 4>), grid, block, stream, d);
      else ADP_LAUNCH((conv_mm4_kernel<false, 1, 32, 4>), grid, block, stream, d);
    }
  }
  if (ADP_LAUNCH_OK() != ADP_OK) return ADP_ERR_LAUNCH;
  if (ks > 1) return adp_conv_splitk_reduce(d, ks, stream);
  return ADP_OK;
}
