// Flash-style multi-head attention core for gfx950 on the exact-f32 matrix cores (v_mfma_f32_32x32x2_f32).
// Stands in for a_unet's AttentionBase (einsum QK^T -> softmax -> einsum PV) reached from
// AttentionItem / CrossAttentionItem at /root/reference/audio_diffusion_pytorch/components.py:92-93.
//
// Operands are CHANNEL-MAJOR, exactly what the 1x1-conv projections produce: q [B, H*D, n], k/v [B, H*D, m]
// (position fastest).  That makes every MFMA fragment a unit-stride read along positions, and lets the score
// tile be produced TRANSPOSED (S^T = K^T Q, key index down the accumulator registers, query index across
// lanes): the softmax reduction over keys is then 16 in-register values + one cross-half shuffle, and the
// probability tile already sits in the B-operand layout of the P*V product -- no LDS round trip, the [n, m]
// score matrix is never materialised (the reference writes B*H*n*m floats: 33.5 MB per item at n = 1024).
//
// Structure (round 2): every WAVE is independent -- it owns 32 queries (forward, dq pass) or 32 keys (dk/dv pass)
// and reads its MFMA operands STRAIGHT from global memory in the two fragment shapes the matrix cores want:
//   "column" fragments  x[d = 2s + hi][p0 + l31]        32 coalesced 128-byte half-wave rows per 32-position tile
//   "row"    fragments  x[d = 32t + l31][p0 + 8k + 4hi .. +3]   four 16-byte pieces per lane, whole 128-B lines used
// K/V (or Q/dO) of one head are 128-512 KB and are re-read by every wave of that head out of L2 -- staging them in
// LDS bought nothing but two workgroup barriers per 32-key tile with all global latency exposed (round 1: 8 TF at
// n = 1024).  Without LDS and barriers the waves free-run, several per SIMD, and hide each other's L2 latency.
// The dk/dv pass splits the QUERY range over several waves when there are few key tiles (cross attention: m = 64
// keys x n = 4096 queries) and sums the partial tiles in a second, deterministic stage.  D <= 64, multiple of 2.
#include <stdlib.h>
#include "adp_rt.h"
#include "adp.h"

namespace {

constexpr int DMAX = 64;

__device__ __forceinline__ int acc_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// Addressing: every tensor of one (batch, head) is a [D, len] slab (len = n or m) behind a wave-uniform base pointer;
// lanes use 32-bit element offsets into it (the host checks D * len < 2^31).  Out-of-range positions / channels are
// read at a CLAMPED in-range address and replaced by zero with a select -- no divergent branches in the loops.

// "column" fragment element x[d][p]: col = hi * len + min(p, len - 1) precomputed per tile, pok = p < len
template <bool D64>
__device__ __forceinline__ float ld_col(const float* x, int len, int s, int hi, int D, int col, bool pok) {
  // the load is UNCONDITIONAL (the address is always in range) and the guard is a select on its result: a load
  // inside the conditional arm makes hipcc branch around every single load and wait for each one separately
  if (D64) {
    const float v = x[2 * s * len + col];
    return pok ? v : 0.0f;
  }
  const int d = 2 * s + hi;
  const float v = x[(d < D ? 2 * s : 0) * len + col];
  return (pok && d < D) ? v : 0.0f;
}

// all D/2 column fragments of one 32-position tile.  Interior tiles (`full`) take plain loads with NO select on the
// result: a select right behind a prefetch load would make the wave wait for the load it was meant to overlap.
template <bool D64>
__device__ __forceinline__ void ld_cols(const float* x, int len, int hi, int l31, int D, int p0, bool full,
                                        float (&out)[DMAX / 2]) {
  if (D64 && full) {
    const float* xp = x + hi * len + p0 + l31;
#pragma unroll
    for (int s = 0; s < DMAX / 2; ++s) out[s] = xp[2 * s * len];
  } else {
    const bool pok = p0 + l31 < len;
    const int col = hi * len + (pok ? p0 + l31 : len - 1);
#pragma unroll
    for (int s = 0; s < DMAX / 2; ++s) out[s] = (D64 || 2 * s < D) ? ld_col<D64>(x, len, s, hi, D, col, pok) : 0.0f;
  }
}

// "row" fragment of one lane: the 16 values x[d][p0 + acc_row(s, hi)], s = 0..15, as four 4-float pieces (piece k =
// positions p0 + 8k + 4hi .. +3).  `full`: the whole 32-position tile is in range and rows are 16-byte tileable.
template <bool D64>
__device__ __forceinline__ void ld_row16(const float* x, int len, int d, int D, int p0, int hi, bool full,
                                         float (&out)[16]) {
  const bool dok = D64 || d < D;
  const float* row = x + (dok ? d : 0) * len;
  if (full) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(row + p0 + 8 * k + 4 * hi);
#pragma unroll
      for (int e = 0; e < 4; ++e) out[4 * k + e] = (D64 || dok) ? v[e] : 0.0f;
    }
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int p = p0 + 8 * k + 4 * hi + e;
        const float v = row[p < len ? p : len - 1];
        out[4 * k + e] = (dok && p < len) ? v : 0.0f;
      }
  }
}

// the 16 per-position scalars a lane needs for its accumulator rows: x[p0 + acc_row(r, hi)], r = 0..15
__device__ __forceinline__ void ld_vec16(const float* x, int p0, int hi, int len, bool full, float (&out)[16]) {
  if (full) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(x + p0 + 8 * k + 4 * hi);
#pragma unroll
      for (int e = 0; e < 4; ++e) out[4 * k + e] = v[e];
    }
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int p = p0 + acc_row(r, hi);
      const float v = x[p < len ? p : len - 1];
      out[r] = p < len ? v : 0.0f;
    }
  }
}

__device__ __forceinline__ bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// ---------------------------------------------------------------------------------------------------
// forward: o[d, i] = sum_j softmax_j(q_i . k_j * scale) v[d, j];  lse[i] = log sum_j exp(s_ij)
// one wave = 32 queries of one (batch, head); 4 independent waves per workgroup
// ---------------------------------------------------------------------------------------------------
template <bool D64>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const float* q, const float* k, const float* v, int H, int D,
                                                       int n, int m, int64_t qbs, int64_t kvbs, float scale,
                                                       float* o, float* lse, int nsplit, int tps, float* part) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hi = lane >> 5, l31 = lane & 31;
  const int64_t b = blockIdx.z, h = blockIdx.y;
  const int qblocks = (n + 127) / 128;
  const int sp = blockIdx.x / qblocks;  // key split: this wave covers key tiles [sp * tps, (sp + 1) * tps)
  const int i0 = (blockIdx.x - sp * qblocks) * 128 + wave * 32;
  if (i0 >= n) return;  // no workgroup-wide synchronisation anywhere below
  const int j_lo = sp * tps * 32, j_hi = (j_lo + tps * 32 < m) ? j_lo + tps * 32 : m;
  const float* qh = q + b * qbs + h * (int64_t)D * n;
  const float* kh = k + b * kvbs + h * (int64_t)D * m;
  const float* vh = v + b * kvbs + h * (int64_t)D * m;
  const int iq = i0 + l31;
  const bool qok = iq < n;
  const bool vec = ((m & 3) == 0) && aligned16(vh);

  // Q fragments for this wave's 32 queries: B operand of S^T = K^T Q, lane (i = l31, kk = hi) -> q[d][i] * scale
  float qf[DMAX / 2];
  {
    const int qcol = hi * n + (qok ? iq : n - 1);
#pragma unroll
    for (int s = 0; s < DMAX / 2; ++s) qf[s] = (D64 || 2 * s < D) ? ld_col<D64>(qh, n, s, hi, D, qcol, qok) * scale : 0.0f;
  }
  f32x16 oacc[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[t][r] = 0.0f;
  float mrun = -3.0e38f, lrun = 0.0f;

  // K column fragments of a tile (A operand of S^T): one 4-byte load per MFMA.  They are requested ONE TILE AHEAD
  // into the other register set, so a tile's 64 MFMAs cover the L2 latency of the next tile's loads.
  auto load_kc = [&](float (&kc)[DMAX / 2], int j0) { ld_cols<D64>(kh, m, hi, l31, D, j0, j0 + 32 <= m, kc); };
  auto tile = [&](const float (&kc)[DMAX / 2], float (&kn)[DMAX / 2], int j0) {
    const bool full = j0 + 32 <= m;
    // V row fragments of THIS tile: needed only after the 32 S MFMAs and the softmax
    float vr[2][16];
#pragma unroll
    for (int t = 0; t < 2; ++t)
      if (D64 || 32 * t < D) ld_row16<D64>(vh, m, 32 * t + l31, D, j0, hi, full && vec, vr[t]);
    if (j0 + 32 < j_hi) load_kc(kn, j0 + 32);
#ifndef ADP_EMULATE
    __builtin_amdgcn_sched_barrier(0);  // keep the loads ahead of the matrix work
#endif
    // S^T tile: rows j (regs), cols i (lanes): A[i'=j][kk=d] = k[d][j], B[kk=d][j'=i] = q[d][i]
    f32x16 st;
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = 0.0f;
#pragma unroll
    for (int s = 0; s < DMAX / 2; ++s)
      if (D64 || 2 * s < D) st = adp_mfma32(kc[s], qf[s], st);
    // online softmax over j: 16 in-register rows + the other half-wave
    float tmax = -3.0e38f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (!full && j0 + acc_row(r, hi) >= m) st[r] = -3.0e38f;
      tmax = fmaxf(tmax, st[r]);
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float mnew = fmaxf(mrun, tmax);
    const float alpha = __expf(mrun - mnew);
    float psum = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = (full || j0 + acc_row(r, hi) < m) ? __expf(st[r] - mnew) : 0.0f;
      st[r] = p;
      psum += p;
    }
    psum += __shfl_xor(psum, 32, 64);
    lrun = lrun * alpha + psum;
    mrun = mnew;
    // O^T[d][i] = alpha * O^T + sum_j v[d][j] P^T[j][i]: A[i'=d][kk=j] = v[d][j(s,hi)], B = own register s
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (D64 || 32 * t < D) {
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[t][r] *= alpha;
#pragma unroll
        for (int s = 0; s < 16; ++s) oacc[t] = adp_mfma32(vr[t][s], st[s], oacc[t]);
      }
    }
  };
  float ka[DMAX / 2], kb[DMAX / 2];
  load_kc(ka, j_lo);
  for (int j0 = j_lo; j0 < j_hi; j0 += 64) {  // two tiles per trip: the register sets swap roles statically
    tile(ka, kb, j0);
    if (j0 + 32 < j_hi) tile(kb, ka, j0 + 32);
  }
  if (nsplit > 1) {
    // partial result of this key range: un-normalised O^T, running maximum and sum; attn_fwd_combine_kernel merges the
    // nsplit partials in split order.  part = [nsplit][B*H][D][n] | m [nsplit][B*H][n] | l [nsplit][B*H][n]
    const int64_t BH = (int64_t)gridDim.z * H, bh = b * H + h;
    float* po = part + ((int64_t)sp * BH + bh) * D * n;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int dd = 32 * t + acc_row(r, hi);
        if (dd < D && qok) po[(int64_t)dd * n + iq] = oacc[t][r];
      }
    if (hi == 0 && qok) {
      float* pm = part + (int64_t)nsplit * BH * D * n + ((int64_t)sp * BH + bh) * n;
      pm[iq] = mrun;
      pm[(int64_t)nsplit * BH * n + iq] = lrun;
    }
    return;
  }
  const float inv = (lrun > 0.0f) ? 1.0f / lrun : 0.0f;
  float* oh = o + (b * H + h) * (int64_t)D * n;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int dd = 32 * t + acc_row(r, hi);
      if (dd < D && qok) oh[dd * n + iq] = oacc[t][r] * inv;
    }
  if (hi == 0 && qok) lse[(b * H + h) * (int64_t)n + iq] = mrun + logf(lrun);
}

// ---------------------------------------------------------------------------------------------------
// forward for FEW KEYS (m <= 64: cross attention over the 64-token embedding), D = 64 (round 6).  In the kernel above one wave owns
// a 32-query tile and walks a serial chain of 128 MFMAs behind ~100 scalar loads whatever n is: 10-14 us per call with the chip idle
// (batch x heads x n / 32 = 32 ... 1024 waves).  Here the FOUR waves of a workgroup share ONE 32-query tile: wave = (key block
// kb, channel half dh).  S^T of key block kb is contracted over the 32 channels of half dh (16 MFMAs) and the two halves meet in
// LDS; the softmax runs per wave on its 32 keys with the row maximum / sum exchanged through LDS; O^T rows 32 dh .. + 31 are
// accumulated over the keys of block kb (16 MFMAs) and the two key blocks meet in LDS: 32 MFMAs and a quarter of the loads per
// wave, four workgroup barriers.  Sums of two partials only: the order cannot matter (a + b = b + a), results are deterministic.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_fwd_fewkeys_kernel(const float* q, const float* k, const float* v, int H, int n, int m,
                                                               int64_t qbs, int64_t kvbs, float scale, float* o, float* lse) {
  constexpr int D = 64;
  __shared__ float xch[4][16 * 64];  // one accumulator tile per wave
  __shared__ float red[2][2][32];    // [max | sum][key block][query]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hi = lane >> 5, l31 = lane & 31;
  const int kb = wave >> 1, dh = wave & 1;
  const int64_t b = blockIdx.z, h = blockIdx.y;
  const int i0 = blockIdx.x * 32, iq = i0 + l31;
  const bool qok = iq < n;
  const float* qh = q + b * qbs + h * (int64_t)D * n;
  const float* kh = k + b * kvbs + h * (int64_t)D * m;
  const float* vh = v + b * kvbs + h * (int64_t)D * m;
  const int j0 = 32 * kb;
  const bool kfull = j0 + 32 <= m, kany = j0 < m;
  // ---- S^T partial over channels 32 dh .. 32 dh + 31: A = k[d][j0 + l31], B = q[d][i] * scale (d = 32 dh + 2 s + hi)
  float kc[16], qf[16];
  {
    const int qcol = (32 * dh + hi) * n + (qok ? iq : n - 1);
    const bool jok = j0 + l31 < m;
    const int kcol = (32 * dh + hi) * m + (jok ? j0 + l31 : m - 1);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const float qv = qh[2 * s * n + qcol], kv = kh[2 * s * m + kcol];
      qf[s] = qok ? qv * scale : 0.0f;
      kc[s] = jok ? kv : 0.0f;
    }
  }
  // V rows of this wave's output half and key block, requested now (needed after the softmax)
  float vr[16];
  ld_row16<true>(vh, m, 32 * dh + l31, D, kany ? j0 : 0, hi, kfull && ((m & 3) == 0) && aligned16(vh), vr);
  f32x16 st;
#pragma unroll
  for (int r = 0; r < 16; ++r) st[r] = 0.0f;
#pragma unroll
  for (int s = 0; s < 16; ++s) st = adp_mfma32(kc[s], qf[s], st);
#pragma unroll
  for (int r = 0; r < 16; ++r) xch[wave][r * 64 + lane] = st[r];
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 16; ++r) st[r] = xch[2 * kb][r * 64 + lane] + xch[2 * kb + 1][r * 64 + lane];  // (channel half 0 + half 1)
  // ---- softmax over the 64 keys: this wave's 32 + the other key block's maximum / sum through LDS
  float tmax = -3.0e38f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    if (!kfull && j0 + acc_row(r, hi) >= m) st[r] = -3.0e38f;
    tmax = fmaxf(tmax, st[r]);
  }
  tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
  if (dh == 0 && hi == 0) red[0][kb][l31] = tmax;
  __syncthreads();
  const float mrow = fmaxf(red[0][0][l31], red[0][1][l31]);
  float psum = 0.0f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float p = (kfull || j0 + acc_row(r, hi) < m) ? __expf(st[r] - mrow) : 0.0f;
    st[r] = p;
    psum += p;
  }
  psum += __shfl_xor(psum, 32, 64);
  if (dh == 0 && hi == 0) red[1][kb][l31] = psum;
  // ---- O^T rows 32 dh + l31 over the keys of block kb: A = v[d][j(s, hi)], B = P^T (own register s)
  f32x16 oacc;
#pragma unroll
  for (int r = 0; r < 16; ++r) oacc[r] = 0.0f;
#pragma unroll
  for (int s = 0; s < 16; ++s) oacc = adp_mfma32(kany ? vr[s] : 0.0f, st[s], oacc);
  __syncthreads();  // (the S tiles have been read by everybody; the sums are there)
  if (kb == 1) {
#pragma unroll
    for (int r = 0; r < 16; ++r) xch[wave][r * 64 + lane] = oacc[r];
  }
  __syncthreads();
  if (kb == 0) {
    const float lrow = red[1][0][l31] + red[1][1][l31];
    const float inv = lrow > 0.0f ? 1.0f / lrow : 0.0f;
    float* oh = o + (b * H + h) * (int64_t)D * n;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int dd = 32 * dh + acc_row(r, hi);
      if (qok) oh[dd * n + iq] = (oacc[r] + xch[2 + dh][r * 64 + lane]) * inv;  // (key block 0 + block 1)
    }
    if (dh == 0 && hi == 0 && qok) lse[(b * H + h) * (int64_t)n + iq] = mrow + logf(lrow);
  }
}

// merge of the forward key split (fixed order): M = max_s m_s, w_s = exp(m_s - M), L = sum_s w_s l_s,
// o[d][i] = sum_s w_s O_s[d][i] / L, lse = M + log L.  One workgroup = 64 queries x 4 channels (blockIdx.z picks the
// channel quad): every thread owns one output element, so the launch is wide instead of long.
__global__ __launch_bounds__(256) void attn_fwd_combine_kernel(const float* part, int nsplit, int D, int64_t n, int64_t BH,
                                                               float* o, float* lse) {
  const int ql = threadIdx.x & 63, d = blockIdx.z * 4 + (threadIdx.x >> 6);
  const int64_t bh = blockIdx.y, i = (int64_t)blockIdx.x * 64 + ql;
  if (i >= n || d >= D) return;
  const float* pm = part + (int64_t)nsplit * BH * D * n + bh * n + i;
  const int64_t sstride = BH * n;  // between splits of m / l
  float mv[8], pv[8];
  float M = -3.0e38f;
  for (int s = 0; s < nsplit; ++s) {
    mv[s] = pm[s * sstride];
    pv[s] = part[(((int64_t)s * BH + bh) * D + d) * n + i];
    M = fmaxf(M, mv[s]);
  }
  float L = 0.0f, a = 0.0f;
  for (int s = 0; s < nsplit; ++s) {
    const float w = __expf(mv[s] - M);
    L += w * pm[(int64_t)nsplit * sstride + s * sstride];
    a += w * pv[s];
  }
  o[(bh * D + d) * n + i] = L > 0.0f ? a / L : 0.0f;
  if (d == 0) lse[bh * n + i] = M + logf(L);
}

// out[b][e] = sum_{sp < nsplit} part[sp * pstride + b * bstride + e], e < cnt (fixed order); blockIdx.y = b
__global__ __launch_bounds__(256) void attn_sum_splits_kernel(const float* part, int nsplit, int64_t pstride, int64_t bstride,
                                                              int64_t cnt, float* out) {
  const float* p = part + blockIdx.y * bstride;
  float* o = out + blockIdx.y * bstride;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < cnt; e += (int64_t)gridDim.x * 256) {
    float v[8];  // nsplit <= 8: all partial loads in flight together, summed in split order
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = p[(u < nsplit ? u : 0) * pstride + e];  // (clamped address, then a select)
    float a = 0.0f;
#pragma unroll
    for (int u = 0; u < 8; ++u) a += u < nsplit ? v[u] : 0.0f;
    o[e] = a;
  }
}

// ---------------------------------------------------------------------------------------------------
// backward, key-major pass: one wave owns 32 keys and a slice of the query tiles:
//   S[i][j] (rows i in regs, cols j across lanes), P = exp(S - lse_i), dP = dO^T V, dS = P (dP - delta_i) scale
//   dv[d][j] += sum_i dO[d][i] P[i][j] ; dk[d][j] += sum_i q[d][i] dS[i][j]
// grid.x = ceil(key tiles x nsplit / 4); split sp covers query tiles [sp * tps, (sp + 1) * tps).  With nsplit > 1 the
// partial [D, 32] tiles go to part[sp] (addressed like dk / dv) and attn_kv_reduce_kernel sums them in split order.
// ---------------------------------------------------------------------------------------------------
// OWN_DELTA (the merged launch, attn_bwd_merged_kernel): delta_i = sum_d dO[d][i] O[d][i] is formed HERE from the dO columns the
// wave holds anyway and 32 more loads of O per tile, and enters the dP tile as one more MFMA step (A = -delta_i in the kk = 0
// slot, B = 1): the key-major pass then needs nothing from the query-major pass and both run in ONE launch.
template <bool D64, bool OWN_DELTA>
__device__ __forceinline__ void attn_bwd_kv_body(const float* q, const float* k, const float* v, const float* o,
                                                 const float* dout, const float* lse, const float* delta, int H, int D,
                                                 int n, int m, int64_t qbs, int64_t kvbs, float scale, int nsplit, int tps,
                                                 int64_t pstride, float* dk, float* dv, int bx) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hi = lane >> 5, l31 = lane & 31;
  const int64_t b = blockIdx.z, h = blockIdx.y;
  const int ktiles = (m + 31) / 32;
  const int wid = bx * 4 + wave;  // wave id within (b, h): key tile fastest
  const int kt = wid % ktiles, sp = wid / ktiles;
  if (sp >= nsplit) return;
  const int j0 = kt * 32;
  const float* qh = q + b * qbs + h * (int64_t)D * n;
  const float* kh = k + b * kvbs + h * (int64_t)D * m;
  const float* vh = v + b * kvbs + h * (int64_t)D * m;
  const float* doh = dout + (b * H + h) * (int64_t)D * n;
  const float* lh = lse + (b * H + h) * (int64_t)n;
  const float* dlh = OWN_DELTA ? lh : delta + (b * H + h) * (int64_t)n;
  const float* oh = o + (b * H + h) * (int64_t)D * n;
  const int jk = j0 + l31;
  const bool kok = jk < m;
  const bool vec = ((n & 3) == 0) && aligned16(qh) && aligned16(doh) && aligned16(lh) && aligned16(dlh);
  // K and V fragments of this wave's 32 keys: B operands (lane (j = l31, kk = hi) -> x[d][j])
  float kf[DMAX / 2], vf[DMAX / 2];
  {
    const int kcol = hi * m + (kok ? jk : m - 1);
#pragma unroll
    for (int s = 0; s < DMAX / 2; ++s) {
      kf[s] = (D64 || 2 * s < D) ? ld_col<D64>(kh, m, s, hi, D, kcol, kok) : 0.0f;
      vf[s] = (D64 || 2 * s < D) ? ld_col<D64>(vh, m, s, hi, D, kcol, kok) : 0.0f;
    }
  }
  f32x16 dka[2], dva[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) dka[t][r] = dva[t][r] = 0.0f;

  const int i_beg = sp * tps * 32;
  const int i_end = (i_beg + tps * 32 < n) ? i_beg + tps * 32 : n;
  // Register plan per 32-query tile (everything else would spill: K/V fragments 64 + accumulators 64 are resident):
  //   top    : dO columns and t = 0 row fragments + lse / delta of THIS tile, Q columns of the NEXT tile
  //   S      : 32 MFMAs on the Q columns requested one tile ago (cover the latency of the loads above)
  //   dP     : 32 MFMAs on the dO columns; then the t = 1 row fragments are requested (covered by the t = 0 products)
  auto load_q = [&](float (&qc)[DMAX / 2], int i0) { ld_cols<D64>(qh, n, hi, l31, D, i0, i0 + 32 <= n, qc); };
  auto tile = [&](const float (&qc)[DMAX / 2], float (&qn)[DMAX / 2], int i0) {
    const bool full = i0 + 32 <= n;
    float dc[DMAX / 2], dor[16], qr[16], ls[16], dl[16];
    ld_cols<D64>(doh, n, hi, l31, D, i0, full, dc);
    ld_row16<D64>(doh, n, l31, D, i0, hi, full && vec, dor);
    ld_row16<D64>(qh, n, l31, D, i0, hi, full && vec, qr);
    ld_vec16(lh, i0, hi, n, full && vec, ls);
    float di = 0.0f;
    if (OWN_DELTA) {
      float oc[DMAX / 2];
      ld_cols<D64>(oh, n, hi, l31, D, i0, full, oc);
#pragma unroll
      for (int s = 0; s < DMAX / 2; ++s)
        if (D64 || 2 * s < D) di = fmaf(dc[s], oc[s], di);
      di += __shfl_xor(di, 32, 64);  // query i0 + l31, both half-waves
#pragma unroll
      for (int r = 0; r < 16; ++r) dl[r] = 0.0f;
    } else {
      ld_vec16(dlh, i0, hi, n, full && vec, dl);
    }
    if (i0 + 32 < i_end) load_q(qn, i0 + 32);
#ifndef ADP_EMULATE
    __builtin_amdgcn_sched_barrier(0);
#endif
    // S tile: A[i'=i][kk=d] = q[d][i], B = kf ; dP tile: A = dO[d][i], B = vf
    f32x16 sa, dpa;
#pragma unroll
    for (int r = 0; r < 16; ++r) sa[r] = dpa[r] = 0.0f;
#pragma unroll
    for (int s = 0; s < DMAX / 2; ++s)
      if (D64 || 2 * s < D) sa = adp_mfma32(qc[s], kf[s], sa);
#pragma unroll
    for (int s = 0; s < DMAX / 2; ++s)
      if (D64 || 2 * s < D) dpa = adp_mfma32(dc[s], vf[s], dpa);
    if (OWN_DELTA) dpa = adp_mfma32(hi == 0 ? -di : 0.0f, 1.0f, dpa);  // dP[i][j] - delta_i for every key column j
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const bool ok = (full || i0 + acc_row(r, hi) < n) && kok;
      const float p = ok ? __expf(sa[r] * scale - ls[r]) : 0.0f;
      sa[r] = p;                                  // P[i][j]
      dpa[r] = p * (dpa[r] - dl[r]) * scale;      // dS[i][j]
    }
#ifndef ADP_EMULATE
    __builtin_amdgcn_sched_barrier(0);
#endif
    float dor1[16], qr1[16];
    if (D64 || 32 < D) {
      ld_row16<D64>(doh, n, 32 + l31, D, i0, hi, full && vec, dor1);
      ld_row16<D64>(qh, n, 32 + l31, D, i0, hi, full && vec, qr1);
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      dva[0] = adp_mfma32(dor[s], sa[s], dva[0]);
      dka[0] = adp_mfma32(qr[s], dpa[s], dka[0]);
    }
    if (D64 || 32 < D) {
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        dva[1] = adp_mfma32(dor1[s], sa[s], dva[1]);
        dka[1] = adp_mfma32(qr1[s], dpa[s], dka[1]);
      }
    }
  };
  float qa[DMAX / 2], qb[DMAX / 2];
  if (i_beg < i_end) load_q(qa, i_beg);
  for (int i0 = i_beg; i0 < i_end; i0 += 64) {
    tile(qa, qb, i0);
    if (i0 + 32 < i_end) tile(qb, qa, i0 + 32);
  }
  float* dkh = (nsplit > 1 ? dk + sp * pstride : dk) + b * kvbs + h * (int64_t)D * m;
  float* dvh = (nsplit > 1 ? dv + sp * pstride : dv) + b * kvbs + h * (int64_t)D * m;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int dd = 32 * t + acc_row(r, hi);
      if (dd < D && kok) {
        dkh[dd * m + jk] = dka[t][r];
        dvh[dd * m + jk] = dva[t][r];
      }
    }
}

template <bool D64>
__global__ __launch_bounds__(256) void attn_bwd_kv_kernel(const float* q, const float* k, const float* v,
                                                          const float* dout, const float* lse, const float* delta,
                                                          int H, int D, int n, int m, int64_t qbs, int64_t kvbs,
                                                          float scale, int nsplit, int tps, int64_t pstride,
                                                          float* dk, float* dv) {
  attn_bwd_kv_body<D64, false>(q, k, v, dout, dout, lse, delta, H, D, n, m, qbs, kvbs, scale, nsplit, tps, pstride, dk, dv,
                               (int)blockIdx.x);
}

// dk / dv [b][e] = sum_{sp < nsplit} part_{k,v}[sp * pstride + b * kvbs + e], e < cnt   (fixed order: deterministic);
// blockIdx.y = 2 * b + (0: k, 1: v)
__global__ __launch_bounds__(256) void attn_kv_reduce_kernel(const float* pk, const float* pv, int nsplit,
                                                             int64_t pstride, int64_t kvbs, int64_t cnt, float* dk,
                                                             float* dv) {
  const int64_t b = blockIdx.y >> 1;
  const float* part = ((blockIdx.y & 1) ? pv : pk) + b * kvbs;
  float* out = ((blockIdx.y & 1) ? dv : dk) + b * kvbs;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < cnt; e += (int64_t)gridDim.x * 256) {
    // the partial copies are independent loads: eight in flight at a time (a plain loop pays each one's latency in
    // turn: 14.5 us for 32 copies of 256 KB), summed in split order
    float s = 0.0f;
    int sp = 0;
    for (; sp + 8 <= nsplit; sp += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = part[(sp + u) * pstride + e];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; sp < nsplit; ++sp) s += part[sp * pstride + e];
    out[e] = s;
  }
}

// The split sums of both backward passes in ONE launch (self attention at batch 1 has both): blockIdx.y < nbq sums the nq partial
// copies of dq (batch element blockIdx.y), the other rows are attn_kv_reduce_kernel's (2 * b + (0: k, 1: v)).  Same fixed orders.
__global__ __launch_bounds__(256) void attn_bwd_reduce_kernel(const float* pq, int nq, int64_t qstride, int64_t qbs, int64_t qcnt,
                                                              float* dq, int nbq, const float* pk, const float* pv, int ns,
                                                              int64_t pstride, int64_t kvbs, int64_t kcnt, float* dk,
                                                              float* dv) {
  if ((int)blockIdx.y < nbq) {
    const float* p = pq + blockIdx.y * qbs;
    float* o = dq + blockIdx.y * qbs;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < qcnt; e += (int64_t)gridDim.x * 256) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = p[(u < nq ? u : 0) * qstride + e];
      float a = 0.0f;
#pragma unroll
      for (int u = 0; u < 8; ++u) a += u < nq ? v[u] : 0.0f;
      o[e] = a;
    }
    return;
  }
  const int y = (int)blockIdx.y - nbq;
  const int64_t b = y >> 1;
  const float* part = ((y & 1) ? pv : pk) + b * kvbs;
  float* out = ((y & 1) ? dv : dk) + b * kvbs;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < kcnt; e += (int64_t)gridDim.x * 256) {
    float s = 0.0f;
    int sp = 0;
    for (; sp + 8 <= ns; sp += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = part[(sp + u) * pstride + e];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; sp < ns; ++sp) s += part[sp * pstride + e];
    out[e] = s;
  }
}

// ---------------------------------------------------------------------------------------------------
// backward, query-major pass: one wave owns 32 queries and loops over key tiles (transposed tiles, as forward):
//   dS^T[j][i] -> dq[d][i] = sum_j k[d][j] dS^T[j][i]
// ---------------------------------------------------------------------------------------------------
template <bool D64>
__device__ __forceinline__ void attn_bwd_q_body(const float* q, const float* k, const float* v, const float* o,
                                                const float* dout, const float* lse, float* delta, int H, int D, int n,
                                                int m, int64_t qbs, int64_t kvbs, float scale, float* dq, int tps,
                                                int64_t pstride, int bx) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hi = lane >> 5, l31 = lane & 31;
  const int64_t b = blockIdx.z, h = blockIdx.y;
  const int qblocks = (n + 127) / 128;
  const int sp = bx / qblocks;  // key split (partial dq tiles go to dq + sp * pstride, summed afterwards)
  const int i0 = (bx - sp * qblocks) * 128 + wave * 32;
  if (i0 >= n) return;
  const int j_lo = sp * tps * 32, j_hi = (j_lo + tps * 32 < m) ? j_lo + tps * 32 : m;
  const float* qh = q + b * qbs + h * (int64_t)D * n;
  const float* kh = k + b * kvbs + h * (int64_t)D * m;
  const float* vh = v + b * kvbs + h * (int64_t)D * m;
  const float* doh = dout + (b * H + h) * (int64_t)D * n;
  const int iq = i0 + l31;
  const bool qok = iq < n;
  const bool vec = ((m & 3) == 0) && aligned16(kh);
  float qf[DMAX / 2], df[DMAX / 2];
  {
    const int qcol = hi * n + (qok ? iq : n - 1);
#pragma unroll
    for (int s = 0; s < DMAX / 2; ++s) {
      qf[s] = (D64 || 2 * s < D) ? ld_col<D64>(qh, n, s, hi, D, qcol, qok) : 0.0f;
      df[s] = (D64 || 2 * s < D) ? ld_col<D64>(doh, n, s, hi, D, qcol, qok) : 0.0f;
    }
  }
  const float li = qok ? lse[(b * H + h) * (int64_t)n + iq] : 0.0f;
  // delta_i = sum_d dO[d][i] O[d][i]: the wave holds the dO columns of its 32 queries anyway (each half-wave every other
  // channel), so the O columns are 32 more loads per lane and the separate delta kernel (a launch per attention layer)
  // is gone; the first key slice publishes delta for the key-major pass, which runs after this one
  float di;
  {
    const float* oh = o + (b * H + h) * (int64_t)D * n;
    const int qcol = hi * n + (qok ? iq : n - 1);
    float part = 0.0f;
#pragma unroll
    for (int s = 0; s < DMAX / 2; ++s)
      if (D64 || 2 * s < D) part = fmaf(df[s], ld_col<D64>(oh, n, s, hi, D, qcol, qok), part);
    di = part + __shfl_xor(part, 32, 64);
    if (delta && sp == 0 && hi == 0 && qok) delta[(b * H + h) * (int64_t)n + iq] = di;
  }
  f32x16 dqa[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) dqa[t][r] = 0.0f;

  // K / V column fragments (A operands of S^T and dP^T) are requested one tile ahead into the other register set
  auto load_c = [&](float (&kc)[DMAX / 2], float (&vc)[DMAX / 2], int j0) {
    ld_cols<D64>(kh, m, hi, l31, D, j0, j0 + 32 <= m, kc);
    ld_cols<D64>(vh, m, hi, l31, D, j0, j0 + 32 <= m, vc);
  };
  auto tile = [&](const float (&kc)[DMAX / 2], const float (&vc)[DMAX / 2], float (&kn)[DMAX / 2],
                  float (&vn)[DMAX / 2], int j0) {
    const bool full = j0 + 32 <= m;
    float kr[2][16];
#pragma unroll
    for (int t = 0; t < 2; ++t)
      if (D64 || 32 * t < D) ld_row16<D64>(kh, m, 32 * t + l31, D, j0, hi, full && vec, kr[t]);
    if (j0 + 32 < j_hi) load_c(kn, vn, j0 + 32);
#ifndef ADP_EMULATE
    __builtin_amdgcn_sched_barrier(0);
#endif
    f32x16 st, dpt;
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = dpt[r] = 0.0f;
#pragma unroll
    for (int s = 0; s < DMAX / 2; ++s)
      if (D64 || 2 * s < D) {
        st = adp_mfma32(kc[s], qf[s], st);    // S^T[j][i]
        dpt = adp_mfma32(vc[s], df[s], dpt);  // dP^T[j][i] = sum_d v[d][j] dO[d][i]
      }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const bool ok = (full || j0 + acc_row(r, hi) < m) && qok;
      const float p = ok ? __expf(st[r] * scale - li) : 0.0f;
      dpt[r] = p * (dpt[r] - di) * scale;  // dS^T[j][i]
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (D64 || 32 * t < D) {
#pragma unroll
        for (int s = 0; s < 16; ++s) dqa[t] = adp_mfma32(kr[t][s], dpt[s], dqa[t]);
      }
    }
  };
  float ka[DMAX / 2], va[DMAX / 2], kb[DMAX / 2], vb[DMAX / 2];
  load_c(ka, va, j_lo);
  for (int j0 = j_lo; j0 < j_hi; j0 += 64) {
    tile(ka, va, kb, vb, j0);
    if (j0 + 32 < j_hi) tile(kb, vb, ka, va, j0 + 32);
  }
  float* dqh = dq + sp * pstride + b * qbs + h * (int64_t)D * n;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int dd = 32 * t + acc_row(r, hi);
      if (dd < D && qok) dqh[dd * n + iq] = dqa[t][r];
    }
}

template <bool D64>
__global__ __launch_bounds__(256) void attn_bwd_q_kernel(const float* q, const float* k, const float* v,
                                                         const float* o, const float* dout, const float* lse,
                                                         float* delta, int H, int D, int n, int m, int64_t qbs,
                                                         int64_t kvbs, float scale, float* dq, int tps,
                                                         int64_t pstride) {
  attn_bwd_q_body<D64>(q, k, v, o, dout, lse, delta, H, D, n, m, qbs, kvbs, scale, dq, tps, pstride, (int)blockIdx.x);
}

// Both passes of the backward in ONE launch: blocks [0, gq) run the query-major pass (dq), blocks [gq, gq + gkv) the key-major
// pass (dk / dv) with its own delta.  At batch 1 each pass alone leaves most of the chip idle (cross attention over 64 keys at
// n = 128: 8 + 16 workgroups), and a dependent launch costs its boundary plus a cold first load.
struct attn_bwd_args {
  const float *q, *k, *v, *o, *dout, *lse;
  int H, D, n, m;
  int64_t qbs, kvbs;
  float scale;
  float* dq;       // query-major pass: destination (or its partial copies), key tiles per split, stride between the copies
  int qtps;
  int64_t qstride;
  float *dk, *dv;  // key-major pass: destination (or partial copies), query split, query tiles per split, stride
  int ns, tps;
  int64_t pstride;
  int gq;
};
template <bool D64>
__global__ __launch_bounds__(256) void attn_bwd_merged_kernel(attn_bwd_args a) {
  if ((int)blockIdx.x < a.gq)
    attn_bwd_q_body<D64>(a.q, a.k, a.v, a.o, a.dout, a.lse, nullptr, a.H, a.D, a.n, a.m, a.qbs, a.kvbs, a.scale, a.dq, a.qtps,
                         a.qstride, (int)blockIdx.x);
  else
    attn_bwd_kv_body<D64, true>(a.q, a.k, a.v, a.o, a.dout, a.lse, nullptr, a.H, a.D, a.n, a.m, a.qbs, a.kvbs, a.scale, a.ns,
                                a.tps, a.pstride, a.dk, a.dv, (int)blockIdx.x - a.gq);
}

// ---------------------------------------------------------------------------------------------------
// backward for FEW KEYS (m <= 64), D = 64 (round 6): the two passes above as ONE launch whose workgroups -- like the few-keys
// forward -- put FOUR waves (key block kb, channel half dh) on what one wave did alone.
//   blocks [0, gq): query-major, one workgroup per 32-query tile.  S^T and dP^T of key block kb are contracted over channel half
//     dh (16 + 16 MFMAs), the halves meet in LDS (with the halves of delta_i), dS^T is formed by both waves of a pair, dq rows
//     32 dh .. + 31 are accumulated over the keys of block kb (16 MFMAs) and the two key blocks meet in LDS.
//   blocks [gq, gq + ns): key-major, one workgroup per slice of query tiles.  Per tile S and dP - delta (the -delta_i column as
//     one more MFMA step per half) are contracted over channel half dh and meet in LDS; wave (kb, dh) then owns the output tiles
//     dv / dk [32 dh .. + 31][keys of block kb] outright (16 + 16 MFMAs per tile): nothing to exchange at the end.
// 48 / 65 MFMAs per wave and tile instead of 192 / 129, a quarter / half of the scalar loads.  Two-term sums only (a + b).
// ---------------------------------------------------------------------------------------------------
struct attn_fk_args {
  const float *q, *k, *v, *o, *dout, *lse;
  int H, n, m;
  int64_t qbs, kvbs;
  float scale;
  float* dq;
  float *dk, *dv;  // destination or its ns partial copies (stride pstride)
  int ns, tps;
  int64_t pstride;
  int gq;
};
__global__ __launch_bounds__(256) void attn_bwd_fewkeys_kernel(attn_fk_args a) {
  constexpr int D = 64;
  __shared__ float xch[4][2][16 * 64];  // two accumulator tiles per wave
  __shared__ float dred[2][32];         // halves of delta_i (query-major role)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hi = lane >> 5, l31 = lane & 31;
  const int kb = wave >> 1, dh = wave & 1;
  const int64_t b = blockIdx.z, h = blockIdx.y;
  const int n = a.n, m = a.m, H = a.H;
  const float scale = a.scale;
  const float* qh = a.q + b * a.qbs + h * (int64_t)D * n;
  const float* kh = a.k + b * a.kvbs + h * (int64_t)D * m;
  const float* vh = a.v + b * a.kvbs + h * (int64_t)D * m;
  const float* doh = a.dout + (b * H + h) * (int64_t)D * n;
  const float* oh = a.o + (b * H + h) * (int64_t)D * n;
  const float* lh = a.lse + (b * H + h) * (int64_t)n;
  const int j0 = 32 * kb;
  const bool kfull = j0 + 32 <= m, kany = j0 < m;
  const int jk = j0 + l31;
  const bool jok = jk < m;
  if ((int)blockIdx.x < a.gq) {
    // =========================== query-major: dq of one 32-query tile ===========================
    const int i0 = blockIdx.x * 32, iq = i0 + l31;
    const bool qok = iq < n;
    float qf[16], df[16], kc[16], vc[16];
    float dpart = 0.0f;
    {
      const int qcol = (32 * dh + hi) * n + (qok ? iq : n - 1);
      const int kcol = (32 * dh + hi) * m + (jok ? jk : m - 1);
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const float qv = qh[2 * s * n + qcol], dv_ = doh[2 * s * n + qcol], ov = oh[2 * s * n + qcol];
        const float kv = kh[2 * s * m + kcol], vv = vh[2 * s * m + kcol];
        qf[s] = qok ? qv : 0.0f;
        df[s] = qok ? dv_ : 0.0f;
        dpart = fmaf(df[s], qok ? ov : 0.0f, dpart);
        kc[s] = jok ? kv : 0.0f;
        vc[s] = jok ? vv : 0.0f;
      }
    }
    dpart += __shfl_xor(dpart, 32, 64);  // this channel half's part of delta_i, query i0 + l31
    float kr[16];                        // K rows of this wave's dq half and key block (needed last)
    ld_row16<true>(kh, m, 32 * dh + l31, D, kany ? j0 : 0, hi, kfull && ((m & 3) == 0) && aligned16(kh), kr);
    const float li = qok ? lh[iq] : 0.0f;
    f32x16 st, dpt;
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = dpt[r] = 0.0f;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      st = adp_mfma32(kc[s], qf[s], st);    // S^T[j][i], this half of the channels
      dpt = adp_mfma32(vc[s], df[s], dpt);  // dP^T[j][i]
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      xch[wave][0][r * 64 + lane] = st[r];
      xch[wave][1][r * 64 + lane] = dpt[r];
    }
    if (kb == 0 && hi == 0) dred[dh][l31] = dpart;
    __syncthreads();
    const float di = dred[0][l31] + dred[1][l31];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float sv = xch[2 * kb][0][r * 64 + lane] + xch[2 * kb + 1][0][r * 64 + lane];
      const float dv_ = xch[2 * kb][1][r * 64 + lane] + xch[2 * kb + 1][1][r * 64 + lane];
      const bool ok = (kfull || j0 + acc_row(r, hi) < m) && qok;
      const float pr = ok ? __expf(sv * scale - li) : 0.0f;
      dpt[r] = pr * (dv_ - di) * scale;  // dS^T[j][i]
    }
    f32x16 dqa;
#pragma unroll
    for (int r = 0; r < 16; ++r) dqa[r] = 0.0f;
#pragma unroll
    for (int s = 0; s < 16; ++s) dqa = adp_mfma32(kany ? kr[s] : 0.0f, dpt[s], dqa);  // dq[32 dh + l31'][i] over the keys of block kb
    __syncthreads();  // (everybody has read the S / dP tiles)
    if (kb == 1) {
#pragma unroll
      for (int r = 0; r < 16; ++r) xch[wave][0][r * 64 + lane] = dqa[r];
    }
    __syncthreads();
    if (kb == 0 && qok) {
      float* dqh = a.dq + b * a.qbs + h * (int64_t)D * n;
#pragma unroll
      for (int r = 0; r < 16; ++r) dqh[(32 * dh + acc_row(r, hi)) * n + iq] = dqa[r] + xch[2 + dh][0][r * 64 + lane];
    }
    return;
  }
  // =========================== key-major: dk / dv over one slice of query tiles ===========================
  const int sp = (int)blockIdx.x - a.gq;
  const int i_beg = sp * a.tps * 32;
  const int i_end = (i_beg + a.tps * 32 < n) ? i_beg + a.tps * 32 : n;
  // K and V fragments of key block kb, channel half dh: B operands (lane (j = l31, kk = hi) -> x[32 dh + 2 s + hi][j])
  float kf[16], vf[16];
  {
    const int kcol = (32 * dh + hi) * m + (jok ? jk : m - 1);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const float kv = kh[2 * s * m + kcol], vv = vh[2 * s * m + kcol];
      kf[s] = jok ? kv : 0.0f;
      vf[s] = jok ? vv : 0.0f;
    }
  }
  f32x16 dka, dva;
#pragma unroll
  for (int r = 0; r < 16; ++r) dka[r] = dva[r] = 0.0f;
  const bool vecq = ((n & 3) == 0) && aligned16(qh) && aligned16(doh) && aligned16(lh);
  for (int i0 = i_beg; i0 < i_end; i0 += 32) {
    const bool full = i0 + 32 <= n;
    const bool iok = i0 + l31 < n;
    float qc[16], dc[16];
    float dpart = 0.0f;
    {
      const int col = (32 * dh + hi) * n + (iok ? i0 + l31 : n - 1);
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const float qv = qh[2 * s * n + col], dv_ = doh[2 * s * n + col], ov = oh[2 * s * n + col];
        qc[s] = iok ? qv : 0.0f;
        dc[s] = iok ? dv_ : 0.0f;
        dpart = fmaf(dc[s], iok ? ov : 0.0f, dpart);
      }
    }
    dpart += __shfl_xor(dpart, 32, 64);  // this half's part of delta_i for query i0 + l31
    float dor[16], qr[16], ls[16];       // rows 32 dh + l31 of dO and q (A operands of dv / dk), lse of the accumulator rows
    ld_row16<true>(doh, n, 32 * dh + l31, D, i0, hi, full && vecq, dor);
    ld_row16<true>(qh, n, 32 * dh + l31, D, i0, hi, full && vecq, qr);
    ld_vec16(lh, i0, hi, n, full && vecq, ls);
    f32x16 sa, dpa;
#pragma unroll
    for (int r = 0; r < 16; ++r) sa[r] = dpa[r] = 0.0f;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      sa = adp_mfma32(qc[s], kf[s], sa);    // S[i][j], this half of the channels
      dpa = adp_mfma32(dc[s], vf[s], dpa);  // dP[i][j]
    }
    dpa = adp_mfma32(hi == 0 ? -dpart : 0.0f, 1.0f, dpa);  // - (this half of delta_i) for every key column
    if (i0 != i_beg) __syncthreads();                      // (the previous tile's exchange has been read)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      xch[wave][0][r * 64 + lane] = sa[r];
      xch[wave][1][r * 64 + lane] = dpa[r];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float sv = xch[2 * kb][0][r * 64 + lane] + xch[2 * kb + 1][0][r * 64 + lane];
      const float dv_ = xch[2 * kb][1][r * 64 + lane] + xch[2 * kb + 1][1][r * 64 + lane];
      const bool ok = (full || i0 + acc_row(r, hi) < n) && jok;
      const float pr = ok ? __expf(sv * scale - ls[r]) : 0.0f;
      sa[r] = pr;                    // P[i][j]
      dpa[r] = pr * dv_ * scale;     // dS[i][j]
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      dva = adp_mfma32(dor[s], sa[s], dva);   // dv[32 dh + l31'][j] += sum_i dO[d][i] P[i][j]
      dka = adp_mfma32(qr[s], dpa[s], dka);   // dk[32 dh + l31'][j] += sum_i q[d][i] dS[i][j]
    }
  }
  if (jok) {
    float* dkh = (a.ns > 1 ? a.dk + sp * a.pstride : a.dk) + b * a.kvbs + h * (int64_t)D * m;
    float* dvh = (a.ns > 1 ? a.dv + sp * a.pstride : a.dv) + b * a.kvbs + h * (int64_t)D * m;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int dd = 32 * dh + acc_row(r, hi);
      dkh[dd * m + jk] = dka[r];
      dvh[dd * m + jk] = dva[r];
    }
  }
}

// query split of the key-major pass: enough waves to cover the chip when there are few key tiles
int64_t kv_nsplit(int64_t B, int64_t H, int64_t n, int64_t m) {
  const int64_t ktiles = (m + 31) / 32, qtiles = (n + 31) / 32;
  int64_t ns = 1024 / (B * H * ktiles);          // target ~1024 waves (4 per CU),
  if (ns > qtiles) ns = qtiles;                  // down to one query tile per wave (a wave is serial: 8 us per tile; the
                                                 // old floor of four tiles left n = 256 cross attention on 32 waves)
  if (ns > 32) ns = 32;                          // and a bounded number of partial copies to sum
  if (ns < 1) ns = 1;
  return ns;
}

// key split of the query-major passes (forward, dq): when B * H * query tiles leaves most SIMDs idle (batch 1: 256 waves
// at n = 1024, 32 at n = 128) the key range is cut into up to 8 slices run by separate waves
int64_t q_nsplit(int64_t B, int64_t H, int64_t n, int64_t m, int64_t* tps_out) {
  const int64_t ktiles = (m + 31) / 32, qtiles = (n + 31) / 32;
  int64_t ns = 1024 / (B * H * qtiles);
  if (ns > 8) ns = 8;
  if (ns > ktiles / 2) ns = ktiles / 2;  // at least two key tiles per slice (cross attention over 64 keys: no split)
  if (ns < 1) ns = 1;
  const int64_t tps = (ktiles + ns - 1) / ns;
  *tps_out = tps;
  return (ktiles + tps - 1) / tps;  // every slice non-empty
}

bool attn_shape_ok(int64_t B, int64_t H, int64_t D, int64_t n, int64_t m) {
  return B > 0 && H > 0 && D >= 2 && D <= DMAX && (D % 2 == 0) && n > 0 && m > 0 && B <= 65535 && H <= 65535 &&
         D * n < ((int64_t)1 << 31) && D * m < ((int64_t)1 << 31);  // 32-bit element offsets inside one head's slab
}

}  // namespace

extern "C" int64_t adp_attn_fwd_ws_bytes(int64_t B, int64_t H, int64_t D, int64_t n, int64_t m) {
  if (!attn_shape_ok(B, H, D, n, m)) return ADP_ERR_SHAPE;
  int64_t tps;
  const int64_t ns = q_nsplit(B, H, n, m, &tps);
  return ns > 1 ? ns * B * H * (D * n + 2 * n) * (int64_t)sizeof(float) : 0;
}

extern "C" int adp_attn_fwd(const float* q, const float* k, const float* v, int64_t B, int64_t H, int64_t D,
                            int64_t n, int64_t m, int64_t q_bstride, int64_t kv_bstride, float* o, float* lse,
                            float* ws, void* stream) {
  if (!q || !k || !v || !o || !lse) return ADP_ERR_NULL;
  if (!attn_shape_ok(B, H, D, n, m)) return ADP_ERR_SHAPE;
  const float scale = 1.0f / sqrtf((float)D);
  int64_t tps = adp_cdiv(m, 32);
  const int64_t ns = ws ? q_nsplit(B, H, n, m, &tps) : 1;
  if (ns == 1) tps = adp_cdiv(m, 32);
  {  // few keys (cross attention over the embedding): four waves per query tile (ADP_ATTN_FEWKEYS=0: the one-wave form, A/B)
    const char* fk = getenv("ADP_ATTN_FEWKEYS");
    if (D == 64 && m <= 64 && ns == 1 && (!fk || fk[0] != '0')) {
      ADP_LAUNCH(attn_fwd_fewkeys_kernel, dim3((unsigned)adp_cdiv(n, 32), (unsigned)H, (unsigned)B), dim3(256), stream, q, k, v,
                 (int)H, (int)n, (int)m, q_bstride, kv_bstride, scale, o, lse);
      return ADP_LAUNCH_OK();
    }
  }
  const dim3 grid((unsigned)(adp_cdiv(n, 128) * ns), (unsigned)H, (unsigned)B);
  if (D == 64)
    ADP_LAUNCH(attn_fwd_kernel<true>, grid, dim3(256), stream, q, k, v, (int)H, (int)D, (int)n, (int)m, q_bstride,
               kv_bstride, scale, o, lse, (int)ns, (int)tps, ws);
  else
    ADP_LAUNCH(attn_fwd_kernel<false>, grid, dim3(256), stream, q, k, v, (int)H, (int)D, (int)n, (int)m, q_bstride,
               kv_bstride, scale, o, lse, (int)ns, (int)tps, ws);
  if (ns > 1)
    ADP_LAUNCH(attn_fwd_combine_kernel, dim3((unsigned)adp_cdiv(n, 64), (unsigned)(B * H), (unsigned)adp_cdiv(D, 4)), dim3(256), stream,
               (const float*)ws, (int)ns, (int)D, n, B * H, o, lse);
  return ADP_LAUNCH_OK();
}

extern "C" int64_t adp_attn_bwd_ws_bytes(int64_t B, int64_t H, int64_t D, int64_t n, int64_t m) {
  if (!attn_shape_ok(B, H, D, n, m)) return ADP_ERR_SHAPE;
  const int64_t ns = kv_nsplit(B, H, n, m);
  int64_t tps;
  const int64_t nq = q_nsplit(B, H, n, m, &tps);
  // delta [B, H, n] + (query-split dk/dv pass) nsplit partial copies of dk and of dv, each B * kv_bstride floats at
  // most 2*H*D*m per batch element (k and v are the two halves of one projection output) + (key-split dq pass) nq
  // partial copies of dq (packed q: q_bstride = H*D*n)
  return (B * H * n + (ns > 1 ? 2 * ns * B * 2 * H * D * m : 0) + (nq > 1 ? nq * B * H * D * n : 0)) * (int64_t)sizeof(float);
}

extern "C" int adp_attn_bwd(const float* q, const float* k, const float* v, const float* o, const float* dout,
                            const float* lse, int64_t B, int64_t H, int64_t D, int64_t n, int64_t m,
                            int64_t q_bstride, int64_t kv_bstride, float* dq, float* dk, float* dv, float* ws,
                            void* stream) {
  if (!q || !k || !v || !o || !dout || !lse || !dq || !dk || !dv || !ws) return ADP_ERR_NULL;
  if (!attn_shape_ok(B, H, D, n, m)) return ADP_ERR_SHAPE;
  const float scale = 1.0f / sqrtf((float)D);
  const int64_t ns = kv_nsplit(B, H, n, m), ktiles = adp_cdiv(m, 32), qtiles = adp_cdiv(n, 32);
  const int64_t tps = adp_cdiv(qtiles, ns);  // query tiles per split
  float* pk = dk;
  float* pv = dv;
  int64_t pstride = 0;
  if (ns > 1) {
    if (kv_bstride > 2 * H * D * m) return ADP_ERR_UNSUPPORTED;  // the partial copies are sized for packed k|v
    pstride = B * kv_bstride;                 // one partial copy, addressed exactly like dk / dv
    pk = ws + B * H * n;
    pv = pk + ns * pstride;
  }
  // query-major pass first: it also computes delta = rowsum(dO * O) for its queries and leaves it in ws[0 .. B*H*n) for the
  // key-major pass.  Key-split when the query tiles alone do not fill the chip (needs packed q: one partial copy is
  // addressed exactly like dq)
  int64_t qtps = ktiles;
  int64_t nq = (q_bstride == H * D * n) ? q_nsplit(B, H, n, m, &qtps) : 1;
  if (nq == 1) qtps = ktiles;
  float* pq = dq;
  int64_t qstride = 0;
  if (nq > 1) {
    pq = ws + B * H * n + (ns > 1 ? 2 * ns * pstride : 0);
    qstride = B * q_bstride;
  }
  {  // few keys: four waves per query tile / per query slice in one launch (ADP_ATTN_FEWKEYS=0: the one-wave forms, A/B)
    const char* fk = getenv("ADP_ATTN_FEWKEYS");
    // (taken while the query tiles are few: at 1024+ the chip is full of one-wave items either way -- hipGraph microbench, us per
    // backward, one-wave forms -> this kernel: batch 1 n = 128 16.4 -> 9.8, 256 16.2 -> 9.5, 1024 20.9 -> 17.0, 4096 52.1 ->
    // 51.3; batch 4 n = 256 20.0 -> 15.9, n = 1024 40.4 -> 45.6, n = 4096 123 -> 177)
    if (D == 64 && m <= 64 && B * H * qtiles <= (B * H <= 8 ? 1024 : 512) && (!fk || fk[0] != '0')) {
      const char* et = getenv("ADP_ATTN_FK_SLICES");
      int64_t ns4 = (et ? atoi(et) : 512) / (B * H);  // ~two workgroups per CU for the key-major role
      if (ns4 > qtiles) ns4 = qtiles;
      if (ns4 > 32) ns4 = 32;
      if (ns4 < 1) ns4 = 1;
      if (ns4 > ns) ns4 = ns;  // (the scratch is sized for kv_nsplit partial copies)
      const int64_t tps4 = adp_cdiv(qtiles, ns4);
      ns4 = adp_cdiv(qtiles, tps4);  // every slice non-empty
      attn_fk_args a;
      a.q = q, a.k = k, a.v = v, a.o = o, a.dout = dout, a.lse = lse;
      a.H = (int)H, a.n = (int)n, a.m = (int)m, a.qbs = q_bstride, a.kvbs = kv_bstride, a.scale = scale;
      a.dq = dq;
      a.dk = ns4 > 1 ? pk : dk, a.dv = ns4 > 1 ? pv : dv;
      a.ns = (int)ns4, a.tps = (int)tps4, a.pstride = pstride, a.gq = (int)qtiles;
      unsigned gx4 = (unsigned)(qtiles + ns4);
#ifdef ADP_ATTN_FK_DEBUG
      if (const char* eo = getenv("ADP_ATTN_FK_ONLY")) {  // timing only (results incomplete): one role alone
        if (eo[0] == 'q') gx4 = (unsigned)qtiles;
        if (eo[0] == 'k') a.gq = 0, gx4 = (unsigned)ns4;
      }
#endif
      ADP_LAUNCH(attn_bwd_fewkeys_kernel, dim3(gx4, (unsigned)H, (unsigned)B), dim3(256), stream, a);
      if (ns4 > 1) {
        const int64_t gx = adp_cdiv(H * D * m, 256);
        ADP_LAUNCH(attn_bwd_reduce_kernel, dim3((unsigned)(gx < 2048 ? gx : 2048), (unsigned)(2 * B)), dim3(256), stream,
                   (const float*)dq, 1, (int64_t)0, q_bstride, H * D * n, dq, 0, (const float*)pk, (const float*)pv, (int)ns4,
                   pstride, kv_bstride, H * D * m, dk, dv);
      }
      return ADP_LAUNCH_OK();
    }
  }
  const dim3 gq2((unsigned)(adp_cdiv(n, 128) * nq), (unsigned)H, (unsigned)B);
  const dim3 gkv((unsigned)adp_cdiv(ktiles * ns, 4), (unsigned)H, (unsigned)B);
  // One launch for both passes while together they are at most one wave per SIMD (batch 1; hipGraph microbench, us per backward,
  // separate -> merged: cross attention over 64 keys n = 128 27.6 -> 16.6, n = 1024 32.3 -> 20.7, n = 4096 (1536 waves) 52.3 ->
  // 53.1; self attention n = 256 35.1 -> 21.7, n = 1024 (2048 waves: the chip is full either way and the key-major pass pays
  // for its O columns) 107 -> 127).  ADP_ATTN_MERGE=0 / 1 forces either form (tests, A/B).
  const char* em = getenv("ADP_ATTN_MERGE");
  const bool merge = em ? em[0] != '0' : (int64_t)(gq2.x + gkv.x) * H * B * 4 <= 1024;
  if (merge) {
    attn_bwd_args a;
    a.q = q, a.k = k, a.v = v, a.o = o, a.dout = dout, a.lse = lse;
    a.H = (int)H, a.D = (int)D, a.n = (int)n, a.m = (int)m, a.qbs = q_bstride, a.kvbs = kv_bstride, a.scale = scale;
    a.dq = pq, a.qtps = (int)qtps, a.qstride = qstride;
    a.dk = pk, a.dv = pv, a.ns = (int)ns, a.tps = (int)tps, a.pstride = pstride, a.gq = (int)gq2.x;
    const dim3 grid(gq2.x + gkv.x, (unsigned)H, (unsigned)B);
    if (D == 64) ADP_LAUNCH(attn_bwd_merged_kernel<true>, grid, dim3(256), stream, a);
    else ADP_LAUNCH(attn_bwd_merged_kernel<false>, grid, dim3(256), stream, a);
    if (nq > 1 || ns > 1) {
      const int64_t gx = nq > 1 ? adp_cdiv(H * D * n, 256) : adp_cdiv(H * D * m, 256);
      const dim3 gr((unsigned)(gx < 2048 ? gx : 2048), (unsigned)((nq > 1 ? B : 0) + (ns > 1 ? 2 * B : 0)));
      ADP_LAUNCH(attn_bwd_reduce_kernel, gr, dim3(256), stream, (const float*)pq, (int)nq, qstride, q_bstride, H * D * n, dq,
                 (int)(nq > 1 ? B : 0), (const float*)pk, (const float*)pv, (int)ns, pstride, kv_bstride, H * D * m, dk, dv);
    }
    return ADP_LAUNCH_OK();
  }
  if (D == 64)
    ADP_LAUNCH(attn_bwd_q_kernel<true>, gq2, dim3(256), stream, q, k, v, o, dout, lse, ws, (int)H, (int)D, (int)n, (int)m,
               q_bstride, kv_bstride, scale, pq, (int)qtps, qstride);
  else
    ADP_LAUNCH(attn_bwd_q_kernel<false>, gq2, dim3(256), stream, q, k, v, o, dout, lse, ws, (int)H, (int)D, (int)n, (int)m,
               q_bstride, kv_bstride, scale, pq, (int)qtps, qstride);
  if (nq > 1)
    ADP_LAUNCH(attn_sum_splits_kernel, dim3((unsigned)adp_cdiv(H * D * n, 256), (unsigned)B), dim3(256), stream,
               (const float*)pq, (int)nq, qstride, q_bstride, H * D * n, dq);
  if (D == 64)
    ADP_LAUNCH(attn_bwd_kv_kernel<true>, gkv, dim3(256), stream, q, k, v, dout, lse, (const float*)ws, (int)H, (int)D,
               (int)n, (int)m, q_bstride, kv_bstride, scale, (int)ns, (int)tps, pstride, pk, pv);
  else
    ADP_LAUNCH(attn_bwd_kv_kernel<false>, gkv, dim3(256), stream, q, k, v, dout, lse, (const float*)ws, (int)H, (int)D,
               (int)n, (int)m, q_bstride, kv_bstride, scale, (int)ns, (int)tps, pstride, pk, pv);
  if (ns > 1)  // dk and dv are [H*D, m] slabs inside each batch stride
    ADP_LAUNCH(attn_kv_reduce_kernel, dim3((unsigned)adp_cdiv(H * D * m, 256), (unsigned)(2 * B)), dim3(256), stream,
               (const float*)pk, (const float*)pv, (int)ns, pstride, kv_bstride, H * D * m, dk, dv);
  return ADP_LAUNCH_OK();
}
