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
// One workgroup = 4 waves = 4 x 32 queries of one (batch, head); K/V tiles of 32 keys are staged in LDS once
// per workgroup and shared by the 4 waves.  D (head features) <= 64, multiple of 2.
#include "adp_rt.h"
#include "adp.h"

namespace {

constexpr int DMAX = 64;
constexpr int KP = 33;  // LDS row stride of a [D][32] tile (+1: column reads down D are conflict-free)

__device__ __forceinline__ int acc_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// stage a [D][32] tile of channel-major src (row stride ld) starting at column c0 into dst[D][KP]; zero past cmax
__device__ __forceinline__ void stage_tile(const float* src, int64_t ld, int64_t c0, int64_t cmax, int D, float* dst) {
  for (int e = threadIdx.x; e < D * 32; e += 256) {
    const int dd = e >> 5, c = e & 31;
    dst[dd * KP + c] = (c0 + c < cmax) ? src[dd * ld + c0 + c] : 0.0f;
  }
}

// ---------------------------------------------------------------------------------------------------
// forward: o[d, i] = sum_j softmax_j(q_i . k_j * scale) v[d, j];  lse[i] = log sum_j exp(s_ij)
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_fwd_kernel(const float* q, const float* k, const float* v, int H, int D,
                                                       int64_t n, int64_t m, int64_t qbs, int64_t kvbs, float scale,
                                                       float* o, float* lse) {
  __shared__ float Ks[DMAX * KP];
  __shared__ float Vs[DMAX * KP];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hi = lane >> 5, l31 = lane & 31;
  const int64_t b = blockIdx.z, h = blockIdx.y;
  const int64_t i0 = (int64_t)blockIdx.x * 128 + wave * 32;
  const float* qh = q + b * qbs + h * D * n;
  const float* kh = k + b * kvbs + h * D * m;
  const float* vh = v + b * kvbs + h * D * m;
  const int64_t iq = i0 + l31;
  const bool qok = iq < n;

  // Q fragments for this wave's 32 queries: B operand of S^T = K^T Q, lane (i = l31, kk = hi) -> q[d][i]
  float qf[DMAX / 2];
#pragma unroll
  for (int s = 0; s < DMAX / 2; ++s) qf[s] = (qok && 2 * s + hi < D) ? qh[(2 * s + hi) * n + iq] * scale : 0.0f;

  f32x16 oacc[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[t][r] = 0.0f;
  float mrun = -3.0e38f, lrun = 0.0f;

  for (int64_t j0 = 0; j0 < m; j0 += 32) {
    __syncthreads();
    stage_tile(kh, m, j0, m, D, Ks);
    stage_tile(vh, m, j0, m, D, Vs);
    __syncthreads();
    // S^T tile: rows j (regs), cols i (lanes): A[i'=j][kk=d] = k[d][j], B[kk=d][j'=i] = q[d][i]
    f32x16 st;
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = 0.0f;
#pragma unroll
    for (int s = 0; s < DMAX / 2; ++s)
      if (2 * s < D) st = adp_mfma32(Ks[(2 * s + hi) * KP + l31], qf[s], st);
    // online softmax over j: 16 in-register rows + the other half-wave
    float tmax = -3.0e38f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (j0 + acc_row(r, hi) >= m) st[r] = -3.0e38f;
      tmax = fmaxf(tmax, st[r]);
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float mnew = fmaxf(mrun, tmax);
    const float alpha = __expf(mrun - mnew);
    float psum = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = (j0 + acc_row(r, hi) < m) ? __expf(st[r] - mnew) : 0.0f;
      st[r] = p;
      psum += p;
    }
    psum += __shfl_xor(psum, 32, 64);
    lrun = lrun * alpha + psum;
    mrun = mnew;
    // O^T[d][i] = alpha * O^T + sum_j v[d][j] P^T[j][i]: A[i'=d][kk=j] = v[d][j(s,hi)], B = own register s
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (32 * t < D) {
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[t][r] *= alpha;
#pragma unroll
        for (int s = 0; s < 16; ++s) oacc[t] = adp_mfma32(Vs[(32 * t + l31) * KP + acc_row(s, hi)], st[s], oacc[t]);
      }
    }
  }
  const float inv = (lrun > 0.0f) ? 1.0f / lrun : 0.0f;
  float* oh = o + b * (int64_t)H * D * n + h * D * n;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int dd = 32 * t + acc_row(r, hi);
      if (dd < D && qok) oh[dd * n + iq] = oacc[t][r] * inv;
    }
  if (hi == 0 && qok) lse[(b * H + h) * n + iq] = mrun + logf(lrun);
}

// delta[b,h,i] = sum_d dO[d,i] * O[d,i]
__global__ __launch_bounds__(256) void attn_delta_kernel(const float* o, const float* dout, int H, int D, int64_t n,
                                                         float* delta) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t bh = blockIdx.y;
  if (i >= n) return;
  const float* op = o + bh * D * n;
  const float* dp = dout + bh * D * n;
  float s = 0.0f;
  for (int dd = 0; dd < D; ++dd) s = fmaf(op[dd * n + i], dp[dd * n + i], s);
  delta[bh * n + i] = s;
}

// ---------------------------------------------------------------------------------------------------
// backward, key-major pass: one wave owns 32 keys and loops over query tiles:
//   S[i][j] (rows i in regs, cols j across lanes), P = exp(S - lse_i), dP = dO^T V, dS = P (dP - delta_i) scale
//   dv[d][j] += sum_i dO[d][i] P[i][j] ; dk[d][j] += sum_i q[d][i] dS[i][j]
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_bwd_kv_kernel(const float* q, const float* k, const float* v,
                                                          const float* dout, const float* lse, const float* delta,
                                                          int H, int D, int64_t n, int64_t m, int64_t qbs,
                                                          int64_t kvbs, float scale, float* dk, float* dv) {
  __shared__ float Qs[DMAX * KP];
  __shared__ float Ds[DMAX * KP];
  __shared__ float Ls[32], Dl[32];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hi = lane >> 5, l31 = lane & 31;
  const int64_t b = blockIdx.z, h = blockIdx.y;
  const int64_t j0 = (int64_t)blockIdx.x * 128 + wave * 32;
  const float* qh = q + b * qbs + h * D * n;
  const float* kh = k + b * kvbs + h * D * m;
  const float* vh = v + b * kvbs + h * D * m;
  const float* doh = dout + (b * H + h) * (int64_t)D * n;
  const int64_t jk = j0 + l31;
  const bool kok = jk < m;
  // K and V fragments of this wave's 32 keys: B operands (lane (j = l31, kk = hi) -> x[d][j])
  float kf[DMAX / 2], vf[DMAX / 2];
#pragma unroll
  for (int s = 0; s < DMAX / 2; ++s) {
    const bool ok = kok && (2 * s + hi < D);
    kf[s] = ok ? kh[(2 * s + hi) * m + jk] : 0.0f;
    vf[s] = ok ? vh[(2 * s + hi) * m + jk] : 0.0f;
  }
  f32x16 dka[2], dva[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) dka[t][r] = dva[t][r] = 0.0f;

  for (int64_t i0 = 0; i0 < n; i0 += 32) {
    __syncthreads();
    stage_tile(qh, n, i0, n, D, Qs);
    stage_tile(doh, n, i0, n, D, Ds);
    if (threadIdx.x < 32) {
      const int64_t i = i0 + threadIdx.x;
      Ls[threadIdx.x] = (i < n) ? lse[(b * H + h) * n + i] : 0.0f;
      Dl[threadIdx.x] = (i < n) ? delta[(b * H + h) * n + i] : 0.0f;
    }
    __syncthreads();
    // S tile: A[i'=i][kk=d] = q[d][i] (unit stride in LDS row), B = kf
    f32x16 sa, dpa;
#pragma unroll
    for (int r = 0; r < 16; ++r) sa[r] = dpa[r] = 0.0f;
#pragma unroll
    for (int s = 0; s < DMAX / 2; ++s)
      if (2 * s < D) {
        sa = adp_mfma32(Qs[(2 * s + hi) * KP + l31], kf[s], sa);
        dpa = adp_mfma32(Ds[(2 * s + hi) * KP + l31], vf[s], dpa);
      }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int il = acc_row(r, hi);
      const bool ok = (i0 + il < n) && kok;
      const float p = ok ? __expf(sa[r] * scale - Ls[il]) : 0.0f;
      sa[r] = p;                                   // P[i][j]
      dpa[r] = p * (dpa[r] - Dl[il]) * scale;      // dS[i][j]
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (32 * t < D) {
#pragma unroll
        for (int s = 0; s < 16; ++s) {
          const int il = acc_row(s, hi);
          dva[t] = adp_mfma32(Ds[(32 * t + l31) * KP + il], sa[s], dva[t]);
          dka[t] = adp_mfma32(Qs[(32 * t + l31) * KP + il], dpa[s], dka[t]);
        }
      }
    }
  }
  float* dkh = dk + b * kvbs + h * D * m;
  float* dvh = dv + b * kvbs + h * D * m;
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

// ---------------------------------------------------------------------------------------------------
// backward, query-major pass: one wave owns 32 queries and loops over key tiles (transposed tiles, as forward):
//   dS^T[j][i] -> dq[d][i] = sum_j k[d][j] dS^T[j][i]
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_bwd_q_kernel(const float* q, const float* k, const float* v,
                                                         const float* dout, const float* lse, const float* delta,
                                                         int H, int D, int64_t n, int64_t m, int64_t qbs,
                                                         int64_t kvbs, float scale, float* dq) {
  __shared__ float Ks[DMAX * KP];
  __shared__ float Vs[DMAX * KP];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hi = lane >> 5, l31 = lane & 31;
  const int64_t b = blockIdx.z, h = blockIdx.y;
  const int64_t i0 = (int64_t)blockIdx.x * 128 + wave * 32;
  const float* qh = q + b * qbs + h * D * n;
  const float* kh = k + b * kvbs + h * D * m;
  const float* vh = v + b * kvbs + h * D * m;
  const float* doh = dout + (b * H + h) * (int64_t)D * n;
  const int64_t iq = i0 + l31;
  const bool qok = iq < n;
  float qf[DMAX / 2], df[DMAX / 2];
#pragma unroll
  for (int s = 0; s < DMAX / 2; ++s) {
    const bool ok = qok && (2 * s + hi < D);
    qf[s] = ok ? qh[(2 * s + hi) * n + iq] : 0.0f;
    df[s] = ok ? doh[(2 * s + hi) * n + iq] : 0.0f;
  }
  const float li = qok ? lse[(b * H + h) * n + iq] : 0.0f;
  const float di = qok ? delta[(b * H + h) * n + iq] : 0.0f;
  f32x16 dqa[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) dqa[t][r] = 0.0f;

  for (int64_t j0 = 0; j0 < m; j0 += 32) {
    __syncthreads();
    stage_tile(kh, m, j0, m, D, Ks);
    stage_tile(vh, m, j0, m, D, Vs);
    __syncthreads();
    f32x16 st, dpt;
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = dpt[r] = 0.0f;
#pragma unroll
    for (int s = 0; s < DMAX / 2; ++s)
      if (2 * s < D) {
        st = adp_mfma32(Ks[(2 * s + hi) * KP + l31], qf[s], st);    // S^T[j][i]
        dpt = adp_mfma32(Vs[(2 * s + hi) * KP + l31], df[s], dpt);  // dP^T[j][i] = sum_d v[d][j] dO[d][i]
      }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const bool ok = (j0 + acc_row(r, hi) < m) && qok;
      const float p = ok ? __expf(st[r] * scale - li) : 0.0f;
      dpt[r] = p * (dpt[r] - di) * scale;  // dS^T[j][i]
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (32 * t < D) {
#pragma unroll
        for (int s = 0; s < 16; ++s)
          dqa[t] = adp_mfma32(Ks[(32 * t + l31) * KP + acc_row(s, hi)], dpt[s], dqa[t]);
      }
    }
  }
  float* dqh = dq + b * qbs + h * D * n;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int dd = 32 * t + acc_row(r, hi);
      if (dd < D && qok) dqh[dd * n + iq] = dqa[t][r];
    }
}

bool attn_shape_ok(int64_t B, int64_t H, int64_t D, int64_t n, int64_t m) {
  return B > 0 && H > 0 && D >= 2 && D <= DMAX && (D % 2 == 0) && n > 0 && m > 0 && B <= 65535 && H <= 65535;
}

}  // namespace

extern "C" int adp_attn_fwd(const float* q, const float* k, const float* v, int64_t B, int64_t H, int64_t D,
                            int64_t n, int64_t m, int64_t q_bstride, int64_t kv_bstride, float* o, float* lse,
                            void* stream) {
  if (!q || !k || !v || !o || !lse) return ADP_ERR_NULL;
  if (!attn_shape_ok(B, H, D, n, m)) return ADP_ERR_SHAPE;
  const float scale = 1.0f / sqrtf((float)D);
  ADP_LAUNCH(attn_fwd_kernel, dim3((unsigned)adp_cdiv(n, 128), (unsigned)H, (unsigned)B), dim3(256), stream, q, k, v,
             (int)H, (int)D, n, m, q_bstride, kv_bstride, scale, o, lse);
  return ADP_LAUNCH_OK();
}

extern "C" int64_t adp_attn_bwd_ws_bytes(int64_t B, int64_t H, int64_t D, int64_t n, int64_t m) {
  if (!attn_shape_ok(B, H, D, n, m)) return ADP_ERR_SHAPE;
  return B * H * n * (int64_t)sizeof(float);
}

extern "C" int adp_attn_bwd(const float* q, const float* k, const float* v, const float* o, const float* dout,
                            const float* lse, int64_t B, int64_t H, int64_t D, int64_t n, int64_t m,
                            int64_t q_bstride, int64_t kv_bstride, float* dq, float* dk, float* dv, float* ws,
                            void* stream) {
  if (!q || !k || !v || !o || !dout || !lse || !dq || !dk || !dv || !ws) return ADP_ERR_NULL;
  if (!attn_shape_ok(B, H, D, n, m)) return ADP_ERR_SHAPE;
  const float scale = 1.0f / sqrtf((float)D);
  ADP_LAUNCH(attn_delta_kernel, dim3((unsigned)adp_cdiv(n, 256), (unsigned)(B * H)), dim3(256), stream, o, dout,
             (int)H, (int)D, n, ws);
  ADP_LAUNCH(attn_bwd_kv_kernel, dim3((unsigned)adp_cdiv(m, 128), (unsigned)H, (unsigned)B), dim3(256), stream, q,
             k, v, dout, lse, (const float*)ws, (int)H, (int)D, n, m, q_bstride, kv_bstride, scale, dk, dv);
  ADP_LAUNCH(attn_bwd_q_kernel, dim3((unsigned)adp_cdiv(n, 128), (unsigned)H, (unsigned)B), dim3(256), stream, q, k,
             v, dout, lse, (const float*)ws, (int)H, (int)D, n, m, q_bstride, kv_bstride, scale, dq);
  return ADP_LAUNCH_OK();
}
