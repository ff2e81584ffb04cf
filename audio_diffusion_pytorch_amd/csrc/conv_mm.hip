// Deep-layer implicit-GEMM Conv1d for gfx950: stride 1, kernel 1 or 3, channel counts that are multiples of 32.
// This is the kernel the MFMA-bound half of the U-Net lives in (ResnetItem ConvBlocks and their data
// gradients at depths 2-8; /root/reference/audio_diffusion_pytorch/components.py:89).
//
// Shape of the machine it is written for:
//   * v_mfma_f32_32x32x2_f32 retires one 32x32x2 tile per 64 cycles per SIMD, so operand traffic is tiny
//     (two VGPRs per MFMA) and the only thing that matters is that every SIMD always has an MFMA to issue.
//     Deep layers have few output tiles (depth 8 at batch 4: 512 tiles of 32x32 for 1024 SIMDs), so the block's
//     waves split K instead: a block owns a BM x BN output tile, NKG wave groups each take BKT/NKG of the 32
//     channels of every staged chunk, and their partial tiles are summed through LDS in a fixed order at the
//     end (deterministic).  That puts 2-4 waves on every SIMD with one block per CU.
//   * both operands are staged with 16-byte global loads and 16-byte LDS stores, register-prefetched one chunk
//     ahead, double-buffered in LDS, one barrier per chunk.  Weights are copied as they lie in memory:
//       forward   As[m][r*KT+t]  (row = 32 channels x KT taps, contiguous in w[M][R][KT]); the A fragment
//                 of lane (m, hi) is 4 channels x KT taps = KT ds_read_b128 (row stride = 4 mod 8 dwords: no
//                 bank conflicts), i.e. 4*KT MFMAs per KT LDS reads;
//       gradient  As[k][m*KT+t]  (row = BM outputs x KT taps, contiguous in w[R][M][KT]); fragment reads are
//                 stride-KT ds_read_b32 (conflict-free for KT = 1, 3).
//     The MFMA K pair is (channel c + 4*hi), which both layouts share.
//   * the GroupNorm+SiLU prologue is applied between the global load and the LDS store, from per-(b,channel)
//     constants kept in LDS; zero padding is applied after the activation, like nn.Conv1d.
//   * the 1-D grid is decoded XCD-aware: workgroup id -> XCD id%8 (round-robin dispatch), so ids are permuted
//     to give each XCD a contiguous range of weight row tiles; its 4 MiB L2 then holds 1/8 of the weights.
#include "adp_rt.h"
#include "adp.h"
#include "conv_internal.h"

namespace {

constexpr int MM_BKT = 32;        // channels per staged chunk (16 for the stride-4 variant); R must divide by it
constexpr int MM_PRO_RMAX = 1024; // channels whose GroupNorm constants fit the LDS table

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

// S: conv stride (1, or kernel = stride = 2 / 4 for DownsampleItem); UP: nearest-upsample factor folded into the X
// loader (UpsampleItem: the [B, C, L*UP] intermediate is never materialised); BKT: channels per staged chunk.
template <int BM, int BN, int NKG, int KT, int S, int UP, bool TR, int PRO, int BKT>
__global__ __launch_bounds__((BM / 32) * (BN / 32) * NKG * 64) void conv_mm_kernel(adp_conv_desc d) {
  constexpr int CPW = BKT / NKG;
  constexpr int NQN = BN / 32, NQ = (BM / 32) * NQN, NW = NQ * NKG, NT = NW * 64;
  constexpr int QK = BKT * KT;
  constexpr int AS = TR ? (BM * KT + 4) : (QK + 4);  // A row stride in floats
  constexpr int AROWS = TR ? BKT : BM;
  constexpr int AQ = (TR ? BM * KT : QK) / 4;        // float4 per A row
  constexpr int XSP = BN * S + 8, XQ = XSP / 4;      // X row: (virtual) positions n0*S-4 .. n0*S+BN*S+3
  constexpr int A_ELEMS = AROWS * AS, X_ELEMS = BKT * XSP;
  constexpr int NA4 = (AROWS * AQ + NT - 1) / NT, NX4 = (BKT * XQ + NT - 1) / NT;
  constexpr int RED = (NKG - 1) * NQ * 1024;
  constexpr int SM = cmax<2 * (A_ELEMS + X_ELEMS), RED>::v;
  static_assert(CPW % 8 == 0, "a wave consumes channels in groups of 8");
  __shared__ __attribute__((aligned(16))) float smem[SM];
  __shared__ float Pa[PRO == 1 ? MM_PRO_RMAX : 1], Pb[PRO == 1 ? MM_PRO_RMAX : 1];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hi = lane >> 5, l31 = lane & 31;
  const int quad = wave % NQ, kg = wave / NQ;
  const int wm0 = (quad / NQN) * 32, wn0 = (quad % NQN) * 32;

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

  const float* xb = d.x + (int64_t)b * R * L;
  const float* wbase = TR ? d.w + (int64_t)m0 * KT : d.w + (int64_t)m0 * R * KT;

  if (PRO == 1) {
    const int cpg = R / (int)d.groups;
    for (int r = tid; r < R; r += NT) {
      const int g = r / cpg;
      const float mean = d.pro_stats[((int64_t)b * d.groups + g) * 2];
      const float ga = (d.pro_gamma ? d.pro_gamma[r] : 1.0f) * d.pro_stats[((int64_t)b * d.groups + g) * 2 + 1];
      Pa[r] = ga;
      Pb[r] = (d.pro_beta ? d.pro_beta[r] : 0.0f) - mean * ga;
    }
  }

  // ---- per-thread staging slots (chunk independent parts).  Slot indices wrap around instead of being
  // guarded: a few threads then stage the same 16 bytes twice, and the loop body stays branch-free.
  int a_src[NA4], a_dst[NA4];
#pragma unroll
  for (int i = 0; i < NA4; ++i) {
    const int e = (tid + i * NT) % (AROWS * AQ);
    const int row = e / AQ, qq = e - row * AQ;
    a_dst[i] = row * AS + 4 * qq;
    a_src[i] = TR ? row * M * KT + 4 * qq : row * R * KT + 4 * qq;
  }
  int x_src[NX4], x_dst[NX4], x_row[NX4];
  bool x_ok[NX4];
#pragma unroll
  for (int i = 0; i < NX4; ++i) {
    const int e = (tid + i * NT) % (BKT * XQ);
    const int rl = e / XQ, pq = e - rl * XQ;
    const int u = n0 * S - 4 + 4 * pq;
    x_dst[i] = rl * XSP + 4 * pq;
    x_ok[i] = (u >= 0 && u < Lv);  // Lv % 4 == 0: a quad is entirely inside or outside the row
    x_src[i] = rl * L + (x_ok[i] ? u / UP : 0);  // nearest upsample: source index = floor(u / UP), exact
    x_row[i] = rl;
  }

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;

  // lane-constant fragment offsets
  const int xfrag = 4 * hi * XSP + (wn0 + l31) * S + 4 - pad;         // + (ci + c) * XSP + t * dil
  const int afrag = TR ? 4 * hi * AS + (wm0 + l31) * KT                // + (ci + c) * AS + (KT - 1 - t)
                       : (wm0 + l31) * AS + 4 * hi * KT;               // + ci * KT + (c * KT + t)

  // Software pipeline, one barrier per chunk:
  //   registers(chunk c) -> LDS[c&1] | barrier | global loads of chunk c+1 -> registers | MFMAs over LDS[c&1]
  // (the store of chunk c+1 overwrites the buffer read by chunk c-1, which every wave has finished before it
  // passed the barrier of chunk c).
  f32x4 ra[NA4], rx[NX4];
  const int nchunks = R / BKT;
  {
    const float* wp = wbase;
#pragma unroll
    for (int i = 0; i < NA4; ++i) ra[i] = *reinterpret_cast<const f32x4*>(wp + a_src[i]);
#pragma unroll
    for (int i = 0; i < NX4; ++i) rx[i] = load_xquad<UP>(xb + x_src[i]);
  }
  if (PRO == 1) __syncthreads();

  for (int c = 0; c < nchunks; ++c) {
    const int r0 = c * BKT;
    float* Ab = smem + (c & 1) * (A_ELEMS + X_ELEMS);
    float* Xb = Ab + A_ELEMS;
    // ---- registers -> LDS, with the prologue
#pragma unroll
    for (int i = 0; i < NA4; ++i) *reinterpret_cast<f32x4*>(Ab + a_dst[i]) = ra[i];
#pragma unroll
    for (int i = 0; i < NX4; ++i) {
      f32x4 v = rx[i];
      if (PRO == 1) {
        const float pa = Pa[r0 + x_row[i]], pb = Pb[r0 + x_row[i]];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = adp_silu_fast(fmaf(v[j], pa, pb));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = x_ok[i] ? v[j] : 0.0f;  // zero padding is applied after the activation
      *reinterpret_cast<f32x4*>(Xb + x_dst[i]) = v;
    }
    __syncthreads();
    // ---- prefetch the next chunk
    if (c + 1 < nchunks) {
      const int rn = r0 + BKT;
      const float* wp = TR ? wbase + (int64_t)rn * M * KT : wbase + rn * KT;
#pragma unroll
      for (int i = 0; i < NA4; ++i) ra[i] = *reinterpret_cast<const f32x4*>(wp + a_src[i]);
      const float* xp = xb + (int64_t)rn * L;
#pragma unroll
      for (int i = 0; i < NX4; ++i) rx[i] = load_xquad<UP>(xp + x_src[i]);
    }
    // ---- matrix cores over this wave's share of the chunk
#pragma unroll
    for (int i8 = 0; i8 < CPW / 8; ++i8) {
      const int ci = kg * CPW + 8 * i8;
      if (!TR) {
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
          for (int t = 0; t < KT; ++t)
            acc = adp_mfma32(av[cc * KT + t], Xb[xfrag + (ci + cc) * XSP + t * dil], acc);
      } else {
#pragma unroll
        for (int cc = 0; cc < 4; ++cc)
#pragma unroll
          for (int t = 0; t < KT; ++t)
            acc = adp_mfma32(Ab[afrag + (ci + cc) * AS + (KT - 1 - t)], Xb[xfrag + (ci + cc) * XSP + t * dil], acc);
      }
    }
  }
  __syncthreads();

  // ---- fixed-order sum of the K groups through LDS (the staging buffers are free after the last barrier)
  if (NKG > 1) {
    if (kg > 0) {
      float* rp = smem + ((kg - 1) * NQ + quad) * 1024 + lane;
#pragma unroll
      for (int r = 0; r < 16; ++r) rp[r * 64] = acc[r];
    }
    __syncthreads();
    if (kg == 0) {
#pragma unroll
      for (int g = 1; g < NKG; ++g) {
        const float* rp = smem + ((g - 1) * NQ + quad) * 1024 + lane;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += rp[r * 64];
      }
    }
  }
  if (kg != 0) return;

  // ---- epilogue (same contract as adp_conv1d's generic kernel)
  const int sp = (int)d.sp;
  const int64_t ebs = d.e_bstride ? d.e_bstride : M;
  const int n = n0 + wn0 + l31;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + wm0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
    const bool ok = (m < M) && (n < N);
    float v = acc[r];
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

template <int BM, int BN, int NKG, int KT, int S, int UP, bool TR, int PRO, int BKT>
int launch_mm(const adp_conv_desc& d, void* stream) {
  const int64_t blocks = (d.M / BM) * adp_cdiv(d.N, BN) * d.B;
  ADP_LAUNCH((conv_mm_kernel<BM, BN, NKG, KT, S, UP, TR, PRO, BKT>), dim3((unsigned)blocks),
             dim3((BM / 32) * (BN / 32) * NKG * 64), stream, d);
  return ADP_LAUNCH_OK();
}

// ~1024 SIMDs want >= 2 waves each; a 64x64 tile carries 4 waves per K group.  Returns NKG*1e6 + BM*1e3 + BN.
// The stride-4 variant stages 16 channels per chunk (its X rows are 4x wider), which caps NKG at 2.
int64_t mm_tile(const adp_conv_desc& d) {
  const int64_t tiles64 = (d.M % 64 == 0) ? (d.M / 64) * adp_cdiv(d.N, 64) * d.B : 0;
  const int64_t kmax = (d.stride == 4) ? 2 : 4;
  if (tiles64 >= 160) {
    if (tiles64 >= 768) return 1064064;
    if (tiles64 >= 384 || kmax == 2) return 2064064;
    return 4064064;
  }
  return kmax * 1000000 + 32064;
}

template <int KT, int S, int UP, bool TR, int PRO, int BKT>
int pick_mm(const adp_conv_desc& d, void* stream) {
  switch (mm_tile(d)) {
    case 1064064: return launch_mm<64, 64, 1, KT, S, UP, TR, PRO, BKT>(d, stream);
    case 2064064: return launch_mm<64, 64, 2, KT, S, UP, TR, PRO, BKT>(d, stream);
    case 2032064: return launch_mm<32, 64, 2, KT, S, UP, TR, PRO, BKT>(d, stream);
    default: break;
  }
  if constexpr (BKT >= 32) {
    if (mm_tile(d) == 4064064) return launch_mm<64, 64, 4, KT, S, UP, TR, PRO, BKT>(d, stream);
    return launch_mm<32, 64, 4, KT, S, UP, TR, PRO, BKT>(d, stream);
  }
  return ADP_ERR_UNSUPPORTED;
}

}  // namespace

bool adp_conv_mm_eligible(const adp_conv_desc& d) {
  if (d.R1 != d.R) return false;
  const bool plain = d.stride == 1 && (d.KT == 1 || d.KT == 3) && d.up == 1;                    // ConvBlock family
  const bool upc = d.stride == 1 && d.KT == 3 && (d.up == 2 || d.up == 4) && !d.transposed && d.prologue == 0;
  const bool down = (d.stride == 2 || d.stride == 4) && d.KT == d.stride && d.up == 1 && d.pad == 0 && d.dil == 1 &&
                    !d.transposed && d.prologue == 0;
  if (!plain && !upc && !down) return false;
  if (d.prologue != 0 && d.prologue != 1) return false;
  if (d.prologue == 1 && d.R > MM_PRO_RMAX) return false;
  if (d.R % MM_BKT != 0 || d.M % 32 != 0 || (d.Lin * d.up) % 4 != 0) return false;
  if (d.pad > 4 || (d.KT - 1) * d.dil - d.pad > 4) return false;  // halo of 4 positions on both sides of the X tile
  if ((d.N - 1) * d.stride + (d.KT - 1) * d.dil - d.pad >= d.Lin * d.up + 4) return false;
  if ((reinterpret_cast<uintptr_t>(d.x) | reinterpret_cast<uintptr_t>(d.w)) & 15) return false;
  if (d.B * d.R * d.Lin >= (int64_t)1 << 31 || d.M * d.R * d.KT >= (int64_t)1 << 31) return false;
  return true;
}

int64_t adp_conv_mm_tile(const adp_conv_desc& d) { return mm_tile(d); }

int adp_conv_mm(const adp_conv_desc& d, void* stream) {
  const bool tr = d.transposed != 0;
  if (d.stride == 2) return pick_mm<2, 2, 1, false, 0, 32>(d, stream);
  if (d.stride == 4) return pick_mm<4, 4, 1, false, 0, 16>(d, stream);
  if (d.up == 2) return pick_mm<3, 1, 2, false, 0, 32>(d, stream);
  if (d.up == 4) return pick_mm<3, 1, 4, false, 0, 32>(d, stream);
  if (d.KT == 3) {
    if (d.prologue == 1) return tr ? pick_mm<3, 1, 1, true, 1, 32>(d, stream) : pick_mm<3, 1, 1, false, 1, 32>(d, stream);
    return tr ? pick_mm<3, 1, 1, true, 0, 32>(d, stream) : pick_mm<3, 1, 1, false, 0, 32>(d, stream);
  }
  if (d.prologue == 1) return tr ? pick_mm<1, 1, 1, true, 1, 32>(d, stream) : pick_mm<1, 1, 1, false, 1, 32>(d, stream);
  return tr ? pick_mm<1, 1, 1, true, 0, 32>(d, stream) : pick_mm<1, 1, 1, false, 0, 32>(d, stream);
}
