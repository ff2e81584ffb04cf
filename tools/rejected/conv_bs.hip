// conv_bs: the deep kernel-3 convolutions (forward and data gradient, >= 256 channels) on the bf16 matrix cores at
// fp32 accuracy ("bf16 split").
//
// gfx950 runs v_mfma_f32_32x32x16_bf16 (K = 16 in 32 cycles) at 16x the rate of the exact-f32
// v_mfma_f32_32x32x2_f32 (K = 2 in 64 cycles).  An fp32 value splits EXACTLY into three bf16 values by truncation,
// x = hi + mid + lo (8 + 8 + 8 significant bits), and a product of two such values is the sum of nine partial
// products of which the six down to 2^-16 are kept: lo*mid, mid*lo and lo*lo (<= 2^-24 relative each) are dropped --
// the size of one fp32 rounding.  Every partial product is an fp32-accumulating bf16 MFMA, so a K = 16 step costs
// 6 x 32 cycles instead of 8 x 64: 2.67x the exact-f32 MFMA rate.  Measured error against an fp64 reference over
// K = 3072 (tools/probe/split_probe.hip, tests/test_kernels.py): 2e-7 relative, the exact-f32 MFMA chain 5e-7.
//
// Structure: 128 output channels x 128 positions per workgroup; 4 MMA waves (2 x 2, a 64 x 64 accumulator block each
// = 4 MFMA tiles) + 4 loader waves.  Per 16-channel chunk the loaders read the fp32 weights / activations from
// global memory (one chunk ahead in registers), split them and write the three bf16 planes to LDS in the MFMA
// operand layout: weights [part][shift][row][16 channels], activations [part][position][16 channels] -- a lane's
// 8-channel fragment is one 16-byte read and a wave's fragment reads cover 1 KiB of consecutive LDS.  Two LDS stages,
// one barrier per chunk.  Small grids are filled by the cross-workgroup K split of conv_mm (partial tiles in d.ws +
// conv_splitk_reduce_kernel, fixed order).
#include "adp_rt.h"
#include "conv_internal.h"

namespace {

constexpr int BS_BM = 128, BS_BN = 128, BS_BK = 16;
constexpr int BS_NMMA = 4, BS_NLD = 4;
constexpr int BS_XROWS = BS_BN + 4;                       // positions n0-1 .. n0+128 (+2 rows of padding)
constexpr int BS_A_WORDS = 3 * 3 * BS_BM * 8;             // [part][shift][row][8 words = 16 bf16]
constexpr int BS_X_WORDS = 3 * BS_XROWS * 8;              // [part][position][8 words]
constexpr int BS_STAGE = BS_A_WORDS + BS_X_WORDS;         // 12384 words; two stages = 99 KB

// exact three-way split: the bits of each part sit in the upper half of a word
__device__ __forceinline__ void bs_split(float x, uint32_t& h, uint32_t& m, uint32_t& l) {
  h = __float_as_uint(x) & 0xffff0000u;
  const float r1 = x - __uint_as_float(h);
  m = __float_as_uint(r1) & 0xffff0000u;
  const float r2 = r1 - __uint_as_float(m);
  l = __float_as_uint(r2) & 0xffff0000u;
}
// two parts (upper halves of a, b) -> one word, a in the low half (bf16 element 0)
__device__ __forceinline__ uint32_t bs_pack(uint32_t a, uint32_t b) { return (a >> 16) | b; }

// four fp32 values (consecutive channels) -> the three 8-byte LDS slots
__device__ __forceinline__ void bs_store4(uint32_t* base, int part_stride, float v0, float v1, float v2, float v3) {
  uint32_t h[4], m[4], l[4];
  bs_split(v0, h[0], m[0], l[0]);
  bs_split(v1, h[1], m[1], l[1]);
  bs_split(v2, h[2], m[2], l[2]);
  bs_split(v3, h[3], m[3], l[3]);
  *reinterpret_cast<uint2*>(base) = make_uint2(bs_pack(h[0], h[1]), bs_pack(h[2], h[3]));
  *reinterpret_cast<uint2*>(base + part_stride) = make_uint2(bs_pack(m[0], m[1]), bs_pack(m[2], m[3]));
  *reinterpret_cast<uint2*>(base + 2 * part_stride) = make_uint2(bs_pack(l[0], l[1]), bs_pack(l[2], l[3]));
}

// TR = false: out[m][n] = sum_{r,t} w[m][r][t] x[r][n + t - 1]        (w: [M][R][3])
// TR = true : out[m][n] = sum_{r,t} w[r][m][t] x[r][n + 1 - t]        (w: [R][M][3], the data gradient)
// Both read the activation tile at LDS position (n - n0) + s, s = 0..2, against weight tap t = s (TR: t = 2 - s).
template <bool TR>
__global__ __launch_bounds__((BS_NMMA + BS_NLD) * 64) void conv_bs_kernel(adp_conv_desc d, int KS) {
  __shared__ __attribute__((aligned(16))) uint32_t smem[2 * BS_STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int M = (int)d.M, R = (int)d.R, L = (int)d.Lin, N = (int)d.N;

  int id = blockIdx.x;
  const int total = gridDim.x;
  if ((total & 7) == 0) id = (id & 7) * (total >> 3) + (id >> 3);  // consecutive tiles of a row share an XCD's L2
  const int ntn = (N + BS_BN - 1) / BS_BN, per_m = ntn * (int)d.B;
  const int ks = blockIdx.y;
  const int mt = id / per_m, rem = id - mt * per_m;
  const int b = rem / ntn, nt = rem - b * ntn;
  const int m0 = mt * BS_BM, n0 = nt * BS_BN;
  const int nch = R / BS_BK;
  const int c_lo = (int)((int64_t)nch * ks / KS), c_hi = (int)((int64_t)nch * (ks + 1) / KS);

  if (wave >= BS_NMMA) {
    // ------------------------------------------------------------------ loader waves
    const int lt = tid - BS_NMMA * 64;  // 0 .. 255
    const int cq = lt & 3;              // channel quad of the chunk
    const int rw = lt >> 2;             // 0 .. 63
    const float* xb = d.x + (int64_t)b * R * L;
    float wa[3][2][12], xa[3][3][4];    // [register stage][task][values]: two chunks in flight beside the one staged
    auto issue = [&](int c, int st) {
      const int c0 = c * BS_BK + cq * 4;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int row = rw + 64 * k;
        if (!TR) {
          const float4* p = reinterpret_cast<const float4*>(d.w + ((int64_t)(m0 + row) * R + c0) * 3);
#pragma unroll
          for (int q = 0; q < 3; ++q) {
            const float4 v = p[q];
            wa[st][k][4 * q] = v.x;
            wa[st][k][4 * q + 1] = v.y;
            wa[st][k][4 * q + 2] = v.z;
            wa[st][k][4 * q + 3] = v.w;
          }
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float* p = d.w + ((int64_t)(c0 + j) * M + m0 + row) * 3;
#pragma unroll
            for (int t = 0; t < 3; ++t) wa[st][k][3 * j + t] = p[t];
          }
        }
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int q = rw + 64 * k;  // LDS position row; k == 2 only covers rows 128, 129
        const int pos = n0 - 1 + q;
        const bool ok = pos >= 0 && pos < L && q < BS_BN + 2;
        const int pc = pos < 0 ? 0 : (pos >= L ? L - 1 : pos);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float v = 0.0f;
          if (k < 2 || rw < 2) v = xb[(int64_t)(c0 + j) * L + pc];
          xa[st][k][j] = ok ? v : 0.0f;
        }
      }
    };
    auto stage = [&](int st, int buf) {
      uint32_t* A = smem + buf * BS_STAGE;
      uint32_t* X = A + BS_A_WORDS;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int row = rw + 64 * k;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          const int s = TR ? 2 - t : t;
          bs_store4(A + (s * BS_BM + row) * 8 + cq * 2, 3 * BS_BM * 8, wa[st][k][t], wa[st][k][3 + t], wa[st][k][6 + t],
                    wa[st][k][9 + t]);
        }
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int q = rw + 64 * k;
        if (k < 2 || rw < 2) bs_store4(X + q * 8 + cq * 2, BS_XROWS * 8, xa[st][k][0], xa[st][k][1], xa[st][k][2], xa[st][k][3]);
      }
    };
    // chunk i of this workgroup: register stage i % 3, LDS stage i % 2 (constant indices: six chunks per trip)
    auto step = [&](int c, int rs, int buf) {
      if (c + 2 < c_hi) issue(c + 2, (rs + 2) % 3);
      stage(rs, buf);
      __syncthreads();
    };
    if (c_lo < c_hi) issue(c_lo, 0);
    if (c_lo + 1 < c_hi) issue(c_lo + 1, 1);
    for (int c = c_lo; c < c_hi; c += 6) {
      step(c, 0, 0);
      if (c + 1 < c_hi) step(c + 1, 1, 1);
      if (c + 2 < c_hi) step(c + 2, 2, 0);
      if (c + 3 < c_hi) step(c + 3, 0, 1);
      if (c + 4 < c_hi) step(c + 4, 1, 0);
      if (c + 5 < c_hi) step(c + 5, 2, 1);
    }
    return;
  }

  // -------------------------------------------------------------------- MMA waves
  const int wm0 = (wave & 1) * 64, wn0 = (wave >> 1) * 64;
  const int l31 = lane & 31, hi = lane >> 5;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  for (int c = c_lo; c < c_hi; ++c) {
    __syncthreads();
    const uint32_t* A = smem + ((c - c_lo) & 1) * BS_STAGE;
    const uint32_t* X = A + BS_A_WORDS;
    // fragments of shift s + 1 are read while the 24 MFMAs of shift s run
    bf16x8 a[2][2][3], bb[2][2][3];
    auto frags = [&](int s, int f) {
#pragma unroll
      for (int p = 0; p < 3; ++p) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
          a[f][i][p] = *reinterpret_cast<const bf16x8*>(A + ((p * 3 + s) * BS_BM + wm0 + i * 32 + l31) * 8 + hi * 4);
#pragma unroll
        for (int j = 0; j < 2; ++j)
          bb[f][j][p] = *reinterpret_cast<const bf16x8*>(X + (p * BS_XROWS + wn0 + j * 32 + l31 + s) * 8 + hi * 4);
      }
    };
    frags(0, 0);
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const int f = s & 1;
      if (s < 2) frags(s + 1, f ^ 1);
      adp_sched_fence();
      // partial products, smallest first: (mid,mid) (hi,lo) (lo,hi) | (hi,mid) (mid,hi) | (hi,hi)
#pragma unroll
      for (int o = 2; o >= 0; --o)
#pragma unroll
        for (int pa = 0; pa <= o; ++pa) {
          const int pb = o - pa;
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = adp_mfma32_bf16(a[f][i][pa], bb[f][j][pb], acc[i][j]);
        }
    }
  }

  // ---- epilogue (store mode 0): raw partial tile for the K split, else bias / out_pre / e_scale / residual
  const int64_t ebs = d.e_bstride ? d.e_bstride : M;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + wn0 + j * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (n >= N) continue;
        float v = acc[i][j][r];
        if (KS > 1) {
          d.ws[(((int64_t)ks * d.B + b) * M + m) * N + n] = v;
          continue;
        }
        const int64_t o = ((int64_t)b * M + m) * N + n;
        if (d.bias) v += d.bias[m];
        if (d.out_pre) d.out_pre[o] = v;
        if (d.e_scale) v *= d.e_scale[b * ebs + m];
        if (d.res) v += d.res[o];
        d.out[o] = v;
      }
    }
}

}  // namespace

// A/B switch while the family is being brought up: ADP_CONV_BS=1 routes eligible convs here.
bool adp_conv_bs_enabled() {
  const char* e = getenv("ADP_CONV_BS");  // read per call: the tests flip it inside one process
  return e != nullptr && e[0] == '1';
}

bool adp_conv_bs_eligible(const adp_conv_desc& d) {
  if (d.KT != 3 || d.stride != 1 || d.up != 1 || d.dil != 1 || d.pad != 1 || d.prologue != 0 || d.store != 0) return false;
  if (d.R1 != d.R || d.N != d.Lin) return false;
  if (d.R % BS_BK != 0 || d.M % BS_BM != 0 || d.R < 256) return false;
  if (reinterpret_cast<uintptr_t>(d.w) & 15) return false;
  if (d.B * d.R * d.Lin >= (int64_t)1 << 31 || d.M * d.R * 3 >= (int64_t)1 << 31) return false;
  return true;
}

int64_t adp_conv_bs_ksplit(const adp_conv_desc& d) {
  const int64_t blocks = (d.M / BS_BM) * adp_cdiv(d.N, BS_BN) * d.B;
  const int64_t nch = d.R / BS_BK;
  int64_t ks = 1;
  while (ks < 16 && blocks * ks < 200 && nch / (ks * 2) >= 4) ks *= 2;
  return ks;
}

int adp_conv_bs(const adp_conv_desc& d, void* stream) {
  const int64_t blocks = (d.M / BS_BM) * adp_cdiv(d.N, BS_BN) * d.B;
  const int KS = d.ws ? (int)adp_conv_bs_ksplit(d) : 1;
  if (d.transposed)
    ADP_LAUNCH((conv_bs_kernel<true>), dim3((unsigned)blocks, (unsigned)KS), dim3((BS_NMMA + BS_NLD) * 64), stream, d, KS);
  else
    ADP_LAUNCH((conv_bs_kernel<false>), dim3((unsigned)blocks, (unsigned)KS), dim3((BS_NMMA + BS_NLD) * 64), stream, d, KS);
  if (ADP_LAUNCH_OK() != ADP_OK) return ADP_ERR_LAUNCH;
  if (KS > 1) return adp_conv_splitk_reduce(d, KS, stream);
  return ADP_OK;
}
