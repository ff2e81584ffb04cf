// Dispatch of the implicit-GEMM Conv1d family (kernel: conv_mm_impl.h; one translation unit per block tile:
// conv_mm_m64.hip, conv_mm_m32.hip).
#include "adp_rt.h"
#include "adp.h"
#include "conv_internal.h"

int adp_conv_mm_m64(const adp_conv_desc& d, void* stream);
int adp_conv_mm_m32(const adp_conv_desc& d, void* stream);

namespace {

constexpr int MM_BKT = 32;        // channels per staged chunk (16 for the stride-4 variant); R must divide by it
constexpr int MM_PRO_RMAX = 1024; // channels whose GroupNorm constants fit the LDS table

// 64 x 64 block tiles (8 MMA waves + 4 loaders) unless that leaves most of the 256 CUs without a block
bool mm_use64(const adp_conv_desc& d) {
  if (d.M % 64 != 0) return false;
  return (d.M / 64) * adp_cdiv(d.N, 64) * d.B >= 200;
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

// NKG * 1e6 + BM * 1e3 + BN of the tile the dispatcher picks
int64_t adp_conv_mm_tile(const adp_conv_desc& d) {
  const int64_t nkg = d.stride == 4 ? 2 : 4;
  return nkg * 1000000 + (mm_use64(d) ? 64064 : 32064);
}

int adp_conv_mm(const adp_conv_desc& d, void* stream) {
  return mm_use64(d) ? adp_conv_mm_m64(d, stream) : adp_conv_mm_m32(d, stream);
}
