// Private to libadp_hip.so: kernel families behind adp_conv1d / adp_conv1d_wgrad (not part of the C-ABI).
#pragma once
#include <stdlib.h>
#include "adp.h"

// conv_mm.hip (+ conv_mm_impl.h, conv_mm_m64/m32.hip): wave-specialised implicit-GEMM conv (stride 1 kernel 1/3,
// kernel = stride 2/4, nearest-upsample loader; channels % 32 == 0)
bool adp_conv_mm_eligible(const adp_conv_desc& d);
int adp_conv_mm(const adp_conv_desc& d, void* stream);
int64_t adp_conv_mm_gnb_entries(const adp_conv_desc& d);  // slices per row of gnb_ab (set d.ws before asking)
int64_t adp_conv_mm_tile(const adp_conv_desc& d);  // NKG * 1000000 + BM * 1000 + BN
int64_t adp_conv_mm_ksplit(const adp_conv_desc& d);  // cross-workgroup K split the dispatcher picks (1 = none)
bool adp_conv_mm_winograd(const adp_conv_desc& d);   // this conv runs conv_mm's Winograd F(2,3) variant (WN)
int adp_conv_mm_nsp(const adp_conv_desc& d);         // 64-position tiles per block of that variant (1, 2 or 4)
bool adp_winograd_enabled();                         // ADP_CONV_WINO switch (shared with the weight gradients)

int adp_conv_splitk_reduce(const adp_conv_desc& d, int64_t ks, void* stream);  // sum of d.ws partial tiles + epilogue
int64_t adp_conv_splitk_gn_entries(const adp_conv_desc& d);  // gn_part slices per row the reduce kernel writes

// wgrad_mm.hip: wave-specialised weight gradient of the same convolutions (channels % 32 == 0)
bool adp_wgrad_mm_eligible(const adp_wgrad_desc& d);
int64_t adp_wgrad_mm_ws_floats(const adp_wgrad_desc& d);
int64_t adp_wgrad_mm_nsplit(const adp_wgrad_desc& d);
int adp_wgrad_mm_n(const adp_wgrad_desc* ds, int n, void* stream);
constexpr int ADP_WGR_BATCH = 8;
int adp_wgrad_reduce_n(const float* const* ws, float* const* dw, float* const* dbias, int n, int64_t nsplit, int64_t cnt,
                       int64_t M, int accumulate, void* stream);
int adp_wgrad_mm(const adp_wgrad_desc& d, void* stream);
int adp_wgrad_reduce(const float* ws, int64_t nsplit, int64_t cnt, int64_t M, float* dw, float* dbias, int accumulate,
                     void* stream);

// conv_mm4.hip: Winograd F(4,3) variant of the same block for the wide kernel-3 'same' convs without prologue (>= 64 channels,
// >= 200 blocks of 32 rows x 128 positions): MMA waves split the six Winograd planes and the chunk's channels
bool adp_conv_mm4_eligible(const adp_conv_desc& d);
int adp_conv_mm4(const adp_conv_desc& d, void* stream);
// ADP_GNB_FAMILIES (A/B, bit mask; default all): which kernel families leave the GroupNorm-backward sums -- 1 conv_mm4 12-wave block,
// 2 conv_mm4 light block, 4 conv_tilek, 8 conv_mm, 16 conv_tile32, 32 split-K reduce
inline bool adp_gnb_family_on(int bit) {
  const char* e = getenv("ADP_GNB_FAMILIES");
  return e == nullptr || (atoi(e) & bit) != 0;
}
int64_t adp_conv_mm4_gnb_entries(const adp_conv_desc& d);  // slices per row of gnb_ab (0: K-split launch)
int64_t adp_conv_mm4_gn_entries(const adp_conv_desc& d);  // two entries (row pairs) per row quad and 128-position tile
int64_t adp_conv_mm4_ksplit(const adp_conv_desc& d);      // cross-workgroup K split (1 = none)

// conv_tile.hip: barrier-free wave-tile kernel (wave-private LDS tile, Winograd F(4,3)) for the HBM-bound 32 -> 32 channel
// kernel-3 ConvBlock convs and their data gradients (depth 1)
bool adp_conv_tile_eligible(const adp_conv_desc& d);
int adp_conv_tile(const adp_conv_desc& d, void* stream);
int64_t adp_conv_tile_gnb_entries(const adp_conv_desc& d);  // slices per row of gnb_ab (0: not a plain data gradient)
int64_t adp_conv_tile_gn_entries(const adp_conv_desc& d);  // GroupNorm partial slices per output row quad (gn_part)

// conv_tilek.hip: the same wave tile for the deep layers whose tiles alone cannot fill the chip (>= 512 channels, <= 320 tiles):
// eight waves of a workgroup split the input channels, 16- or 32-row tiles, no cross-workgroup K split / reduce launch
bool adp_conv_tilek_eligible(const adp_conv_desc& d);
int adp_conv_tilek(const adp_conv_desc& d, void* stream);
// conv_tilek1.hip: the 1x1 sibling (attention projections at batch 1: K split inside the workgroup instead of across workgroups)
bool adp_conv_tilek1_eligible(const adp_conv_desc& d);
int adp_conv_tilek1(const adp_conv_desc& d, void* stream);
int64_t adp_conv_tilek1_gn_entries(const adp_conv_desc& d);  // one GroupNorm partial entry per row quad and 64-position tile
int64_t adp_conv_tilek_gnb_entries(const adp_conv_desc& d);  // slices per row of gnb_ab
int64_t adp_conv_tilek_gn_entries(const adp_conv_desc& d);  // one GroupNorm partial entry per row quad and 64-position tile

// conv_direct.hip: VALU direct convolution for the narrow (2-8 channel) ends of the U-Net
bool adp_conv_direct_eligible(const adp_conv_desc& d);
int adp_conv_direct(const adp_conv_desc& d, void* stream);

// wgrad_direct.hip: VALU streaming weight gradient for the narrow layers (M * R <= 256)
bool adp_wgrad_direct_eligible(const adp_wgrad_desc& d);
int64_t adp_wgrad_direct_ws_floats(const adp_wgrad_desc& d);
int adp_wgrad_direct(const adp_wgrad_desc& d, void* stream);
