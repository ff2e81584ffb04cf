// conv_mm block tile 32 x 64 (output channels x positions); one translation unit per tile so the instantiations
// compile in parallel.  Kernel: conv_mm_impl.h, dispatch: conv_mm.hip.
#include "conv_mm_impl.h"

int adp_conv_mm_m32(const adp_conv_desc& d, void* stream) { return run_tile<32>(d, stream); }
