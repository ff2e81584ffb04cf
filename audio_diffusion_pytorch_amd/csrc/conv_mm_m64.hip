// conv_mm block tile 64 x 64 (output channels x positions); one translation unit per tile so the instantiations
// compile in parallel.  Kernel: conv_mm_impl.h, dispatch: conv_mm.hip.
#include "conv_mm_impl.h"

int adp_conv_mm_m64(const adp_conv_desc& d, void* stream) { return run_tile<64>(d, stream); }
