// placeholder until the flash-style attention kernels land (see include/adp.h)
#include "adp_rt.h"
#include "adp.h"
extern "C" int adp_attn_fwd(const float*, const float*, const float*, int64_t, int64_t, int64_t, int64_t, int64_t,
                            int64_t, int64_t, float*, float*, void*) { return ADP_ERR_UNSUPPORTED; }
extern "C" int64_t adp_attn_bwd_ws_bytes(int64_t, int64_t, int64_t, int64_t, int64_t) { return ADP_ERR_UNSUPPORTED; }
extern "C" int adp_attn_bwd(const float*, const float*, const float*, const float*, const float*, const float*,
                            int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, float*, float*, float*,
                            float*, void*) { return ADP_ERR_UNSUPPORTED; }
