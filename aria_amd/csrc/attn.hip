// placeholder until the flash-attention kernels land (next commit)
#include "aria_device.h"
#include "aria_hip.h"
extern "C" {
int aria_attn_fwd(const void*, const void*, const void*, void*, float*, const int32_t*, int64_t, int64_t, int64_t, int64_t, int64_t,
                  int64_t, int64_t, int64_t, float, int, void*) { return ARIA_ERR_UNSUPPORTED; }
int aria_attn_bwd(const void*, const void*, const void*, const void*, const void*, const float*, float*, void*, void*, void*,
                  const int32_t*, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t,
                  float, int, void*) { return ARIA_ERR_UNSUPPORTED; }
}
