// Shared host-side helpers of libaria_hip.so.
#include "aria_device.h"
#include "aria_hip.h"

int aria_check_launch() {
#ifdef ARIA_EMU
    return ARIA_OK;
#else
    return hipGetLastError() == hipSuccess ? ARIA_OK : ARIA_ERR_LAUNCH;
#endif
}

extern "C" int aria_abi_version(void) { return ARIA_ABI_VERSION; }
