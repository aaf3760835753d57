// Hardware-semantics probes (used by tests/test_gpu_probes.py to re-verify on a real MI355X the lane layouts
// that tests/emu/hip_emu.h assumes).  Not on the hot path.
#include "aria_device.h"
#include "aria_hip.h"

#ifndef ARIA_EMU
namespace {
typedef short s16x4v __attribute__((ext_vector_type(4)));
// LDS holds lds[i] = i (u16).  mode 0: lane address = lane*8 bytes; mode 1: lane address = (lane&15)*128 + (lane>>4)*8 bytes.
__global__ void probe_tr16_kernel(unsigned short* out, int mode) {
    __shared__ unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int l = threadIdx.x;
    const int off = mode == 0 ? l * 4 : ((l & 15) * 64 + (l >> 4) * 4);
    s16x4v v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4v*)(lds + off));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)v[j];
}
}  // namespace
#endif

extern "C" int aria_probe_tr16(void* out, int mode, void* stream) {
#ifdef ARIA_EMU
    (void)out; (void)mode; (void)stream;
    return ARIA_ERR_UNSUPPORTED;
#else
    hipLaunchKernelGGL(probe_tr16_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), static_cast<unsigned short*>(out), mode);
    return aria_check_launch();
#endif
}
