// How fast does one CU move 16 bytes per lane from the L2 -- as LDS-DMA (global_load_lds_dwordx4), as plain global_load_dwordx4 into
// registers, and as plain loads followed by ds_write_b128?  Every workgroup re-reads its own 64 KiB (L2-resident after the first pass),
// 8 loads per wave per iteration, counted vmcnt so that 8 stay in flight.  Prints shader-clock cycles per wave-instruction per CU.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/src/ta_rate.hip -o build/abl/ta_rate && build/abl/ta_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k(const char* src, unsigned long long* out, int iters, int rowbytes) {
    extern __shared__ char smem[];
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    // wave w, piece s: 8 "rows" of `rowbytes` stride, 128 contiguous bytes per row (the rc operand pattern), or fully linear if rowbytes == 128
    const char* base = src + (size_t)blockIdx.x * 65536;
    u32x4 keep[2][8];
    const char* g[8];
    int piece[8];
    for (int s = 0; s < 8; ++s) {
        piece[s] = (w * 8 + s) % 64;
        g[s] = base + (size_t)((piece[s] * 8 + (l >> 3)) % 512) * rowbytes % 65536 + (l & 7) * 16;
        keep[0][s] = keep[1][s] = u32x4{0, 0, 0, 0};
    }
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                if (MODE == 0)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g[s],
                                                     (__attribute__((address_space(3))) void*)(smem + piece[s] * 1024), 16, 0, 0);
                else
                    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(keep[half][s]) : "v"(g[s]) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // the previous batch of 8 has landed, this one stays in flight
            if (MODE == 1) {
#pragma unroll
                for (int s = 0; s < 8; ++s) asm volatile("" ::"v"(keep[half ^ 1][s]));
            } else if (MODE == 2) {
#pragma unroll
                for (int s = 0; s < 8; ++s) *reinterpret_cast<u32x4*>(smem + piece[s] * 1024 + l * 16) = keep[half ^ 1][s];
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

template <int MODE, int WAVES>
void run(const char* name, const char* src, unsigned long long* out, int rowbytes) {
    const int iters = 2000, grid = 256;
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    hipLaunchKernelGGL((k<MODE, WAVES>), dim3(grid), dim3(WAVES * 64), 65536, 0, src, out, 50, rowbytes);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((k<MODE, WAVES>), dim3(grid), dim3(WAVES * 64), 65536, 0, src, out, iters, rowbytes);
    hipEventRecord(b);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    std::vector<unsigned long long> h(grid);
    hipMemcpy(h.data(), out, grid * 8, hipMemcpyDeviceToHost);
    double cyc = 0;
    for (auto v : h) cyc += double(v);
    cyc /= grid;
    const double instr = double(iters) * 8 * WAVES;
    printf("%-44s waves %d rowbytes %5d: %7.1f counter ticks / wave-instruction / CU, %6.2f ns, %6.1f GB/s per CU, chip %5.2f TB/s\n", name, WAVES,
           rowbytes, cyc / instr, ms * 1e6 / instr, 1024.0 / (ms * 1e6 / instr), 1024.0 / (ms * 1e6 / instr) * 256 / 1000);
}

int main() {
    char* src;
    unsigned long long* out;
    hipMalloc(&src, 256 * 65536 + 65536);
    hipMemset(src, 1, 256 * 65536 + 65536);
    hipMalloc(&out, 4096);
    for (int rb : {128, 5120}) {
        run<0, 4>("LDS-DMA global_load_lds_dwordx4", src, out, rb);
        run<0, 8>("LDS-DMA global_load_lds_dwordx4", src, out, rb);
        run<1, 4>("global_load_dwordx4 -> VGPR", src, out, rb);
        run<1, 8>("global_load_dwordx4 -> VGPR", src, out, rb);
        run<2, 4>("global_load_dwordx4 -> VGPR -> ds_write_b128", src, out, rb);
        run<2, 8>("global_load_dwordx4 -> VGPR -> ds_write_b128", src, out, rb);
    }
    return 0;
}
