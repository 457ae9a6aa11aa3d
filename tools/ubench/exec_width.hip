// Does a wave64 VALU instruction cost less issue time when only some quarter-waves have active lanes?  (The entropy kernels run
// wave-uniform arithmetic on the VALU: if 16 active lanes were cheaper than 64, masking would buy throughput.)
// Waves per SIMD x active lanes -> VALU wave-instructions per cycle per SIMD.  hipcc -O3 --offload-arch=gfx950 exec_width.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__global__ void __launch_bounds__(64) chain(uint32_t *out, int iters, int active) {
    uint32_t a = threadIdx.x, b = blockIdx.x + 1, c = 3, d = 7;
    if (static_cast<int>(threadIdx.x) < active) { // EXEC = the first `active` lanes for the whole loop
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int k = 0; k < 16; k++) { // four independent chains: issue-bound, not latency-bound
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));
                asm volatile("v_xor_b32 %0, %0, %1" : "+v"(b) : "v"(c));
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(c) : "v"(d));
                asm volatile("v_xor_b32 %0, %0, %1" : "+v"(d) : "v"(a));
            }
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = a ^ b ^ c ^ d;
}
int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    uint32_t *d;
    hipMalloc(&d, sizeof(uint32_t) * 64 * cus * 64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    const int iters = 20000;
    printf("CUs %d, clock %.0f MHz\n", cus, p.clockRate / 1e3);
    for (int wps : {1, 2, 4, 8})
        for (int active : {64, 32, 16, 1}) {
            const int blocks = cus * 4 * wps; // one 64-thread block per wave slot
            chain<<<blocks, 64>>>(d, 100, active);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            chain<<<blocks, 64>>>(d, iters, active);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            const double insts = static_cast<double>(iters) * 64 * wps; // VALU wave-instructions per SIMD
            printf("waves/SIMD %d active lanes %2d: %.3f ms -> %.3f VALU instr per cycle per SIMD (at %.0f MHz)\n", wps, active, ms, insts / (ms * 1e-3 * p.clockRate * 1e3),
                   p.clockRate / 1e3);
        }
    return 0;
}
