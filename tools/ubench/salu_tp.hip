// Micro-benchmark: aggregate issue throughput per CU of scalar / vector / mixed chains vs resident waves (diagnostics only).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)
#define REP256(x) REP64(x) REP64(x) REP64(x) REP64(x)
__global__ void k(uint64_t *out, int mode, int iters) {
    uint32_t s = __builtin_amdgcn_readfirstlane(iters), s2 = s + 1;
    uint32_t v = threadIdx.x, v2 = v + 1;
    uint64_t w0 = wall_clock64();
    for (int it = 0; it < iters; it++) {
        if (mode == 0) asm volatile(REP256("s_add_u32 %0, %0, 1\n") : "+s"(s));
        else if (mode == 1) asm volatile(REP256("v_add_u32 %0, %0, 1\n") : "+v"(v));
        else if (mode == 2) asm volatile(REP256("s_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 1\n") : "+s"(s), "+v"(v));
        else if (mode == 3) asm volatile(REP256("s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n") : "+s"(s), "+s"(s2)); // 2 independent chains
        else if (mode == 4) asm volatile(REP256("v_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 1\n") : "+v"(v), "+v"(v2));
    }
    uint64_t w1 = wall_clock64();
    if (threadIdx.x == 0) out[blockIdx.x * 2] = w0, out[blockIdx.x * 2 + 1] = w1 + (s + s2 + v + v2 == 12345 ? 1 : 0);
}
int main() {
    uint64_t *d;
    hipMalloc(&d, 16384 * 16);
    std::vector<uint64_t> h(16384 * 2);
    const char *names[] = {"s_add chain", "v_add chain", "s_add+v_add (2 instr)", "2 indep s_add (2 instr)", "2 indep v_add (2 instr)"};
    for (int mode = 0; mode < 5; mode++)
        for (int wpc : {1, 4, 8, 16, 32}) {
            const int blocks = 256 * wpc, iters = 100;
            for (int rep = 0; rep < 2; rep++) {
                hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, d, mode, iters);
                hipDeviceSynchronize();
            }
            hipMemcpy(h.data(), d, blocks * 16, hipMemcpyDeviceToHost);
            uint64_t t0 = ~0ull, t1 = 0;
            double sum = 0;
            for (int b = 0; b < blocks; b++) t0 = std::min(t0, h[2 * b]), t1 = std::max(t1, h[2 * b + 1]), sum += (h[2 * b + 1] - h[2 * b]);
            const double instr_per_wave = 256.0 * iters * (mode >= 2 ? 2 : 1);
            const double span_ns = (t1 - t0) * 10.0, avg_ns = sum / blocks * 10.0;
            printf("%-26s waves/CU %2d: per-wave %.2f ns/instr; aggregate %.2f instr/ns/CU (= %.2f per cycle @2.39GHz)\n", names[mode], wpc, avg_ns / instr_per_wave,
                   instr_per_wave * wpc / span_ns, instr_per_wave * wpc / span_ns / 2.39);
        }
    return 0;
}
