// Micro-benchmark: issue cost of dependent scalar / cross-lane instruction chains on gfx950 (diagnostics only).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)
#define REP256(x) REP64(x) REP64(x) REP64(x) REP64(x)

__global__ void k(uint64_t *out, int mode, int iters) {
    uint32_t s = __builtin_amdgcn_readfirstlane(iters);
    uint32_t v = threadIdx.x;
    uint64_t t0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; it++) {
        if (mode == 0) { // 256 dependent s_add
            asm volatile(REP256("s_add_u32 %0, %0, 1\n") : "+s"(s));
        } else if (mode == 1) { // readlane -> s_add -> (lane select) readlane
            asm volatile(REP256("v_readlane_b32 %0, %1, %0\n s_and_b32 %0, %0, 63\n") : "+s"(s) : "v"(v));
        } else if (mode == 2) { // writelane + readlane
            asm volatile(REP256("s_mov_b32 m0, %0\n v_writelane_b32 %1, %0, m0\n v_readlane_b32 %0, %1, %0\n s_and_b32 %0, %0, 63\n") : "+s"(s), "+v"(v));
        } else if (mode == 3) { // taken branches
            asm volatile(REP256("s_cmp_eq_u32 %0, %0\n s_cbranch_scc1 1f\n s_add_u32 %0, %0, 1\n 1:\n s_add_u32 %0, %0, 1\n") : "+s"(s)::"scc");
        } else if (mode == 4) { // dependent VALU
            asm volatile(REP256("v_add_u32 %0, %0, 1\n") : "+v"(v));
        } else if (mode == 5) { // valu -> readfirstlane -> salu -> valu
            asm volatile(REP256("v_readfirstlane_b32 %0, %1\n s_add_u32 %0, %0, 1\n v_add_u32 %1, %0, %1\n") : "+s"(s), "+v"(v));
        } else if (mode == 6) { // s_cselect chain with compare
            asm volatile(REP256("s_cmp_lt_u32 %0, 7\n s_cselect_b32 %0, %0, 3\n") : "+s"(s)::"scc");
        } else if (mode == 7) { // not-taken branches
            asm volatile(REP256("s_cmp_lg_u32 %0, %0\n s_cbranch_scc1 1f\n s_add_u32 %0, %0, 1\n 1:\n") : "+s"(s)::"scc");
        }
    }
    uint64_t t1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0) {
        out[blockIdx.x * 4 + 0] = t1 - t0;
        out[blockIdx.x * 4 + 1] = w1 - w0;
        out[blockIdx.x * 4 + 2] = s + v;
    }
}
int main() {
    uint64_t *d, h[4];
    hipMalloc(&d, 4096 * 32);
    const char *names[] = {"s_add chain (1 instr)", "readlane+s_and (2)", "m0,writelane,readlane,s_and (4)", "cmp+taken branch+add (3)", "v_add chain (1)", "readfirstlane,s_add,v_add (3)", "s_cmp+s_cselect (2)", "cmp+nottaken branch+add (3)"};
    for (int blocks : {1, 256 * 16}) {
        for (int mode = 0; mode < 8; mode++) {
            const int iters = 200;
            hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, d, mode, iters);
            hipDeviceSynchronize();
            hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, d, mode, iters);
            hipDeviceSynchronize();
            hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
            double units = 256.0 * iters;
            printf("blocks %5d  %-36s  %.2f shader-clocks/unit  %.2f ns/unit  (clock %.0f MHz)\n", blocks, names[mode], h[0] / units, h[1] * 10.0 / units,
                   h[0] / (h[1] * 10.0) * 1000.0);
        }
    }
    return 0;
}
