// K4's horizontal 6-tap on random window rows, GPU vs host (bring-up aid).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
__device__ __forceinline__ uint32_t align8(uint32_t hi, uint32_t lo, uint32_t sh) { return __builtin_amdgcn_alignbyte(hi, lo, sh); }
__global__ void k(const uint32_t *in, int *out) {
    const int t = threadIdx.x + blockIdx.x * 64;
    const uint32_t a = in[3 * t] ^ 0x80808080u, b = in[3 * t + 1] ^ 0x80808080u, c = in[3 * t + 2] ^ 0x80808080u;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t A = i ? align8(b, a, i) : a, Bq = i ? align8(c, b, i) : b;
        out[4 * t + i] = __builtin_amdgcn_sdot4(static_cast<int>(A), 0x1414FB01, __builtin_amdgcn_sdot4(static_cast<int>(Bq), 0x000001FB, 4096, false), false);
    }
}
int main() {
    const int N = 4096;
    uint32_t *h = (uint32_t *)malloc(N * 12), *d_in;
    int *o = (int *)malloc(N * 16), *d_out;
    srand(1);
    for (int i = 0; i < 3 * N; i++) h[i] = (uint32_t)rand() ^ ((uint32_t)rand() << 16);
    (void)hipMalloc(&d_in, N * 12), (void)hipMalloc(&d_out, N * 16);
    (void)hipMemcpy(d_in, h, N * 12, hipMemcpyHostToDevice);
    k<<<N / 64, 64>>>(d_in, d_out);
    (void)hipMemcpy(o, d_out, N * 16, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < N; t++) {
        const uint8_t *w = (const uint8_t *)(h + 3 * t);
        for (int i = 0; i < 4; i++) {
            int want = w[i] - 5 * w[i + 1] + 20 * w[i + 2] + 20 * w[i + 3] - 5 * w[i + 4] + w[i + 5];
            if (want != o[4 * t + i] && bad++ < 8) printf("t %d i %d want %d got %d (diff %d) bytes %d %d %d %d %d %d\n", t, i, want, o[4 * t + i], o[4 * t + i] - want, w[i], w[i + 1], w[i + 2], w[i + 3], w[i + 4], w[i + 5]);
        }
    }
    printf("bad %d of %d\n", bad, 4 * N);
    return 0;
}
