// tools/ubench/fetch_calib.hip -- TEST TOOLING: what rocprofv3's FETCH_SIZE counts for the access patterns of this repository's kernels.
// MI355X_MICROARCH.md: "On gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read ... Other access widths are
// uncalibrated: calibrate on a known byte count in your own access pattern".  Three kernels read the same buffer exactly once:
//   k_stream : lane i reads 16 bytes at 16 i, the wavefront a contiguous 1 KB (the calibrated case: the counter should show half the bytes);
//   k_rows64 : K5's pattern -- every lane owns one row of a picture (pitch 1920) and reads 64 contiguous bytes of it (4 x dwordx4), then the
//              next 64 bytes, ...: a wavefront touches 64 different lines per load instruction, each lane half a 128-byte line per visit;
//   k_rows16 : the same rows, 16 bytes per visit (one dwordx4).
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/ubench/fetch_calib.hip -o /tmp/fetch_calib &&
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/calib -- /tmp/fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
#define PITCH 1920
#define ROWS_PER_PIC 1088
__global__ void k_stream(const v4u *src, size_t n16, uint32_t *sink) {
    uint32_t acc = 0;
    for (size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x; i < n16; i += static_cast<size_t>(gridDim.x) * blockDim.x) {
        const v4u v = src[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
// one wavefront per 64 rows of a picture; grid = (pictures, 17 groups of 64 rows), block = 64; VISIT = bytes per lane per visit (64 or 16)
template <int VISIT>
__global__ void k_rows(const uint8_t *src, uint32_t *sink) {
    const int lane = threadIdx.x & 63;
    const size_t pic = blockIdx.x, row = static_cast<size_t>(blockIdx.y) * 64 + lane;
    uint32_t acc = 0;
    if (row < ROWS_PER_PIC) {
        const uint8_t *p = src + (pic * ROWS_PER_PIC + row) * PITCH;
        for (int x = 0; x < PITCH; x += VISIT) {
#pragma unroll
            for (int k = 0; k < VISIT; k += 16) {
                const v4u v = *reinterpret_cast<const v4u *>(p + x + k);
                acc += v.x ^ v.y ^ v.z ^ v.w;
            }
            __builtin_amdgcn_s_sleep(8); // (spread the visits of a line in time a little, as the filter steps do)
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
int main() {
    const size_t pics = 256, bytes = pics * ROWS_PER_PIC * PITCH; // 535 MB: larger than the Infinity Cache
    uint8_t *buf;
    uint32_t *sink;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) return 1;
    hipMemset(buf, 1, bytes);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(k_stream, dim3(4096), dim3(256), 0, 0, reinterpret_cast<const v4u *>(buf), bytes / 16, sink);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(k_rows<64>, dim3(pics, 17), dim3(64), 0, 0, buf, sink);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(k_rows<16>, dim3(pics, 17), dim3(64), 0, 0, buf, sink);
    hipDeviceSynchronize();
    printf("bytes read by each kernel: %zu\n", bytes);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
