// tools/hoststub/hip/hip_runtime.h -- TEST TOOLING, never part of the product build.
// A "null device" stand-in for the handful of HIP runtime calls the host side of libh264mi makes, so that mi_api.cpp
// + mi_parse.cpp compile with g++ under AddressSanitizer / UBSan and their host logic (NAL / header parsing, picture
// boundaries, DPB and reference lists, slice-group maps, batching, staging-buffer layout, error paths) can be driven
// with damaged streams on a machine without a GPU (tools/host_asan.sh).  Device memory is host memory, copies are
// memcpy, kernels are NOT run (every status word stays 0 = "no error", the pixels stay whatever the allocation held),
// so this says nothing about the kernels: it checks that the host never reads or writes out of bounds while preparing them.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unordered_map>

#define __global__
#define __device__
#define __host__
#define __launch_bounds__(...)

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorInvalidValue = 1 };
typedef struct hipstub_stream *hipStream_t;
typedef struct hipstub_event *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocDefault = 0 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
struct hipDeviceProp_t { int multiProcessorCount; size_t totalGlobalMem; char gcnArchName[64]; };
struct dim3 { uint32_t x, y, z; dim3(uint32_t a = 1, uint32_t b = 1, uint32_t c = 1) : x(a), y(b), z(c) {} };

namespace hipstub { template <class... A> inline void sink(A &&...) {} }

static inline const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "ok" : e == hipErrorOutOfMemory ? "out of memory (stub)" : "error (stub)"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
static inline hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) { memset(p, 0, sizeof *p); p->multiProcessorCount = 256; p->totalGlobalMem = size_t(288) << 30; strcpy(p->gcnArchName, "gfx950"); return hipSuccess; }
// an allocation limit, so that a configuration meant for 288 GB fails cleanly instead of taking the host down
static inline size_t &hipstub_budget() { static size_t b = size_t(8) << 30; return b; }
static inline std::unordered_map<void *, size_t> &hipstub_sizes() { static std::unordered_map<void *, size_t> m; return m; }
static inline std::mutex &hipstub_lock() { static std::mutex m; return m; }
static inline hipError_t hipMalloc(void **p, size_t n) {
    std::lock_guard<std::mutex> g(hipstub_lock());
    if (n > hipstub_budget()) { *p = nullptr; return hipErrorOutOfMemory; }
    *p = calloc(n ? n : 1, 1);  // zeroed: the status words no kernel writes read as "no error"
    if (!*p) return hipErrorOutOfMemory;
    hipstub_budget() -= n;
    hipstub_sizes()[*p] = n;
    return hipSuccess;
}
template <class T> static inline hipError_t hipMalloc(T **p, size_t n) { return hipMalloc(reinterpret_cast<void **>(p), n); }
static inline hipError_t hipFree(void *p) {
    if (p) {
        std::lock_guard<std::mutex> g(hipstub_lock());
        auto it = hipstub_sizes().find(p);
        if (it != hipstub_sizes().end()) { hipstub_budget() += it->second; hipstub_sizes().erase(it); }
    }
    free(p);
    return hipSuccess;
}
static inline hipError_t hipHostMalloc(void **p, size_t n, unsigned = 0) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
template <class T> static inline hipError_t hipHostMalloc(T **p, size_t n, unsigned f = 0) { return hipHostMalloc(reinterpret_cast<void **>(p), n, f); }
static inline hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpy2D(void *d, size_t dpitch, const void *s, size_t spitch, size_t width, size_t height, hipMemcpyKind) {
    for (size_t r = 0; r < height; r++) memmove(static_cast<char *>(d) + r * dpitch, static_cast<const char *>(s) + r * spitch, width);
    return hipSuccess;
}
static inline hipError_t hipMemset(void *d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t = nullptr) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemset2DAsync(void *d, size_t pitch, int v, size_t width, size_t height, hipStream_t = nullptr) {
    for (size_t r = 0; r < height; r++) memset(static_cast<char *>(d) + r * pitch, v, width);
    return hipSuccess;
}
static inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = reinterpret_cast<hipStream_t>(malloc(1)); return hipSuccess; }
static inline hipError_t hipExtStreamCreateWithCUMask(hipStream_t *s, uint32_t, const uint32_t *) { *s = reinterpret_cast<hipStream_t>(malloc(1)); return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned = 0) { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t *e) { *e = reinterpret_cast<hipEvent_t>(malloc(1)); return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
static inline hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t = nullptr) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
#define hipFuncSetAttribute(fn, attr, val) (hipSuccess)
#define hipLaunchKernelGGL(kernel, ...) hipstub::sink(__VA_ARGS__)
