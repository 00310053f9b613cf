// hip_emu.h -- a minimal HOST emulation of the HIP constructs the vocoder
// kernels use, so that the very same kernel source can be compiled with g++
// and stepped through with gdb / run under the CPU test-suite.
//
// DEVELOPMENT HARNESS ONLY.  The product library (libvocoder_hip.so) is built
// by hipcc for gfx950 and never includes this file; nothing under vocoder_amd/
// loads the emulated build.  One std::thread per HIP thread, one std::barrier
// per workgroup, workgroups run one after another.
#pragma once
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#include <barrier>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>

#define VOC_EMU 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)

struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

struct alignas(16) double2 {
    double x, y;
};
static inline double2 make_double2(double x, double y) { return double2{x, y}; }
struct alignas(8) float2 {
    float x, y;
};
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
// device code resolves fma(float, float, float) to the single-precision instruction
static inline float fma(float a, float b, float c) { return std::fmaf(a, b, c); }

namespace emu {
struct BlockCtx {
    std::barrier<> *bar;
    unsigned char *smem;
};
inline thread_local BlockCtx tls{nullptr, nullptr};
}  // namespace emu

inline thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;

static inline void __syncthreads() { emu::tls.bar->arrive_and_wait(); }

typedef int hipError_t;
typedef void *hipStream_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorInvalidValue = 1 };
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };

static inline const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "success" : "emulated HIP error"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
static inline hipError_t hipMalloc(void **p, size_t n)
{
    *p = std::aligned_alloc(256, (n + 255) / 256 * 256 + 256);
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
static inline hipError_t hipFree(void *p) { std::free(p); return hipSuccess; }
static inline hipError_t hipHostMalloc(void **p, size_t n, unsigned) { return hipMalloc(p, n); }
static inline hipError_t hipHostFree(void *p) { std::free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { std::memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { std::memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpy2DAsync(void *d, size_t dp, const void *s, size_t sp, size_t w, size_t h, hipMemcpyKind, hipStream_t)
{
    for (size_t r = 0; r < h; ++r) std::memmove((char *)d + r * dp, (const char *)s + r * sp, w);
    return hipSuccess;
}
static inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { std::memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemset(void *d, int v, size_t n) { std::memset(d, v, n); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
// events: everything runs synchronously here, so they order nothing
typedef void *hipEvent_t;
enum { hipEventDisableTiming = 2 };
static inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = (void *)1; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
// a "device" array is an ordinary host array here: its address is its name
#define HIP_SYMBOL(x) (&(x))
template <class T>
static inline hipError_t hipGetSymbolAddress(void **p, T *sym) { *p = (void *)sym; return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }

template <class Kern, class... Args>
static inline void hipLaunchKernelGGL(Kern kernel, dim3 grid, dim3 block, size_t shmem, hipStream_t, Args... args)
{
    const unsigned nthr = block.x * block.y * block.z;
    std::unique_ptr<unsigned char, void (*)(void *)> smem(
        (unsigned char *)std::aligned_alloc(256, (shmem + 255) / 256 * 256 + 256), std::free);
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                std::barrier<> bar(nthr);
                std::vector<std::thread> th;
                th.reserve(nthr);
                for (unsigned t = 0; t < nthr; ++t)
                    th.emplace_back([&, t]() {
                        threadIdx = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
                        blockIdx = dim3(bx, by, bz);
                        blockDim = block;
                        gridDim = grid;
                        emu::tls = emu::BlockCtx{&bar, smem.get()};
                        kernel(args...);
                    });
                for (auto &t : th) t.join();
            }
}
