// f32_sincos.hip -- accuracy and issue cost of the hardware sine / cosine (v_sin_f32, v_cos_f32: argument in turns)
// on gfx950, against double-precision sin / cos of the same float argument, over [-0.5, 0.5) turns.
// Development aid: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void err_kernel(double *maxerr, int n_per_thread)
{
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)gridDim.x * blockDim.x * n_per_thread;
    double es = 0, ec = 0;
    for (int i = 0; i < n_per_thread; ++i) {
        const long long j = gid * n_per_thread + i;
        const float x = (float)((double)j / (double)total - 0.5);
        const float s = __builtin_amdgcn_sinf(x), c = __builtin_amdgcn_cosf(x);
        const double a = 6.283185307179586476925 * (double)x;
        es = fmax(es, fabs((double)s - sin(a)));
        ec = fmax(ec, fabs((double)c - cos(a)));
    }
    // one value per thread; reduced on the host
    maxerr[2 * gid] = es;
    maxerr[2 * gid + 1] = ec;
}

template <int KIND>
__global__ void __launch_bounds__(256) rate_kernel(float *out, float seed, int reps)
{
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = seed + 0.001f * (threadIdx.x + i);
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a[i]));
                if (KIND == 1) asm volatile("v_sin_f32 %0, %0" : "+v"(a[i]));
                if (KIND == 2) asm volatile("v_cos_f32 %0, %0" : "+v"(a[i]));
                if (KIND == 3) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND>
int rate(const char *name, float *out)
{
    const int reps = 1 << 16, blocks = 256 * 8;
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    rate_kernel<KIND><<<blocks, 256>>>(out, 0.1f, 1 << 10);
    CHK(hipEventRecord(e0));
    rate_kernel<KIND><<<blocks, 256>>>(out, 0.1f, reps);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    const double n = (double)blocks * 4 * reps * 16;
    printf("%-12s %.3e wave-instructions/s\n", name, n / (ms * 1e-3));
    return 0;
}

int main()
{
    const int blocks = 1024, threads = 256, per = 64;  // 2^24 arguments
    double *d, *h = new double[2 * blocks * threads];
    CHK(hipMalloc(&d, sizeof(double) * 2 * blocks * threads));
    err_kernel<<<blocks, threads>>>(d, per);
    CHK(hipMemcpy(h, d, sizeof(double) * 2 * blocks * threads, hipMemcpyDeviceToHost));
    double es = 0, ec = 0;
    for (int i = 0; i < blocks * threads; ++i) {
        es = fmax(es, h[2 * i]);
        ec = fmax(ec, h[2 * i + 1]);
    }
    printf("max |v_sin_f32 - sin| = %.3e   max |v_cos_f32 - cos| = %.3e   over 2^24 arguments in [-0.5, 0.5) turns\n", es, ec);
    float *out;
    CHK(hipMalloc(&out, sizeof(float) * 256 * 8 * 256));
    if (rate<0>("v_fma_f32", out) || rate<1>("v_sin_f32", out) || rate<2>("v_cos_f32", out) || rate<3>("v_rcp_f32", out)) return 1;
    return 0;
}
