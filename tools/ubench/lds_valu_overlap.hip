// lds_valu_overlap.hip -- do LDS instructions take issue time from the vector ALU of their SIMD?
// Every wave runs REPS x (16 independent v_fma_f64 + NL LDS instructions of one kind); the table shows
// shader cycles per iteration as seen by one wave at 1, 2 and 3 waves per SIMD.  If the LDS instructions
// were free for the vector ALU, the time at 3 waves per SIMD would stay that of the 16 fma alone.
// Development aid: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int REPS = 2048;
typedef double v2d __attribute__((ext_vector_type(2)));

// KIND 0: fma only; 1: + 2 ds_write_b128; 2: + 2 ds_read_b128; 3: + 1 write + 1 read b128; 4: + 4 ds_write_b64; 5: + 4 ds_read_b64;
// 6: + 2 global_load_dwordx4 (same line every time: cache hits); 7: + 2 global_store_dwordx4
template <int KIND>
__global__ void __launch_bounds__(256) bench(double *out, unsigned long long *cyc, double seed, const v2d *g, v2d *gs)
{
    extern __shared__ v2d lds[];
    double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const double m = 1.0000001, c = 1e-9;
    v2d w0 = {a0, a1}, w1 = {a2, a3}, r0 = w0, r1 = w1;
    v2d *p = lds + threadIdx.x;  // 16 bytes per lane, contiguous: no bank conflicts
    const v2d *gp = g + threadIdx.x;
    v2d *gsp = gs + (size_t)blockIdx.x * 256 + threadIdx.x;
    p[0] = w0;
    p[256] = w1;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < REPS; ++r) {
#define F(x) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(m), "v"(c));
        F(a0) F(a1) F(a2) F(a3) F(a4) F(a5) F(a6) F(a7)
        if (KIND == 1 || KIND == 3) asm volatile("ds_write_b128 %0, %1" : : "v"((unsigned)(size_t)p), "v"(w0) : "memory");
        if (KIND == 1) asm volatile("ds_write_b128 %0, %1 offset:4096" : : "v"((unsigned)(size_t)p), "v"(w1) : "memory");
        if (KIND == 2 || KIND == 3) asm volatile("ds_read_b128 %0, %1" : "=v"(r0) : "v"((unsigned)(size_t)p) : "memory");
        if (KIND == 2) asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(r1) : "v"((unsigned)(size_t)p) : "memory");
        if (KIND == 4) asm volatile("ds_write_b64 %0, %1\n ds_write_b64 %0, %2 offset:8\n ds_write_b64 %0, %3 offset:4096\n ds_write_b64 %0, %4 offset:4104"
                                    : : "v"((unsigned)(size_t)p), "v"(w0.x), "v"(w0.y), "v"(w1.x), "v"(w1.y) : "memory");
        if (KIND == 5) asm volatile("ds_read_b64 %0, %4\n ds_read_b64 %1, %4 offset:8\n ds_read_b64 %2, %4 offset:4096\n ds_read_b64 %3, %4 offset:4104"
                                    : "=v"(r0.x), "=v"(r0.y), "=v"(r1.x), "=v"(r1.y) : "v"((unsigned)(size_t)p) : "memory");
        if (KIND == 6) asm volatile("global_load_dwordx4 %0, %2, off\n global_load_dwordx4 %1, %2, off offset:2048" : "=v"(r0), "=v"(r1) : "v"(gp) : "memory");
        if (KIND == 7) asm volatile("global_store_dwordx4 %0, %1, off\n global_store_dwordx4 %0, %2, off offset:2048" : : "v"(gsp), "v"(w0), "v"(w1) : "memory");
        F(a0) F(a1) F(a2) F(a3) F(a4) F(a5) F(a6) F(a7)
#undef F
        if (KIND != 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const int gi = blockIdx.x * blockDim.x + threadIdx.x;
    out[gi] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + r0.x + r0.y + r1.x + r1.y;
    if ((threadIdx.x & 63) == 0) cyc[gi >> 6] = t1 - t0;
}

template <int KIND>
int run(const char *name, double *out, unsigned long long *cyc, const v2d *g, v2d *gs)
{
    printf("%-44s", name);
    for (int wps = 1; wps <= 3; ++wps) {
        const int grid = 256 * wps;
        hipLaunchKernelGGL(bench<KIND>, dim3(grid), dim3(256), 8192 + 4096, 0, out, cyc, 1.0, g, gs);
        CHK(hipDeviceSynchronize());
        hipLaunchKernelGGL(bench<KIND>, dim3(grid), dim3(256), 8192 + 4096, 0, out, cyc, 1.0, g, gs);
        CHK(hipDeviceSynchronize());
        std::vector<unsigned long long> h(grid * 4);
        CHK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
        std::sort(h.begin(), h.end());
        const double per_iter = (double)h[h.size() / 2] / REPS;
        printf("  %d w/SIMD: %7.1f cyc/iter = %6.1f per SIMD", wps, per_iter, per_iter / wps);
    }
    printf("\n");
    return 0;
}

int main()
{
    double *out;
    unsigned long long *cyc;
    v2d *g, *gs;
    CHK(hipMalloc(&out, 768 * 256 * 8));
    CHK(hipMalloc(&cyc, 768 * 4 * 8));
    CHK(hipMalloc(&g, 1 << 20));
    CHK(hipMalloc(&gs, (size_t)768 * 256 * 16 + 8192));
    CHK(hipMemset(g, 0, 1 << 20));
    run<0>("16 v_fma_f64", out, cyc, g, gs);
    run<1>("16 v_fma_f64 + 2 ds_write_b128", out, cyc, g, gs);
    run<2>("16 v_fma_f64 + 2 ds_read_b128", out, cyc, g, gs);
    run<3>("16 v_fma_f64 + 1 write + 1 read b128", out, cyc, g, gs);
    run<4>("16 v_fma_f64 + 4 ds_write_b64", out, cyc, g, gs);
    run<5>("16 v_fma_f64 + 4 ds_read_b64", out, cyc, g, gs);
    run<6>("16 v_fma_f64 + 2 global_load_dwordx4 (hits)", out, cyc, g, gs);
    run<7>("16 v_fma_f64 + 2 global_store_dwordx4", out, cyc, g, gs);
    return 0;
}
