// fp64_sustained.hip -- what the whole chip sustains on fp64 vector instructions over tens of milliseconds
// (long enough for the power management to settle): every wave runs REPS x 16 independent v_fma_f64; the
// table shows wave-instructions per second, the clock that rate corresponds to at 4 cycles per instruction and
// SIMD, and TFLOP/s, at 1, 2 and 3 waves per SIMD; then the same with half of the instructions v_add_f64 / v_mul_f64.
// Development aid: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

#define BODY8(INS) INS(a0) INS(a1) INS(a2) INS(a3) INS(a4) INS(a5) INS(a6) INS(a7) INS(a0) INS(a1) INS(a2) INS(a3) INS(a4) INS(a5) INS(a6) INS(a7)

template <int KIND>
__global__ void __launch_bounds__(256) bench(double *out, int reps, double seed)
{
    double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    double m = 1.0000001, c = 1e-9;
    for (int r = 0; r < reps; ++r) {
        if (KIND == 0) {
#define INS(x) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(m), "v"(c));
            BODY8(INS)
#undef INS
        } else if (KIND == 1) {
#define INS(x) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x) : "v"(c));
            BODY8(INS)
#undef INS
        } else if (KIND == 2) {
#define INS(x) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(m));
            BODY8(INS)
#undef INS
        } else {  // 32-bit integer adds: the same issue slots, little arithmetic power
            unsigned i0 = (unsigned)a0;
#define INS(x) asm volatile("v_add_u32 %0, %0, %1" : "+v"(i0) : "v"(r));
            BODY8(INS)
#undef INS
            a0 = (double)i0;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int KIND>
int run(const char *name, double *out, int reps)
{
    printf("%-14s", name);
    for (int wps = 1; wps <= 3; ++wps) {
        const int grid = 256 * wps;
        hipEvent_t e0, e1;
        CHK(hipEventCreate(&e0));
        CHK(hipEventCreate(&e1));
        hipLaunchKernelGGL(bench<KIND>, dim3(grid), dim3(256), 0, 0, out, reps, 1.0);
        CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL(bench<KIND>, dim3(grid), dim3(256), 0, 0, out, reps, 1.0);
        CHK(hipEventRecord(e1));
        CHK(hipDeviceSynchronize());
        float ms = 0;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        const double instr = (double)grid * 4 * reps * 16.0;  // wave-instructions
        const double rate = instr / (ms * 1e-3);
        printf("  %dw: %6.1f ms %.3e wi/s = %.3f GHz at 4 cyc, %.1f TFLOP/s(fma)", wps, ms, rate, rate * 4 / 1024 / 1e9, rate * 128 / 1e12);
    }
    printf("\n");
    return 0;
}

int main(int argc, char **argv)
{
    double *out;
    CHK(hipMalloc(&out, 768 * 256 * 8));
    const int reps = argc > 1 ? atoi(argv[1]) : 400000;  // 6.4 M instructions per wave: ~15 ms per wave per SIMD
    run<0>("v_fma_f64", out, reps);
    run<1>("v_add_f64", out, reps);
    run<2>("v_mul_f64", out, reps);
    run<3>("v_add_u32", out, reps);
    run<0>("v_fma_f64", out, reps * 4);
    return 0;
}
