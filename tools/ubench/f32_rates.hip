// f32_rates.hip -- issue cost of the 32-bit and packed 32-bit vector instructions the single-precision kernels
// are made of, on gfx950.  Every wave runs REPS x 16 independent instructions of one kind; the table shows
// wave-instructions per second over the whole chip at 2 and 4 waves per SIMD and the time per instruction
// relative to v_fma_f32.   Development aid: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int REPS = 1 << 18;

typedef float f2 __attribute__((ext_vector_type(2)));

#define R16(I) I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7) I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7)

template <int KIND>
__global__ void __launch_bounds__(256) bench(float *out, float seed)
{
    float a[8];
    f2 p[8];
    for (int i = 0; i < 8; ++i) {
        a[i] = seed + threadIdx.x + i;
        p[i] = f2{a[i], a[i] + 0.5f};
    }
    float m = 1.0001f, c = 1e-5f;
    f2 pm = {1.0001f, 0.9999f}, pc = {1e-5f, 2e-5f};
    unsigned k = (unsigned)seed;
    for (int r = 0; r < REPS; ++r) {
        if (KIND == 0) {
#define I(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
            R16(I)
#undef I
        } else if (KIND == 1) {
#define I(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pm), "v"(pc));
            R16(I)
#undef I
        } else if (KIND == 2) {
#define I(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pm));
            R16(I)
#undef I
        } else if (KIND == 3) {
#define I(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc));
            R16(I)
#undef I
        } else if (KIND == 4) {
#define I(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            R16(I)
#undef I
        } else if (KIND == 5) {
#define I(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
            R16(I)
#undef I
        } else if (KIND == 6) {
#define I(i) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(a[i]) : "v"(k));
            R16(I)
#undef I
        } else if (KIND == 7) {
#define I(i) asm volatile("v_floor_f32 %0, %0" : "+v"(a[i]));
            R16(I)
#undef I
        } else if (KIND == 8) {
#define I(i) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(m));
            R16(I)
#undef I
        } else if (KIND == 9) {
#define I(i) asm volatile("v_pk_mov_b32 %0, %1, %2" : "=v"(p[i]) : "v"(pm), "v"(pc));
            R16(I)
#undef I
        } else if (KIND == 10) {
#define I(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(m) : "vcc");
            R16(I)
#undef I
        } else if (KIND == 11) {
#define I(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pm), "v"(pc));
            R16(I)
#undef I
        } else if (KIND == 12) {  // packed fma with swizzled halves (op_sel), the shape of a complex product
#define I(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "+v"(p[i]) : "v"(pm), "v"(pc));
            R16(I)
#undef I
        } else if (KIND == 13) {
#define I(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            R16(I)
#undef I
        } else if (KIND == 14) {
#define I(i) asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(a[i]) : "v"(k));
            R16(I)
#undef I
        } else if (KIND == 15) {
#define I(i) asm volatile("v_frexp_exp_i32_f32 %0, %1" : "=v"(k) : "v"(a[i]));
            R16(I)
#undef I
        }
    }
    float s = (float)k;
    for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static double base_rate[2];

template <int KIND>
int run(const char *name, float *out)
{
    printf("%-36s", name);
    int col = 0;
    for (int wps : {2, 4}) {
        const int grid = 256 * wps;  // 256 CUs x 4 SIMDs x wps waves = grid x 4 waves
        hipEvent_t e0, e1;
        CHK(hipEventCreate(&e0));
        CHK(hipEventCreate(&e1));
        hipLaunchKernelGGL(bench<KIND>, dim3(grid), dim3(256), 0, 0, out, 1.0f);
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL(bench<KIND>, dim3(grid), dim3(256), 0, 0, out, 1.0f);
        CHK(hipEventRecord(e1));
        CHK(hipEventSynchronize(e1));
        float ms;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        const double rate = (double)grid * 4 * REPS * 16 / (ms * 1e-3);
        if (KIND == 0) base_rate[col] = rate;
        printf("  %d waves/SIMD: %.3e /s (x%.2f)", wps, rate, base_rate[col] / rate);
        ++col;
    }
    printf("\n");
    return 0;
}

int main()
{
    float *out;
    CHK(hipMalloc(&out, 1024 * 256 * sizeof(float)));
    printf("wave-instructions per second over the chip, and time per instruction relative to v_fma_f32\n");
    if (run<0>("v_fma_f32", out)) return 1;
    if (run<1>("v_pk_fma_f32", out)) return 1;
    if (run<2>("v_pk_mul_f32", out)) return 1;
    if (run<3>("v_pk_add_f32", out)) return 1;
    if (run<12>("v_pk_fma_f32 op_sel / neg", out)) return 1;
    if (run<11>("v_fma_f64", out)) return 1;
    if (run<4>("v_rcp_f32", out)) return 1;
    if (run<5>("v_sqrt_f32", out)) return 1;
    if (run<6>("v_ldexp_f32", out)) return 1;
    if (run<7>("v_floor_f32", out)) return 1;
    if (run<8>("v_mov_b32", out)) return 1;
    if (run<9>("v_pk_mov_b32", out)) return 1;
    if (run<10>("v_cndmask_b32 (vcc)", out)) return 1;
    if (run<13>("v_max_f32", out)) return 1;
    if (run<14>("v_cvt_f32_i32", out)) return 1;
    if (run<15>("v_frexp_exp_i32_f32", out)) return 1;
    CHK(hipFree(out));
    return 0;
}
