// valu_rates.hip -- issue cost of the vector instructions the vocoder kernels are made of, on gfx950.
// For each instruction kind, every wave runs REPS x 16 independent instructions (8 chains x 2) and
// stamps s_memtime around the loop; the table shows shader cycles per wave-instruction as seen by
// one wave, at 1, 2 and 3 waves per SIMD (grids of 256 / 512 / 768 workgroups of 256 threads), and the
// resulting cycles per instruction per SIMD.   Development aid: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int REPS = 2048;

#define BODY8(INS)                                                                  \
    INS(a0) INS(a1) INS(a2) INS(a3) INS(a4) INS(a5) INS(a6) INS(a7)                 \
    INS(a0) INS(a1) INS(a2) INS(a3) INS(a4) INS(a5) INS(a6) INS(a7)

template <int KIND>
__global__ void __launch_bounds__(256) bench(double *out, unsigned long long *cyc, double seed, int iseed)
{
    double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    double m = 1.0000001, c = 1e-9;
    unsigned i0 = iseed + threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3, i4 = i0 + 4, i5 = i0 + 5, i6 = i0 + 6, i7 = i0 + 7;
    float f0 = (float)a0, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, f4 = f0 + 4, f5 = f0 + 5, f6 = f0 + 6, f7 = f0 + 7;
    float fm = 1.0001f, fc = 1e-5f;
    unsigned k = iseed, r = iseed * 7;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < REPS; ++r) {
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
        } else if (KIND == 3) {  // 32-bit integer add
            asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                         "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"
                         "v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                         "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"
                         : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(k));
        } else if (KIND == 4) {  // select on vcc
            asm volatile("v_cmp_gt_u32 vcc, %8, %0\n"
                         "v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n"
                         "v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %5, %5, %6, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n v_cndmask_b32 %7, %7, %0, vcc\n"
                         "v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n"
                         "v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %5, %5, %6, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n"
                         : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(k) : "vcc");
        } else if (KIND == 5) {
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(fm), "v"(fc));
        } else if (KIND == 6) {
#define INS(x) asm volatile("v_rcp_f64 %0, %0" : "+v"(x));
            BODY8(INS)
#undef INS
        } else if (KIND == 7) {
#define INS(x) asm volatile("v_rsq_f64 %0, %0" : "+v"(x));
            BODY8(INS)
#undef INS
        } else if (KIND == 8) {
#define INS(x) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(x) : "v"(k));
            BODY8(INS)
#undef INS
        } else if (KIND == 9) {
#define INS(x) asm volatile("v_floor_f64 %0, %0" : "+v"(x));
            BODY8(INS)
#undef INS
        } else if (KIND == 10) {
#define INS(x) asm volatile("v_max_f64 %0, %0, %1" : "+v"(x) : "v"(c));
            BODY8(INS)
#undef INS
        } else if (KIND == 11) {  // compare into vcc (result unused)
#define INS(x) asm volatile("v_cmp_gt_f64 vcc, %0, %1" : : "v"(x), "v"(c) : "vcc");
            BODY8(INS)
#undef INS
        } else if (KIND == 12) {  // half f64 fma, half 32-bit integer: do they share issue?
            asm volatile("v_fma_f64 %0, %0, %8, %9\n v_add_u32 %4, %4, %10\n v_fma_f64 %1, %1, %8, %9\n v_add_u32 %5, %5, %10\n"
                         "v_fma_f64 %2, %2, %8, %9\n v_add_u32 %6, %6, %10\n v_fma_f64 %3, %3, %8, %9\n v_add_u32 %7, %7, %10\n"
                         "v_fma_f64 %0, %0, %8, %9\n v_add_u32 %4, %4, %10\n v_fma_f64 %1, %1, %8, %9\n v_add_u32 %5, %5, %10\n"
                         "v_fma_f64 %2, %2, %8, %9\n v_add_u32 %6, %6, %10\n v_fma_f64 %3, %3, %8, %9\n v_add_u32 %7, %7, %10\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(m), "v"(c), "v"(k));
        } else if (KIND == 13) {
#define INS(x) asm volatile("v_frexp_exp_i32_f64 %0, %1" : "=v"(i0) : "v"(x));
            BODY8(INS)
#undef INS
        } else if (KIND == 14) {  // 64-bit move
#define INS(x) asm volatile("v_mov_b64 %0, %1" : "=v"(x) : "v"(m));
            BODY8(INS)
#undef INS
        } else if (KIND == 15) {  // dependent chain of f64 fma (latency)
            asm volatile("v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n"
                         "v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n"
                         "v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n"
                         "v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n"
                         : "+v"(a0) : "v"(m), "v"(c));
        } else if (KIND == 16) {
#define INS(x) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(x) : "v"(k));
            BODY8(INS)
#undef INS
        } else if (KIND == 17) {  // v_bfi_b32
            asm volatile("v_bfi_b32 %0, %8, %0, %1\n v_bfi_b32 %1, %8, %1, %2\n v_bfi_b32 %2, %8, %2, %3\n v_bfi_b32 %3, %8, %3, %4\n"
                         "v_bfi_b32 %4, %8, %4, %5\n v_bfi_b32 %5, %8, %5, %6\n v_bfi_b32 %6, %8, %6, %7\n v_bfi_b32 %7, %8, %7, %0\n"
                         "v_bfi_b32 %0, %8, %0, %1\n v_bfi_b32 %1, %8, %1, %2\n v_bfi_b32 %2, %8, %2, %3\n v_bfi_b32 %3, %8, %3, %4\n"
                         "v_bfi_b32 %4, %8, %4, %5\n v_bfi_b32 %5, %8, %5, %6\n v_bfi_b32 %6, %8, %6, %7\n v_bfi_b32 %7, %8, %7, %0\n"
                         : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(k));
        } else if (KIND == 18) {  // select on vcc, independent destinations, condition written once outside
            asm volatile("v_cndmask_b32 %0, %8, %9, vcc\n v_cndmask_b32 %1, %8, %9, vcc\n v_cndmask_b32 %2, %8, %9, vcc\n v_cndmask_b32 %3, %8, %9, vcc\n"
                         "v_cndmask_b32 %4, %8, %9, vcc\n v_cndmask_b32 %5, %8, %9, vcc\n v_cndmask_b32 %6, %8, %9, vcc\n v_cndmask_b32 %7, %8, %9, vcc\n"
                         "v_cndmask_b32 %0, %8, %9, vcc\n v_cndmask_b32 %1, %8, %9, vcc\n v_cndmask_b32 %2, %8, %9, vcc\n v_cndmask_b32 %3, %8, %9, vcc\n"
                         "v_cndmask_b32 %4, %8, %9, vcc\n v_cndmask_b32 %5, %8, %9, vcc\n v_cndmask_b32 %6, %8, %9, vcc\n v_cndmask_b32 %7, %8, %9, vcc\n"
                         : "=v"(i0), "=v"(i1), "=v"(i2), "=v"(i3), "=v"(i4), "=v"(i5), "=v"(i6), "=v"(i7) : "v"(k), "v"(r) : "vcc");
        } else if (KIND == 19) {  // select on an SGPR pair (e64)
            asm volatile("v_cndmask_b32 %0, %8, %9, s[20:21]\n v_cndmask_b32 %1, %8, %9, s[20:21]\n v_cndmask_b32 %2, %8, %9, s[20:21]\n v_cndmask_b32 %3, %8, %9, s[20:21]\n"
                         "v_cndmask_b32 %4, %8, %9, s[20:21]\n v_cndmask_b32 %5, %8, %9, s[20:21]\n v_cndmask_b32 %6, %8, %9, s[20:21]\n v_cndmask_b32 %7, %8, %9, s[20:21]\n"
                         "v_cndmask_b32 %0, %8, %9, s[20:21]\n v_cndmask_b32 %1, %8, %9, s[20:21]\n v_cndmask_b32 %2, %8, %9, s[20:21]\n v_cndmask_b32 %3, %8, %9, s[20:21]\n"
                         "v_cndmask_b32 %4, %8, %9, s[20:21]\n v_cndmask_b32 %5, %8, %9, s[20:21]\n v_cndmask_b32 %6, %8, %9, s[20:21]\n v_cndmask_b32 %7, %8, %9, s[20:21]\n"
                         : "=v"(i0), "=v"(i1), "=v"(i2), "=v"(i3), "=v"(i4), "=v"(i5), "=v"(i6), "=v"(i7) : "v"(k), "v"(r) : "s20", "s21");
        } else if (KIND == 20) {  // compare + two selects, the pattern of a 64-bit conditional
            asm volatile("v_cmp_gt_f64 vcc, %0, %8\n v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %5, %5, %4, vcc\n"
                         "v_cmp_gt_f64 vcc, %1, %8\n v_cndmask_b32 %6, %6, %7, vcc\n v_cndmask_b32 %7, %7, %6, vcc\n"
                         "v_cmp_gt_f64 vcc, %2, %8\n v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %5, %5, %4, vcc\n"
                         "v_cmp_gt_f64 vcc, %3, %8\n v_cndmask_b32 %6, %6, %7, vcc\n v_cndmask_b32 %7, %7, %6, vcc\n"
                         "v_cmp_gt_f64 vcc, %0, %8\n v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %5, %5, %4, vcc\n v_add_u32 %6, %6, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(c) : "vcc");
        } else if (KIND == 21) {
            asm volatile("v_xor_b32 %0, %0, %8\n v_xor_b32 %1, %1, %8\n v_xor_b32 %2, %2, %8\n v_xor_b32 %3, %3, %8\n"
                         "v_lshlrev_b32 %4, 3, %4\n v_lshlrev_b32 %5, 3, %5\n v_and_b32 %6, %6, %8\n v_and_b32 %7, %7, %8\n"
                         "v_sub_u32 %0, %0, %8\n v_sub_u32 %1, %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n"
                         "v_add3_u32 %4, %4, %8, %5\n v_add3_u32 %5, %5, %8, %6\n v_and_or_b32 %6, %6, %8, %7\n v_and_or_b32 %7, %7, %8, %0\n"
                         : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(k));
        } else if (KIND == 22) {
#define INS(x) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(x) : "v"(m), "v"(c));
            BODY8(INS)
#undef INS
        } else if (KIND == 23) {  // f64 fma with an s_nop after each
#define INS(x) asm volatile("v_fma_f64 %0, %0, %1, %2\n s_nop 0" : "+v"(x) : "v"(m), "v"(c));
            BODY8(INS)
#undef INS
        } else if (KIND == 24) {  // 64-bit shift-add (address arithmetic)
            unsigned long long u0 = i0, u1 = i1, u2 = i2, u3 = i3;
            asm volatile("v_lshl_add_u64 %0, %0, 0, %4\n v_lshl_add_u64 %1, %1, 0, %4\n v_lshl_add_u64 %2, %2, 0, %4\n v_lshl_add_u64 %3, %3, 0, %4\n"
                         "v_lshl_add_u64 %0, %0, 0, %4\n v_lshl_add_u64 %1, %1, 0, %4\n v_lshl_add_u64 %2, %2, 0, %4\n v_lshl_add_u64 %3, %3, 0, %4\n"
                         "v_lshl_add_u64 %0, %0, 0, %4\n v_lshl_add_u64 %1, %1, 0, %4\n v_lshl_add_u64 %2, %2, 0, %4\n v_lshl_add_u64 %3, %3, 0, %4\n"
                         "v_lshl_add_u64 %0, %0, 0, %4\n v_lshl_add_u64 %1, %1, 0, %4\n v_lshl_add_u64 %2, %2, 0, %4\n v_lshl_add_u64 %3, %3, 0, %4\n"
                         : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(t0));
            i0 = (unsigned)u0; i1 = (unsigned)u1; i2 = (unsigned)u2; i3 = (unsigned)u3;
        } else if (KIND == 25) {  // f64 fma with an SGPR-pair constant operand
#define INS(x) asm volatile("v_fma_f64 %0, %0, %1, s[20:21]" : "+v"(x) : "v"(m) : "s20", "s21");
            BODY8(INS)
#undef INS
        } else if (KIND == 26) {  // f64 fma and vcc select alternating, vcc written once per 16
            asm volatile("v_cmp_gt_f64 vcc, %0, %8\n"
                         "v_fma_f64 %0, %0, %9, %8\n v_cndmask_b32 %4, %4, %5, vcc\n v_fma_f64 %1, %1, %9, %8\n v_cndmask_b32 %5, %5, %6, vcc\n"
                         "v_fma_f64 %2, %2, %9, %8\n v_cndmask_b32 %6, %6, %7, vcc\n v_fma_f64 %3, %3, %9, %8\n v_cndmask_b32 %7, %7, %4, vcc\n"
                         "v_fma_f64 %0, %0, %9, %8\n v_cndmask_b32 %4, %4, %5, vcc\n v_fma_f64 %1, %1, %9, %8\n v_cndmask_b32 %5, %5, %6, vcc\n"
                         "v_fma_f64 %2, %2, %9, %8\n v_cndmask_b32 %6, %6, %7, vcc\n v_fma_f64 %3, %3, %9, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(c), "v"(m) : "vcc");
        } else if (KIND == 27) {  // one compare, four selects, three f64 fma: the shape of a folded 64-bit pair select
            asm volatile("v_cmp_gt_f64 vcc, %0, %8\n v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %5, %5, %6, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n v_cndmask_b32 %7, %7, %4, vcc\n"
                         "v_fma_f64 %0, %0, %9, %8\n v_fma_f64 %1, %1, %9, %8\n v_fma_f64 %2, %2, %9, %8\n"
                         "v_cmp_gt_f64 vcc, %1, %8\n v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %5, %5, %6, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n v_cndmask_b32 %7, %7, %4, vcc\n"
                         "v_fma_f64 %0, %0, %9, %8\n v_fma_f64 %1, %1, %9, %8\n v_fma_f64 %2, %2, %9, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(c), "v"(m) : "vcc");
        } else if (KIND == 28) {  // the same with the condition in an SGPR pair
            asm volatile("v_cmp_gt_f64 s[20:21], %0, %8\n v_cndmask_b32 %4, %4, %5, s[20:21]\n v_cndmask_b32 %5, %5, %6, s[20:21]\n v_cndmask_b32 %6, %6, %7, s[20:21]\n v_cndmask_b32 %7, %7, %4, s[20:21]\n"
                         "v_fma_f64 %0, %0, %9, %8\n v_fma_f64 %1, %1, %9, %8\n v_fma_f64 %2, %2, %9, %8\n"
                         "v_cmp_gt_f64 s[20:21], %1, %8\n v_cndmask_b32 %4, %4, %5, s[20:21]\n v_cndmask_b32 %5, %5, %6, s[20:21]\n v_cndmask_b32 %6, %6, %7, s[20:21]\n v_cndmask_b32 %7, %7, %4, s[20:21]\n"
                         "v_fma_f64 %0, %0, %9, %8\n v_fma_f64 %1, %1, %9, %8\n v_fma_f64 %2, %2, %9, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(c), "v"(m) : "s20", "s21");
        } else if (KIND == 29) {  // 32-bit integer compare into vcc + selects
            asm volatile("v_cmp_gt_u32 vcc, %0, %8\n v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %5, %5, %6, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n"
                         "v_cmp_gt_u32 vcc, %1, %8\n v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %5, %5, %6, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n"
                         "v_cmp_gt_u32 vcc, %2, %8\n v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %5, %5, %6, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n"
                         "v_cmp_gt_u32 vcc, %3, %8\n v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %5, %5, %6, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n"
                         : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(k) : "vcc");
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    out[g] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (double)(i0 + i1 + i2 + i3 + i4 + i5 + i6 + i7) + (double)(f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7);
    if ((threadIdx.x & 63) == 0) cyc[g >> 6] = t1 - t0;
}

template <int KIND>
int run(const char *name, double *out, unsigned long long *cyc)
{
    printf("%-34s", name);
    for (int wps = 1; wps <= 3; ++wps) {
        const int grid = 256 * wps;
        hipEvent_t e0, e1;
        CHK(hipEventCreate(&e0));
        CHK(hipEventCreate(&e1));
        hipLaunchKernelGGL(bench<KIND>, dim3(grid), dim3(256), 0, 0, out, cyc, 1.0, 3);
        CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL(bench<KIND>, dim3(grid), dim3(256), 0, 0, out, cyc, 1.0, 3);
        CHK(hipEventRecord(e1));
        CHK(hipDeviceSynchronize());
        float ms = 0;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> h(grid * 4);
        CHK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
        std::sort(h.begin(), h.end());
        const double med = (double)h[h.size() / 2];
        const double per_wave = med / (REPS * 16.0);
        printf("  %d w/SIMD: %6.2f cyc/instr/wave = %5.2f cyc/SIMD (%.3f ms)", wps, per_wave, per_wave / wps, ms);
    }
    printf("\n");
    return 0;
}

int main()
{
    double *out;
    unsigned long long *cyc;
    CHK(hipMalloc(&out, 768 * 256 * 8));
    CHK(hipMalloc(&cyc, 768 * 4 * 8));
    run<0>("v_fma_f64", out, cyc);
    run<1>("v_add_f64", out, cyc);
    run<2>("v_mul_f64", out, cyc);
    run<10>("v_max_f64", out, cyc);
    run<8>("v_ldexp_f64", out, cyc);
    run<9>("v_floor_f64", out, cyc);
    run<11>("v_cmp_gt_f64", out, cyc);
    run<13>("v_frexp_exp_i32_f64", out, cyc);
    run<16>("v_cvt_f64_u32", out, cyc);
    run<6>("v_rcp_f64", out, cyc);
    run<7>("v_rsq_f64", out, cyc);
    run<14>("v_mov_b64", out, cyc);
    run<3>("v_add_u32", out, cyc);
    run<4>("v_cndmask_b32", out, cyc);
    run<17>("v_bfi_b32", out, cyc);
    run<18>("v_cndmask_b32 vcc, indep", out, cyc);
    run<19>("v_cndmask_b32 sgpr pair", out, cyc);
    run<20>("cmp_f64 + 2 cndmask (x5) + add", out, cyc);
    run<21>("int mix xor/shl/and/sub/mov/add3", out, cyc);
    run<22>("v_fmac_f64", out, cyc);
    run<23>("v_fma_f64 + s_nop 0", out, cyc);
    run<24>("v_lshl_add_u64", out, cyc);
    run<25>("v_fma_f64 with SGPR operand", out, cyc);
    run<26>("fma_f64 / cndmask vcc alternating", out, cyc);
    run<27>("cmp_f64 vcc + 4 cndmask + 3 fma", out, cyc);
    run<28>("cmp_f64 sgpr + 4 cndmask + 3 fma", out, cyc);
    run<29>("cmp_u32 vcc + 3 cndmask", out, cyc);
    run<5>("v_fma_f32", out, cyc);
    run<12>("v_fma_f64 + v_add_u32 alternating", out, cyc);
    run<15>("v_fma_f64 dependent chain", out, cyc);
    return 0;
}
