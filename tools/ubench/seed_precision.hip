// seed_precision.hip -- worst relative error of the v_rcp_f64 / v_rsq_f64 seeds on gfx950 over a dense sweep
// of [0.5, 4), and of the refinement sequences built on them.  Development aid.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>

__global__ void sweep(double *err, int n)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    double e_rcp = 0, e_rsq = 0, e_r1 = 0, e_q1 = 0, e_s1 = 0;
    for (int i = g; i < n; i += gridDim.x * blockDim.x) {
        const double b = 0.5 + 3.5 * ((double)i + 0.37) / (double)n;
        const double r0 = __builtin_amdgcn_rcp(b);
        e_rcp = fmax(e_rcp, fabs(fma(-b, r0, 1.0)));               // |1 - b r0| = relative error of the seed
        const double r1 = fma(fma(-b, r0, 1.0), r0, r0);              // one Newton step
        e_r1 = fmax(e_r1, fabs(fma(-b, r1, 1.0)));
        const double a = 0.3 * b;                                     // quotient a / b with one step + one residual correction
        const double q = a * r1;
        const double q1 = fma(fma(-b, q, a), r1, q);
        e_q1 = fmax(e_q1, fabs(q1 - a / b) / (a / b));
        const double y0 = __builtin_amdgcn_rsq(b);
        e_rsq = fmax(e_rsq, fabs(fma(-b * y0, y0, 1.0)) * 0.5);       // relative error of the seed
        double gq = b * y0, h = 0.5 * y0;                             // one coupled step, one correction
        const double rr = fma(-h, gq, 0.5);
        gq = fma(gq, rr, gq);
        h = fma(h, rr, h);
        gq = fma(fma(-gq, gq, b), h, gq);
        e_s1 = fmax(e_s1, fabs(gq - sqrt(b)) / sqrt(b));
    }
    err[g * 5 + 0] = e_rcp; err[g * 5 + 1] = e_rsq; err[g * 5 + 2] = e_r1; err[g * 5 + 3] = e_q1; err[g * 5 + 4] = e_s1;
}

int main()
{
    const int threads = 256 * 1024, n = 1 << 30;
    double *d;
    if (hipMalloc(&d, threads * 5 * 8) != hipSuccess) return 1;
    hipLaunchKernelGGL(sweep, dim3(1024), dim3(256), 0, 0, d, n);
    std::vector<double> h(threads * 5);
    if (hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    double m[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < threads; ++i) for (int k = 0; k < 5; ++k) m[k] = fmax(m[k], h[i * 5 + k]);
    const char *names[5] = {"v_rcp_f64 seed", "v_rsq_f64 seed", "rcp + 1 Newton step", "quotient: 1 step + 1 correction (vs a/b)", "sqrt: 1 coupled step + 1 correction (vs sqrt)"};
    for (int k = 0; k < 5; ++k) printf("%-48s max rel err %.3e = 2^%.1f\n", names[k], m[k], log2(m[k] > 0 ? m[k] : 1e-300));
    return 0;
}
