// voc_kernels_bluestein_big.h -- frame lengths above 4096 that are not powers of two, and every frame
// length above 16384 (N <= 32768).
//
// The reference's FFT (vector-fftw, vocoder/src/Vocoder.hs:49-50,115) accepts any length.  The chirp-z
// transform of voc_kernels_bluestein.h needs a power-of-two transform of L >= 2N - 1 points, which for
// these lengths (L = 16384 to 65536 complex points) no longer fits in LDS.  Here it runs through
// device memory, L = L1 * P with P = 8192 (the largest LDS-resident transform) and L1 = 2, 4 or 8:
//     A[k2 + P k1] = sum_n1 W_L1^(n1 k1) [ W_L^(n1 k2) sum_n2 a[n1 + L1 n2] W_P^(n2 k2) ]
// i.e. L1 interleaved P-point transforms (one workgroup each, first kernel), a twiddle, and an
// L1-point transform across them (a few registers per thread, second kernel).  The second kernel also
// multiplies by the transform of the chirp, conjugates, and starts the second L-point transform with
// its own L1-point step and twiddle, so that the third kernel -- again L1 P-point transforms per frame
// -- ends with the convolution in natural order and finishes the frame (spectrum, or windowed samples).
// Three launches and 4 * 16 L bytes of device-memory traffic per transform; the per-bin arithmetic
// (polar conversion and phase chain, Vocoder.hs:158-163,186-190) runs on spectra in device memory,
// one thread per (channel, bin) walking the frames in order.
#pragma once

#include "voc_kernels_bluestein.h"

namespace voc {

constexpr int kBigLog2P = 13, kBigP = 1 << kBigLog2P;

struct BlueBigTab {
    const double2 *chirp;  // [N]  w[n] = exp(i pi n^2 / N)
    const double2 *bfft;   // [L]  L-point FFT of b, b[m] = b[L-m] = w[m] for m < N, else 0
    const double2 *wl;     // [L]  exp(-2 pi i m / L)
    double inv_l, n_d;
    int L1;                // L = L1 * kBigP
};

enum { BIG_SRC_RAW = 0, BIG_SRC_FRAME = 1, BIG_SRC_SPEC = 2 };
enum { BIG_DST_SPEC = 0, BIG_DST_FRAME = 1, BIG_DST_RAW = 2 };

// Where transform input n of frame `b` comes from.
struct BigSrc {
    const double *x;         // RAW / FRAME: frame (c, f) starts at x + c*chan_stride + f*frame_stride
    long long chan_stride, frame_stride;
    const double *spec;      // SPEC: row (c*F + f - spec_base) of [..][K] complex, the half spectrum of a real frame
    long long F;             // frames per channel
    long long first;         // index c*F + f of the first frame of this launch
    long long spec_base;
};

// First pass: interleaved sequence n1 of frame b.  u[(b - first) * L + n1 + L1 k2] = W_L^(n1 k2) FFT_P(a[n1 + L1 .])[k2]
template <int SRC>
__global__ void __launch_bounds__(Cfg<kBigLog2P>::T) blue_big_first_kernel(Params p, BlueBigTab b, BigSrc s, double2 *u)
{
    constexpr int P = kBigP, T = Cfg<kBigLog2P>::T;
    VOC_SMEM(double2, lds);
    const int tid = threadIdx.x;
    const int L1 = b.L1, L = L1 * P;
    const long long fl = blockIdx.x / L1;  // frame of this launch
    const int n1 = (int)(blockIdx.x % L1);
    const long long bf = s.first + fl, c = bf / s.F, f = bf % s.F;
    const int N = p.N, half = p.N / 2;
    const double *fr = SRC == BIG_SRC_SPEC ? nullptr : s.x + c * s.chan_stride + f * s.frame_stride;
    const double *sp = SRC == BIG_SRC_SPEC ? s.spec + (bf - s.spec_base) * 2 * p.K : nullptr;
    fft_passes<P, T, 1>(lds, p.twP, tid, [&](int i, int j) -> double2 {
        const int n = n1 + L1 * (i + j * (P / 8));
        if (n >= N) return make_double2(0.0, 0.0);
        const double2 w = b.chirp[n];
        if (SRC == BIG_SRC_SPEC) {  // conj(Xfull[n]) conj(w[n]), Xfull the Hermitian extension (real bins: imaginary part ignored)
            const int k = n < p.K ? n : N - n;
            double2 X = make_double2(sp[2 * k], sp[2 * k + 1]);
            if (k == 0 || 2 * k == N) X.y = 0.0;
            if (n >= p.K) X.y = -X.y;
            return cconj(cmul(X, w));
        }
        double v;
        if (SRC == BIG_SRC_RAW) {
            v = fr[n];
        } else {  // window, zero padding, rewind: u[n] = padded[(n + N div 2) mod N]  (Vocoder.hs:118-139)
            int q = n + half;
            if (q >= N) q -= N;
            const int idx = q - p.hd;
            v = (idx >= 0 && idx < p.W) ? p.window[idx] * fr[idx] : 0.0;
        }
        return make_double2(v * w.x, -(v * w.y));
    });
    double2 *dst = u + fl * L;
    for (int k2 = tid; k2 < P; k2 += T) {
        const int e = (int)(((long long)n1 * k2) % L);
        dst[n1 + L1 * k2] = cmul(lds[lds_pad(k2)], b.wl[e]);
    }
}

// L1-point DFT in registers, natural order in and out (L1 = 2, 4 or 8; entries from L1 on are zero and stay unused)
VOC_DEV void small_dft(double2 (&v)[8], int L1)
{
    if (L1 == 2) {
        const double2 t = v[0];
        v[0] = cadd(t, v[1]);
        v[1] = csub(t, v[1]);
    } else if (L1 == 4) {
        double2 q[4] = {v[0], v[1], v[2], v[3]};
        dft<4>(q);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = q[j];
    } else {
        dft<8>(v);
    }
}

// Second pass, one thread per (frame, k2): finishes the first transform across the L1 sequences,
// multiplies by the chirp's transform, conjugates, and does the L1-point step and twiddle of the second
// transform:  u[k2' + L1 k2] = W_L^(k2 k2') sum_k1 conj(A[k2 + P k1] B[k2 + P k1]) W_L1^(k1 k2')
__global__ void blue_big_middle_kernel(BlueBigTab b, double2 *u, long long frames)
{
    constexpr int P = kBigP;
    const int L1 = b.L1, L = L1 * P;
    const long long total = frames * P;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
        const int k2 = (int)(g % P);
        double2 *row = u + (g / P) * L + (long long)L1 * k2;
        double2 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = j < L1 ? row[j] : make_double2(0.0, 0.0);
        small_dft(v, L1);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (j < L1) v[j] = cconj(cmul(v[j], b.bfft[k2 + P * j]));
        small_dft(v, L1);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (j < L1) row[j] = cmul(v[j], b.wl[(int)(((long long)k2 * j) % L)]);
    }
}

struct BigDst {
    double *spec;            // SPEC: row (c*rpc + f - row_base) of [..][K] complex
    double *out;             // FRAME: [(c*F + f)*W + i]; RAW: [(c*F + f)*N + n]
    long long first;         // index c*F + f of the first frame of this launch
    long long F, rpc, row_base;
};

// Third pass: sequence k2' of the second transform; its outputs r = k2' + L1 k1' are the convolution in
// natural order (conjugated, times L), finished per element as the in-LDS kernels finish theirs.
template <int DST>
__global__ void __launch_bounds__(Cfg<kBigLog2P>::T) blue_big_last_kernel(Params p, BlueBigTab b, const double2 *u, BigDst d)
{
    constexpr int P = kBigP, T = Cfg<kBigLog2P>::T;
    VOC_SMEM(double2, lds);
    const int tid = threadIdx.x;
    const int L1 = b.L1, L = L1 * P;
    const long long fl = blockIdx.x / L1;
    const int c2 = (int)(blockIdx.x % L1);
    const long long row = d.first + fl;  // c*F + f
    const long long srow = (row / d.F) * d.rpc + (row % d.F) - d.row_base;
    const int N = p.N, half = p.N / 2;
    const double2 *src = u + fl * L;
    fft_passes<P, T, 1>(lds, p.twP, tid, [&](int i, int j) -> double2 { return src[c2 + L1 * (i + j * (P / 8))]; });
    for (int k1 = tid; k1 < P; k1 += T) {
        const int r = c2 + L1 * k1;
        const double2 y = lds[lds_pad(k1)];
        if (DST == BIG_DST_SPEC) {
            if (r < p.K) {  // X[r] = conj(w[r]) c[r],  c[r] = conj(y) / L
                const double2 t = cmul(b.chirp[r], y);
                double2 X = make_double2(t.x * b.inv_l, -(t.y * b.inv_l));
                if (r == 0 || 2 * r == N) X.y = 0.0;  // real bins of a real transform
                d.spec[srow * 2 * p.K + 2 * r] = X.x;
                d.spec[srow * 2 * p.K + 2 * r + 1] = X.y;
            }
        } else if (r < N) {  // DFT(conj Xfull)[r] = conj(w[r]) conj(y) / L, so x[r] = Re(w[r] y) / (L N)
            const double2 w = b.chirp[r];
            const double xr = (fma(w.x, y.x, -(w.y * y.y)) * b.inv_l) * p.inv_n;
            if (DST == BIG_DST_RAW) {
                d.out[row * N + r] = xr;
            } else {  // rewind, cutCenter, window: frame[i] = w[i] x[(i + off + N div 2) mod N]  (Vocoder.hs:194-200)
                int i = r - p.off - half;
                if (i < 0) i += N;
                if (i < 0) i += N;
                if (i < p.W) d.out[row * p.W + i] = p.window[i] * xr;
            }
        }
    }
}

// analysisStep on spectra in device memory (Vocoder.hs:158-163): one thread per (channel, bin) walks the
// frames [f_lo, f_hi) of its channel in order.  phase[c*K + k] holds the previous raw phase on entry and the
// last one on return; spec rows are numbered from spec_first.
__global__ void spec_rows_to_polar_kernel(Params p, double n_d, const double *spec, long long spec_first, double *phase,
                                          double *mag, double *inc, long long ldk, long long rpc, long long F,
                                          long long b_lo, long long b_hi, double vol, long long channels)
{
    const long long total = channels * p.K;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
        const int k = (int)(g % p.K);
        const long long c = g / p.K;
        long long lo = c * F, hi = lo + F;
        lo = lo > b_lo ? lo : b_lo;
        hi = hi < b_hi ? hi : b_hi;
        if (lo >= hi) continue;
        double prev = phase[g];
        for (long long bf = lo; bf < hi; ++bf) {
            const double *s = spec + (bf - spec_first) * 2 * p.K;
            double m, ph;
            hs_polar(s[2 * k], s[2 * k + 1], m, ph);
            const long long row = c * rpc + (bf - c * F);
            mag[row * ldk + k] = m * vol;
            inc[row * ldk + k] = phase_inc_any(p, n_d, k, ph - prev);
            prev = ph;
        }
        phase[g] = prev;
    }
}

// synthesisStep into spectra in device memory (Vocoder.hs:186-190), the same walk.
__global__ void polar_rows_to_spec_kernel(Params p, SynArgs a, double *phase, double *spec, long long spec_first,
                                          long long b_lo, long long b_hi, long long channels)
{
    const long long total = channels * p.K;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
        const int k = (int)(g % p.K);
        const long long c = g / p.K;
        long long lo = c * a.F, hi = lo + a.F;
        lo = lo > b_lo ? lo : b_lo;
        hi = hi < b_hi ? hi : b_hi;
        if (lo >= hi) continue;
        double ph = phase[g];
        for (long long bf = lo; bf < hi; ++bf) {
            const long long f = bf - c * a.F;
            const double2 A = synthesize_bin(p, mag_row_of(a, c, f)[k], inc_row_of(a, c, f)[k], ph);
            double *s = spec + (bf - spec_first) * 2 * p.K;
            s[2 * k] = A.x;
            s[2 * k + 1] = A.y;
        }
        phase[g] = ph;
    }
}

}  // namespace voc
