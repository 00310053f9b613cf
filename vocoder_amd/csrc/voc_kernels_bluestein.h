// voc_kernels_bluestein.h -- frame lengths that are not powers of two (2 <= N <= 4096).
//
// The reference's FFT (vector-fftw, vocoder/src/Vocoder.hs:49-50,115) accepts any length.  The
// radix-2/4/8 kernels of voc_kernels.h cover powers of two; every other length runs here as a
// chirp-z (Bluestein) transform on top of the same LDS-resident power-of-two FFT:
//     X[k] = conj(w[k]) * sum_n (x[n] conj(w[n])) w[k-n],      w[n] = exp(i pi n^2 / N),
// i.e. a cyclic convolution of length L = 2^m >= 2N-1 done with two L-point FFTs per transform (the
// transform of the chirp is a per-context table).  The backward transform is the forward one of
// the conjugated, Hermitian-extended spectrum, conjugated again.  Everything around the transform
// -- window, zero padding, rewind, polar conversion, phase increment, phase accumulation, cut,
// window -- is the code of the generic kernels with index arithmetic for a general N.  Overlap-add
// of these lengths goes through the frames-in-HBM path (sum_frames_kernel).
#pragma once

#include "voc_kernels.h"

namespace voc {

struct BlueTab {
    const double2 *chirp;  // [N]  w[n] = exp(i pi n^2 / N)
    const double2 *bfft;   // [L]  L-point FFT of b, b[m] = b[L-m] = w[m] for m < N, else 0
    double inv_l;          // 1 / L
    double n_d;            // N as a double
};

// lds holds the L-point FFT of a[n] = x[n] conj(w[n]); on return lds[pad(m)] = conj(L * c[m]) with
// c = a (*) b the cyclic convolution: multiply by FFT(b), conjugate, forward FFT (= conjugated
// unnormalised backward FFT).
template <int LOG2L>
VOC_DEV void blue_convolve(double2 *lds, const Params &p, const BlueTab &b, int tid)
{
    constexpr int L = 1 << LOG2L, T = Cfg<LOG2L>::T;
    for (int m = tid; m < L; m += T) {
        const double2 z = cmul(lds[lds_pad(m)], b.bfft[m]);
        lds[lds_pad(m)] = cconj(z);
    }
    __syncthreads();
    fft_passes<L, T, 1>(lds, p.twP, tid, LdsLoader<L>{lds});
}

// calcPhaseInc for a general N: the division by N is a division (Vocoder.hs:169-173)
VOC_DEV double phase_inc_any(const Params &p, double n_d, int k, double ph_diff)
{
    const double omega = ((kTwoPi * (double)k) * p.hop_d) / n_d;
    const double s = omega + hs_wrap(ph_diff - omega);
    return p.hop_pow2 ? s * p.inv_hop : s / p.hop_d;
}

constexpr int kBlueBins = 3;  // bins per thread: K <= L/4 + 1 and T >= L/8 threads

template <int LOG2L, bool RAW, bool SPEC_OUT>
__global__ void __launch_bounds__(Cfg<LOG2L>::T) analysis_blue_kernel(Params p, AnaArgs a, BlueTab b)
{
    constexpr int L = 1 << LOG2L, T = Cfg<LOG2L>::T;
    VOC_SMEM(double2, lds);
    const int tid = threadIdx.x;
    const long long c = blockIdx.x / a.tiles;
    const int tile = blockIdx.x % a.tiles;
    const long long f0 = (long long)tile * a.tile_len;
    const long long f1 = (f0 + a.tile_len < a.F) ? f0 + a.tile_len : a.F;
    const double *xc = a.x + c * a.chan_stride;
    const int N = p.N, half = p.N / 2;

    double prev[kBlueBins];
    if (!SPEC_OUT && f0 == 0) {
#pragma unroll
        for (int q = 0; q < kBlueBins; ++q) {
            const int k = tid + q * T;
            if (k < p.K) prev[q] = a.phase_in[c * p.K + k];
        }
    }
    const long long fs = (!SPEC_OUT && f0 > 0) ? f0 - 1 : f0;  // one frame early to recover its raw phase
    for (long long f = fs; f < f1; ++f) {
        const double *fr = xc + f * a.frame_stride;
        const bool emit = f >= f0;
        fft_passes<L, T, 1>(lds, p.twP, tid, [&](int i, int j) -> double2 {
            const int n = i + j * (L / 8);
            if (n >= N) return make_double2(0.0, 0.0);
            double v;
            if (RAW) {
                v = fr[n];
            } else {  // window, zero padding, rewind: u[n] = padded[(n + N div 2) mod N]  (Vocoder.hs:118-139)
                int s = n + half;
                if (s >= N) s -= N;
                const int idx = s - p.hd;
                v = (idx >= 0 && idx < p.W) ? p.window[idx] * fr[idx] : 0.0;
            }
            const double2 w = b.chirp[n];
            return make_double2(v * w.x, -(v * w.y));
        });
        blue_convolve<LOG2L>(lds, p, b, tid);
        const long long row = c * a.rpc + f;
#pragma unroll
        for (int q = 0; q < kBlueBins; ++q) {
            const int k = tid + q * T;
            if (k < p.K) {
                // X[k] = conj(w[k]) c[k],  c[k] = conj(lds) / L
                const double2 t = cmul(b.chirp[k], lds[lds_pad(k)]);
                double2 X = make_double2(t.x * b.inv_l, -(t.y * b.inv_l));
                if (k == 0 || 2 * k == N) X.y = 0.0;  // real bins of a real transform
                if (SPEC_OUT) {
                    double *s = a.spec + row * 2 * p.K;
                    s[2 * k] = X.x;
                    s[2 * k + 1] = X.y;
                } else {
                    double m, ph;
                    hs_polar(X.x, X.y, m, ph);
                    if (emit) {
                        a.mag[row * a.ldk + k] = m * a.vol;
                        a.inc[row * a.ldk + k] = phase_inc_any(p, b.n_d, k, ph - prev[q]);
                    }
                    prev[q] = ph;
                }
            }
        }
        __syncthreads();  // the next frame's first pass writes the buffer read above
    }
    if (!SPEC_OUT && f1 == a.F && f1 > f0) {
#pragma unroll
        for (int q = 0; q < kBlueBins; ++q) {
            const int k = tid + q * T;
            if (k < p.K) a.phase_out[c * p.K + k] = prev[q];
        }
    }
}

// OUT_MODE: SYN_OUT_FRAMES or SYN_OUT_RAW.
template <int LOG2L, int IN_MODE, int OUT_MODE>
__global__ void __launch_bounds__(Cfg<LOG2L>::T) synthesis_blue_kernel(Params p, SynArgs a, BlueTab b)
{
    constexpr int L = 1 << LOG2L, T = Cfg<LOG2L>::T;
    VOC_SMEM(double2, lds);
    const int tid = threadIdx.x;
    const long long c = blockIdx.x / a.tiles;
    const int tile = blockIdx.x % a.tiles;
    const long long f0 = (long long)tile * a.tile_len;
    const long long f1 = (f0 + a.tile_len < a.F) ? f0 + a.tile_len : a.F;
    const int N = p.N, half = p.N / 2;

    double ph[kBlueBins];
    if (IN_MODE == SYN_IN_POLAR) {
        const double *pi = (tile == 0) ? a.phase_in + c * p.K : a.carry + (c * a.tiles + tile) * p.K;
#pragma unroll
        for (int q = 0; q < kBlueBins; ++q) {
            const int k = tid + q * T;
            if (k < p.K) ph[q] = pi[k];
        }
    }
    for (long long f = f0; f < f1; ++f) {
        const long long row = c * a.F + f;
        // the frame's spectrum X[0..K), staged in the (still unused) transform buffer
#pragma unroll
        for (int q = 0; q < kBlueBins; ++q) {
            const int k = tid + q * T;
            if (k < p.K) {
                double2 A;
                if (IN_MODE == SYN_IN_POLAR) {
                    A = synthesize_bin(p, mag_row_of(a, c, f)[k], inc_row_of(a, c, f)[k], ph[q]);
                } else {
                    const double *s = a.spec + row * 2 * p.K;
                    A = make_double2(s[2 * k], s[2 * k + 1]);
                }
                if (k == 0 || 2 * k == N) A.y = 0.0;  // C2R ignores the imaginary parts of the real bins
                lds[lds_pad(k)] = A;
            }
        }
        __syncthreads();
        // forward chirp-z transform of conj(Xfull), Xfull[n] = X[n] for n < K, conj(X[N-n]) above
        // (every load below happens before the first barrier inside fft_passes)
        fft_passes<L, T, 1>(lds, p.twP, tid, [&](int i, int j) -> double2 {
            const int n = i + j * (L / 8);
            if (n >= N) return make_double2(0.0, 0.0);
            const double2 X = n < p.K ? lds[lds_pad(n)] : cconj(lds[lds_pad(N - n)]);
            return cconj(cmul(X, b.chirp[n]));  // conj(X) conj(w)
        });
        blue_convolve<LOG2L>(lds, p, b, tid);
        // DFT(conj Xfull)[n] = conj(w[n]) conj(lds)/L, so x[n] = Re(w[n] lds[n]) / (L N)
        for (int n = tid; n < N; n += T) {
            const double2 v = lds[lds_pad(n)], w = b.chirp[n];
            const double xr = (fma(w.x, v.x, -(w.y * v.y)) * b.inv_l) * p.inv_n;
            if (OUT_MODE == SYN_OUT_RAW) {
                a.out[row * N + n] = xr;
            } else {  // rewind, cutCenter, window: frame[i] = w[i] x[(i + off + N div 2) mod N]  (Vocoder.hs:194-200)
                int i = n - p.off - half;
                if (i < 0) i += N;
                if (i < 0) i += N;
                if (i < p.W) a.out[row * p.W + i] = p.window[i] * xr;
            }
        }
        __syncthreads();
    }
    if (IN_MODE == SYN_IN_POLAR && f1 == a.F && f1 > f0) {
#pragma unroll
        for (int q = 0; q < kBlueBins; ++q) {
            const int k = tid + q * T;
            if (k < p.K) a.phase_out[c * p.K + k] = ph[q];
        }
    }
}

}  // namespace voc
