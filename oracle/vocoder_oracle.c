/*
 * vocoder_oracle.c -- CPU restatement of the tilk/vocoder STFT round-trip path.
 *
 * THIS FILE IS TEST INFRASTRUCTURE.  It is the checker that the HIP path is
 * compared with; it is never linked into, imported by or called from the
 * shipped library.  Only the test-suite, the smoke check and the benchmark's
 * CPU-baseline leg may load it.
 *
 * Pinning status
 *   - framing / overlap-add (orc_gen_frames_of_e_*, orc_sum_frames_e_*,
 *     orc_frames_of_s_*, orc_sum_frames_s_*): PINNED by the reference's own
 *     list models (vocoder-conduit/test/main.hs:18-28,
 *     vocoder-dunai/test/main.hs:28-38,54-63), restated independently in the
 *     test-suite and checked on random integer inputs.
 *   - FFT / phase arithmetic (everything following Vocoder.hs): PARITY
 *     UNPINNED.  The reference ships no golden vector, known-answer test or
 *     fixture for it, and neither GHC nor FFTW exists in the build
 *     environment.  The restatement is cross-checked against pocketfft
 *     (numpy.fft), a long-double direct DFT and analytic known answers.
 *
 * Each function cites the reference lines it follows (paths are relative to
 * the reference checkout).  Build with -ffp-contract=off so that every
 * multiply and add rounds separately, as the reference's do.
 *
 * The FFT itself lives in a third-party dependency of the reference that is
 * not in its tree: Hackage `vector-fftw` (>=0.1.3.8 && <0.2,
 * vocoder/vocoder.cabal:28) over the system libfftw3.  Its published
 * conventions are restated here: `dftR2C` is the unnormalised forward
 * transform X[k] = sum_n x[n] exp(-2 pi i k n / N), k = 0..N/2, with the
 * imaginary parts of the DC bin (and, for even N, the Nyquist bin) exactly +0;
 * `Invertible.dftC2R` is the inverse of that, i.e. the unnormalised backward
 * transform multiplied by 1/N, reading N/2+1 bins and ignoring the imaginary
 * parts of the DC and Nyquist bins.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

static const double ORC_PI = 3.141592653589793238462643383279502884; /* GHC `pi :: Double` */

/* ------------------------------------------------------------------------ */
/* FFT (stand-in for vector-fftw / libfftw3; call sites Vocoder.hs:115,145,196) */
/* ------------------------------------------------------------------------ */

static int is_pow2(int n) { return n > 0 && (n & (n - 1)) == 0; }

/* In-place iterative radix-2 decimation-in-time complex FFT, sign = -1 forward,
 * +1 backward, unnormalised.  re/im are separate arrays of length n (n = 2^m). */
static void fft_pow2(int n, double *re, double *im, int sign)
{
    int i, j, len;
    double *tw_re, *tw_im;
    /* bit reversal */
    for (i = 1, j = 0; i < n; i++) {
        int bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) {
            double t = re[i]; re[i] = re[j]; re[j] = t;
            t = im[i]; im[i] = im[j]; im[j] = t;
        }
    }
    /* twiddles exp(sign * 2 pi i k / n), k < n/2, rounded from long double; the table of
     * the last size used is kept (an FFT plan would hold it) */
    {
        static __thread int cached_n = 0; /* per thread: the timing harness runs streams on several threads */
        static __thread double *c_re = NULL, *c_im = NULL;
        if (cached_n != n) {
            free(c_re);
            free(c_im);
            c_re = (double *)malloc(sizeof(double) * (size_t)(n / 2 + 1));
            c_im = (double *)malloc(sizeof(double) * (size_t)(n / 2 + 1));
            for (i = 0; i < n / 2; i++) {
                long double a = 2.0L * 3.141592653589793238462643383279502884L * (long double)i / (long double)n;
                c_re[i] = (double)cosl(a);
                c_im[i] = (double)-sinl(a);
            }
            cached_n = n;
        }
        tw_re = c_re;
        tw_im = c_im;
    }
    for (len = 2; len <= n; len <<= 1) {
        int half = len >> 1, step = n / len;
        for (i = 0; i < n; i += len) {
            for (j = 0; j < half; j++) {
                double wr = tw_re[j * step], wi = sign < 0 ? tw_im[j * step] : -tw_im[j * step];
                double xr = re[i + j + half], xi = im[i + j + half];
                double tr = xr * wr - xi * wi;
                double ti = xr * wi + xi * wr;
                re[i + j + half] = re[i + j] - tr;
                im[i + j + half] = im[i + j] - ti;
                re[i + j] += tr;
                im[i + j] += ti;
            }
        }
    }
}

/* Direct O(n^2) DFT with long-double accumulation, any n. */
static void dft_any(int n, const double *in_re, const double *in_im, double *out_re, double *out_im, int sign)
{
    int k, j;
    long double *c = (long double *)malloc(sizeof(long double) * (size_t)n);
    long double *s = (long double *)malloc(sizeof(long double) * (size_t)n);
    for (j = 0; j < n; j++) {
        long double a = 2.0L * 3.141592653589793238462643383279502884L * (long double)j / (long double)n;
        c[j] = cosl(a);
        s[j] = sign < 0 ? -sinl(a) : sinl(a);
    }
    for (k = 0; k < n; k++) {
        long double ar = 0.0L, ai = 0.0L;
        for (j = 0; j < n; j++) {
            int idx = (int)(((int64_t)k * (int64_t)j) % n);
            long double xr = in_re[j], xi = in_im ? in_im[j] : 0.0L;
            ar += xr * c[idx] - xi * s[idx];
            ai += xr * s[idx] + xi * c[idx];
        }
        out_re[k] = (double)ar;
        out_im[k] = (double)ai;
    }
    free(c);
    free(s);
}

/* `execute (plan dftR2C n)`: n reals -> n/2+1 interleaved complex, unnormalised. */
ORC_API void orc_fft_r2c(int n, const double *in, double *out)
{
    int k, K = n / 2 + 1;
    double *re = (double *)malloc(sizeof(double) * (size_t)n * 2);
    double *im = re + n;
    if (is_pow2(n)) {
        memcpy(re, in, sizeof(double) * (size_t)n);
        memset(im, 0, sizeof(double) * (size_t)n);
        fft_pow2(n, re, im, -1);
    } else {
        dft_any(n, in, NULL, re, im, -1);
    }
    for (k = 0; k < K; k++) {
        out[2 * k] = re[k];
        out[2 * k + 1] = im[k];
    }
    /* r2c convention: DC (and Nyquist for even n) are purely real, +0 imaginary. */
    out[1] = 0.0;
    if (n % 2 == 0) out[2 * (K - 1) + 1] = 0.0;
    free(re);
}

/* `execute (plan dftC2R n)` from Numeric.FFT.Vector.Invertible: n/2+1 complex
 * -> n reals, backward transform multiplied by 1/n. */
ORC_API void orc_fft_c2r(int n, const double *in, double *out)
{
    int k, K = n / 2 + 1;
    double scale = 1.0 / (double)n;
    double *re = (double *)malloc(sizeof(double) * (size_t)n * 4);
    double *im = re + n, *ore = re + 2 * n, *oim = re + 3 * n;
    for (k = 0; k < K; k++) {
        re[k] = in[2 * k];
        im[k] = in[2 * k + 1];
    }
    im[0] = 0.0;
    if (n % 2 == 0) im[K - 1] = 0.0;
    for (k = K; k < n; k++) { /* Hermitian extension */
        re[k] = re[n - k];
        im[k] = -im[n - k];
    }
    if (is_pow2(n)) {
        fft_pow2(n, re, im, +1);
        for (k = 0; k < n; k++) out[k] = re[k] * scale;
    } else {
        dft_any(n, re, im, ore, oim, +1);
        for (k = 0; k < n; k++) out[k] = ore[k] * scale;
    }
    free(re);
}

/* Second opinion: forward real DFT entirely in long double, any n. */
ORC_API void orc_dft_r2c_ld(int n, const double *in, double *out)
{
    int k, K = n / 2 + 1;
    double *re = (double *)malloc(sizeof(double) * (size_t)n * 2);
    double *im = re + n;
    dft_any(n, in, NULL, re, im, -1);
    for (k = 0; k < K; k++) {
        out[2 * k] = re[k];
        out[2 * k + 1] = im[k];
    }
    out[1] = 0.0;
    if (n % 2 == 0) out[2 * (K - 1) + 1] = 0.0;
    free(re);
}

/* ------------------------------------------------------------------------ */
/* Vocoder.hs                                                               */
/* ------------------------------------------------------------------------ */

/* applyWindow = V.zipWith (*)                                 Vocoder.hs:118-119 */
ORC_API void orc_apply_window(int w, const double *window, const double *frame, double *out)
{
    int i;
    for (i = 0; i < w; i++) out[i] = window[i] * frame[i];
}

/* rewind: splitAt (len div 2), swap halves                    Vocoder.hs:124-125 */
ORC_API void orc_rewind(int n, const double *in, double *out)
{
    int h = n / 2, i;
    for (i = 0; i < n - h; i++) out[i] = in[h + i];
    for (i = 0; i < h; i++) out[n - h + i] = in[i];
}

/* addZeroPadding: halfdiff = diff - diff div 2 zeros in front  Vocoder.hs:128-139
 * returns -1 where the reference calls `error` (input longer than target). */
ORC_API int orc_add_zero_padding(int len, int l, const double *v, double *out)
{
    int diff = len - l, halfdiff, i;
    if (diff < 0) return -1;
    halfdiff = diff - diff / 2;
    for (i = 0; i < halfdiff; i++) out[i] = 0.0;
    for (i = 0; i < l; i++) out[halfdiff + i] = v[i];
    for (i = halfdiff + l; i < len; i++) out[i] = 0.0;
    return 0;
}

/* cutCenter: take len from offset (length - len) div 2         Vocoder.hs:198-200 */
ORC_API void orc_cut_center(int n, int len, const double *in, double *out)
{
    int off = (n - len) / 2, i;
    for (i = 0; i < len; i++) out[i] = in[off + i];
}

/* doFFT = execute . rewind . addZeroPadding N . applyWindow    Vocoder.hs:143-145 */
ORC_API int orc_do_fft(int n, int w, const double *window, const double *frame, double *out)
{
    double *a = (double *)malloc(sizeof(double) * (size_t)(w + 2 * n + 2));
    double *b = a + w, *c = b + n;
    int rc;
    orc_apply_window(w, window, frame, a);
    rc = orc_add_zero_padding(n, w, a, b);
    if (rc == 0) {
        orc_rewind(n, b, c);
        orc_fft_r2c(n, c, out);
    }
    free(a);
    return rc;
}

/* doIFFT = applyWindow . cutCenter W . rewind . execute        Vocoder.hs:194-196
 * (the same left rotation is applied again, not its inverse: for odd N this
 * leaves a one-sample circular shift, as in the reference) */
ORC_API void orc_do_ifft(int n, int w, const double *window, const double *spec, double *out)
{
    double *a = (double *)malloc(sizeof(double) * (size_t)(2 * n + w + 2));
    double *b = a + n, *c = b + n;
    orc_fft_c2r(n, spec, a);
    orc_rewind(n, a, b);
    orc_cut_center(n, w, b, c);
    orc_apply_window(w, window, c, out);
    free(a);
}

/* Data.Complex.magnitude: scaleFloat k (sqrt (sqr (scaleFloat mk x) + sqr (scaleFloat mk y)))
 * with k = max (exponent x) (exponent y); `exponent` follows the frexp convention
 * and is 0 for 0.                                                Vocoder.hs:218 */
static double hs_magnitude(double x, double y)
{
    int ex = 0, ey = 0, k;
    double sx, sy;
    if (x != 0.0) (void)frexp(x, &ex);
    if (y != 0.0) (void)frexp(y, &ey);
    k = ex > ey ? ex : ey;
    sx = ldexp(x, -k);
    sy = ldexp(y, -k);
    return ldexp(sqrt(sx * sx + sy * sy), k);
}

static int is_neg_zero(double v) { return v == 0.0 && signbit(v); }

/* GHC's default `atan2` of class RealFloat (base, GHC.Float), on libm atan. */
static double hs_atan2(double y, double x)
{
    if (x > 0.0) return atan(y / x);
    if (x == 0.0 && y > 0.0) return ORC_PI / 2;
    if (x < 0.0 && y > 0.0) return ORC_PI + atan(y / x);
    if ((x <= 0.0 && y < 0.0) || (x < 0.0 && is_neg_zero(y)) || (is_neg_zero(x) && is_neg_zero(y)))
        return -hs_atan2(-y, x);
    if (y == 0.0 && (x < 0.0 || is_neg_zero(x))) return ORC_PI;
    if (x == 0.0 && y == 0.0) return y;
    return x + y;
}

/* Data.Complex.phase: phase (0 :+ 0) = 0 ; phase (x:+y) = atan2 y x */
static double hs_phase(double x, double y)
{
    if (x == 0.0 && y == 0.0) return 0.0;
    return hs_atan2(y, x);
}

ORC_API double orc_magnitude(double re, double im) { return hs_magnitude(re, im); }
ORC_API double orc_phase(double re, double im) { return hs_phase(re, im); }

/* frameFromComplex = V.map magnitude &&& V.map phase           Vocoder.hs:217-218 */
ORC_API void orc_frame_from_complex(int k, const double *spec, double *mag, double *ph)
{
    int i;
    for (i = 0; i < k; i++) {
        mag[i] = hs_magnitude(spec[2 * i], spec[2 * i + 1]);
        ph[i] = hs_phase(spec[2 * i], spec[2 * i + 1]);
    }
}

/* frameToComplex = zipWith mkPolar ; mkPolar r t = r*cos t :+ r*sin t   Vocoder.hs:213-214 */
ORC_API void orc_frame_to_complex(int k, const double *mag, const double *ph, double *spec)
{
    int i;
    for (i = 0; i < k; i++) {
        spec[2 * i] = mag[i] * cos(ph[i]);
        spec[2 * i + 1] = mag[i] * sin(ph[i]);
    }
}

/* Data.Fixed.mod': n - fromInteger f * d, f = floor (toRational n / toRational d).
 * The floor is of the EXACT quotient; it is recovered here from the rounded
 * quotient by checking the sign of the exactly-rounded remainder (fma). */
static double hs_mod(double n, double d)
{
    double f = floor(n / d);
    if (fma(-f, d, n) < 0.0) f -= 1.0;
    else if (fma(-(f + 1.0), d, n) >= 0.0) f += 1.0;
    return n - f * d;
}

/* wrap e = (e+pi) `mod'` (2*pi) - pi                            Vocoder.hs:166-167 */
ORC_API double orc_wrap(double e) { return hs_mod(e + ORC_PI, 2 * ORC_PI) - ORC_PI; }

/* calcPhaseInc eN hop k ph_diff = (omega + wrap (ph_diff - omega)) / hop
 * omega = (2*pi*k*hop) / eN, evaluated left to right           Vocoder.hs:169-173 */
ORC_API double orc_calc_phase_inc(int n, int hop, int k, double ph_diff)
{
    double omega = (2 * ORC_PI * (double)k * (double)hop) / (double)n;
    return (omega + orc_wrap(ph_diff - omega)) / (double)hop;
}

/* analysisStep                                                  Vocoder.hs:158-163 */
ORC_API void orc_analysis_step(int hop, int n, int k, const double *prev_ph, const double *spec,
                               double *ph_out, double *mag, double *ph_inc)
{
    int i;
    for (i = 0; i < k; i++) {
        double m = hs_magnitude(spec[2 * i], spec[2 * i + 1]);
        double p = hs_phase(spec[2 * i], spec[2 * i + 1]);
        double d = p - prev_ph[i];
        mag[i] = m;
        ph_inc[i] = orc_calc_phase_inc(n, hop, i, d);
        ph_out[i] = p; /* written last: ph_out may alias prev_ph */
    }
}

/* analysisBlock par prev_ph vec = analysisStep H N prev_ph (doFFT par vec)   Vocoder.hs:153-154 */
ORC_API int orc_analysis_block(int n, int w, int hop, const double *window, const double *prev_ph,
                               const double *frame, double *ph_out, double *mag, double *ph_inc)
{
    int K = n / 2 + 1, rc;
    double *spec = (double *)malloc(sizeof(double) * 2 * (size_t)K);
    rc = orc_do_fft(n, w, window, frame, spec);
    if (rc == 0) orc_analysis_step(hop, n, K, prev_ph, spec, ph_out, mag, ph_inc);
    free(spec);
    return rc;
}

/* analysisStage = mapAccumL (analysisBlock par)                 Vocoder.hs:149-150
 * frames: nframes x w, mag / ph_inc: nframes x K, phase_io: K (in: initial, out: final) */
ORC_API int orc_analysis_stage(int n, int w, int hop, const double *window, double *phase_io,
                               const double *frames, int64_t nframes, double *mag, double *ph_inc)
{
    int K = n / 2 + 1;
    int64_t j;
    for (j = 0; j < nframes; j++) {
        int rc = orc_analysis_block(n, w, hop, window, phase_io, frames + j * w, phase_io,
                                    mag + j * K, ph_inc + j * K);
        if (rc) return rc;
    }
    return 0;
}

/* synthesisStep: new_ph = ph + ph_inc * hop ; frameToComplex (mag, new_ph)   Vocoder.hs:186-190 */
ORC_API void orc_synthesis_step(int hop, int k, const double *ph, const double *mag, const double *ph_inc,
                                double *ph_out, double *spec)
{
    int i;
    for (i = 0; i < k; i++) {
        double t = ph_inc[i] * (double)hop;
        double np = ph[i] + t;
        spec[2 * i] = mag[i] * cos(np);
        spec[2 * i + 1] = mag[i] * sin(np);
        ph_out[i] = np;
    }
}

/* synthesisBlock = (id *** doIFFT par) . synthesisStep H ph     Vocoder.hs:181-182 */
ORC_API void orc_synthesis_block(int n, int w, int hop, const double *window, const double *ph,
                                 const double *mag, const double *ph_inc, double *ph_out, double *frame_out)
{
    int K = n / 2 + 1;
    double *spec = (double *)malloc(sizeof(double) * 2 * (size_t)K);
    orc_synthesis_step(hop, K, ph, mag, ph_inc, ph_out, spec);
    orc_do_ifft(n, w, window, spec, frame_out);
    free(spec);
}

/* synthesisStage = mapAccumL (synthesisBlock par)               Vocoder.hs:177-178 */
ORC_API void orc_synthesis_stage(int n, int w, int hop, const double *window, double *phase_io,
                                 const double *mag, const double *ph_inc, int64_t nframes, double *frames_out)
{
    int K = n / 2 + 1;
    int64_t j;
    for (j = 0; j < nframes; j++)
        orc_synthesis_block(n, w, hop, window, phase_io, mag + j * K, ph_inc + j * K, phase_io,
                            frames_out + j * w);
}

/* volumeCoeff = hop / V.sum (V.map (**2) window)                Vocoder.hs:209-210
 * (**) on Double is libm pow; V.sum is a left fold from 0. */
ORC_API double orc_volume_coeff(int hop, int w, const double *window)
{
    double s = 0.0;
    int i;
    for (i = 0; i < w; i++) s = s + pow(window[i], 2.0);
    return (double)hop / s;
}

/* addFrames f1 f2 = frameFromComplex (zipWith (+) (frameToComplex f1) (frameToComplex f2))   Vocoder.hs:220-222 */
ORC_API void orc_add_frames(int k, const double *mag1, const double *inc1, const double *mag2, const double *inc2,
                            double *mag_out, double *inc_out)
{
    int i;
    for (i = 0; i < k; i++) {
        double re = mag1[i] * cos(inc1[i]) + mag2[i] * cos(inc2[i]);
        double im = mag1[i] * sin(inc1[i]) + mag2[i] * sin(inc2[i]);
        mag_out[i] = hs_magnitude(re, im);
        inc_out[i] = hs_phase(re, im);
    }
}

/* ------------------------------------------------------------------------ */
/* Vocoder/Window.hs                                                        */
/* ------------------------------------------------------------------------ */

enum {
    ORC_WIN_BOX = 0, ORC_WIN_TRIANGLE = 1, ORC_WIN_HAMMING = 2, ORC_WIN_HANN = 3,
    ORC_WIN_GEN_BLACKMAN = 4, ORC_WIN_BLACKMAN = 5, ORC_WIN_EXACT_BLACKMAN = 6,
    ORC_WIN_LANCZOS = 7, ORC_WIN_FLATTOP = 8
};

static double win_fn(int kind, double alpha, double x)
{
    switch (kind) {
    case ORC_WIN_BOX: return 1.0;                                            /* Window.hs:32-33 */
    case ORC_WIN_TRIANGLE: return 2 * (0.5 - fabs(x - 0.5));                 /* Window.hs:36-37 */
    case ORC_WIN_HAMMING: return 25.0 / 46.0 - 21.0 / 46.0 * cos(2 * ORC_PI * x); /* Window.hs:40-44 */
    case ORC_WIN_HANN: return 0.5 * (1 - cos(2 * ORC_PI * x));               /* Window.hs:47-48 */
    case ORC_WIN_BLACKMAN: alpha = 0.16; /* fallthrough */                   /* Window.hs:59-60 */
    case ORC_WIN_GEN_BLACKMAN: {                                             /* Window.hs:51-56 */
        double p = 2 * ORC_PI * x, a0 = (1 - alpha) / 2, a1 = 0.5, a2 = alpha / 2;
        return a0 - a1 * cos(p) + a2 * cos(2 * p);
    }
    case ORC_WIN_EXACT_BLACKMAN: {                                           /* Window.hs:63-68 */
        double p = 2 * ORC_PI * x;
        return 7938.0 / 18608.0 - 9240.0 / 18608.0 * cos(p) + 1430.0 / 18608.0 * cos(2 * p);
    }
    case ORC_WIN_LANCZOS: {                                                  /* Window.hs:71-75 */
        double s = 2 * x - 1;
        return s == 0.0 ? 1.0 : sin(ORC_PI * s) / (ORC_PI * s);
    }
    case ORC_WIN_FLATTOP:                                                    /* Window.hs:78-85 */
        return 0.21557895 - 0.41663158 * cos(2 * ORC_PI * x) + 0.277263158 * cos(4 * ORC_PI * x)
               - 0.083578947 * cos(6 * ORC_PI * x) + 0.006947368 * cos(8 * ORC_PI * x);
    default: return 0.0 / 0.0;
    }
}

/* makeWindow f n = generate n (\k -> f (k / (n - 1)))           Window.hs:28-29 */
ORC_API void orc_make_window(int kind, double alpha, int n, double *out)
{
    int k;
    for (k = 0; k < n; k++) out[k] = win_fn(kind, alpha, (double)k / ((double)n - 1));
}

/* ------------------------------------------------------------------------ */
/* Framing and overlap-add: instantiated for double and int64                */
/* ------------------------------------------------------------------------ */

/* Publication schedule of sumFramesE (Frames.hs:49-53), data left out: published[f] = chunks yielded
 * once frame f has been awaited.  `process` appends the frame at offset `sofar`; `process2` publishes
 * while a whole chunk lies before that offset (sofar >= chunkSize), then the offset moves on by a hop. */
ORC_API void orc_sum_frames_e_schedule(int64_t chunk_size, int64_t hop, int64_t nframes, int64_t *published)
{
    int64_t sofar = 0, nch = 0, f;
    for (f = 0; f < nframes; f++) {
        while (sofar >= chunk_size) {
            nch++;
            sofar -= chunk_size;
        }
        sofar += hop;
        published[f] = nch;
    }
}


#define T double
#define SUF(name) name##_f64
#include "frames_generic.inc"
#undef T
#undef SUF

#define T int64_t
#define SUF(name) name##_i64
#include "frames_generic.inc"
#undef T
#undef SUF

/* ------------------------------------------------------------------------ */
/* Whole pipelines                                                          */
/* ------------------------------------------------------------------------ */

/* Number of frames genFramesOfE yields for a total buffer of len samples:
 * offsets 0, H, .. <= len - W                                   Frames.hs:34-35 */
ORC_API int64_t orc_num_frames(int64_t len, int w, int hop)
{
    if (len < w) return 0;
    return (len - w) / hop + 1;
}

/* Output length of sumFramesE chunk hop over nframes frames of w samples:
 * chunk count = lastIndex div chunk + 1      vocoder-conduit/test/main.hs:25-28 */
ORC_API int64_t orc_ola_length(int64_t nframes, int w, int hop, int64_t chunk)
{
    int64_t last;
    if (nframes <= 0) return 0;
    last = (nframes - 1) * hop + w - 1;
    return (last / chunk + 1) * chunk;
}

/* Vocoder.Conduit `processWith q phs idFilter` for one channel  Conduit.hs:46-53:
 *   genFramesOfE W H q  |>  analysis  |>  volumeFix (mag * volumeCoeff)
 *   |>  (identity filter)  |>  synthesis  |>  sumFramesE chunk H
 * x holds q ++ input (len samples).  y must hold orc_ola_length(F, W, H, chunk)
 * samples.  ana_phase / syn_phase: K each, in = initial, out = final.
 * leftover receives x[F*H ..] (Frames.hs:36-37); returns its length via *leftover_len.
 * Optionally stores the STFT (mag after volumeFix, ph_inc), F x K each. */
ORC_API int orc_roundtrip(int n, int w, int hop, const double *window, const double *x, int64_t len,
                          int64_t chunk, double *ana_phase, double *syn_phase, double *y,
                          double *leftover, int64_t *leftover_len, double *stft_mag, double *stft_inc)
{
    int K = n / 2 + 1, i;
    int64_t F = orc_num_frames(len, w, hop), j, ylen = orc_ola_length(F, w, hop, chunk);
    double vc = orc_volume_coeff(hop, w, window);
    double *mag = (double *)malloc(sizeof(double) * (size_t)(2 * K + w));
    double *inc = mag + K, *fr = inc + K;
    for (j = 0; j < ylen; j++) y[j] = 0.0;
    for (j = 0; j < F; j++) {
        int rc = orc_analysis_block(n, w, hop, window, ana_phase, x + j * hop, ana_phase, mag, inc);
        if (rc) { free(mag); return rc; }
        for (i = 0; i < K; i++) mag[i] = mag[i] * vc;      /* volumeFix, Conduit.hs:34-35 */
        if (stft_mag) memcpy(stft_mag + j * K, mag, sizeof(double) * (size_t)K);
        if (stft_inc) memcpy(stft_inc + j * K, inc, sizeof(double) * (size_t)K);
        orc_synthesis_block(n, w, hop, window, syn_phase, mag, inc, syn_phase, fr);
        /* sumFramesE: out[i] = sum over live frames, oldest first (Frames.hs:44-45);
         * accumulating frames in arrival order performs the same left fold. */
        for (i = 0; i < w; i++) y[j * hop + i] = y[j * hop + i] + fr[i];
    }
    if (leftover_len) {
        /* dropcnt = ((len - W) div H) * H + H with flooring div; a non-positive
         * count keeps the whole buffer                          Frames.hs:36-37 */
        int64_t q = len - w, d = q / hop, dropcnt, keep;
        if ((q % hop != 0) && (q < 0)) d -= 1; /* floor division */
        dropcnt = d * hop + hop;
        if (dropcnt < 0) dropcnt = 0;
        if (dropcnt > len) dropcnt = len;
        keep = len - dropcnt;
        if (leftover) memcpy(leftover, x + dropcnt, sizeof(double) * (size_t)keep);
        *leftover_len = keep;
    }
    free(mag);
    return 0;
}
