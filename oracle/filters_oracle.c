/* filters_oracle.c -- CPU restatement of the reference's frequency-domain filters and playSpeed.
 *
 * TEST INFRASTRUCTURE ONLY (part of libvocoder_oracle.so; see vocoder_oracle.c).  Follows
 *   vocoder/src/Vocoder/Filter.hs:67-157          (filters on one STFT frame)
 *   vocoder/src/Vocoder/Window.hs:28-60            (box / Blackman kernel of `envelope`)
 *   vocoder-conduit/src/Vocoder/Conduit/Filter.hs:122-136   (playSpeed)
 * one definition per function, scalar, in the reference's evaluation order.
 *
 * PARITY UNPINNED: the reference has no test, golden vector or fixture for any filter.  The
 * restatement is checked against independent numpy formulations and hand-computed cases
 * (tests/test_oracle_filters.py).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#define ORC_API __attribute__((visibility("default")))
#define ORC_PI 3.141592653589793238462643383279502884 /* GHC `pi :: Double` */

enum {
    ORC_F_AMPLIFY = 0, ORC_F_LOWPASS_BRICKWALL = 1, ORC_F_HIGHPASS_BRICKWALL = 2, ORC_F_BANDPASS_BRICKWALL = 3,
    ORC_F_BANDSTOP_BRICKWALL = 4, ORC_F_LOWPASS_BUTTERWORTH = 5, ORC_F_HIGHPASS_BUTTERWORTH = 6,
    ORC_F_BANDPASS_BUTTERWORTH = 7, ORC_F_BANDSTOP_BUTTERWORTH = 8
};

/* butterworthGain n t x = 1 / sqrt (1 + (x / t)**(2 * n))                        Filter.hs:99-100 */
static double butterworth_gain(double n, double t, double x) { return 1 / sqrt(1 + pow(x / t, 2 * n)); }

/* the function each named filter hands to linearAmplitudeFilter                   Filter.hs:80-116 */
static double gain_fn(int kind, const double *q, double x)
{
    switch (kind) {
    case ORC_F_AMPLIFY: return q[0];                                              /* :80-81 const k */
    case ORC_F_LOWPASS_BRICKWALL: return x <= q[0] ? 1.0 : 0.0;                   /* :84-85 */
    case ORC_F_HIGHPASS_BRICKWALL: return x >= q[0] ? 1.0 : 0.0;                  /* :88-89 */
    case ORC_F_BANDPASS_BRICKWALL: return (x >= q[0] && x <= q[1]) ? 1.0 : 0.0;   /* :92-93 */
    case ORC_F_BANDSTOP_BRICKWALL: return (x <= q[0] || x >= q[1]) ? 1.0 : 0.0;   /* :96-97 */
    case ORC_F_LOWPASS_BUTTERWORTH: return butterworth_gain(q[0], q[1], x);       /* :103-104 */
    case ORC_F_HIGHPASS_BUTTERWORTH: return butterworth_gain(-q[0], q[1], x);     /* :107-108 */
    case ORC_F_BANDPASS_BUTTERWORTH:                                              /* :111-112 */
        return butterworth_gain(q[0], q[2], x) * butterworth_gain(-q[0], q[1], x);
    case ORC_F_BANDSTOP_BUTTERWORTH:                                              /* :115-116 */
        return butterworth_gain(-q[0], q[1], x) + butterworth_gain(q[0], q[2], x);
    default: return 0.0 / 0.0;
    }
}

/* linearAmplitudeFilter f step (mag, ph_inc) = (zipWith (*) mag (generate K (\k -> f (step * k))), ph_inc)
 *                                                                                 Filter.hs:67-68,76-77 */
ORC_API void orc_filter_linear(int kind, const double *params, double step, int k, double *mag_io)
{
    int i;
    for (i = 0; i < k; i++) mag_io[i] = mag_io[i] * gain_fn(kind, params, step * (double)i);
}

/* interpolate n v                                                                 Filter.hs:118-125 */
ORC_API void orc_interpolate(double n, const double *v, int len, double *out)
{
    int x;
    for (x = 0; x < len; x++) {
        double xp = n * (double)x;
        double fl = floor(xp);
        if (fl + 1 >= (double)len) {
            out[x] = 0;
        } else {
            long i = (long)fl;
            double k = xp - (double)i;
            out[x] = (1 - k) * v[i] + k * v[i + 1];
        }
    }
}

/* pitchShiftInterpolate n _ (mag, ph_inc) = (interpolate n mag, map (/n) (interpolate n ph_inc))
 *                                                                                 Filter.hs:128-129 */
ORC_API void orc_filter_pitch_shift(double n, int k, double *mag_io, double *inc_io)
{
    double *t = (double *)malloc(sizeof(double) * (size_t)k);
    int i;
    orc_interpolate(n, mag_io, k, t);
    for (i = 0; i < k; i++) mag_io[i] = t[i];
    orc_interpolate(n, inc_io, k, t);
    for (i = 0; i < k; i++) inc_io[i] = t[i] / n;
    free(t);
}

/* convolution ker mag                                                             Filter.hs:132-137 */
ORC_API void orc_convolution(const double *ker, int len, const double *mag, int k, double *out)
{
    int h = len / 2, glen = k + 2 * h, i, j;
    double *gmag = (double *)calloc((size_t)glen + 1, sizeof(double)); /* replicate h 0 ++ mag ++ replicate h 0 */
    double s = 0;
    for (i = 0; i < k; i++) gmag[h + i] = mag[i];
    for (i = 0; i < len; i++) s = s + ker[i];                          /* V.sum ker */
    for (j = 0; j < k; j++) {
        double acc = 0;                                                /* V.sum of the imap'ed kernel */
        for (i = 0; i < len; i++) acc = acc + ker[i] * gmag[i + j] / s;
        out[j] = acc;
    }
    free(gmag);
}

/* convolutionFilter ker = amplitudeFilter0 (\_ -> convolution ker)                Filter.hs:72-73,140-141 */
ORC_API void orc_filter_convolution(const double *ker, int len, int k, double *mag_io, double *inc_io)
{
    double *t = (double *)malloc(sizeof(double) * (size_t)k);
    int i;
    orc_convolution(ker, len, mag_io, k, t);
    for (i = 0; i < k; i++) { mag_io[i] = t[i]; inc_io[i] = 0; }
    free(t);
}

/* the kernel of `envelope`: boxWindow for ksize <= 3, blackmanWindow otherwise   Filter.hs:148, Window.hs:28-60 */
ORC_API void orc_envelope_kernel(int ksize, double *ker)
{
    int i;
    for (i = 0; i < ksize; i++) {
        if (ksize <= 3) {
            ker[i] = 1;
        } else {
            double x = (double)i / ((double)ksize - 1);
            double p = 2 * ORC_PI * x, a = 0.16, a0 = (1 - a) / 2, a1 = 0.5, a2 = a / 2;
            ker[i] = a0 - a1 * cos(p) + a2 * cos(2 * p);
        }
    }
}

/* envelope ksize = map ((+(-ee)) . exp) . convolution ker . map (log . (+ee)),  ee = 2**(-24)
 *                                                                                 Filter.hs:144-148 */
ORC_API void orc_envelope(int ksize, const double *mag, int k, double *out)
{
    const double ee = pow(2, -24);
    double *ker = (double *)malloc(sizeof(double) * (size_t)ksize);
    double *lg = (double *)malloc(sizeof(double) * (size_t)k);
    int i;
    orc_envelope_kernel(ksize, ker);
    for (i = 0; i < k; i++) lg[i] = log(mag[i] + ee);
    orc_convolution(ker, ksize, lg, k, out);
    for (i = 0; i < k; i++) out[i] = exp(out[i]) + (-ee);
    free(lg);
    free(ker);
}

/* envelopeFilter ksize = amplitudeFilter0 (\_ -> envelope ksize)                  Filter.hs:151-152 */
ORC_API void orc_filter_envelope(int ksize, int k, double *mag_io, double *inc_io)
{
    double *t = (double *)malloc(sizeof(double) * (size_t)k);
    int i;
    orc_envelope(ksize, mag_io, k, t);
    for (i = 0; i < k; i++) { mag_io[i] = t[i]; inc_io[i] = 0; }
    free(t);
}

/* playSpeed coeff, coeff = num/den: the conduit's two loops run on the list of input frame
 * indices.  `c` is kept as an exact fraction c_num/den.  Writes the input index each emitted
 * frame repeats (l !! 0 = the frame awaited last); returns the number of frames emitted, or
 * -1 if `cap` is too small.                                                Conduit/Filter.hs:122-136 */
ORC_API int64_t orc_play_speed(int64_t num, int64_t den, int64_t nframes_in, int64_t *src, int64_t cap)
{
    __int128 c = 0;           /* f [] 0 */
    int64_t next = 0, head = -1, out = 0;
    for (;;) {
        if (c < (__int128)den) {              /* f: c < 1 -> await */
            if (next >= nframes_in) break;    /* Nothing: leftovers pushed back, conduit ends */
            head = next++;                    /* i : l */
            c += (__int128)num;               /* c + coeff */
        } else {                              /* g: c >= 1 -> yield (l !! 0), c - 1 */
            while (c >= (__int128)den) {
                if (out >= cap) return -1;
                src[out++] = head;
                c -= (__int128)den;
            }
        }
    }
    return out;
}
