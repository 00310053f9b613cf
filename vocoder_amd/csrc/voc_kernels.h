// voc_kernels.h -- gfx950 kernels of the phase-vocoder round trip.
//
// One workgroup owns one (channel, run of consecutive frames) tile and walks its
// frames in order.  A frame of N reals is transformed as an M = N/2 point complex
// FFT (even samples -> re, odd samples -> im) held in LDS, Stockham radix-8/4/2
// passes with the butterflies in registers, followed by the real-FFT split.
// Everything around the FFT (window, zero pad, rewind, polar conversion, phase
// increment, phase accumulation, polar->complex, cut, window, overlap-add) is
// fused into the same kernel so that a frame touches HBM once on the way in
// and once on the way out.
//
// Arithmetic notes: the file is compiled with -ffp-contract=off; every fused
// multiply-add below is an explicit fma().  Operations whose rounding sequence
// the reference fixes (phase increment, wrap, phase accumulation, magnitude
// scaling, overlap-add order) are written as separate multiplies and adds in
// the reference's order.
#pragma once

namespace voc {

#define VOC_DEV __device__ __forceinline__

#ifdef VOC_EMU
#define VOC_SMEM(type, name) type *name = reinterpret_cast<type *>(emu::tls.smem)
#else
#define VOC_SMEM(type, name)                                                        \
    extern __shared__ __attribute__((aligned(16))) unsigned char voc_smem_raw[];    \
    type *name = reinterpret_cast<type *>(voc_smem_raw)
#endif

constexpr double kPi = 3.141592653589793238462643383279502884;  // GHC `pi :: Double`
constexpr double kTwoPi = 2.0 * kPi;
constexpr double kInvTwoPi = 0.15915494309189533576888376337251436;
constexpr double kSqrtHalf = 0.70710678118654752440084436210484904;

// The kernels of this file exist in two precisions: R = double (the reference's arithmetic, bit
// for bit where the reference fixes it) and R = float (a reduced-precision variant with its own,
// looser, parity bar).  cplx<R> is the matching two-component vector type.
template <class R> struct CplxOf;
template <> struct CplxOf<double> { using type = double2; };
template <> struct CplxOf<float> { using type = float2; };
template <class R> using cplx = typename CplxOf<R>::type;
VOC_DEV double2 mk2(double x, double y) { return make_double2(x, y); }
VOC_DEV float2 mk2(float x, float y) { return make_float2(x, y); }

template <class R>
struct ParamsT {
    int N, W, H, K, M;
    int hd;        // zeros addZeroPadding puts in front: diff - diff div 2        (Vocoder.hs:138)
    int off;       // cutCenter offset (N - W) div 2                               (Vocoder.hs:200)
    int hop_pow2;  // H is a power of two: x / H == x * (1/H) exactly
    R hop_d, inv_hop, inv_n;
    const R *window;       // [W]
    const R *window_n;     // [W]  window[i] * (1/N): exact where N is a power of two (contexts only; null elsewhere)
    const cplx<R> *twP;    // [M-1]    per-pass twiddles, see fft_passes
    const cplx<R> *twS;    // [M/2+1]  exp(-2 pi i k / N)
    const cplx<R> *cs64;   // [64]     (cos, sin)(2 pi j / 64): the circle in 64 steps (double-precision contexts; else null)
    double omega1;         // 2 pi H / N when H and N are powers of two (the scaling is then exact), else 0
    // single-precision contexts with omega1 != 0: omega1 as the sum of two floats, and N/H - 1 (bin k advances
    // by (k mod N/H) / (N/H) of a turn per hop, modulo whole turns)
    float om1_hi, om1_lo;
    int turn_mask;
};
using Params = ParamsT<double>;

template <int LOG2M>
struct Cfg {
    static constexpr int M = 1 << LOG2M;
    static constexpr int T = (M / 8) < 64 ? 64 : ((M / 8) > 1024 ? 1024 : (M / 8));  // threads per frame
    static constexpr int NP = ((M / 2) / T) < 1 ? 1 : ((M / 2) / T);                 // (k, M-k) bin pairs per thread
    static constexpr int NQ = (M / T) < 1 ? 1 : (M / T);                             // complex points per thread
    static constexpr int LDS_CPLX = M + (M >> 3) + 2;
};

// ------------------------------------------------------------------ complex helpers
template <class C> VOC_DEV C cadd(C a, C b) { return mk2(a.x + b.x, a.y + b.y); }
template <class C> VOC_DEV C csub(C a, C b) { return mk2(a.x - b.x, a.y - b.y); }
template <class C> VOC_DEV C cmul(C a, C b)
{
    return mk2(fma(a.x, b.x, -(a.y * b.y)), fma(a.x, b.y, a.y * b.x));
}
template <class C> VOC_DEV C cmul_conj(C a, C b)  // a * conj(b)
{
    return mk2(fma(a.x, b.x, a.y * b.y), fma(a.y, b.x, -(a.x * b.y)));
}
template <class C> VOC_DEV C mul_mi(C a) { return mk2(a.y, -a.x); }  // a * (-i)
template <class C> VOC_DEV C cconj(C a) { return mk2(a.x, -a.y); }
// a * exp(-i pi/4) and a * exp(-3 i pi/4)
template <class C> VOC_DEV C mul_w8(C a)
{
    using R = decltype(a.x);
    return mk2((a.x + a.y) * (R)kSqrtHalf, (a.y - a.x) * (R)kSqrtHalf);
}
template <class C> VOC_DEV C mul_w8_3(C a)
{
    using R = decltype(a.x);
    return mk2((a.y - a.x) * (R)kSqrtHalf, -((a.x + a.y) * (R)kSqrtHalf));
}

// two floats in one 64-bit register pair: the operand type of the packed instructions
#ifdef VOC_EMU
struct f32x2 { float x, y; };
VOC_DEV f32x2 operator+(f32x2 a, f32x2 b) { return f32x2{a.x + b.x, a.y + b.y}; }
VOC_DEV f32x2 operator-(f32x2 a, f32x2 b) { return f32x2{a.x - b.x, a.y - b.y}; }
VOC_DEV f32x2 operator*(f32x2 a, f32x2 b) { return f32x2{a.x * b.x, a.y * b.y}; }
VOC_DEV f32x2 operator-(f32x2 a) { return f32x2{-a.x, -a.y}; }
VOC_DEV f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return f32x2{fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y)}; }
#else
typedef float f32x2 __attribute__((ext_vector_type(2)));
VOC_DEV f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
#endif
VOC_DEV f32x2 pk2(float a, float b)
{
    f32x2 r;
    r.x = a;
    r.y = b;
    return r;
}
VOC_DEV f32x2 splat2(float c) { return pk2(c, c); }

// Single-precision complex arithmetic on packed registers: a complex number is one (re, im) register pair, a
// complex sum one v_pk_add_f32, a product two instructions (the swaps and sign changes of i are operand
// modifiers of the packed instructions).  Same roundings as the component-wise forms above.
#ifndef VOC_F32_CPLX_PACKED
#define VOC_F32_CPLX_PACKED 1
#endif
#if VOC_F32_CPLX_PACKED
VOC_DEV f32x2 v2(float2 a) { return pk2(a.x, a.y); }
VOC_DEV float2 c2(f32x2 a) { return make_float2(a.x, a.y); }
// (a.y, a.x): an operand modifier of the consuming instruction.  A sign change of one component is not a modifier
// the compiler forms, so those are products with (1, -1) / (-1, 1) fused into the sum they feed: x + (1, -1) y
// rounds like x.x + y.x, x.y - y.y.
#ifdef VOC_EMU
VOC_DEV f32x2 swap2(f32x2 a) { return f32x2{a.y, a.x}; }
#else
VOC_DEV f32x2 swap2(f32x2 a) { return __builtin_shufflevector(a, a, 1, 0); }
#endif
VOC_DEV f32x2 sign_pm() { return pk2(1.0f, -1.0f); }
VOC_DEV f32x2 sign_mp() { return pk2(-1.0f, 1.0f); }
VOC_DEV float2 cadd(float2 a, float2 b) { return c2(v2(a) + v2(b)); }
VOC_DEV float2 csub(float2 a, float2 b) { return c2(v2(a) - v2(b)); }
VOC_DEV float2 cmul(float2 a, float2 b)
{
    // (a.x b.x - a.y b.y, a.x b.y + a.y b.x)
    const f32x2 an = splat2(a.y) * sign_mp();
    return c2(fma2(splat2(a.x), v2(b), an * swap2(v2(b))));
}
VOC_DEV float2 cmul_conj(float2 a, float2 b)
{
    // (a.x b.x + a.y b.y, a.y b.x - a.x b.y)
    const f32x2 bn = splat2(b.y) * sign_pm();
    return c2(fma2(v2(a), splat2(b.x), swap2(v2(a)) * bn));
}
VOC_DEV float2 mul_w8(float2 a) { return c2(fma2(swap2(v2(a)), sign_pm(), v2(a)) * splat2((float)kSqrtHalf)); }
VOC_DEV float2 mul_w8_3(float2 a) { return c2(fma2(swap2(v2(a)), sign_pm(), -v2(a)) * splat2((float)kSqrtHalf)); }
#endif
VOC_DEV int lds_pad(int i) { return i + (i >> 3); }

template <class C> VOC_DEV void bf2(C &a, C &b)
{
    C t = a;
    a = cadd(t, b);
    b = csub(t, b);
}

// (a, b) <- (a - i b, a + i b): bf2 after b <- -i b
template <class C> VOC_DEV void bf2_mi(C &a, C &b)
{
    b = mul_mi(b);
    bf2(a, b);
}
#if VOC_F32_CPLX_PACKED
VOC_DEV void bf2_mi(float2 &a, float2 &b)
{
    const f32x2 t = v2(a), u = swap2(v2(b));
    a = c2(fma2(u, sign_pm(), t));
    b = c2(fma2(u, sign_mp(), t));
}
#endif

// Forward DFT of RAD points in place, natural order in and out.
template <class C> VOC_DEV void dft_n(C (&v)[2]) { bf2(v[0], v[1]); }

template <class C> VOC_DEV void dft_n(C (&v)[4])
{
    bf2(v[0], v[2]);
    bf2(v[1], v[3]);
    bf2(v[0], v[1]);
    bf2_mi(v[2], v[3]);
    C t = v[1];  // bit-reversed -> natural
    v[1] = v[2];
    v[2] = t;
}

template <class C> VOC_DEV void dft_n(C (&v)[8])
{
    bf2(v[0], v[4]);
    bf2(v[1], v[5]);
    bf2(v[2], v[6]);
    bf2(v[3], v[7]);
    // twiddles W8^1, W8^2, W8^3 on the lower half
    v[5] = mul_w8(v[5]);
    v[7] = mul_w8_3(v[7]);
    bf2(v[0], v[2]);
    bf2(v[1], v[3]);
    bf2_mi(v[4], v[6]);
    bf2(v[5], v[7]);
    bf2(v[0], v[1]);
    bf2_mi(v[2], v[3]);
    bf2(v[4], v[5]);
    bf2_mi(v[6], v[7]);
    // registers now hold y0 y4 y2 y6 y1 y5 y3 y7
    C t;
    t = v[1]; v[1] = v[4]; v[4] = t;
    t = v[3]; v[3] = v[6]; v[6] = t;
}

template <int RAD, class C> VOC_DEV void dft(C (&v)[RAD]) { dft_n(v); }

// One Stockham pass of radix RAD at stride NS over M points.  Butterfly i reads
// in[i + j*M/RAD], multiplies by exp(-2 pi i j k / (NS*RAD)) with k = i mod NS, and
// writes out[(i-k)*RAD + k + j*NS].  Reads come through `ld` (LDS or a fused global
// loader for the first pass); writes go to LDS in place, fenced by two barriers.
//
// Twiddles: twP holds, pass after pass, the factors of every pass with NS > 1 as
// [j-1][k] (k fastest), the pass at stride NS starting at entry NS - 1 -- so for a fixed j
// consecutive lanes read consecutive 16-byte entries, and the seven loads of a butterfly
// share one per-lane offset.
// LDS: element e lives at e + e/8 (one 16-byte pad per 8 elements keeps the stride-8 stores
// of the first pass off a single bank); when a stride is a multiple of 8 the padded index
// is linear in j, which the code spells out so that the j-th access is base + constant.
template <int M, int T, int NS, class C, class Loader>
VOC_DEV void fft_passes(C *lds, const C *__restrict__ twP, int tid, Loader ld)
{
    constexpr int REM = M / NS;
    constexpr int RAD = REM >= 8 ? 8 : REM;
    constexpr int NBF = M / RAD;
    constexpr int NB = (NBF + T - 1) / T;
    constexpr bool LIN_ST = (NS % 8 == 0) || (NS == 1 && RAD == 8);
    C v[NB][RAD];
#pragma unroll
    for (int q = 0; q < NB; ++q) {
        const int i = tid + q * T;
        if ((NBF % T == 0) || i < NBF) {
#pragma unroll
            for (int j = 0; j < RAD; ++j) v[q][j] = ld(i, j);
            if (NS > 1) {
                const unsigned k = (unsigned)i & (unsigned)(NS - 1);
#pragma unroll
                for (int j = 1; j < RAD; ++j) {
                    const C *tj = twP + (NS - 1) + (j - 1) * NS;  // uniform base of row j
                    v[q][j] = cmul(v[q][j], tj[k]);
                }
            }
            dft<RAD>(v[q]);
        }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NB; ++q) {
        const int i = tid + q * T;
        if ((NBF % T == 0) || i < NBF) {
            const int k = i & (NS - 1);
            const int j0 = (i - k) * RAD + k;
            if (LIN_ST) {
                C *dst = lds + lds_pad(j0);
#pragma unroll
                for (int j = 0; j < RAD; ++j) dst[j * (NS + NS / 8)] = v[q][j];
            } else {
#pragma unroll
                for (int j = 0; j < RAD; ++j) lds[lds_pad(j0 + j * NS)] = v[q][j];
            }
        }
    }
    __syncthreads();
    if constexpr (NS * RAD < M) {
        constexpr int NBF2 = M / ((M / (NS * RAD)) >= 8 ? 8 : (M / (NS * RAD)));
        fft_passes<M, T, NS * RAD>(lds, twP, tid, [lds](int i, int j) {
            if (NBF2 % 8 == 0) return (lds + lds_pad(i))[j * (NBF2 + NBF2 / 8)];
            return lds[lds_pad(i + j * NBF2)];
        });
    }
}

// first-pass loader over LDS: element i + j*NBF of the M-point array
template <int M, class C = double2>
struct LdsLoader {
    const C *lds;
    static constexpr int RAD0 = M >= 8 ? 8 : M;
    static constexpr int NBF = M / RAD0;
    VOC_DEV C operator()(int i, int j) const
    {
        if (NBF % 8 == 0) return (lds + lds_pad(i))[j * (NBF + NBF / 8)];
        return lds[lds_pad(i + j * NBF)];
    }
};

// ------------------------------------------------------------------ per-bin arithmetic
//
// The transcendental functions are written out here instead of calling the device math library:
// the path spends most of its instructions in them, and each is needed only on a restricted
// domain.  All are accurate to about one ulp (checked against 60-digit arithmetic by
// tools/gen_poly.py and the parity tests); inputs outside the fast domain (huge, tiny, inf, nan)
// take the library routine.

VOC_DEV double approx_rcp(double b)
{
#ifdef VOC_EMU
    return 1.0 / b;
#else
    return __builtin_amdgcn_rcp(b);
#endif
}

// a / b for b in [2^-4, 2^4], |a| <= b (and a = 0 with any normal b > 0): hardware reciprocal seed
// (relative error below 2^-24 on gfx950, measured over a dense sweep by tools/ubench/seed_precision.hip),
// one Newton step (2^-48), one residual correction of the quotient -- on that sweep the bits of the
// correctly rounded quotient.
// LEAN (here and in sqrt_scaled, atan_poly, hs_polar): the form the fused kernels use.  It stops one
// step earlier -- no residual correction of the quotient and of the square root (2^-48 instead of
// correctly rounded), a degree-8 instead of a degree-10 arctangent polynomial (4e-15 instead of 5e-18
// before rounding) -- which leaves magnitudes within 3e-15 relative and phases within 3e-14 rad of the
// full form's (measured against the CPU restatement), five orders below the 1e-9 the path is held to,
// for six instructions per bin less.
template <bool LEAN = false>
VOC_DEV double div_pos(double a, double b)
{
    double r = approx_rcp(b);
    r = fma(fma(-b, r, 1.0), r, r);
    const double q = a * r;
    if (LEAN) return q;
    return fma(fma(-b, q, a), r, q);
}

VOC_DEV int frexp_exponent(double v)  // exponent e with |v| = m 2^e, 0.5 <= m < 1; 0 for v = 0
{
#ifdef VOC_EMU
    int e = 0;
    if (v != 0.0) (void)frexp(v, &e);
    return e;
#else
    return __builtin_amdgcn_frexp_exp(v);
#endif
}

// sqrt(s) for s = 0 or 0.25 <= s < 2 (a sum of two squares of numbers scaled into [0, 1), the larger
// one at least 0.5): reciprocal-square-root seed (relative error below 2^-24 on gfx950), one coupled
// Newton step on (sqrt, 1/(2 sqrt)), one residual correction -- the device library's own sequence
// without its rescaling of tiny arguments, its special-case tests and its second correction, none
// of which changes a bit on this domain (tools/ubench/seed_precision.hip).  For s = 0 the seed is taken of
// a tiny positive number instead: it stays finite, every product with s is an exact zero and the
// result is 0 without a test.
template <bool LEAN = false>
VOC_DEV double sqrt_scaled(double s)
{
#ifdef VOC_EMU
    return sqrt(s);
#else
    const double y = __builtin_amdgcn_rsq(fmax(s, 0x1p-1000));
    double g = s * y;
    double h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    if (LEAN) return g;
    h = fma(h, r, h);
    return fma(fma(-g, g, s), h, g);
#endif
}

// One Horner step p*z + c with the coefficient read from a scalar register pair.  Written as an
// instruction because the compiler otherwise selects the two-address multiply-add, which needs the
// coefficient copied into a vector register first: one extra instruction per step and a vector
// register pair per coefficient held across the loop.
VOC_DEV double horner(double p, double z, double c)
{
#ifdef VOC_EMU
    return fma(p, z, c);
#else
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(p), "v"(z), "s"(c));
    return r;
#endif
}

// atan(t) = t + t z P(z), z = t^2, for |t| <= tan(pi/8): degree-10 minimax P (tools/gen_poly.py),
// relative error 5e-18 before rounding.
template <bool LEAN = false>
VOC_DEV double atan_poly(double t)
{
    const double z = t * t;
    if (LEAN) {  // degree 8 (tools/gen_poly.py), relative error 3.9e-15 before rounding
        double q = -0x1.bf5790a65fcfbp-6;
        q = horner(q, z, 0x1.a7d43558287d4p-5);
        q = horner(q, z, -0x1.0c6176a21253cp-4);
        q = horner(q, z, 0x1.3a9faa15e1c1cp-4);
        q = horner(q, z, -0x1.745668abcbcd7p-4);
        q = horner(q, z, 0x1.c71c3a088a42bp-4);
        q = horner(q, z, -0x1.249248af9e3c5p-3);
        q = horner(q, z, 0x1.99999998dc43dp-3);
        q = horner(q, z, -0x1.55555555553c1p-2);
        return fma(t * z, q, t);
    }
    double p = -0x1.3b08efe0aa3adp-6;
    p = horner(p, z, 0x1.41bf291d28cbdp-5);
    p = horner(p, z, -0x1.a0bb5679f2cabp-5);
    p = horner(p, z, 0x1.dfed1a54859f0p-5);
    p = horner(p, z, -0x1.10fae2404e73fp-4);
    p = horner(p, z, 0x1.3b126b14c8afbp-4);
    p = horner(p, z, -0x1.745d0b879bfcfp-4);
    p = horner(p, z, 0x1.c71c7187c0787p-4);
    p = horner(p, z, -0x1.249249243a720p-3);
    p = horner(p, z, 0x1.99999999993a8p-3);
    p = horner(p, z, -0x1.5555555555555p-2);
    return fma(t * z, p, t);
}

// frameFromComplex for one bin (Vocoder.hs:217-218): Data.Complex.magnitude and phase of x + iy.
//
// Both start from the pair scaled by 2^-e, e = exponent of max(|x|, |y|), exactly as
// `magnitude` prescribes (scaleFloat k (sqrt (sqr (scaleFloat (-k) x) + sqr (scaleFloat (-k) y)))):
// power-of-two scaling commutes with every rounding, so the bits match the reference's whenever
// its own intermediate squares neither overflow nor underflow.
//
// phase: 0 for 0:+0, else atan2 y x in (-pi, pi] with atan2(+-0, x<0) = +-pi, as GHC's and IEEE's.
// The angle phi of (|x|, |y|) in [0, pi/2] comes from its half: tan(phi/2) = |y| / (r + |x|) with the
// r = sqrt(x^2 + y^2) the magnitude already needs, a quotient in [0, 1] -- so there is no swap of the
// two components and no octant to undo.  The quotient is folded once at tan(pi/8),
// atan t = pi/4 + atan((t-1)/(t+1)), BEFORE the one division.  Every case distinction is arithmetic
// on a 0.0 / 1.0 or +-2.0 factor (one 32-bit select each: only the high word of such a constant
// varies) instead of a select between two doubles:
//   f  = 1 when folded:      num = |y| - f (r+|x|),  den = (r+|x|) + f |y|,  phi/2 = atan(num/den) + f pi/4
//   s2 = -2 for x < 0, else 2:   atan2(|y|, x) = s2 phi/2 + (pi/2 - s2 pi/4)        [= phi or pi - phi]
// (f = 0 and s2 = 2 reproduce the plain values exactly: the products are exact and the added zero
// changes nothing), then the sign of y.  Accurate to about an ulp like the direct form; an exactly
// real or imaginary bin still gives exactly 0, +-pi/2 or +-pi.  Finite inputs only: an infinite or
// NaN component gives NaN.
//
// The zero bin needs no test of its own: its divisor is replaced by a tiny positive number, so the
// quotient is 0, and "x negative" is decided by x < 0, false for -0.  The result is a zero with the
// sign of y (the reference's is +0 for every zero bin).
template <bool LEAN = false>
VOC_DEV void hs_polar(double x, double y, double &mag, double &ph)
{
    const double ax = fabs(x), ay = fabs(y);
    const int e = frexp_exponent(fmax(ax, ay));
    const double sx = ldexp(ax, -e), sy = ldexp(ay, -e);
    const double a = sx * sx, b = sy * sy;
    const double r = sqrt_scaled<LEAN>(LEAN ? fma(sy, sy, a) : a + b);  // in [0.5, 1.42), or 0
    mag = ldexp(r, e);
    const double d0 = r + sx;             // in [0.5, 2.42), or 0
    const double f = (sy > 0.41421356237309503 * d0) ? 1.0 : 0.0;
    const double num = fma(-f, d0, sy);
    const double den = fmax(fma(f, sy, d0), 0x1p-1000);  // in [0.5, 3.84) unless the bin is zero
    const double half = fma(f, 0.78539816339744828, atan_poly<LEAN>(div_pos<LEAN>(num, den)));
    const double s2 = (x < 0.0) ? -2.0 : 2.0;
    const double off = fma(-s2, 0.78539816339744828, 1.5707963267948966);  // 0 or pi, exactly
    ph = copysign(fma(s2, half, off), y);
}

VOC_DEV double hs_magnitude(double x, double y)
{
    double m, p;
    hs_polar(x, y, m, p);
    return m;
}

VOC_DEV double hs_phase(double x, double y)
{
    double m, p;
    hs_polar(x, y, m, p);
    return p;
}

// sin and cos of an UNWRAPPED phase: n = round(x 2/pi) by the 1.5*2^52 trick, two-step Cody-Waite
// reduction with fma (x - n*hi is exact: both are multiples of 2^-52 and the difference is below
// 1), then the classical degree-13 / degree-14 kernels on [-pi/4, pi/4].  Accurate to about an ulp
// for |x| < 2^50 (a phase that large has a resolution of 0.25 rad).
//
// The plain form returns NaN beyond 2^50 and for inf / nan.  LEAN gives the same bits on the valid
// range with fewer instructions and no branch: there is no range test (inf and nan still come out
// as NaN through the reduction, a finite phase beyond 2^50 gives some finite value of no
// meaning -- at 1024 samples a hop that takes 10^9 hours of audio), the quadrant's parity is one
// 32-bit test, and the quadrant signs are bits of q moved onto the sign bits (sine: bit 1; cosine:
// bit 1 xor bit 0) instead of two negate-and-select pairs.  Without the branches the bins of a
// loop body become one scheduling region; where registers are scarce that region is scheduled worse
// than the separate ones, so the caller chooses (the specialised synthesis kernel: lean from N = 2048 up).
template <bool LEAN = false>
VOC_DEV void fast_sincos(double x, double *sn, double *cs)
{
    if (!LEAN) x = (fabs(x) < 0x1p50) ? x : (x - x) / (x - x);
    const double big = 0x1.8p52;
    const double tn = fma(x, 0.63661977236758138, big);
    const double n = tn - big;
    const unsigned q = (unsigned)__builtin_bit_cast(unsigned long long, tn);
    double r = fma(-n, 1.5707963267948966, x);
    r = fma(-n, 6.123233995736766e-17, r);
    const double z = r * r;
    double ps = 1.58969099521155010221e-10;
    ps = fma(ps, z, -2.50507602534068634195e-08);
    ps = fma(ps, z, 2.75573137070700676789e-06);
    ps = fma(ps, z, -1.98412698298579493134e-04);
    ps = fma(ps, z, 8.33333333332248946124e-03);
    ps = fma(ps, z, -1.66666666666666324348e-01);
    const double s = fma(r * z, ps, r);
    double pc = -1.13596475577881948265e-11;
    pc = fma(pc, z, 2.08757232129817482790e-09);
    pc = fma(pc, z, -2.75573143513906633035e-07);
    pc = fma(pc, z, 2.48015872894767294178e-05);
    pc = fma(pc, z, -1.38888888888741095749e-03);
    pc = fma(pc, z, 4.16666666666666019037e-02);
    const double c = fma(z * z, pc, fma(-0.5, z, 1.0));
    const bool odd = (q & 1u) != 0u;
    const double s1 = odd ? c : s;
    const double c1 = odd ? s : c;
    if (LEAN) {  // quadrants 2, 3 negate the sine, 1, 2 the cosine
        const unsigned long long s1b = __builtin_bit_cast(unsigned long long, s1), c1b = __builtin_bit_cast(unsigned long long, c1);
        const unsigned fs = (q << 30) & 0x80000000u;  // bit 1 of q
        const unsigned f1 = q << 31;                  // bit 0 of q
        const unsigned shi = (unsigned)(s1b >> 32) ^ fs, chi = ((unsigned)(c1b >> 32) ^ fs) ^ f1;
        *sn = __builtin_bit_cast(double, ((unsigned long long)shi << 32) | (s1b & 0xffffffffull));
        *cs = __builtin_bit_cast(double, ((unsigned long long)chi << 32) | (c1b & 0xffffffffull));
    } else {
        *sn = (q & 2u) ? -s1 : s1;
        *cs = ((q + 1u) & 2u) ? -c1 : c1;
    }
}

// sin and cos of an UNWRAPPED phase from a table of the circle in 64 steps and two short series:
// n = round(x 32/pi), r = x - n pi/32 (two-step Cody-Waite as above, |r| <= pi/64), (C, S) = table[n mod 64],
// cos x = C cos r - S sin r, sin x = S cos r + C sin r with sin r to r^7 and cos r to r^6 (truncation errors
// 7e-18 and 8e-16).  Same accuracy as fast_sincos to within an ulp or two, for |x| < 2^46, with a third fewer
// vector instructions: no quadrant logic, half the polynomial steps, one 16-byte LDS read.
// tab: this lane's column of the table -- entry j at tab[16 j]: sixteen copies side by side keep the sixteen
// lanes that share an LDS read cycle on different banks whatever their indices.
VOC_DEV void fast_sincos_tab(double x, const double2 *tab, double *sn, double *cs)
{
    const double big = 0x1.8p52;
    const double tn = fma(x, 10.185916357881302, big);
    const double n = tn - big;
    const unsigned q = (unsigned)__builtin_bit_cast(unsigned long long, tn);
    double r = fma(-n, 0x1.921fb54442d18p-4, x);
    r = fma(-n, 0x1.1a62633145c07p-58, r);
    const double2 e = tab[(q & 63u) * 16u];
    const double z = r * r;
    double ps = fma(z, -1.0 / 5040.0, 1.0 / 120.0);
    ps = fma(z, ps, -1.0 / 6.0);
    const double s = fma(r * z, ps, r);
    double pc = fma(z, -1.0 / 720.0, 1.0 / 24.0);
    pc = fma(z, pc, -0.5);
    const double c = fma(z, pc, 1.0);
    *cs = fma(e.x, c, -(e.y * s));
    *sn = fma(e.y, c, e.x * s);
}

// wrap e = (e + pi) `mod'` (2 pi) - pi with mod' n d = n - floor(n/d) * d and the floor taken
// of the exact quotient (Vocoder.hs:166-167, Data.Fixed).  The floor f0 of the rounded quotient is
// the exact floor or one off; the remainder r0 = n - f0 * 2pi (one fma: exactly rounded, so its sign
// and its comparison with 2 pi are those of the exact remainder unless it lies within half an ulp
// of 2 pi) then falls in [-2pi, 0) or [2pi, 4pi) instead of [0, 2pi), and floor(r0 / 2pi) is the
// correction -- no compare, no select.  Only a remainder within an ulp of 2 pi (probability
// ~2^-52 per call) can still be taken for a full turn: the result is then one turn off, -pi - tiny
// instead of pi - tiny.  As an angle that is the same, but the increment built from it,
// (omega + wrap) / H, is then 2 pi / H smaller than the reference's for that one bin and frame -- which
// a filter that rescales or interpolates increments does not undo.  Accepted at that probability
// (the metric workload makes 6e9 calls per pass: one such bin in about 10^6 passes).
#ifndef VOC_WRAP8
#define VOC_WRAP8 1
#endif
VOC_DEV double hs_wrap(double e)
{
    const double n = e + kPi;
    const double f0 = floor(n * kInvTwoPi);
    const double r0 = fma(-f0, kTwoPi, n);
#if VOC_WRAP8
    // the corrected remainder itself: n - f 2pi in one rounding where the reference's n - fl(f 2pi) has
    // two (they differ by at most an ulp of n: 5e-13 rad at bin 2048), with the same floor f
    return fma(-floor(r0 * kInvTwoPi), kTwoPi, r0) - kPi;
#else
    const double f = f0 + floor(r0 * kInvTwoPi);
    const double t = f * kTwoPi;
    return (n - t) - kPi;
#endif
}

// ---- single precision: the same definitions on float.
VOC_DEV float approx_rcp(float b)
{
#ifdef VOC_EMU
    return 1.0f / b;
#else
    return __builtin_amdgcn_rcpf(b);
#endif
}

VOC_DEV int frexp_exponent(float v)
{
#ifdef VOC_EMU
    int e = 0;
    if (v != 0.0f) (void)frexpf(v, &e);
    return e;
#else
    return __builtin_amdgcn_frexp_expf(v);
#endif
}

// Same construction as the double version: shared power-of-two scaling, one division placed
// after the fold at tan(pi/8), atan(t) = t + t z P(z) with a degree-4 minimax P (error 2.5e-9).
template <bool LEAN = false>
VOC_DEV void hs_polar(float x, float y, float &mag, float &ph)
{
    const float ax = fabsf(x), ay = fabsf(y);
    const int e = frexp_exponent(fmaxf(ax, ay));
    const float sx = ldexpf(ax, -e), sy = ldexpf(ay, -e);
    const bool is_zero = ax == 0.0f && ay == 0.0f;
#ifdef VOC_EMU
    mag = ldexpf(sqrtf(sx * sx + sy * sy), e);
#else
    mag = ldexpf(__builtin_amdgcn_sqrtf(sx * sx + sy * sy), e);
#endif
    const float mx = fmaxf(sx, sy), mn = fminf(sx, sy);
    const bool fold = mn > 0.41421356f * mx;
    const float num = fold ? mn - mx : mn;
    const float den = fold ? mn + mx : mx;
    const float r = approx_rcp(den);
    float q = num * r;
    q = fmaf(fmaf(-den, q, num), r, q);
    const float z = q * q;
    float p = -0.06466829776763916f;
    p = fmaf(p, z, 0.10748856514692307f);
    p = fmaf(p, z, -0.14264507591724396f);
    p = fmaf(p, z, 0.19999559223651886f);
    p = fmaf(p, z, -0.3333333134651184f);
    float t = fmaf(q * z, p, q);
    t = fold ? t + 0.78539816f : t;
    t = (ay > ax) ? 1.57079633f - t : t;
    t = __builtin_signbit(x) ? 3.14159265f - t : t;
    ph = is_zero ? 0.0f : copysignf(t, y);
}

// sin and cos of a phase of moderate size (the float variant keeps its phases wrapped): Cody-Waite
// reduction by pi/2 in two fma steps, minimax kernels on [-pi/4, pi/4] (errors 1.2e-8 / 8e-10).
template <bool LEAN = false>
VOC_DEV void fast_sincos(float x, float *sn, float *cs)
{
    const float n = rintf(x * 0.636619772f);
    const int q = (int)n;
    float r = fmaf(-n, 1.57079637f, x);
    r = fmaf(-n, -4.37113883e-8f, r);
    const float z = r * r;
    float ps = -0.00019588008581195027f;
    ps = fmaf(ps, z, 0.008332748897373676f);
    ps = fmaf(ps, z, -0.166666641831398f);
    const float s = fmaf(r * z, ps, r);
    float pc = 2.454804052831605e-05f;
    pc = fmaf(pc, z, -0.001388830365613103f);
    pc = fmaf(pc, z, 0.0416666641831398f);
    const float c = fmaf(z * z, pc, fmaf(-0.5f, z, 1.0f));
    const float s1 = (q & 1) ? c : s;
    const float c1 = (q & 1) ? s : c;
    *sn = (q & 2) ? -s1 : s1;
    *cs = ((q + 1) & 2) ? -c1 : c1;
}

VOC_DEV float hs_wrap(float e)
{
    const float pi = (float)kPi, two_pi = (float)kTwoPi;
    const float n = e + pi;
    const float f0 = floorf(n * (float)kInvTwoPi);
    const bool below = fmaf(-f0, two_pi, n) < 0.0f;
    const bool above = fmaf(-(f0 + 1.0f), two_pi, n) >= 0.0f;
    const float f = below ? f0 - 1.0f : (above ? f0 + 1.0f : f0);
    return (n - f * two_pi) - pi;
}

// calcPhaseInc (Vocoder.hs:169-173): omega = ((2 pi k) H) / N ; (omega + wrap (d - omega)) / H
// With H and N powers of two, multiplying by H and by 1/N only moves the exponent, so one product
// with the constant 2 pi H / N has the bits of ((2 pi k) H) / N.
template <class R>
VOC_DEV double bin_omega(const ParamsT<R> &p, int k)
{
    if (p.omega1 != 0.0) return (double)k * p.omega1;
    asm volatile("");  // keeps this (uniform) branch a branch: as a select both sides would be computed
    const double t = (kTwoPi * (double)k) * (double)p.hop_d;
    // the reference divides by N (Vocoder.hs:173); a product with 1/N has the same bits only for a power of two
    if ((p.N & (p.N - 1)) == 0) return t * (double)p.inv_n;
    asm volatile("");
    return t / (double)p.N;
}

VOC_DEV double phase_inc(const ParamsT<double> &p, int k, double ph_diff)
{
    const double omega = bin_omega(p, k);
    const double s = omega + hs_wrap(ph_diff - omega);
    return p.hop_pow2 ? s * p.inv_hop : s / p.hop_d;
}

// The same for callers that know at compile time that H and N are powers of two (the bits are
// those of phase_inc; the two uniform branches are gone).
VOC_DEV double phase_inc_pow2(const ParamsT<double> &p, int k, double ph_diff)
{
    const double omega = (double)k * p.omega1;
    return (omega + hs_wrap(ph_diff - omega)) * p.inv_hop;
}

// ---- the single-precision variant's phase arithmetic, on the 32-bit units.
// A float has a resolution of 2.4e-4 rad at the few thousand radians omega reaches, so the two places where
// such a quantity meets one of a few radians keep omega as the sum of two floats and reduce by whole turns with
// a two-word 2 pi (the earlier form widened both to double: the same values to within the float rounding of the
// result, at twice the issue cost per instruction).
constexpr float kTwoPiHiF = 6.28318548202514648f;      // float(2 pi)
constexpr float kTwoPiLoF = -1.74845553146951715e-7f;  // 2 pi - float(2 pi)
constexpr float kInvTwoPiF = 0.159154943091895336f;
constexpr float kPiF = 3.14159274101257324f;

// x minus the nearest whole number of turns: in [-pi, pi] to within an ulp, for |x| up to about 2^20
VOC_DEV float wrap_near_f(float x)
{
    const float n = rintf(x * kInvTwoPiF);
    return fmaf(-n, kTwoPiLoF, fmaf(-n, kTwoPiHiF, x));
}

// x of a few turns at most into [-pi, pi) (floored, like the reference's wrap)
VOC_DEV float wrap_floor_f(float x)
{
    const float n = floorf((x + kPiF) * kInvTwoPiF);
    return fmaf(-n, kTwoPiLoF, fmaf(-n, kTwoPiHiF, x));
}

// calcPhaseInc for H and N powers of two: omega = k omega1 as two floats (the product's rounding error is
// recovered by an fma), its whole turns dropped before it meets the phase difference
VOC_DEV float phase_inc_pow2(const ParamsT<float> &p, int k, float ph_diff)
{
    const float kf = (float)k, km = (float)(k & p.turn_mask);
    const float w = wrap_floor_f(ph_diff - km * p.om1_hi);            // (k mod N/H) omega1 < 2 pi
    const float om_hi = kf * p.om1_hi;
    const float om_lo = fmaf(kf, p.om1_lo, fmaf(kf, p.om1_hi, -om_hi));
    return (om_hi + (w + om_lo)) * p.inv_hop;
}

// any other hop: omega and the sum are formed in double (a float 2 pi alone would bias every frame's increment
// by omega * 3e-8) and rounded to float once.
VOC_DEV float phase_inc(const ParamsT<float> &p, int k, float ph_diff)
{
    if (p.omega1 != 0.0) return phase_inc_pow2(p, k, ph_diff);        // (one arithmetic per context, whichever kernel runs)
    asm volatile("");
    const double omega = bin_omega(p, k);
    const double s = omega + hs_wrap((double)ph_diff - omega);
    return (float)(p.hop_pow2 ? s * (double)p.inv_hop : s / (double)p.hop_d);
}

// ---- single precision, two bins at a time.
// A plain fp32 instruction issues at the rate of an fp64 one on this chip; only the packed forms (v_pk_fma_f32,
// v_pk_mul_f32, v_pk_add_f32) do two lanes' worth per issue slot.  The specialised kernels work on bin pairs
// (k, M - k) whose two bins go through the same arithmetic, so that arithmetic is written here on two-lane
// vectors: products and sums pack; comparisons, selects, conversions, floor / rint and the reciprocal and
// square-root seeds stay one per bin, and the case distinctions are turned into 0 / 1 or +-1 factors so that
// only the factor is selected.  Same approximations as the one-bin float versions, to within float rounding.
#ifndef VOC_F32_PACKED
#define VOC_F32_PACKED 1
#endif
#ifndef VOC_F32_HW_SINCOS
#define VOC_F32_HW_SINCOS 1
#endif
#ifndef VOC_F32_ATAN_REFINE
#define VOC_F32_ATAN_REFINE 1
#endif
// a kernel argument used as one lane-uniform float (kept in its scalar register: without this the compiler
// gathers neighbouring arguments into vectors through the stack)
VOC_DEV float uniform_f(float v)
{
#ifndef VOC_EMU
    asm("" : "+s"(v));
#endif
    return v;
}
VOC_DEV unsigned f32_bits(float v) { return __builtin_bit_cast(unsigned, v); }
VOC_DEV float f32_of(unsigned b) { return __builtin_bit_cast(float, b); }

// frameFromComplex of the bins xa + i ya and xb + i yb, the magnitudes times `scale`.  atan(min / max) of
// (|x|, |y|) with the fold at tan(pi/8) before the division; the octant is undone by two +-1 factors.
// A zero bin gives magnitude 0 and phase 0 (pi for x = -0).  VOC_F32_POLAR_SCALED: the pair (max, min) is first
// scaled by the power of two that brings the larger into [1, 2) -- a product with a float assembled from the
// exponent bits, and the magnitude is scaled back by the float with just those bits -- so that the squares neither
// overflow nor underflow (seven instructions per pair); without it a bin must stay within 1e-19 .. 1e19 for its
// magnitude to come out, which single-precision audio does by twelve orders of magnitude on either side.
#ifndef VOC_F32_POLAR_SCALED
#define VOC_F32_POLAR_SCALED 0
#endif
VOC_DEV void hs_polar_pair(float xa, float ya, float xb, float yb, float scale, float &ma, float &pa, float &mb, float &pb)
{
    const f32x2 mx = pk2(fmaxf(fabsf(xa), fabsf(ya)), fmaxf(fabsf(xb), fabsf(yb)));
    const f32x2 mn = pk2(fminf(fabsf(xa), fabsf(ya)), fminf(fabsf(xb), fabsf(yb)));
#if VOC_F32_POLAR_SCALED
    const unsigned ea = f32_bits(mx.x) & 0x7f800000u, eb = f32_bits(mx.y) & 0x7f800000u;
    const f32x2 down = pk2(f32_of(0x7f000000u - ea), f32_of(0x7f000000u - eb));  // 2^-e (2^127 for a zero bin: 0 stays 0)
    const f32x2 up = pk2(f32_of(ea), f32_of(eb));                                  // 2^e
    const f32x2 MX = mx * down, MN = mn * down;                                    // MX in [1, 2) or 0
#else
    const f32x2 MX = mx, MN = mn;
#endif
    const f32x2 ss = fma2(MN, MN, MX * MX);
#ifdef VOC_EMU
    const f32x2 r = pk2(sqrtf(ss.x), sqrtf(ss.y));
#else
    const f32x2 r = pk2(__builtin_amdgcn_sqrtf(ss.x), __builtin_amdgcn_sqrtf(ss.y));
#endif
#if VOC_F32_POLAR_SCALED
    const f32x2 mag = (r * up) * splat2(scale);
#else
    const f32x2 mag = r * splat2(scale);
#endif
    const f32x2 lim = MX * splat2(0.41421356f);
    const f32x2 f = pk2(MN.x > lim.x ? 1.0f : 0.0f, MN.y > lim.y ? 1.0f : 0.0f);
    const f32x2 num = fma2(-f, MX, MN);
    const f32x2 d0 = fma2(f, MN, MX);
    constexpr float tiny = VOC_F32_POLAR_SCALED ? 1e-30f : 1e-37f;   // (a zero bin: 0 / tiny = 0)
    const f32x2 den = pk2(fmaxf(d0.x, tiny), fmaxf(d0.y, tiny));
    const f32x2 rc = pk2(approx_rcp(den.x), approx_rcp(den.y));
    f32x2 q = num * rc;
    if (VOC_F32_ATAN_REFINE) q = fma2(fma2(-den, q, num), rc, q);
    const f32x2 z = q * q;
    f32x2 p = splat2(-0.06466829776763916f);
    p = fma2(p, z, splat2(0.10748856514692307f));
    p = fma2(p, z, splat2(-0.14264507591724396f));
    p = fma2(p, z, splat2(0.19999559223651886f));
    p = fma2(p, z, splat2(-0.3333333134651184f));
    f32x2 t = fma2(q * z, p, q);
    t = fma2(f, splat2(0.78539816f), t);
    // |y| > |x|: pi/2 - t;  x < 0 (or -0): pi - t;  then the sign of y
    const f32x2 s1 = pk2(fabsf(ya) > fabsf(xa) ? -1.0f : 1.0f, fabsf(yb) > fabsf(xb) ? -1.0f : 1.0f);
    t = fma2(s1, t, fma2(-s1, splat2(0.78539816f), splat2(0.78539816f)));
    const f32x2 s2 = pk2(copysignf(1.0f, xa), copysignf(1.0f, xb));
    t = fma2(s2, t, fma2(-s2, splat2(1.57079633f), splat2(1.57079633f)));
    ma = mag.x;
    mb = mag.y;
    pa = copysignf(t.x, ya);
    pb = copysignf(t.y, yb);
}

// phase_inc_pow2 of two bins whose numbers are given as floats (kf) together with kmhi = (k mod N/H) om1_hi
VOC_DEV void phase_inc_pow2_pair_kf(const ParamsT<float> &p, f32x2 kf, f32x2 kmhi, float da, float db, float &ia, float &ib)
{
    const f32x2 hi = splat2(uniform_f(p.om1_hi));
    const f32x2 x = pk2(da, db) - kmhi;                                            // ph_diff - (k mod N/H) omega1: product, then difference, as phase_inc_pow2
    const f32x2 u = (x + splat2(kPiF)) * splat2(kInvTwoPiF);
    const f32x2 n = pk2(floorf(u.x), floorf(u.y));
    const f32x2 w = fma2(-n, splat2(kTwoPiLoF), fma2(-n, splat2(kTwoPiHiF), x));   // wrap_floor_f
    const f32x2 om_hi = kf * hi;
    const f32x2 om_lo = fma2(kf, splat2(uniform_f(p.om1_lo)), fma2(kf, hi, -om_hi));
    const f32x2 r = (om_hi + (w + om_lo)) * splat2(uniform_f(p.inv_hop));
    ia = r.x;
    ib = r.y;
}

// phase_inc_pow2 of the bins ka and kb
VOC_DEV void phase_inc_pow2_pair(const ParamsT<float> &p, int ka, int kb, float da, float db, float &ia, float &ib)
{
    const f32x2 kf = pk2((float)ka, (float)kb), km = pk2((float)(ka & p.turn_mask), (float)(kb & p.turn_mask));
    phase_inc_pow2_pair_kf(p, kf, km * splat2(uniform_f(p.om1_hi)), da, db, ia, ib);
}

// The bin pairs of a thread of the specialised kernels are (t + m T, M - t - m T), T a multiple of N/H: as floats
// (exact) the two numbers are fa and M - fa with fa = t + m T, and k mod N/H is the same for every m.  What does not
// depend on m is formed once per frame; a pair then costs two instructions where the conversions of its two bin
// numbers and of their remainders cost eight.  Double precision has no use for it (its phase_inc_pow2 takes integers).
template <class R> struct PairBinsT {
    VOC_DEV void frame(const ParamsT<R> &, int) {}
};
template <> struct PairBinsT<float> {
    float ft;
    f32x2 kmhi;
    VOC_DEV void frame(const ParamsT<float> &p, int lt)
    {
        ft = (float)lt;
        kmhi = pk2((float)(lt & p.turn_mask), (float)((0 - lt) & p.turn_mask)) * splat2(uniform_f(p.om1_hi));
    }
};
// synthesisStep of two bins: phase steps (phase_step for float), then mkPolar; LEAN is accepted and ignored
VOC_DEV void synthesize_pair(const ParamsT<float> &p, float mga, float ica, float &pha, float mgb, float icb, float &phb,
                             float2 &A, float2 &B)
{
    const f32x2 t = pk2(ica, icb) * splat2(uniform_f((float)p.hop_d));
    const f32x2 n1 = pk2(rintf(t.x * kInvTwoPiF), rintf(t.y * kInvTwoPiF));
    const f32x2 tn = fma2(-n1, splat2(kTwoPiLoF), fma2(-n1, splat2(kTwoPiHiF), t));   // wrap_near_f
    const f32x2 s = pk2(pha, phb) + tn;
    const f32x2 u = (s + splat2(kPiF)) * splat2(kInvTwoPiF);
    const f32x2 n2 = pk2(floorf(u.x), floorf(u.y));
    const f32x2 ph = fma2(-n2, splat2(kTwoPiLoF), fma2(-n2, splat2(kTwoPiHiF), s));   // wrap_floor_f
    pha = ph.x;
    phb = ph.y;
#if VOC_F32_HW_SINCOS
    // sine and cosine from the hardware (v_sin_f32 / v_cos_f32: argument in turns, within 1.3e-7 of the exact values
    // over [-0.5, 0.5], measured; two issue slots each, against 30 for range reduction, two polynomials and quadrants)
    const f32x2 tu = ph * splat2(kInvTwoPiF);
#ifdef VOC_EMU
    const f32x2 sg = pk2(sinf(6.28318530717958647692f * tu.x), sinf(6.28318530717958647692f * tu.y));
    const f32x2 cg = pk2(cosf(6.28318530717958647692f * tu.x), cosf(6.28318530717958647692f * tu.y));
#else
    const f32x2 sg = pk2(__builtin_amdgcn_sinf(tu.x), __builtin_amdgcn_sinf(tu.y));
    const f32x2 cg = pk2(__builtin_amdgcn_cosf(tu.x), __builtin_amdgcn_cosf(tu.y));
#endif
#else
    // fast_sincos for float, both bins
    const f32x2 v = ph * splat2(0.636619772f);
    const f32x2 n = pk2(rintf(v.x), rintf(v.y));
    const int qa = (int)n.x, qb = (int)n.y;
    f32x2 r = fma2(-n, splat2(1.57079637f), ph);
    r = fma2(-n, splat2(-4.37113883e-8f), r);
    const f32x2 z = r * r;
    f32x2 ps = splat2(-0.00019588008581195027f);
    ps = fma2(ps, z, splat2(0.008332748897373676f));
    ps = fma2(ps, z, splat2(-0.166666641831398f));
    const f32x2 sn = fma2(r * z, ps, r);
    f32x2 pc = splat2(2.454804052831605e-05f);
    pc = fma2(pc, z, splat2(-0.001388830365613103f));
    pc = fma2(pc, z, splat2(0.0416666641831398f));
    const f32x2 cs = fma2(z * z, pc, fma2(splat2(-0.5f), z, splat2(1.0f)));
    // quadrants: odd ones swap sine and cosine; 2, 3 negate the sine, 1, 2 the cosine (sign bits moved in place)
    const float s1a = (qa & 1) ? cs.x : sn.x, c1a = (qa & 1) ? sn.x : cs.x;
    const float s1b = (qb & 1) ? cs.y : sn.y, c1b = (qb & 1) ? sn.y : cs.y;
    const f32x2 sg = pk2(f32_of(f32_bits(s1a) ^ (((unsigned)qa << 30) & 0x80000000u)), f32_of(f32_bits(s1b) ^ (((unsigned)qb << 30) & 0x80000000u)));
    const f32x2 cg = pk2(f32_of(f32_bits(c1a) ^ ((((unsigned)qa + 1u) << 30) & 0x80000000u)),
                         f32_of(f32_bits(c1b) ^ ((((unsigned)qb + 1u) << 30) & 0x80000000u)));
#endif
    const f32x2 mg = pk2(mga, mgb);
    const f32x2 re = mg * cg, im = mg * sg;
    A = make_float2(re.x, im.x);
    B = make_float2(re.y, im.y);
}

// analysisStep for one bin (Vocoder.hs:158-163).  `prev` is replaced by the new raw phase.
template <class R>
VOC_DEV void analyze_bin(const ParamsT<R> &p, int k, cplx<R> X, R &prev, bool emit, R vol, R *mag, R *inc)
{
    R m, ph;
    hs_polar(X.x, X.y, m, ph);
    if (emit) {
        *mag = m * vol;
        *inc = phase_inc(p, k, ph - prev);
    }
    prev = ph;
}

// The synthesis phase chain, one step (Vocoder.hs:190): ph + (ph_inc * H), two roundings, never
// wrapped -- in double.  The float variant wraps the sum into [-pi, pi): an unwrapped float phase
// grows past 2^24 ulp-of-a-radian within a few thousand frames and would lose the phase entirely.
VOC_DEV double phase_step(double ph, double t) { return ph + t; }
// (t = ph_inc * H reaches thousands of radians, where a float has a resolution of a quarter of a milliradian:
// its whole turns are dropped first, with a two-word 2 pi, and only then does it meet the wrapped phase)
VOC_DEV float phase_step(float ph, float t) { return wrap_floor_f(ph + wrap_near_f(t)); }

// synthesisStep for one bin (Vocoder.hs:186-190): the phase step, then mkPolar.
template <class R, bool LEAN = false>
VOC_DEV cplx<R> synthesize_bin(const ParamsT<R> &p, R mag, R inc, R &ph)
{
    const R t = inc * p.hop_d;
    ph = phase_step(ph, t);
    R s, c;
    fast_sincos<LEAN>(ph, &s, &c);
    return mk2(mag * c, mag * s);
}

// Two bins at a time: the double-precision kernels (and float builds without VOC_F32_PACKED) take the one-bin
// functions twice, float takes the packed forms above.
// (the magnitudes come back multiplied by `scale`)
template <bool LEAN>
VOC_DEV void polar_pair(double2 Xa, double2 Xb, double scale, double &ma, double &pa, double &mb, double &pb)
{
    hs_polar<LEAN>(Xa.x, Xa.y, ma, pa);
    hs_polar<LEAN>(Xb.x, Xb.y, mb, pb);
    ma *= scale;
    mb *= scale;
}
template <bool LEAN>
VOC_DEV void polar_pair(float2 Xa, float2 Xb, float scale, float &ma, float &pa, float &mb, float &pb)
{
    if (VOC_F32_PACKED) {
        hs_polar_pair(Xa.x, Xa.y, Xb.x, Xb.y, scale, ma, pa, mb, pb);
    } else {
        hs_polar<LEAN>(Xa.x, Xa.y, ma, pa);
        hs_polar<LEAN>(Xb.x, Xb.y, mb, pb);
        ma *= scale;
        mb *= scale;
    }
}
VOC_DEV double bins_ft(const PairBinsT<double> &) { return 0.0; }
VOC_DEV float bins_ft(const PairBinsT<float> &b) { return b.ft; }
// (fa: the first bin number as a float, float kernels only; ka + kb = M)
VOC_DEV void inc_pow2_pair(const ParamsT<double> &p, const PairBinsT<double> &, double, int, int ka, int kb, double da, double db, double &ia, double &ib)
{
    ia = phase_inc_pow2(p, ka, da);
    ib = phase_inc_pow2(p, kb, db);
}
VOC_DEV void inc_pow2_pair(const ParamsT<float> &p, const PairBinsT<float> &bins, float fa, int M, int ka, int kb, float da, float db, float &ia, float &ib)
{
    if (VOC_F32_PACKED) {
        const f32x2 kf = fma2(splat2(fa), pk2(1.0f, -1.0f), pk2(0.0f, (float)M));
        phase_inc_pow2_pair_kf(p, kf, bins.kmhi, da, db, ia, ib);
    } else {
        ia = phase_inc_pow2(p, ka, da);
        ib = phase_inc_pow2(p, kb, db);
    }
}
VOC_DEV void inc_pow2_pair(const ParamsT<double> &p, int ka, int kb, double da, double db, double &ia, double &ib)
{
    ia = phase_inc_pow2(p, ka, da);
    ib = phase_inc_pow2(p, kb, db);
}
VOC_DEV void inc_pow2_pair(const ParamsT<float> &p, int ka, int kb, float da, float db, float &ia, float &ib)
{
    if (VOC_F32_PACKED) {
        phase_inc_pow2_pair(p, ka, kb, da, db, ia, ib);
    } else {
        ia = phase_inc_pow2(p, ka, da);
        ib = phase_inc_pow2(p, kb, db);
    }
}
template <bool LEAN>
VOC_DEV void synth_pair(const ParamsT<double> &p, double mga, double ica, double &pha, double mgb, double icb, double &phb,
                        double2 &A, double2 &B)
{
    A = synthesize_bin<double, LEAN>(p, mga, ica, pha);
    B = synthesize_bin<double, LEAN>(p, mgb, icb, phb);
}
template <bool LEAN>
VOC_DEV void synth_pair(const ParamsT<float> &p, float mga, float ica, float &pha, float mgb, float icb, float &phb,
                        float2 &A, float2 &B)
{
    if (VOC_F32_PACKED) {
        synthesize_pair(p, mga, ica, pha, mgb, icb, phb, A, B);
    } else {
        A = synthesize_bin<float, LEAN>(p, mga, ica, pha);
        B = synthesize_bin<float, LEAN>(p, mgb, icb, phb);
    }
}

// the same with the sine and cosine from the table in LDS (fast_sincos_tab)
VOC_DEV double2 synthesize_bin_tab(const ParamsT<double> &p, double mag, double inc, double &ph, const double2 *tab)
{
    ph = phase_step(ph, inc * p.hop_d);
    double s, c;
    fast_sincos_tab(ph, tab, &s, &c);
    return mk2(mag * c, mag * s);
}

// ------------------------------------------------------------------ analysis

template <class R>
struct AnaArgsT {
    const R *x;             // frame (c, f) starts at x + c*chan_stride + f*frame_stride
    long long chan_stride;  // [C][F][W]: F*W ; raw signal: samples per channel row
    long long frame_stride; // [C][F][W]: W   ; raw signal: H
    const R *phase_in;      // [C][K]
    R *phase_out;           // [C][K]
    R *mag, *inc;           // [(c*F + f)*ldk + k]
    R *spec;                // SPEC_OUT: [(c*F + f)*K + k] complex
    long long ldk;
    long long F;
    long long rpc;          // rows per channel in mag / inc / spec: F, or more when the caller's arrays hold other rows too
    int tile_len, tiles;
    R vol;                  // factor folded into mag (volumeFix), 1 for the plain stage
};
using AnaArgs = AnaArgsT<double>;

// RAW: plain R2C of N contiguous reals (no window / pad / rewind).  SPEC_OUT: write the complex
// spectrum instead of (mag, ph_inc).
template <int LOG2M, bool RAW, bool SPEC_OUT, class R = double>
__global__ void __launch_bounds__(Cfg<LOG2M>::T, 4) analysis_kernel(ParamsT<R> p, AnaArgsT<R> a)
{
    constexpr int M = Cfg<LOG2M>::M, T = Cfg<LOG2M>::T, NP = Cfg<LOG2M>::NP;
    using C2 = cplx<R>;
    VOC_SMEM(C2, lds);
    const int tid = threadIdx.x;
    const long long c = blockIdx.x / a.tiles;
    const int tile = blockIdx.x % a.tiles;
    const long long f0 = (long long)tile * a.tile_len;
    const long long f1 = (f0 + a.tile_len < a.F) ? f0 + a.tile_len : a.F;
    const R *xc = a.x + c * a.chan_stride;
    const bool own_half = (tid == T - 1);  // the thread that also owns bin M/2

    R prev[2 * NP + 1];
    if (!SPEC_OUT && f0 == 0) {
        const R *pi = a.phase_in + c * p.K;
#pragma unroll
        for (int m = 0; m < NP; ++m) {
            const int pr = tid + m * T;
            if (pr < M / 2) {
                prev[2 * m] = pi[pr];
                prev[2 * m + 1] = pi[M - pr];
            }
        }
        if (own_half) prev[2 * NP] = pi[M / 2];
    }

    const long long fs = (!SPEC_OUT && f0 > 0) ? f0 - 1 : f0;  // one frame early to recover its raw phase
    for (long long f = fs; f < f1; ++f) {
        const R *fr = xc + f * a.frame_stride;
        const bool emit = f >= f0;
        // ---- first pass reads straight from global: window, zero pad, rewind, even/odd packing
        auto loader = [&](int i, int j) -> C2 {
            const int m = i + j * (M / (M >= 8 ? 8 : M));
            if (RAW) return mk2(fr[2 * m], fr[2 * m + 1]);
            const int i0 = ((2 * m + M) & (p.N - 1)) - p.hd;       // rewind: v[n] = padded[(n + N/2) mod N]
            const int i1 = ((2 * m + 1 + M) & (p.N - 1)) - p.hd;
            C2 z;
            z.x = (i0 >= 0 && i0 < p.W) ? p.window[i0] * fr[i0] : (R)0;
            z.y = (i1 >= 0 && i1 < p.W) ? p.window[i1] * fr[i1] : (R)0;
            return z;
        };
        fft_passes<M, T, 1>(lds, p.twP, tid, loader);
        // ---- real-FFT split + per-bin analysis.  Z[k] is at lds[pad(k)].
        const long long row = c * a.rpc + f;
#pragma unroll
        for (int m = 0; m < NP; ++m) {
            const int pr = tid + m * T;
            if (pr < M / 2) {
                C2 Xa, Xb;  // bins pr and M - pr
                if (pr == 0) {
                    const C2 z0 = lds[lds_pad(0)];
                    Xa = mk2(z0.x + z0.y, (R)0);
                    Xb = mk2(z0.x - z0.y, (R)0);
                } else {
                    const C2 P = lds[lds_pad(pr)], Q = lds[lds_pad(M - pr)];
                    const C2 E = mk2((R)0.5 * (P.x + Q.x), (R)0.5 * (P.y - Q.y));
                    const C2 O = mk2((R)0.5 * (P.y + Q.y), (R)0.5 * (Q.x - P.x));  // (P - conj Q) / (2i)
                    const C2 U = cmul(p.twS[pr], O);
                    Xa = cadd(E, U);
                    Xb = cconj(csub(E, U));
                }
                if (SPEC_OUT) {
                    R *s = a.spec + row * 2 * p.K;
                    s[2 * pr] = Xa.x;
                    s[2 * pr + 1] = Xa.y;
                    s[2 * (M - pr)] = Xb.x;
                    s[2 * (M - pr) + 1] = Xb.y;
                } else {
                    R mg, ic;
                    analyze_bin(p, pr, Xa, prev[2 * m], emit, a.vol, &mg, &ic);
                    if (emit) {
                        a.mag[row * a.ldk + pr] = mg;
                        a.inc[row * a.ldk + pr] = ic;
                    }
                    analyze_bin(p, M - pr, Xb, prev[2 * m + 1], emit, a.vol, &mg, &ic);
                    if (emit) {
                        a.mag[row * a.ldk + (M - pr)] = mg;
                        a.inc[row * a.ldk + (M - pr)] = ic;
                    }
                }
            }
        }
        if (own_half) {
            const C2 X = cconj(lds[lds_pad(M / 2)]);  // W^(M/2) = -i
            if (SPEC_OUT) {
                R *s = a.spec + row * 2 * p.K;
                s[M] = X.x;
                s[M + 1] = X.y;
            } else {
                R mg, ic;
                analyze_bin(p, M / 2, X, prev[2 * NP], emit, a.vol, &mg, &ic);
                if (emit) {
                    a.mag[row * a.ldk + M / 2] = mg;
                    a.inc[row * a.ldk + M / 2] = ic;
                }
            }
        }
        // the next frame's first LDS write sits behind the barrier inside fft_passes
    }

    if (!SPEC_OUT && f1 == a.F && f1 > f0) {
        R *po = a.phase_out + c * p.K;
#pragma unroll
        for (int m = 0; m < NP; ++m) {
            const int pr = tid + m * T;
            if (pr < M / 2) {
                po[pr] = prev[2 * m];
                po[M - pr] = prev[2 * m + 1];
            }
        }
        if (own_half) po[M / 2] = prev[2 * NP];
    }
}

// ------------------------------------------------------------------ synthesis

enum { SYN_IN_POLAR = 0, SYN_IN_SPEC = 1 };
enum { SYN_OUT_FRAMES = 0, SYN_OUT_RAW = 1, SYN_OUT_OLA = 2 };

template <class R>
struct SynArgsT {
    const R *mag, *inc;       // [(c*F + f)*ldk + k]
    long long ldk;
    const R *spec;            // IN_SPEC: [(c*F + f)*K + k] complex
    const R *phase_in;        // [C][K]           chain start of tile 0
    const R *carry;           // [C][tiles][K]    chain start of tiles > 0
    R *phase_out;             // [C][K]
    R *out;                   // FRAMES: [(c*F + f)*W + i]; RAW: [(c*F + f)*N + n]; OLA: [c*out_stride + s]
    long long out_stride;
    long long out_len;        // OLA: samples per channel, last chunk zero padded
    long long F;
    int tile_len, tiles, halo;
    long long unsplit;        // the first `unsplit` channels run as one tile each (no carry); the others in `tiles` tiles
    // Optional frame maps (playSpeed without a copy): frame f of a channel takes its moduli from row
    // mag_rows[f] of that channel's F_mag rows, its increments from row inc_rows[f] of F_inc rows.
    // Null map: row f; F_mag / F_inc of 0: F.
    const long long *mag_rows, *inc_rows;
    long long F_mag, F_inc;
    // Optional filter applied to a row as it is read (the double-precision overlap-add kernel of voc_kernels_fast.h
    // built with FILT): pitchShiftInterpolate n (Filter.hs:118-129; 0: none) followed by one table of gains
    // (linearAmplitudeFilter, Filter.hs:76-77; null: none) -- the chain `pitch >> gain` without a pass of its own
    // over the STFT.  pitch_rcp: 1/n where that is exact (n a power of two), else 0 (the increments are divided).
    double pitch_n, pitch_rcp;
    const double *gain;
};
using SynArgs = SynArgsT<double>;

// interpolate n v at bin k (Filter.hs:118-125) from the two entries it reads: position xp = n k, entries
// floor(xp) and floor(xp) + 1, weights (1 - fr) and fr; 0 where xp >= K - 1.  The same operations in the same
// order as the filter kernel's interpolate_at, so the bits are those of the filter pass.
struct PitchTap {
    int i;        // first entry (0 where the bin is out of range, so that loads stay inside the row)
    double fr;
    bool valid;
};
VOC_DEV PitchTap pitch_tap(double n, int k, int K)
{
    const double xp = n * (double)k;
    PitchTap t;
    t.valid = xp < (double)(K - 1);
    const double fl = floor(xp);
    t.i = t.valid ? (int)fl : 0;
    t.fr = xp - (double)t.i;
    return t;
}
VOC_DEV double pitch_mix(const PitchTap &t, double v0, double v1)
{
    const double lo = (1.0 - t.fr) * v0;
    const double hi = t.fr * v1;
    return t.valid ? lo + hi : 0.0;
}

// Row index of frame f of channel c in an array of `rows_per_channel` rows per channel, through an
// optional map whose entries are clamped into the array (a bad entry reads a wrong row, never
// outside the array).
VOC_DEV long long mapped_row(long long c, long long f, const long long *map, long long rows_per_channel, long long F)
{
    const long long n = rows_per_channel ? rows_per_channel : F;
    long long r = f;
    if (map) {
        r = map[f];
        r = r < 0 ? 0 : (r >= n ? n - 1 : r);
    }
    return c * n + r;
}
template <class R>
VOC_DEV const R *mag_row_of(const SynArgsT<R> &a, long long c, long long f)
{
    return a.mag + mapped_row(c, f, a.mag_rows, a.F_mag, a.F) * a.ldk;
}
template <class R>
VOC_DEV const R *inc_row_of(const SynArgsT<R> &a, long long c, long long f)
{
    return a.inc + mapped_row(c, f, a.inc_rows, a.F_inc, a.F) * a.ldk;
}

template <int LOG2M, int IN_MODE, int OUT_MODE, class R = double>
__global__ void __launch_bounds__(Cfg<LOG2M>::T, 4) synthesis_kernel(ParamsT<R> p, SynArgsT<R> a)
{
    constexpr int M = Cfg<LOG2M>::M, T = Cfg<LOG2M>::T, NP = Cfg<LOG2M>::NP, NQ = Cfg<LOG2M>::NQ;
    using C2 = cplx<R>;
    VOC_SMEM(C2, lds);
    R *ring = reinterpret_cast<R *>(lds + Cfg<LOG2M>::LDS_CPLX);  // OLA partial sums, W values
    const int tid = threadIdx.x;
    const bool whole = (long long)blockIdx.x < a.unsplit;  // leading channels: one tile over all frames
    const long long rest = (long long)blockIdx.x - a.unsplit;
    const long long c = whole ? (long long)blockIdx.x : a.unsplit + rest / a.tiles;
    const int tile = whole ? 0 : (int)(rest % a.tiles);
    const long long f0 = (long long)tile * a.tile_len;
    const long long f1 = (whole || f0 + a.tile_len >= a.F) ? a.F : f0 + a.tile_len;
    const bool own_half = (tid == T - 1);
    // OLA: frames that reach into this tile's first output sample are recomputed, in order,
    // so that every output sample is summed oldest frame first with no cross-tile exchange.
    const long long halo = (OUT_MODE == SYN_OUT_OLA) ? ((long long)a.halo < f0 ? (long long)a.halo : f0) : 0;
    const long long fs = f0 - halo;

    R ph[2 * NP + 1];
    if (IN_MODE == SYN_IN_POLAR) {
        const R *pi = (tile == 0) ? a.phase_in + c * p.K : a.carry + ((c - a.unsplit) * a.tiles + tile) * p.K;
#pragma unroll
        for (int m = 0; m < NP; ++m) {
            const int pr = tid + m * T;
            if (pr < M / 2) {
                ph[2 * m] = pi[pr];
                ph[2 * m + 1] = pi[M - pr];
            }
        }
        if (own_half) ph[2 * NP] = pi[M / 2];
    }
    int base = 0;
    if (OUT_MODE == SYN_OUT_OLA) {
        for (int i = tid; i < p.W; i += T) ring[i] = (R)0;
    }

    for (long long f = fs; f < f1; ++f) {
        const long long row = c * a.F + f;
        const R *mrow = nullptr, *irow = nullptr;
        if (IN_MODE == SYN_IN_POLAR) {
            mrow = mag_row_of(a, c, f);
            irow = inc_row_of(a, c, f);
        }
        // ---- spectrum of this frame -> half-length complex input Z (stored re<->im swapped so
        //      that the forward FFT below performs the backward transform)
#pragma unroll
        for (int m = 0; m < NP; ++m) {
            const int pr = tid + m * T;
            if (pr < M / 2) {
                C2 A, B;  // X[pr], X[M - pr]
                if (IN_MODE == SYN_IN_POLAR) {
                    A = synthesize_bin(p, mrow[pr], irow[pr], ph[2 * m]);
                    B = synthesize_bin(p, mrow[M - pr], irow[M - pr], ph[2 * m + 1]);
                } else {
                    const R *s = a.spec + row * 2 * p.K;
                    A = mk2(s[2 * pr], s[2 * pr + 1]);
                    B = mk2(s[2 * (M - pr)], s[2 * (M - pr) + 1]);
                }
                if (pr == 0) {  // DC and Nyquist: imaginary parts ignored
                    lds[lds_pad(0)] = mk2(A.x - B.x, A.x + B.x);  // swapped (im, re)
                } else {
                    const C2 S = mk2(A.x + B.x, A.y - B.y);   // A + conj B
                    const C2 D = mk2(A.x - B.x, A.y + B.y);   // A - conj B
                    const C2 Tw = cmul_conj(D, p.twS[pr]);             // W^(-k) D
                    // Z[k] = S + i Tw ; Z[M-k] = conj S + i conj Tw
                    lds[lds_pad(pr)] = mk2(S.y + Tw.x, S.x - Tw.y);
                    lds[lds_pad(M - pr)] = mk2(Tw.x - S.y, S.x + Tw.y);
                }
            }
        }
        if (own_half) {
            C2 A;
            if (IN_MODE == SYN_IN_POLAR) {
                A = synthesize_bin(p, mrow[M / 2], irow[M / 2], ph[2 * NP]);
            } else {
                const R *s = a.spec + row * 2 * p.K;
                A = mk2(s[M], s[M + 1]);
            }
            lds[lds_pad(M / 2)] = mk2((R)-2 * A.y, (R)2 * A.x);  // Z = 2 conj A, swapped
        }
        __syncthreads();
        fft_passes<M, T, 1>(lds, p.twP, tid, LdsLoader<M, C2>{lds});
        // ---- lds[pad(m)] = swapped z[m]: v[2m] = z.y / N, v[2m+1] = z.x / N  (C2R scaled by 1/N)
        const bool emit = f >= f0;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int m = tid + q * T;
            if (m < M) {
                const C2 z = lds[lds_pad(m)];
                const R v0 = z.y * p.inv_n, v1 = z.x * p.inv_n;
                if (OUT_MODE == SYN_OUT_RAW) {
                    a.out[row * p.N + 2 * m] = v0;
                    a.out[row * p.N + 2 * m + 1] = v1;
                } else {
                    // rewind again, cutCenter, window: frame[i] = w[i] * v[(i + off + N/2) mod N]
                    const int i0 = (2 * m - M - p.off) & (p.N - 1);
                    const int i1 = (2 * m + 1 - M - p.off) & (p.N - 1);
                    if (OUT_MODE == SYN_OUT_FRAMES) {
                        if (i0 < p.W) a.out[row * p.W + i0] = p.window[i0] * v0;
                        if (i1 < p.W) a.out[row * p.W + i1] = p.window[i1] * v1;
                    } else {
                        if (i0 < p.W) {
                            int r = base + i0;
                            if (r >= p.W) r -= p.W;
                            ring[r] = ring[r] + p.window[i0] * v0;
                        }
                        if (i1 < p.W) {
                            int r = base + i1;
                            if (r >= p.W) r -= p.W;
                            ring[r] = ring[r] + p.window[i1] * v1;
                        }
                    }
                }
            }
        }
        __syncthreads();
        if (OUT_MODE == SYN_OUT_OLA) {
            // the first H samples of the ring are complete: emit (if owned) and retire them
            R *yc = a.out + c * a.out_stride;
            for (int i = tid; i < p.H; i += T) {
                R s = (R)0;
                if (i < p.W) {
                    int r = base + i;
                    if (r >= p.W) r -= p.W;
                    s = ring[r];
                    ring[r] = (R)0;
                }
                const long long pos = f * p.H + i;
                if (emit && pos < a.out_len) yc[pos] = s;
            }
            base += p.H % p.W;
            if (base >= p.W) base -= p.W;
            // ring updates of the next frame happen behind the barriers of its FFT
        }
    }

    if (f1 == a.F && f1 > f0) {
        if (IN_MODE == SYN_IN_POLAR) {
            R *po = a.phase_out + c * p.K;
#pragma unroll
            for (int m = 0; m < NP; ++m) {
                const int pr = tid + m * T;
                if (pr < M / 2) {
                    po[pr] = ph[2 * m];
                    po[M - pr] = ph[2 * m + 1];
                }
            }
            if (own_half) po[M / 2] = ph[2 * NP];
        }
        if (OUT_MODE == SYN_OUT_OLA) {
            // tail after the last frame: the ring's remaining W - H partial sums, then zeros up to
            // the end of the last chunk (sumFramesE pads it: Frames.hs:45-47)
            __syncthreads();
            R *yc = a.out + c * a.out_stride;
            const long long start = a.F * p.H;
            for (long long i = tid; start + i < a.out_len; i += T) {
                R s = (R)0;
                if (i < p.W - p.H) {
                    int r = base + (int)i;
                    if (r >= p.W) r -= p.W;
                    s = ring[r];
                }
                yc[start + i] = s;
            }
        }
    }
}

// Sequential phase chain per (channel, bin): ph <- ph + ph_inc * H frame after frame
// (synthesisStep, Vocoder.hs:190), recording the chain value each synthesis tile starts from.
// Tile t >= 1 starts computing at frame t*tile_len - halo.
template <class R>
__global__ void phase_carry_kernel(ParamsT<R> p, const R *__restrict__ inc, long long ldk, long long F,
                                   const R *__restrict__ phase_in, R *__restrict__ carry, int tile_len,
                                   int tiles, int halo, long long channels, const long long *__restrict__ rows,
                                   long long F_inc)
{
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= channels * p.K) return;
    const long long c = g / p.K;
    const int k = (int)(g % p.K);
    R ph = phase_in[c * p.K + k];
    if (rows) {  // frames through a map (`inc` then points at the first of these channels' F_inc rows each)
        long long next = (long long)tile_len - halo;
        int t = 1;
        for (long long f = 0; f < F && t < tiles; ++f) {
            ph = phase_step(ph, inc[mapped_row(c, f, rows, F_inc, F) * ldk + k] * p.hop_d);
            if (f + 1 == next) {
                carry[(c * tiles + t) * p.K + k] = ph;
                ++t;
                next += tile_len;
            }
        }
        return;
    }
    const R *src = inc + c * F * ldk + k;
    long long next = (long long)tile_len - halo;  // frame count after which tile 1 starts
    int t = 1;
    long long f = 0;
    // sixteen loads in flight per thread (the chain is latency bound when there are few channels);
    // the additions stay in frame order
    for (; f + 16 <= F && t < tiles; f += 16) {
        R dd[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) dd[u] = src[(f + u) * ldk];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            ph = phase_step(ph, dd[u] * p.hop_d);
            if (f + u + 1 == next) {
                if (t < tiles) carry[(c * tiles + t) * p.K + k] = ph;
                ++t;
                next += tile_len;
            }
        }
    }
    for (; f < F && t < tiles; ++f) {
        const R d = src[f * ldk] * p.hop_d;
        ph = phase_step(ph, d);
        if (f + 1 == next) {
            carry[(c * tiles + t) * p.K + k] = ph;
            ++t;
            next += tile_len;
        }
    }
}

// The same chain on increments that pass through pitchShiftInterpolate n as they are read (the rows in memory are
// the unfiltered ones; see SynArgsT::pitch_n): what the carry pass computes when the filter pass is folded into
// synthesis.
__global__ void phase_carry_pitch_kernel(ParamsT<double> p, const double *__restrict__ inc, long long ldk, long long F,
                                         const double *__restrict__ phase_in, double *__restrict__ carry, int tile_len,
                                         int tiles, int halo, long long channels, double pitch_n, double pitch_rcp)
{
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= channels * p.K) return;
    const long long c = g / p.K;
    const int k = (int)(g % p.K);
    const PitchTap tp = pitch_tap(pitch_n, k, p.K);
    double ph = phase_in[c * p.K + k];
    const double *src = inc + c * F * ldk + tp.i;
    long long next = (long long)tile_len - halo;
    int t = 1;
    for (long long f = 0; f < F && t < tiles; ++f) {
        const double v = pitch_mix(tp, src[f * ldk], src[f * ldk + 1]);
        const double d = pitch_rcp != 0.0 ? v * pitch_rcp : v / pitch_n;
        ph = phase_step(ph, d * p.hop_d);
        if (f + 1 == next) {
            carry[(c * tiles + t) * p.K + k] = ph;
            ++t;
            next += tile_len;
        }
    }
}

// The synthesis phase chain alone: phase_io <- fold (ph + ph_inc * H) over the F frames, in frame
// order (synthesisStep, Vocoder.hs:190).  Used where only the phase a run of frames ends on is
// needed (time-sharded runs exchange it between segments).
template <class R = double>
__global__ void phase_advance_kernel(ParamsT<R> p, const R *__restrict__ inc, long long ldk, long long F,
                                     R *__restrict__ phase_io, long long channels, long long rpc)
{
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= channels * p.K) return;
    const long long c = g / p.K;
    const int k = (int)(g % p.K);
    R ph = phase_io[g];
    const R *src = inc + c * rpc * ldk + k;  // rpc rows per channel, the first F of them folded
    long long f = 0;
    for (; f + 16 <= F; f += 16) {
        R dd[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) dd[u] = src[(f + u) * ldk];
#pragma unroll
        for (int u = 0; u < 16; ++u) ph = phase_step(ph, dd[u] * p.hop_d);
    }
    for (; f < F; ++f) ph = phase_step(ph, src[f * ldk] * p.hop_d);
    phase_io[g] = ph;
}

// rows[r][0 .. width) *= factor, rows `pitch` elements apart (volumeFix on samples, Dunai.hs:41-42)
template <class R = double>
__global__ void scale_rows_kernel(R *__restrict__ rows, long long pitch, long long width, long long total, R factor)
{
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
        const long long r = g / width, i = g % width;
        rows[r * pitch + i] = rows[r * pitch + i] * factor;
    }
}

// Interleaved <-> planar samples (vocoder-audio frames all channels interleaved and splits them
// per frame, Audio.hs:51-55; here the split happens once for the whole buffer).
__global__ void deinterleave_kernel(const double *__restrict__ inter, long long n, int channels,
                                    double *__restrict__ planar, long long stride)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        for (int c = 0; c < channels; ++c) planar[c * stride + i] = inter[i * channels + c];
}

__global__ void interleave_kernel(const double *__restrict__ planar, long long stride, long long n, int channels,
                                  double *__restrict__ inter)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        for (int c = 0; c < channels; ++c) inter[i * channels + c] = planar[c * stride + i];
}

// 16-bit PCM at the file edge.  The reference reads and writes sound files through libsndfile's
// double interface (sourceSnd / sinkSnd, vocoder-audio/example/VocoderFile.hs:158,168-173): a
// 16-bit sample s is read as s / 32768; a double x is written as lrint(x * 32767) kept to its low
// 16 bits (libsndfile's default), or, with clipping requested, as lrint(x * 32768) saturated to
// [-32768, 32767].  lrint rounds halves to even.
__global__ void pcm16_to_planar_kernel(const short *__restrict__ inter, long long n, int channels,
                                       double *__restrict__ planar, long long stride)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        for (int c = 0; c < channels; ++c) planar[c * stride + i] = (double)inter[i * channels + c] * 0x1p-15;
}

VOC_DEV short pcm16_of(double x, bool clip)
{
    if (clip) {
        const double v = x * 32768.0;
        if (v >= 32767.0) return (short)32767;
        if (v <= -32768.0) return (short)-32768;
        return (short)(long long)rint(v);
    }
    const double v = rint(x * 32767.0);
    // the low 16 bits of the integer: exact for |v| < 2^62, anything beyond (or NaN) gives 0
    const long long w = (fabs(v) < 0x1p62) ? (long long)v : 0;
    return (short)(unsigned short)((unsigned long long)w & 0xffffull);
}

__global__ void planar_to_pcm16_kernel(const double *__restrict__ planar, long long stride, long long n, int channels,
                                       int clip, short *__restrict__ inter)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        for (int c = 0; c < channels; ++c) inter[i * channels + c] = pcm16_of(planar[c * stride + i], clip != 0);
}

// ------------------------------------------------------------------ step-level, framing, overlap-add

// analysisStep over B independent frames (Vocoder.hs:158-163)
template <class R = double>
__global__ void analysis_step_kernel(ParamsT<R> p, const R *prev_ph, const R *spec, R *ph_out, R *mag, R *inc, long long total)
{
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
        const int k = (int)(g % p.K);
        R prev = prev_ph[g], mg, ic;
        analyze_bin<R>(p, k, mk2(spec[2 * g], spec[2 * g + 1]), prev, true, (R)1, &mg, &ic);
        mag[g] = mg;
        inc[g] = ic;
        ph_out[g] = prev;
    }
}

// synthesisStep over B independent frames (Vocoder.hs:186-190)
template <class R = double>
__global__ void synthesis_step_kernel(ParamsT<R> p, const R *ph, const R *mag, const R *inc, R *ph_out, R *spec, long long total)
{
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
        R cur = ph[g];
        const cplx<R> X = synthesize_bin(p, mag[g], inc[g], cur);
        spec[2 * g] = X.x;
        spec[2 * g + 1] = X.y;
        ph_out[g] = cur;
    }
}

// addFrames (Vocoder.hs:220-222): polar -> complex, add, complex -> polar
template <class R = double>
__global__ void add_frames_kernel(const R *mag1, const R *inc1, const R *mag2, const R *inc2, R *mag_out, R *inc_out, long long total)
{
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
        R s1, c1, s2, c2;
        fast_sincos(inc1[g], &s1, &c1);
        fast_sincos(inc2[g], &s2, &c2);
        const R re = mag1[g] * c1 + mag2[g] * c2;
        const R im = mag1[g] * s1 + mag2[g] * s2;
        hs_polar(re, im, mag_out[g], inc_out[g]);
    }
}

// frameFromComplex (Vocoder.hs:217-218) over `total` bins
__global__ void frame_from_complex_kernel(const double *spec, double *mag, double *ph, long long total)
{
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x)
        hs_polar(spec[2 * g], spec[2 * g + 1], mag[g], ph[g]);
}

// frameToComplex (Vocoder.hs:213-214) over `total` bins
__global__ void frame_to_complex_kernel(const double *mag, const double *ph, double *spec, long long total)
{
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
        double s, c;
        fast_sincos(ph[g], &s, &c);
        spec[2 * g] = mag[g] * c;
        spec[2 * g + 1] = mag[g] * s;
    }
}

// mkBinary op1 op2 (VocoderJack.hs:107-110): `--multiply` = (*) on the moduli, (+) on the increments;
// `--divide` = (/) and (-)
template <bool DIVIDE>
__global__ void binary_frames_kernel(const double *mag1, const double *inc1, const double *mag2, const double *inc2,
                                     double *mag_out, double *inc_out, long long total)
{
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
        const double m1 = mag1[g], m2 = mag2[g], i1 = inc1[g], i2 = inc2[g];
        mag_out[g] = DIVIDE ? m1 / m2 : m1 * m2;
        inc_out[g] = DIVIDE ? i1 - i2 : i1 + i2;
    }
}

// genFramesOfE on a whole buffer: frame j = x[j*H .. j*H + W)  (Frames.hs:34-35)
__global__ void frames_of_kernel(const double *x, long long x_stride, double *frames, long long F, int W, int H,
                                 long long total)
{
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
        const int i = (int)(g % W);
        const long long cf = g / W;
        const long long f = cf % F, c = cf / F;
        frames[g] = x[c * x_stride + f * H + i];
    }
}

// sumFramesE: y[s] = sum over frames j covering s of frame_j[s - j*H], oldest frame first, folded
// from 0 (Frames.hs:44-45); samples past the last frame's end are the zero padding of the last chunk.
template <class R>
__global__ void sum_frames_kernel(const R *frames, long long F, int W, int H, R *y, long long y_stride,
                                  long long out_len, long long total)
{
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
        const long long s = g % out_len, c = g / out_len;
        long long jlo = (s - W + H) / H;  // ceil((s - W + 1) / H) for s - W + 1 > 0
        if (s - W + 1 <= 0) jlo = 0;
        long long jhi = s / H;
        if (jhi > F - 1) jhi = F - 1;
        R acc = (R)0;
        for (long long j = jlo; j <= jhi; ++j) acc = acc + frames[(c * F + j) * W + (s - j * H)];
        y[c * y_stride + s] = acc;
    }
}

}  // namespace voc
