// voc_filters.h -- frequency-domain filters applied to the STFT where it rests in HBM between the
// analysis and the synthesis kernel (vocoder/src/Vocoder/Filter.hs), and the frame-axis gather
// of playSpeed (vocoder-conduit/src/Vocoder/Conduit/Filter.hs:122-136).
//
// A filter chain (composeFilters, Filter.hs:54-55) is a short program of row operations.  One
// workgroup takes one STFT row (one frame of one channel: K moduli and K phase increments),
// keeps it in registers, runs every operation of the chain on it in order and writes it back, so
// a chain of any length reads and writes the STFT once; operations that read neighbouring bins
// exchange the row through LDS.
//
// Rounding follows the reference operation by operation (compiled with -ffp-contract=off):
// gains are one multiply per filter in chain order, interpolation is (1-k)*v[i] + k*v[i+1] with
// both products rounded, the convolution divides every term by the kernel sum inside the
// left-to-right sum.
#pragma once

namespace voc {

enum { FOP_GAIN = 0, FOP_PITCH = 1, FOP_CONV = 2, FOP_ENVELOPE = 3, FOP_RANDOM_PHASE = 4 };
constexpr int kMaxFilterOps = 12;

struct FilterOp {
    int kind;
    int len;              // kernel length (convolution, envelope)
    double a;             // pitch coefficient n | kernel sum s
    const double *table;  // gain per bin [K] | kernel [len]
    unsigned long long seed;
};

struct FilterProg {
    int n;
    // what the chain as a whole needs from / changes in memory (filled in by the host): a gain-only
    // chain never touches ph_inc, a random-phase chain never touches the moduli and never reads ph_inc
    int load_mag, store_mag, load_inc, store_inc;
    FilterOp op[kMaxFilterOps];
};

// interpolate n v at index x (Filter.hs:118-125)
VOC_DEV double interpolate_at(const double *v, int K, double n, int x)
{
    const double xp = n * (double)x;
    const bool inside = xp < (double)(K - 1);  // else i + 1 >= length v: 0
    // (no branch: out of range the two reads go to the start of the row and their mix is dropped)
    const int i = inside ? (int)floor(xp) : 0;
    const double k = xp - (double)i;
    const double lo = (1.0 - k) * v[i];
    const double hi = k * v[i + 1];
    return inside ? lo + hi : 0.0;
}

// convolution ker mag at bin k (Filter.hs:132-137): gmag is mag with h zeros on both sides
VOC_DEV double convolve_at(const double *mag, int K, const double *ker, int len, double s, int k)
{
    const int h = len / 2;
    double acc = 0.0;
    for (int i = 0; i < len; ++i) {
        const int j = i + k - h;
        const double g = (j >= 0 && j < K) ? mag[j] : 0.0;
        acc = acc + (ker[i] * g) / s;
    }
    return acc;
}

// Philox-4x32-10 (Salmon et al., SC'11): counter (c0..c3), key (k0, k1) -> first two output words.
// Each round's two 32 x 32 -> 64 bit products are written as one 64-bit multiplication each (one
// v_mad_u64_u32 instead of a v_mul_hi_u32 / v_mul_lo_u32 pair: integer multiplications issue at a quarter of
// the rate of other vector instructions and are all this generator costs).
VOC_DEV void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1, unsigned *r0,
                           unsigned *r1)
{
#pragma unroll
    for (int round = 0; round < 10; ++round) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    *r0 = c0;
    *r1 = c1;
}

// convolve_at with every term multiplied by 1/s and corrected once instead of divided: a few
// ulp-of-a-term cheaper than the exact quotient; used inside the envelope only, whose log / exp are
// not bit-reproducible across math libraries anyway.
VOC_DEV double convolve_fast_at(const double *mag, int K, const double *ker, int len, double s, double rs, int k)
{
    const int h = len / 2;
    double acc = 0.0;
    for (int i = 0; i < len; ++i) {
        const int j = i + k - h;
        const double a = ker[i] * ((j >= 0 && j < K) ? mag[j] : 0.0);
        const double q = a * rs;
        acc = acc + fma(fma(-s, q, a), rs, q);
    }
    return acc;
}

// log(x) for a positive normal x (the envelope takes it of a modulus plus 2^-24): the classical reduction
// x = 2^k m, sqrt(1/2) <= m < sqrt(2), f = m - 1, s = f / (2 + f), log(m) = f - (f^2/2 - s (f^2/2 + R(s^2)))
// with the degree-14 even polynomial R of fdlibm's e_log.c (public coefficients; error below 1 ulp), without the
// device library's tests for zero, negative, infinite and subnormal arguments: 40 instructions where its log has 95.
VOC_DEV double log_pos(double x)
{
#ifdef VOC_EMU
    int k;
    double m = frexp(x, &k);
#else
    int k = __builtin_amdgcn_frexp_exp(x);
    double m = __builtin_amdgcn_frexp_mant(x);
#endif
    const bool low = m < 0.70710678118654752440;   // m in [1/2, 1): bring it into [sqrt(1/2), sqrt(2))
    m = low ? m + m : m;
    k = low ? k - 1 : k;
    const double f = m - 1.0;
    const double s = div_pos<false>(f, 2.0 + f);
    const double dk = (double)k;
    const double z = s * s, w = z * z;
    const double t1 = w * fma(w, fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
    const double t2 = z * fma(w, fma(w, fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01), 2.857142874366239149e-01), 6.666666666666735130e-01);
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    return dk * 6.93147180369123816490e-01 - ((hfsq - fma(s, hfsq + R, dk * 1.90821492927058770002e-10)) - f);
}

// rows too long for two staging arrays in LDS
constexpr bool filter_one_array(int log2m) { return log2m >= 14; }

template <int LOG2M>
struct FiltCfg {
    static constexpr int M = 1 << LOG2M;
    static constexpr int T = (M / 8) < 64 ? 64 : ((M / 8) > 1024 ? 1024 : (M / 8));  // threads per row
    static constexpr int NE = M / T + 1;  // bins per thread: k = tid + T*j, the last slot holds bin M (thread 0 only)
};

// One workgroup per STFT row at a time.  The row lives in registers (thread t holds bins t + T*j);
// operations that read neighbouring bins stage it through LDS (2 * (K+1) doubles, only allocated
// when the chain holds such an operation; rows of more than 8193 bins -- LOG2M = 14, frame lengths above
// 16384 -- have room for one array only and stage the moduli and the increments one after the other).
// S: the type the STFT is stored in.  The arithmetic is the reference's, in double, whatever S is: a float
// STFT (single-precision contexts) is widened on load and rounded once on store.
// ONLY >= 0: a build for chains made of that one kind of operation.  The registers of the kernel are those of its
// hungriest case (240: two waves per SIMD); the envelope alone fits three (166 registers, none spilled) and gains a
// sixth from the third (3.6 -> 3.0 ms per 180 k rows of 2049 bins).
template <int ONLY>
constexpr int filter_min_waves()  // waves per SIMD the build is made for (registers per thread: 512 / that)
{
    return ONLY == FOP_ENVELOPE ? 3 : ((ONLY == FOP_RANDOM_PHASE || ONLY == FOP_GAIN) ? 4 : 1);
}
template <int LOG2M, class S = double, int ONLY = -1>
__global__ __launch_bounds__(FiltCfg<LOG2M>::T, (filter_min_waves<ONLY>())) void filter_rows_kernel(Params p, FilterProg prog,
                                                                        S *__restrict__ mag,
                                                                        S *__restrict__ inc, long long ldk,
                                                                        long long rows, long long F, long long rpc,
                                                                        long long first_frame, long long first_channel)
{
    using Cf = FiltCfg<LOG2M>;
    constexpr int T = Cf::T, NE = Cf::NE;
    const int K = p.K;  // at most M + 1 (M + 1 for power-of-two frame lengths)
    VOC_SMEM(double, lds);
    constexpr bool ONE = filter_one_array(LOG2M);
    double *lm = lds, *li = ONE ? lds : lds + ((K + 1) & ~1);
    const int tid = threadIdx.x;
    const double ee = 5.9604644775390625e-08;  // 2**(-24), Filter.hs:147
    for (long long row = blockIdx.x; row < rows; row += gridDim.x) {
        // row = channel * F + frame; the arrays hold rpc >= F rows per channel
        const long long prow = rpc == F ? row : (row / F) * rpc + row % F;
        S *mrow = mag + prow * ldk, *irow = inc + prow * ldk;
        // (loads and table reads of slots past the last bin are clamped onto the last bin instead of predicated:
        //  a predicated access is a block of its own with its own wait, and the values are never stored)
        double m[NE], ic[NE];
#pragma unroll
        for (int j = 0; j < NE; ++j) {
            const int k = tid + T * j, kc = k < K ? k : K - 1;
            m[j] = 0.0;
            ic[j] = 0.0;
            if (prog.load_mag) m[j] = (double)mrow[kc];
            if (prog.load_inc) ic[j] = (double)irow[kc];
        }
        for (int o = 0; o < prog.n; ++o) {
            const FilterOp op = prog.op[o];
            switch (ONLY >= 0 ? ONLY : op.kind) {
            case FOP_GAIN:  // linearAmplitudeFilter (Filter.hs:76-77): mag * gain, phase increments kept
#pragma unroll
                for (int j = 0; j < NE; ++j) {
                    const int k = tid + T * j;
                    m[j] = m[j] * op.table[k < K ? k : K - 1];
                }
                break;
            case FOP_PITCH:  // pitchShiftInterpolate (Filter.hs:128-129)
#pragma unroll
                for (int j = 0; j < NE; ++j) {
                    const int k = tid + T * j;
                    if (k < K) {
                        lm[k] = m[j];
                        if (!ONE) li[k] = ic[j];
                    }
                }
                __syncthreads();
#pragma unroll
                for (int j = 0; j < NE; ++j) {
                    const int k = tid + T * j;
                    m[j] = interpolate_at(lm, K, op.a, k);  // (0 without a read for every k past the row)
                    if (!ONE) ic[j] = interpolate_at(li, K, op.a, k) / op.a;
                }
                __syncthreads();
                if (ONE) {
#pragma unroll
                    for (int j = 0; j < NE; ++j) {
                        const int k = tid + T * j;
                        if (k < K) li[k] = ic[j];
                    }
                    __syncthreads();
#pragma unroll
                    for (int j = 0; j < NE; ++j) {
                        const int k = tid + T * j;
                        if (k < K) ic[j] = interpolate_at(li, K, op.a, k) / op.a;
                    }
                    __syncthreads();
                }
                break;
            case FOP_CONV:  // convolutionFilter (Filter.hs:140-141): amplitudeFilter0 zeroes ph_inc
#pragma unroll
                for (int j = 0; j < NE; ++j) {
                    const int k = tid + T * j;
                    if (k < K) lm[k] = m[j];
                }
                __syncthreads();
#pragma unroll
                for (int j = 0; j < NE; ++j) {
                    const int k = tid + T * j;
                    if (k < K) m[j] = convolve_at(lm, K, op.table, op.len, op.a, k);
                    ic[j] = 0.0;
                }
                __syncthreads();
                break;
            case FOP_ENVELOPE: {  // envelopeFilter (Filter.hs:144-152)
                const double rs = 1.0 / op.a;
                const int h = op.len / 2;
                // Both staging arrays together hold the row of logarithms with its zero margins (the reference
                // pads the convolved array with len/2 zeros on either side, Filter.hs:132-137), so a tap is one
                // LDS read and one multiply-add with the kernel value times 1/s -- no bounds test, no division.
                // (Rows too long for two arrays, or kernels longer than a row, take the term-by-term form.)
                if (!ONE && op.len <= K && T * NE + op.len <= 2 * ((K + 1) & ~1)) {
#pragma unroll
                    for (int j = 0; j < NE; ++j) {
                        const int k = tid + T * j;
                        if (k < K) lm[h + k] = log_pos(m[j] + ee);
                    }
                    for (int i = tid; i < op.len; i += T) lm[i < h ? i : K + i] = 0.0;  // h in front, len - h behind
                    __syncthreads();
                    double acc[NE];
#pragma unroll
                    for (int j = 0; j < NE; ++j) acc[j] = 0.0;
                    const double *row = lm + tid;
#pragma unroll 4
                    for (int i = 0; i < op.len; ++i) {
                        const double kr = op.table[i] * rs;
                        // (no test per bin here: slots past the last bin read on inside the two arrays -- T NE + len
                        //  <= 2 (K + 1) -- and their sums are never stored; with a test every tap of every bin becomes
                        //  a predicated block of its own, one LDS read waited for at a time)
#pragma unroll
                        for (int j = 0; j < NE; ++j) acc[j] = fma(kr, row[T * j + i], acc[j]);
                    }
#pragma unroll
                    for (int j = 0; j < NE; ++j) {
                        if (tid + T * j < K) m[j] = exp(acc[j]) + (-ee);
                        ic[j] = 0.0;
                    }
                    __syncthreads();
                    break;
                }
#pragma unroll
                for (int j = 0; j < NE; ++j) {
                    const int k = tid + T * j;
                    if (k < K) lm[k] = log_pos(m[j] + ee);
                }
                __syncthreads();
#pragma unroll
                for (int j = 0; j < NE; ++j) {
                    const int k = tid + T * j;
                    if (k < K) m[j] = exp(convolve_fast_at(lm, K, op.table, op.len, op.a, rs, k)) + (-ee);
                    ic[j] = 0.0;
                }
                __syncthreads();
            } break;
            case FOP_RANDOM_PHASE: {  // randomPhaseFilter (Filter.hs:156-157): ph_inc uniform on [0, 2 pi)
                // The reference draws fresh numbers for every frame it is handed, however the frames are
                // chunked.  The counter is therefore the frame's position in the whole signal, not in this
                // call: (bin | position in the chain << 16, channel, frame low, frame high).
                const unsigned long long chan = (unsigned long long)(first_channel + row / F);
                const unsigned long long frame = (unsigned long long)(first_frame + row % F);
#pragma unroll
                for (int j = 0; j < NE; ++j) {
                    const int k = tid + T * j;
                    unsigned r0, r1;
                    philox4x32_10((unsigned)k | ((unsigned)o << 16), (unsigned)chan, (unsigned)frame, (unsigned)(frame >> 32),
                                  (unsigned)op.seed, (unsigned)(op.seed >> 32), &r0, &r1);
                    const unsigned long long bits = (((unsigned long long)r0 << 32) | r1) >> 11;
                    ic[j] = ((double)bits * 1.1102230246251565e-16) * kTwoPi;  // 53 random bits * 2^-53 * 2 pi
                }
            } break;
            default: break;
            }
        }
#pragma unroll
        for (int j = 0; j < NE; ++j) {
            const int k = tid + T * j;
            if (k < K) {
                if (prog.store_mag) mrow[k] = (S)m[j];
                if (prog.store_inc) irow[k] = (S)ic[j];
            }
        }
    }
}

// playSpeed: output frame j of every channel is input frame src[j] (Conduit/Filter.hs:122-136).
template <class S = double>
__global__ void gather_frames_kernel(const S *__restrict__ mag_in, const S *__restrict__ inc_in, long long ldk,
                                     long long F_in, const long long *__restrict__ src, long long F_out,
                                     S *__restrict__ mag_out, S *__restrict__ inc_out, long long total)
{
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
        const long long k = g % ldk;
        const long long row = g / ldk;
        const long long j = row % F_out, c = row / F_out;
        const long long from = (c * F_in + src[j]) * ldk + k;
        mag_out[g] = mag_in[from];
        inc_out[g] = inc_in[from];
    }
}

// playSpeed on a run of frames of a longer stream (Conduit/Filter.hs:122-136): the filter yields frame i
// floor((i+1) num / den) - floor(i num / den) times, so output frame t of the stream is input frame
// floor(((t+1) den - 1) / num).  Output rows out_first .. out_first + G of every channel are copied from
// the input rows held from in_first on; rows of a channel are in_rpc / out_rpc rows apart.
template <class S = double>
__global__ void replicate_frames_kernel(const S *__restrict__ mag_in, const S *__restrict__ inc_in,
                                        long long in_rpc, long long in_first, long long ldk, long long num, long long den,
                                        long long out_first, long long G, S *__restrict__ mag_out,
                                        S *__restrict__ inc_out, long long out_rpc, long long total)
{
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
        const long long k = g % ldk;
        const long long row = g / ldk;
        const long long j = row % G, c = row / G;
        const long long i = ((out_first + j + 1) * den - 1) / num - in_first;
        const long long from = (c * in_rpc + i) * ldk + k, to = (c * out_rpc + j) * ldk + k;
        mag_out[to] = mag_in[from];
        inc_out[to] = inc_in[from];
    }
}

}  // namespace voc
