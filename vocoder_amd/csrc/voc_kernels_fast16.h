// voc_kernels_fast16.h -- kernels for N = 4096, 8192 and 16384 with SIXTEEN complex points per
// thread (M = N/2 complex points, T = M/16 threads per frame, 8 (k, M-k) bin pairs per thread).
//
// Same arithmetic per bin and the same frame-to-frame state as the eight-point kernels of
// voc_kernels_fast.h; what changes is the transform.  Those kernels are bound by instruction issue, and a
// 16-byte LDS write keeps its SIMD from issuing for about as long as seven fp64 instructions, so the
// lever is the number of times a frame crosses LDS: with sixteen points per thread the M-point transform
// is radix 16 * 16 * (M/256) -- three passes, TWO exchanges -- where eight points per thread need four
// passes and three exchanges (M = 2048: 8 * 8 * 8 * 4).  N = 16384 (M = 8192) takes a fourth pass of radix 2.  The hand-over at both ends stays: the last pass
// leaves element tid + n T with thread tid, which owns the bin pairs (tid + m T, M - tid - m T) and, in
// synthesis, the output samples of exactly those elements.
//
// A workgroup is T = 128 (N = 4096), 256 (N = 8192) or 512 (N = 16384) threads and holds one frame in M * 16 bytes of LDS
// and nothing else: previous phases (analysis), accumulated phases and overlap-add partial sums
// (synthesis) live in registers.  Four (two) workgroups fit a CU: two waves per SIMD, 256 registers each.
#pragma once

#include "voc_kernels_fast.h"

// Single precision: a thread's state is half as many registers, so three waves per SIMD fit (168 registers each)
// where a workgroup is at most four waves; the eight-wave workgroups of N = 16384 stay at two.
#ifndef VOC_F32_16_WAVES
#define VOC_F32_16_WAVES 3      // analysis
#endif
#ifndef VOC_F32_SYN16_WAVES
#define VOC_F32_SYN16_WAVES 3   // synthesis
#endif
// Whether single-precision contexts take these kernels (VOC_FAST16_F32_ANA / VOC_FAST16_F32_SYN override):
// analysis yes (N = 4096: 20.6 -> 19.1 ms on the metric shape); synthesis only where there is no eight-point
// kernel (N = 16384), as in double precision.
#ifndef VOC_FAST16_F32_ANA_DEFAULT
#define VOC_FAST16_F32_ANA_DEFAULT 1
#endif
#ifndef VOC_FAST16_F32_SYN_DEFAULT
#define VOC_FAST16_F32_SYN_DEFAULT (-1)
#endif

namespace voc {

template <class R, int LOG2M> constexpr int fast16_waves(int waves_double, int waves_float)
{
    return sizeof(R) == 4 ? (LOG2M == 13 ? 2 : waves_float) : waves_double;
}

template <int LOG2M>
struct Fast16Cfg {
    static_assert(LOG2M >= 11 && LOG2M <= 13, "sixteen points per thread: N = 4096, 8192 or 16384");
    static constexpr int M = 1 << LOG2M;
    static constexpr int T = M / 16;
    static constexpr int K = M + 1;
    static constexpr bool PASS4 = LOG2M == 13;           // M = 8192: radix 16 * 16 * 16 * 2
    static constexpr int R3 = PASS4 ? 16 : M / 256;      // radix of the third pass
    static constexpr int NB3 = 16 / R3;                  // third-pass butterflies per thread
    static constexpr int NC = PASS4 ? 2 : NB3;           // twiddle bases of the passes after the second
    template <class R> static constexpr int lds_bytes() { return M * 2 * (int)sizeof(R); }
};

// Where complex element i lives in LDS: its low three bits XORed by bits 4..6.  Reads of a pass are
// contiguous over the lanes, writes are 16 (first pass) or 1 (later passes: 16 j + k, k = lane mod 16)
// elements apart over the lanes; a 16-byte write is served in groups of 8 lanes over 32 banks, a read
// in groups of 16 lanes over 64 banks, and this placement keeps both free of bank conflicts.  It is
// additive in multiples of 128, the smallest thread count here.
VOC_DEV int lds_sw16(int i) { return i ^ ((i >> 4) & 7); }

// where element tid + n T of the last pass' result sits in the thread's registers: butterfly q of the
// thread (q < NB3) produces the elements tid + (q + j NB3) T; with a fourth pass (radix 2, eight butterflies
// per thread) butterfly q produces the elements tid + (q + 8 j) T
template <int LOG2M>
VOC_DEV constexpr int nat16(int n)
{
    constexpr int R3 = Fast16Cfg<LOG2M>::R3, NB3 = Fast16Cfg<LOG2M>::NB3;
    if (Fast16Cfg<LOG2M>::PASS4) return (n % 8) * 2 + n / 8;
    return (n % NB3) * R3 + n / NB3;
}

// Forward DFT of sixteen points (TW: of v[j] w^j), natural order in and out: the even and the odd points
// as two eight-point transforms, then X[k], X[k + 8] = E[k] +- g_k O[k] with g_k = w W16^k -- the
// product with g_k folded into the butterfly (bf2_tw).  With TW the eight-point transforms carry the
// factors (w^2)^m, whose powers are formed once for both.
template <bool TW, class C2>
VOC_DEV void dft16(C2 (&v)[16], C2 w1)
{
    using R = decltype(v[0].x);
    constexpr R c1 = (R)0.92387953251128675613, s1 = (R)0.38268343236508977173, kh = (R)kSqrtHalf;
    C2 e[8], o[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        e[m] = v[2 * m];
        o[m] = v[2 * m + 1];
    }
    C2 a, b;
    if constexpr (TW) {
        const C2 w2 = csqr(w1);
        dft8_tw(e, w2);
        dft8_tw(o, w2);
        const C2 g1 = cmul(w1, mk2(c1, -s1)), g2 = mul_w8(w1), g3 = cmul(w1, mk2(s1, -c1));
        const C2 g[8] = {w1, g1, g2, g3, mul_mi(w1), mul_mi(g1), mul_mi(g2), mul_mi(g3)};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            bf2_tw(e[k], o[k], g[k], a, b);
            v[k] = a;
            v[k + 8] = b;
        }
    } else {
        dft<8>(e);
        dft<8>(o);
        const C2 g[8] = {mk2((R)1, (R)0), mk2(c1, -s1), mk2(kh, -kh), mk2(s1, -c1),
                         mk2((R)0, (R)-1), mk2(-s1, -c1), mk2(-kh, -kh), mk2(-c1, -s1)};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (k == 0) {
                a = cadd(e[k], o[k]);
                b = csub(e[k], o[k]);
            } else if (k == 4) {
                a = e[k];
                b = o[k];
                bf2_mi(a, b);
            } else {
                bf2_tw(e[k], o[k], g[k], a, b);
            }
            v[k] = a;
            v[k + 8] = b;
        }
    }
}

template <int LOG2M, class C2>
struct Fast16Tw {
    C2 b;                               // pass at stride 16: exp(-2 pi i (tid mod 16) / 256)
    C2 c[Fast16Cfg<LOG2M>::NC];         // pass at stride 256: exp(-2 pi i (tid + q T) / M), one per butterfly (with a fourth
                                        // pass: exp(-2 pi i (tid mod 256) / 4096), then exp(-2 pi i tid / M) for the pass at stride 4096)
    C2 s;                               // exp(-2 pi i tid / N): base of the split twiddles
};

template <int LOG2M, class R>
VOC_DEV void fast16_tw_init(const ParamsT<R> &p, int tid, Fast16Tw<LOG2M, cplx<R>> &t)
{
    constexpr int M = Fast16Cfg<LOG2M>::M, T = Fast16Cfg<LOG2M>::T;
    // twS[k] = exp(-2 pi i k / N), k <= M/2
    t.b = p.twS[(tid & 15) * (M / 128)];
    if constexpr (Fast16Cfg<LOG2M>::PASS4) {
        t.c[0] = p.twS[(tid & 255) * (M / 2048)];
        t.c[1] = p.twS[2 * tid];
    } else {
#pragma unroll
        for (int q = 0; q < Fast16Cfg<LOG2M>::NB3; ++q) t.c[q] = p.twS[2 * ((tid + q * T) & 255)];
    }
    t.s = p.twS[tid];
}

// at the top of every frame: see fast_tw_touch
template <int LOG2M, class C2>
VOC_DEV void fast16_tw_touch(Fast16Tw<LOG2M, C2> &t)
{
    opaque(t.b);
#pragma unroll
    for (int q = 0; q < Fast16Cfg<LOG2M>::NC; ++q) opaque(t.c[q]);
    opaque(t.s);
}

// split twiddle of bin pair slot m: exp(-2 pi i (tid + m T) / N) = s exp(-2 pi i m / 32); thread 0's slot 0
// holds the self-paired bin M/2 instead, whose factor is -i exactly
template <class C2>
VOC_DEV C2 split_tw16(C2 s, int m, bool self_pair)
{
    using R = decltype(s.x);
    constexpr double cs[8] = {1.0, 0.98078528040323044913, 0.92387953251128675613, 0.83146961230254523708,
                              0.70710678118654752440, 0.55557023301960222474, 0.38268343236508977173, 0.19509032201612826785};
    const C2 w = m == 0 ? s : cmul(s, mk2((R)cs[m], (R)-cs[8 - m]));
    return self_pair ? mk2((R)0, (R)-1) : w;
}

template <int LOG2M, class C2>
VOC_DEV void load16(const C2 *lds, int lt, C2 (&v)[16])
{
    constexpr int T = Fast16Cfg<LOG2M>::T;
    const C2 *src = lds + lds_sw16(lt);
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = src[j * T];
}

// first pass (stride 1): butterfly lt writes the elements 16 lt + j; their swizzle bits are lt mod 8
template <int LOG2M, class C2>
VOC_DEV void store16_first(C2 *lds, int lt, const C2 (&v)[16])
{
    C2 *dst = lds + lt * 16;
    const int s = lt & 7;
#pragma unroll
    for (int j = 0; j < 16; ++j) dst[(j & 8) + ((j & 7) ^ s)] = v[j];
}

// second pass (stride 16): butterfly lt writes the elements 256 (lt / 16) + k + 16 j, k = lt mod 16; their
// swizzle bits are j mod 8
template <int LOG2M, class C2>
VOC_DEV void store16_second(C2 *lds, int lt, const C2 (&v)[16])
{
    const int k = lt & 15;
    C2 *dst = lds + (lt - k) * 16;
#pragma unroll
    for (int j = 0; j < 16; ++j) dst[16 * j + (k ^ (j & 7))] = v[j];
}

// third pass (stride 256): butterfly lt + q T reads the elements lt + (q + j NB3) T and leaves its results,
// the same elements of the transform, in place: element lt + n T at v[nat16(n)].
// With a fourth pass (M = 8192): the third pass' butterfly lt (radix 16) writes the elements
// 4096 (lt / 256) + k + 256 j, k = lt mod 256 (their swizzle bits are those of k), and the fourth pass'
// butterflies lt + q T, q = 0..7 (radix 2, stride 4096) read the elements lt + q T and lt + (q + 8) T and leave
// them in place.
template <int LOG2M, class C2>
VOC_DEV void pass3_16(C2 *lds, const Fast16Tw<LOG2M, C2> &t, int lt, C2 (&v)[16])
{
    using R = decltype(v[0].x);
    constexpr int T = Fast16Cfg<LOG2M>::T, R3 = Fast16Cfg<LOG2M>::R3, NB3 = Fast16Cfg<LOG2M>::NB3;
    const C2 *src = lds + lds_sw16(lt);
    if constexpr (Fast16Cfg<LOG2M>::PASS4) {
        load16<LOG2M>(lds, lt, v);
        dft16<true>(v, t.c[0]);
        __syncthreads();
        {
            const int k = lt & 255;
            C2 *dst = lds + (lt - k) * 16 + lds_sw16(k);
#pragma unroll
            for (int j = 0; j < 16; ++j) dst[256 * j] = v[j];
        }
        __syncthreads();
        // factor of butterfly lt + q T: exp(-2 pi i (lt + q T) / M) = c[1] exp(-2 pi i q / 16)
        constexpr double cs[5] = {1.0, 0.92387953251128675613, 0.70710678118654752440, 0.38268343236508977173, 0.0};
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const C2 g = q <= 4 ? mk2((R)cs[q], (R)-cs[4 - q]) : mk2((R)-cs[8 - q], (R)-cs[q - 4]);
            const C2 w = q == 0 ? t.c[1] : cmul(t.c[1], g);
            C2 a, b;
            bf2_tw(src[q * T], src[(q + 8) * T], w, a, b);
            v[2 * q] = a;
            v[2 * q + 1] = b;
        }
    } else {
#pragma unroll
        for (int n = 0; n < 16; ++n) v[nat16<LOG2M>(n)] = src[n * T];
        if constexpr (R3 == 16) {
            dft16<true>(v, t.c[0]);
        } else {
#pragma unroll
            for (int q = 0; q < NB3; ++q) {
                C2 w[R3];
#pragma unroll
                for (int j = 0; j < R3; ++j) w[j] = v[q * R3 + j];
                dft8_tw(w, t.c[q]);
#pragma unroll
                for (int j = 0; j < R3; ++j) v[q * R3 + j] = w[j];
            }
        }
    }
}

#ifndef VOC_FFT16_FENCE
#define VOC_FFT16_FENCE 1
#endif

struct NoMark { VOC_DEV void operator()(int) const {} };
// Passes 2 and 3 of the transform whose first-pass butterflies are in v.  `early` is called before the last
// pass reads its input (loads the caller uses right after the transform), `prefetch` after the last
// butterfly (the caller's next-frame loads).  The results stay in registers; the caller puts a barrier
// between this call and its next write to the transform's LDS area.
template <int LOG2M, class C2, class Early, class Prefetch, class Mark = NoMark>
VOC_DEV void fft16_after_first(C2 *lds, const Fast16Tw<LOG2M, C2> &t, int lt, C2 (&v)[16], Early early, Prefetch prefetch, Mark mark = Mark())
{
    mark(1);
    __syncthreads();
    mark(2);
    store16_first<LOG2M>(lds, lt, v);
    mark(3);
    __syncthreads();
    mark(4);
    load16<LOG2M>(lds, lt, v);
    dft16<true>(v, t.b);
    mark(5);
    __syncthreads();
    mark(6);
    store16_second<LOG2M>(lds, lt, v);
    mark(7);
    __syncthreads();
    mark(8);
    early();
    pass3_16<LOG2M>(lds, t, lt, v);
    if (VOC_FFT16_FENCE) sched_fence();  // the results are complete here: what follows does not stretch the butterflies' live ranges
    prefetch();
}

#ifndef VOC_ANA16_WIN_AT
#define VOC_ANA16_WIN_AT 5  // analysis: the bin pair (of eight) in front of which the next frame's window values are requested
#endif

#ifndef VOC_PRIO16_ANA_T
#define VOC_PRIO16_ANA_T 3  // analysis: window product and transform
#define VOC_PRIO16_ANA_B 0  // analysis: bin pairs
#endif
#ifndef VOC_PRIO16_SYN_B
#define VOC_PRIO16_SYN_B 2  // synthesis: bin pairs
#define VOC_PRIO16_SYN_T 1  // synthesis: transform
#define VOC_PRIO16_SYN_O 0  // synthesis: window and overlap-add
#endif

// VEC2: W == N and every frame starts 16-byte aligned, so a thread's two adjacent samples and window values
// are one 16-byte load each; otherwise they are loaded one by one, with the zero padding of a short window.
// SLIDE > 0 (needs VEC2 and frames cut from one signal H = SLIDE * 2T samples apart): the thread keeps its
// sixteen sample pairs of the current frame in registers, ordered by position in the frame (slot u: the pair
// at frame positions 2 tid + 2 u T), and loads only the SLIDE newest pairs per frame, during the transform of
// the frame before.  SLIDE == 0: all sixteen pairs are loaded at the top of every frame.
// All builds do the same arithmetic: which one runs does not change a bit of the result.
template <int LOG2M, bool VEC2, int SLIDE, class R = double>
__global__ void __launch_bounds__(Fast16Cfg<LOG2M>::T, (fast16_waves<R, LOG2M>(2, VOC_F32_16_WAVES))) analysis_fast16_kernel(ParamsT<R> p, AnaArgsT<R> a)
{
    using C2 = cplx<R>;
    constexpr int M = Fast16Cfg<LOG2M>::M, T = Fast16Cfg<LOG2M>::T;
    static_assert(SLIDE >= 0 && SLIDE <= 8 && (SLIDE == 0 || VEC2), "hop of at most half a frame");
    constexpr int NEW = SLIDE > 0 ? SLIDE : 1;  // (array length: never zero)
    VOC_SMEM(C2, lds);
    const int tid = threadIdx.x;
    const long long c = blockIdx.x / a.tiles;
    const int tile = blockIdx.x % a.tiles;
    const long long f0 = (long long)tile * a.tile_len;
    const long long f1 = (f0 + a.tile_len < a.F) ? f0 + a.tile_len : a.F;
    const R *xc = a.x + c * a.chan_stride;

    // previous raw phases of the thread's eight bin pairs (bins pe = tid + m T -- thread 0, slot 0: M/2 --
    // and M - pe); phr: bin 0 in thread 0, bin M in thread T/2
    R pha[8], phb[8], phr = (R)0;
#pragma unroll
    for (int m = 0; m < 8; ++m) pha[m] = phb[m] = (R)0;
    if (f0 == 0) {
        const R *pi = a.phase_in + c * p.K;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const int pe = (m == 0 && tid == 0) ? M / 2 : tid + m * T;
            pha[m] = pi[pe];
            phb[m] = pi[M - pe];
        }
        if (tid == 0) phr = pi[0];
        if (tid == T / 2) phr = pi[M];
    }

    Fast16Tw<LOG2M, C2> tw;
    fast16_tw_init<LOG2M>(p, tid, tw);

    C2 xs[16], xn[NEW];
    auto sample_pair = [&](const R *fr, int u, int lt) -> C2 {
        if (VEC2) return at_bytes(reinterpret_cast<const C2 *>(fr) + u * T, (unsigned)lt * (unsigned)sizeof(C2), 0);
        const int j0 = 2 * (lt + u * T) - p.hd, j1 = j0 + 1;  // position in the padded frame -> in the input frame
        return mk2((j0 >= 0 && j0 < p.W) ? fr[j0] : (R)0, (j1 >= 0 && j1 < p.W) ? fr[j1] : (R)0);
    };
    const R half_vol = (R)0.5 * a.vol;  // exact; undoes the doubling of split_pair_x2
    const long long fs = f0 > 0 ? f0 - 1 : f0;  // one frame early to recover its raw phase
    // The window is requested again every frame (sixteen register pairs cannot stay): during the last bin
    // pairs of the frame before, when the transform's registers are free.
    C2 ws[16];
    auto load_window = [&](int lt) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (VEC2) {
                ws[u] = at_bytes(reinterpret_cast<const C2 *>(p.window) + u * T, (unsigned)lt * (unsigned)sizeof(C2), 0);
            } else {
                const int j0 = 2 * (lt + u * T) - p.hd, j1 = j0 + 1;
                ws[u] = mk2((j0 >= 0 && j0 < p.W) ? p.window[j0] : (R)0, (j1 >= 0 && j1 < p.W) ? p.window[j1] : (R)0);
            }
        }
    };
    {
        const R *fr = xc + fs * a.frame_stride;
#pragma unroll
        for (int u = 0; u < 16; ++u) xs[u] = sample_pair(fr, u, tid);
        load_window(tid);
    }
    VOC_STAMP_DECL;
    auto mark = [&](int i) { (void)i; VOC_STAMP(i); };
    for (long long f = fs; f < f1; ++f) {
        const bool emit = f >= f0;
        VOC_STAMP_FRAME;
        mark(0);
        fast16_tw_touch<LOG2M>(tw);
        wave_prio<VOC_PRIO16_ANA_T, LOG2M>();
        const int lw = opaque_int(tid);
        if constexpr (SLIDE == 0) {
            if (f > fs) {
                const R *fr = xc + f * a.frame_stride;
#pragma unroll
                for (int u = 0; u < 16; ++u) xs[u] = sample_pair(fr, u, lw);
            }
        }
        C2 v[16];
        // the first pass reads the frame rewound and even/odd packed: its element tid + j T is slot (j + 8) mod 16
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int u = (j + 8) & 15;
            v[j] = mk2(ws[u].x * xs[u].x, ws[u].y * xs[u].y);
        }
        dft16<false>(v, mk2((R)1, (R)0));
        fft16_after_first<LOG2M>(lds, tw, lw, v, NoHook(), [&]() {
            if (SLIDE > 0 && f + 1 < f1) {  // the pairs of the next frame that are not already in registers
                const R *fr = xc + (f + 1) * a.frame_stride;
                const int lp = opaque_int(tid);
#pragma unroll
                for (int i = 0; i < SLIDE; ++i) xn[i] = sample_pair(fr, 16 - SLIDE + i, lp);
            }
        }, mark);
        mark(9);

        wave_prio<VOC_PRIO16_ANA_B, LOG2M>();
        const long long row = c * a.rpc + f;
        R *mrow = a.mag + row * a.ldk, *irow = a.inc + row * a.ldk;
        // Element tid + n T of the spectrum is in v[nat16(n)].  The thread's bin pairs are (tid + m T,
        // M - tid - m T): the first of each pair is here (n = m), the second is the n = 15 - m of thread
        // T - tid (thread 0: its own n = 16 - m).  So the upper eight results go through LDS, and the
        // first, because thread 0's first pair is the self-paired bin M/2 and DC / Nyquist need Z[0].
        const int lt = opaque_int(tid);
        const bool self0 = lt == 0;
        const unsigned pe0 = self0 ? M / 2 : (unsigned)lt;
        PairBinsT<R> bins;
        if constexpr (SLIDE > 0) bins.frame(p, lt);
        __syncthreads();  // the last pass' input has been read by everyone
        mark(10);
        {
            C2 *zw = lds + lds_sw16(lt);
#pragma unroll
            for (int n = 8; n < 16; ++n) zw[n * T] = v[nat16<LOG2M>(n)];
            zw[0] = v[nat16<LOG2M>(0)];
        }
        mark(11);
        __syncthreads();
        mark(12);
        const C2 *zhi = lds + lds_sw16(M - lt);
        constexpr int SZ = (int)sizeof(R);
        // byte offsets in an STFT row: slot 0 has its own pair; slots 1..3 sit around slot 2, slots 4..7 around slot 6
        const unsigned o_pe0 = pe0 * SZ, o_q0 = (unsigned)(M * SZ) - o_pe0;
        const unsigned o_peA = (unsigned)(lt + 2 * T) * SZ, o_qA = (unsigned)(M * SZ) - o_peA;
        const unsigned o_peB = (unsigned)(lt + 6 * T) * SZ, o_qB = (unsigned)(M * SZ) - o_peB;
        // (the second bin of a pair is read from LDS while the pair before it is worked on)
        C2 Qn = lds[lds_sw16(M - pe0)];
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const unsigned pe = m == 0 ? pe0 : (unsigned)(lt + m * T);
            const unsigned q = M - pe;
            const unsigned ope = m == 0 ? o_pe0 : (m < 4 ? o_peA : o_peB), oq = m == 0 ? o_q0 : (m < 4 ? o_qA : o_qB);
            const int ipe = m == 0 ? 0 : (m < 4 ? (m - 2) : (m - 6)) * T * SZ, iq = -ipe;
            C2 P, w;
            const C2 Q = Qn;
            if (m < 7) Qn = zhi[-(m + 1) * T];
            if (m == VOC_ANA16_WIN_AT) load_window(lt);
            if (m == 0) {
                P = lds[lds_sw16(pe)];
                w = split_tw16(tw.s, 0, self0);
            } else {
                P = v[nat16<LOG2M>(m)];
                w = split_tw16(tw.s, m, false);
            }
            C2 Xa, Xb;
            split_pair_x2(P, Q, w, Xa, Xb);
            R ma, mb, na, nb;
            polar_pair<VOC_POLAR_LEAN != 0>(Xa, Xb, half_vol, ma, na, mb, nb);
            if (emit) {
                at_bytes(mrow, ope, ipe) = ma;
                at_bytes(mrow, oq, iq) = mb;
                if constexpr (SLIDE > 0) {  // H = SLIDE * N/16: a power of two, like N
                    R ia, ib;
                    inc_pow2_pair(p, bins, m == 0 ? (self0 ? (R)(M / 2) : bins_ft(bins)) : bins_ft(bins) + (R)(m * T), M, (int)pe, (int)q, na - pha[m], nb - phb[m], ia, ib);
                    at_bytes(irow, ope, ipe) = ia;
                    at_bytes(irow, oq, iq) = ib;
                } else {
                    at_bytes(irow, ope, ipe) = phase_inc(p, pe, na - pha[m]);
                    at_bytes(irow, oq, iq) = phase_inc(p, q, nb - phb[m]);
                }
            }
            pha[m] = na;
            phb[m] = nb;
            sched_fence();
        }
        // DC and Nyquist bins are real: phase 0 or pi, magnitude |x|; one each for two different waves
        if (tid == 0 || tid == T / 2) {
            const C2 z0 = lds[0];
            const bool dc = tid == 0;
            const R xr = dc ? z0.x + z0.y : z0.x - z0.y;
            const int kb = dc ? 0 : M;
            const R pn = xr < (R)0 ? (R)kPi : (R)0;
            if (emit) {
                mrow[kb] = fabs(xr) * a.vol;
                irow[kb] = SLIDE > 0 ? phase_inc_pow2(p, kb, pn - phr) : phase_inc(p, kb, pn - phr);
            }
            phr = pn;
        }
        mark(13);
        if constexpr (SLIDE > 0) {  // slide the sample window by one hop
#pragma unroll
            for (int u = 0; u < 16 - SLIDE; ++u) xs[u] = xs[u + SLIDE];
#pragma unroll
            for (int i = 0; i < SLIDE; ++i) xs[16 - SLIDE + i] = xn[i];
        }
        // the next frame's first LDS write sits behind the first barrier of fft16_after_first
    }

    VOC_STAMP_FLUSH(0);
    if (f1 == a.F && f1 > f0) {
        R *po = a.phase_out + c * p.K;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const int pe = (m == 0 && tid == 0) ? M / 2 : tid + m * T;
            po[pe] = pha[m];
            po[M - pe] = phb[m];  // (thread 0, m = 0: the same bin and the same value twice)
        }
        if (tid == 0) po[0] = phr;
        if (tid == T / 2) po[M] = phr;
    }
}

#ifndef VOC_SYN16_WAVES
#define VOC_SYN16_WAVES 2
#endif
#ifndef VOC_SYN16_ROW_PF
#define VOC_SYN16_ROW_PF 0  // synthesis: bin pairs (of eight) whose STFT row entries are requested a frame ahead
#endif
#ifndef VOC_SYN16_ROW_PF_F32
#define VOC_SYN16_ROW_PF_F32 0  // the same on float (more than none spills there too)
#endif
#ifndef VOC_SYN16_ROW_AHEAD
#define VOC_SYN16_ROW_AHEAD 4  // synthesis: further bin pairs whose row entries are in flight ahead of their use
#endif
#ifndef VOC_SYN16_WIN_AHEAD
#define VOC_SYN16_WIN_AHEAD 2  // synthesis: steps (of four) of the overlap-add whose window values are in flight ahead of their use
#endif

// Synthesis with the overlap-add in registers.  Requires W == N and H = NHP * 2T (ALIGNED: 16-byte aligned y rows;
// otherwise the emitted pairs are stored as two values each):
// thread t owns the sample PAIRS 2 (t + j T), + 1, j = 0..15, of the window, and the partial sums shift by
// NHP pair slots per frame.
template <int LOG2M, int NHP, class R = double, bool MAPPED = false, bool ALIGNED = true>
__global__ void __launch_bounds__(Fast16Cfg<LOG2M>::T, (fast16_waves<R, LOG2M>(VOC_SYN16_WAVES, VOC_F32_SYN16_WAVES))) synthesis_fast16_kernel(ParamsT<R> p, SynArgsT<R> a)
{
    using C2 = cplx<R>;
    constexpr int M = Fast16Cfg<LOG2M>::M, T = Fast16Cfg<LOG2M>::T;
    constexpr int NP = 16;                 // sample pairs per thread
    constexpr bool FRAMES = NHP == 16;     // no overlap-add: every frame is written whole, [C][F][W] (synthesisStage)
    constexpr int NK = NP - NHP;           // pairs whose partial sums outlive the frame
    constexpr int NT = NK > 0 ? NK : 1;
    VOC_SMEM(C2, lds);
    const int tid = threadIdx.x;
    const bool whole = (long long)blockIdx.x < a.unsplit;  // leading channels: one tile over all frames
    const long long rest = (long long)blockIdx.x - a.unsplit;
    const long long c = whole ? (long long)blockIdx.x : a.unsplit + rest / a.tiles;
    const int tile = whole ? 0 : (int)(rest % a.tiles);
    const long long f0 = (long long)tile * a.tile_len;
    const long long f1 = (whole || f0 + a.tile_len >= a.F) ? a.F : f0 + a.tile_len;
    const long long halo = FRAMES ? 0 : ((long long)a.halo < f0 ? (long long)a.halo : f0);
    const long long fs = f0 - halo;

    // accumulated phases of the thread's eight bin pairs, and of DC / Nyquist in thread 0
    R pha[8], phb[8], ph0 = (R)0, phm = (R)0;
    {
        const R *pi = (tile == 0) ? a.phase_in + c * p.K : a.carry + ((c - a.unsplit) * a.tiles + tile) * p.K;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const unsigned pe = (m == 0 && tid == 0) ? M / 2 : (unsigned)(tid + m * T);
            pha[m] = pi[pe];
            phb[m] = pi[M - pe];
        }
        if (tid == 0) {
            ph0 = pi[0];
            phm = pi[M];
        }
    }
    Fast16Tw<LOG2M, C2> tw;
    fast16_tw_init<LOG2M>(p, tid, tw);
    C2 tail[NT];  // overlap-add partial sums that later frames still add to
#pragma unroll
    for (int j = 0; j < NT; ++j) tail[j] = mk2((R)0, (R)0);

    constexpr int SZ = (int)sizeof(R);
    // STFT row of a frame: (mag, ph_inc) of the thread's sixteen bins, and of DC / Nyquist in thread 0.  The
    // entries of the first PF bin pairs are requested during the previous frame's transform (a wave has
    // registers for that many), the others at the top of the frame, behind the bin pairs that are already here.
    constexpr int PF = sizeof(R) == 4 ? VOC_SYN16_ROW_PF_F32 : VOC_SYN16_ROW_PF;
    R mga[8], mgb[8], ica[8], icb[8], mg0 = (R)0, mgm = (R)0, ic0 = (R)0, icm = (R)0;
    auto load_pairs = [&](long long f, int lt, int m0, int m1, bool real_bins) {
        const R *mrow = MAPPED ? mag_row_of(a, c, f) : a.mag + (c * a.F + f) * a.ldk;
        const R *irow = MAPPED ? inc_row_of(a, c, f) : a.inc + (c * a.F + f) * a.ldk;
        const unsigned o_pe0 = (lt == 0 ? (unsigned)(M / 2) : (unsigned)lt) * SZ, o_q0 = (unsigned)(M * SZ) - o_pe0;
        const unsigned o_peA = (unsigned)(lt + 2 * T) * SZ, o_qA = (unsigned)(M * SZ) - o_peA;
        const unsigned o_peB = (unsigned)(lt + 6 * T) * SZ, o_qB = (unsigned)(M * SZ) - o_peB;
#pragma unroll
        for (int m = m0; m < m1; ++m) {
            const unsigned ope = m == 0 ? o_pe0 : (m < 4 ? o_peA : o_peB), oq = m == 0 ? o_q0 : (m < 4 ? o_qA : o_qB);
            const int ipe = m == 0 ? 0 : (m < 4 ? (m - 2) : (m - 6)) * T * SZ, iq = -ipe;
            mga[m] = at_bytes(mrow, ope, ipe);
            ica[m] = at_bytes(irow, ope, ipe);
            mgb[m] = at_bytes(mrow, oq, iq);
            icb[m] = at_bytes(irow, oq, iq);
        }
        if (real_bins && tid == 0) {
            mg0 = mrow[0];
            ic0 = irow[0];
            mgm = mrow[M];
            icm = irow[M];
        }
    };
    if (PF > 0) load_pairs(fs, opaque_int(tid), 0, PF, false);
    VOC_STAMP_DECL;
    auto mark = [&](int i) { (void)i; VOC_STAMP(i); };
    for (long long f = fs; f < f1; ++f) {
        VOC_STAMP_FRAME;
        mark(0);
        const int lt = opaque_int(tid);  // keeps this frame's index arithmetic inside the loop
        fast16_tw_touch<LOG2M>(tw);
        wave_prio<VOC_PRIO16_SYN_B, LOG2M>();
        // (the rest of the row arrives AHEAD bin pairs before its use)
        constexpr int AHEAD = VOC_SYN16_ROW_AHEAD;
        load_pairs(f, lt, PF, (PF + AHEAD < 8 ? PF + AHEAD : 8), true);
        const bool self0 = lt == 0;
        const unsigned pe0 = self0 ? M / 2 : (unsigned)lt;
        // The first pass reads the elements tid + j T.  Those with j < 8 are this thread's own Z[pe]; the
        // others are the Z[M - pe] of thread T - tid (thread 0: its own).  Only the latter go through LDS.
        C2 *zhi = lds + lds_sw16(M - lt);
        C2 za[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const bool self = m == 0 && self0;
            if (m + PF + AHEAD < 8) load_pairs(f, lt, m + PF + AHEAD, m + PF + AHEAD + 1, false);
            C2 A, B;
            synth_pair<true>(p, mga[m], ica[m], pha[m], mgb[m], icb[m], phb[m], A, B);
            C2 Za, Zb;
            merge_pair(A, B, split_tw16(tw.s, m, self), Za, Zb);
            za[m] = Za;
            if (m == 0) lds[lds_sw16(M - pe0)] = Zb;
            else zhi[-m * T] = Zb;
            sched_fence();  // one bin pair at a time
        }
        if (tid == 0) {  // DC and Nyquist: only the real parts enter the C2R transform
            const C2 A = synthesize_bin<R, true>(p, mg0, ic0, ph0);
            const C2 B = synthesize_bin<R, true>(p, mgm, icm, phm);
            za[0] = mk2(A.x - B.x, A.x + B.x);  // element 0; thread 0's slot 0 was bin M/2, which it reads back as element 8 T
        }
        if (PF > 0 && f + 1 < f1) load_pairs(f + 1, lt, 0, PF, false);
        wave_prio<VOC_PRIO16_SYN_T, LOG2M>();
        mark(13);
        __syncthreads();
        mark(14);
        C2 v[16];
        {
            const C2 *zlo = lds + lds_sw16(lt);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                v[j] = za[j];
                v[j + 8] = zlo[(j + 8) * T];
            }
        }
        dft16<false>(v, mk2((R)1, (R)0));
        // Overlap-add in four steps of four sample pairs, the partial sums updated in place (pair j's new sum
        // becomes slot j - NHP), so that what is live shrinks as the transform's results are consumed.  Window
        // values (times 1/N: exact, N is a power of two; uniform bases T pairs apart plus one lane offset) are
        // requested VOC_SYN16_WIN_AHEAD steps ahead, the first ones before the last pass of the transform.
        C2 wq[4][4];
        const unsigned lane16 = (unsigned)lt * (unsigned)sizeof(C2);
        auto load_win = [&](int q) {
#pragma unroll
            for (int j = 0; j < 4; ++j) wq[q][j] = at_bytes(reinterpret_cast<const C2 *>(p.window_n) + (4 * q + j) * T, lane16, 0);
        };
        constexpr int WA = VOC_SYN16_WIN_AHEAD;
        fft16_after_first<LOG2M>(lds, tw, lt, v, [&]() {
#pragma unroll
            for (int q = 0; q < WA; ++q) load_win(q);
        }, NoHook(), mark);
        mark(9);
        const bool emit = f >= f0;
        wave_prio<VOC_PRIO16_SYN_O, LOG2M>();
        // v holds the swapped transform: sample 2i = z.y / N, 2i + 1 = z.x / N of element (i + M/2) mod M
        C2 out[NHP];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (q + WA < 4) load_win(q + WA);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int j = 4 * q + jj;
                const C2 z = v[nat16<LOG2M>((j + 8) & 15)];
                const C2 w = wq[q][jj];
                const C2 old = j < NK ? tail[j] : mk2((R)0, (R)0);
                const C2 cur = mk2(old.x + w.x * z.y, old.y + w.y * z.x);
                if (j < NHP) out[j] = cur;
                else tail[j - NHP] = cur;
            }
            sched_fence();
        }
        // (without this the compiler moves the loads and sums of the emitted pairs into the branch below)
#pragma unroll
        for (int j = 0; j < NHP; ++j) opaque(out[j]);
        if (emit) {
            R *dst = FRAMES ? a.out + (c * a.F + f) * p.W : a.out + c * a.out_stride + f * p.H;
            if constexpr (ALIGNED) {
                C2 *yc = reinterpret_cast<C2 *>(dst);
#pragma unroll
                for (int j = 0; j < NHP; ++j) at_bytes(yc + j * T, lane16, 0) = out[j];
            } else {  // rows that are not 16-byte aligned: the pairs as two values each
                R *ys = dst + 2 * lt;
#pragma unroll
                for (int j = 0; j < NHP; ++j) {
                    ys[2 * j * T] = out[j].x;
                    ys[2 * j * T + 1] = out[j].y;
                }
            }
        }
        mark(10);
        __syncthreads();  // the transform's LDS area is rewritten by the next frame's spectrum
        mark(11);
    }
    VOC_STAMP_FLUSH(1);

    if (f1 == a.F && f1 > f0) {
        R *po = a.phase_out + c * p.K;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const unsigned pe = (m == 0 && tid == 0) ? M / 2 : (unsigned)(tid + m * T);
            po[pe] = pha[m];
            po[M - pe] = phb[m];  // (thread 0, m = 0: the same bin and the same value twice)
        }
        if (tid == 0) {
            po[0] = ph0;
            po[M] = phm;
        }
        // tail after the last frame: the remaining N - H partial sums, then the zero padding of the last
        // chunk (sumFramesE, Frames.hs:45-47)
        if constexpr (!FRAMES) {
            R *yc = a.out + c * a.out_stride;
            const long long start = a.F * p.H;
#pragma unroll
            for (int j = 0; j < NK; ++j) {
                const long long pos = start + 2 * (tid + j * T);
                if (pos < a.out_len) yc[pos] = tail[j].x;
                if (pos + 1 < a.out_len) yc[pos + 1] = tail[j].y;
            }
            for (long long pos = start + 2 * NK * T + tid; pos < a.out_len; pos += T) yc[pos] = (R)0;
        }
    }
}

}  // namespace voc
