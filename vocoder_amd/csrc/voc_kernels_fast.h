// voc_kernels_fast.h -- the hot-path kernels for 1024 <= N <= 8192 (M = N/2 complex points,
// T = M/8 threads per frame: 8 complex points and 4 (k, M-k) bin pairs per thread).
//
// Same arithmetic as the generic kernels in voc_kernels.h, with one exception: analysis uses the shorter
// polar conversion (LEAN, see div_pos there), whose magnitudes lie within 3e-15 relative and whose phases
// within 3e-14 rad of the full form's -- against bars of 1e-9.  What differs is how the work is laid on the machine.  These kernels
// are bound by instruction issue: a SIMD spends about four fifths of its cycles on fp64 vector
// instructions and most of the rest on LDS instructions (a 16-byte LDS write keeps its SIMD from issuing
// vector instructions for about as long as seven of them), so the design counts instructions:
//   - nothing is loaded that can be formed: each thread keeps one twiddle base per pass and multiplies the
//     powers out, and the products with them are folded into the butterflies (bf2_tw);
//   - nothing goes through LDS that its own thread uses next: the last pass leaves element tid + n T with
//     thread tid, which owns the bin pairs (tid + m T, M - tid - m T) and, in synthesis, the output samples
//     of exactly those elements (nat_idx, KEEP_LAST) -- half of the spectrum of analysis, half of the
//     first-pass input of synthesis and the whole input of the overlap-add never leave their registers;
//   - every memory access is a uniform base, one 32-bit lane offset and an immediate (at_bytes);
//   - the next frame's samples (analysis) are requested during the current frame's last pass, always BEFORE
//     the stores of the current stage (loads and stores retire in issue order);
//   - a wave's priority follows its progress through the frame (wave_prio).
// State carried from frame to frame: the previous raw phases of analysis sit in LDS behind the transform
// area, the two bins of a pair side by side; its samples slide through registers (slot u of a thread: the
// pair at frame positions 2 tid + 2 u T).  Synthesis with overlap-add (N >= 2048, double) is built for two
// waves per SIMD instead of three: its accumulated phases, its overlap-add partial sums (pair slots that
// shift by H / 2T per frame), its window and the STFT row of the next frame all live in registers.
#pragma once

#include "voc_kernels.h"

#ifndef VOC_FMA_BF
#define VOC_FMA_BF 1  // twiddle products folded into the first butterfly stage: a = t + w u as two fused multiply-adds per component, b = 2 t - a
#endif

#ifndef VOC_FMA_MORE
#define VOC_FMA_MORE 0  // (measured: synthesis -0.6 %, analysis unchanged; left off) products that feed one sum directly (window times sample, modulus times cosine / sine, window times output sample, the half-turn factors inside the radix-8 butterfly) are fused into that sum
#endif

#ifndef VOC_WS_PREFETCH
#define VOC_WS_PREFETCH 0  // fetch the window with the next frame's samples (aligned case)
#endif

#ifndef VOC_SYN_TW_LDS
#define VOC_SYN_TW_LDS 0  // synthesis: twiddle rows of the passes at stride 8 and 64 read from LDS instead of formed by multiplication
#endif

#ifndef VOC_SYN_SINCOS_TAB
#define VOC_SYN_SINCOS_TAB 0  // (measured: 98 vector instructions per thread and frame fewer, synthesis unchanged at 27.0 ms; left off) two-wave synthesis: sine and cosine from a 64-step table of the circle in LDS (the area the register-held phases left free) and short series
#endif

#ifndef VOC_SYN_PHASE_REGS
#define VOC_SYN_PHASE_REGS 1  // synthesis: accumulated phases in registers instead of LDS where the row prefetch is off
#endif

#ifndef VOC_SYN_ROW_PREFETCH
#define VOC_SYN_ROW_PREFETCH 2  // synthesis, next frame's STFT row loaded during this frame's transform: 0 never, 1 always, 2 by size
#endif

#ifndef VOC_SYN_FENCE
#define VOC_SYN_FENCE 1
#endif

#ifndef VOC_SYN_WIN_EARLY
#define VOC_SYN_WIN_EARLY 8  // synthesis: how many of the eight window pairs of the overlap-add are requested before the transform's last pass
#endif

#ifndef VOC_TW_OPAQUE_TOP
#define VOC_TW_OPAQUE_TOP 1  // twiddle bases marked changed once per frame, in place, instead of a marked copy at every use
#endif

#ifndef VOC_SYN_WIN_SCALED
#define VOC_SYN_WIN_SCALED 1  // overlap-add multiplies by window / N (exact: N is a power of two) instead of scaling every sample by 1/N first
#endif

#ifndef VOC_ANA_PHP_PAIRS
#define VOC_ANA_PHP_PAIRS 1  // analysis: previous phases of a bin pair side by side in LDS (one 16-byte access per pair)
#endif

#ifndef VOC_POLAR_LEAN
#define VOC_POLAR_LEAN 1  // analysis: the shorter polar conversion (see div_pos in voc_kernels.h)
#endif


#ifndef VOC_LAST_IN_REGS
#define VOC_LAST_IN_REGS 1  // the transform's last results (analysis, synthesis) and first inputs (synthesis) stay in registers where the same thread uses them
#endif

#ifndef VOC_POLAR_UNROLL
#define VOC_POLAR_UNROLL 1  // bin pairs in flight in the analysis polar loop
#endif

#define VOC_PRAGMA_(x) _Pragma(#x)
#define VOC_UNROLL(n) VOC_PRAGMA_(unroll n)

#ifndef VOC_SYN_FAT
#define VOC_SYN_FAT 1  // synthesis with overlap-add, N >= 2048, double: budgeted for two waves per SIMD (256 registers): the next STFT row is requested a frame ahead and the window stays in registers
#endif

#ifndef VOC_FAST_WAVES
#define VOC_FAST_WAVES 3  // waves per SIMD the fast kernels are register-budgeted for
#endif

namespace voc {

// Diagnostic builds (-DVOC_STAMPS): every wave sums the shader-clock time it spends between the marks of the
// frame loop and adds its sums to a device table the launcher prints.  A mark drains the wave's LDS and scalar
// memory counters first (the clock is read through the scalar memory path), so the marked build is slower
// than the product and shows where a frame's time goes, not how long a frame takes.
#if defined(VOC_STAMPS) && !defined(VOC_EMU)
constexpr int kStampSlots = 16;
__device__ unsigned long long voc_stamps[2][kStampSlots + 1];
#define VOC_STAMP_DECL unsigned st_acc[kStampSlots] = {}; unsigned long long st_prev = __builtin_readcyclecounter(); unsigned st_frames = 0
#define VOC_STAMP(i) do { const unsigned long long t_ = __builtin_readcyclecounter(); st_acc[i] += (unsigned)(t_ - st_prev); st_prev = t_; } while (0)
#define VOC_STAMP_FRAME ++st_frames
#define VOC_STAMP_FLUSH(kid) do { if ((threadIdx.x & 63) == 0) { for (int i_ = 0; i_ < kStampSlots; ++i_) atomicAdd(&voc_stamps[kid][i_], (unsigned long long)st_acc[i_]); atomicAdd(&voc_stamps[kid][kStampSlots], (unsigned long long)st_frames); } } while (0)
#else
#define VOC_STAMP_DECL
#define VOC_STAMP(i)
#define VOC_STAMP_FRAME
#define VOC_STAMP_FLUSH(kid)
#endif

template <int LOG2M>
struct FastCfg {
    static constexpr int M = 1 << LOG2M;
    static constexpr int T = M / 8;
    static constexpr int K = M + 1;
    static constexpr int R4 = M / 512;               // radix of the fourth pass (1: there is none)
    static constexpr int NB4 = R4 > 1 ? 8 / R4 : 1;  // fourth-pass butterflies per thread
    static constexpr int LDS_CPLX = M;               // swizzled, not padded: see lds_sw
    static constexpr int LDS_BYTES = LDS_CPLX * 16 + ((K + 1) & ~1) * 8;  // double
    template <class R> static constexpr int lds_bytes() { return (LDS_CPLX * 2 + ((K + 1) & ~1)) * (int)sizeof(R); }
};

// real-FFT split of the pair (Z[pe], Z[q]), q = M - pe:  X[pe] = E + W^pe O,  X[q] = conj(E - W^pe O)
// with E = (P + conj Q)/2, O = (P - conj Q)/(2i).  Returned are 2 X[pe] and 2 X[q]: the halvings are
// left out (exact powers of two: the phase is unchanged, the magnitude is exactly doubled and the
// caller folds the 1/2 into its volume factor).
template <class C2>
VOC_DEV void split_pair_x2(C2 P, C2 Q, C2 tw, C2 &Xa, C2 &Xb)
{
    using R = decltype(P.x);
    const C2 E = mk2(P.x + Q.x, P.y - Q.y);
    const C2 O = mk2(P.y + Q.y, Q.x - P.x);
#if VOC_FMA_BF
    // E + W O as fused multiply-adds, E - W O as 2 E - (E + W O): ten instructions instead of twelve
    Xa = mk2(fma(tw.x, O.x, fma(-tw.y, O.y, E.x)), fma(tw.x, O.y, fma(tw.y, O.x, E.y)));
    Xb = mk2(fma((R)2, E.x, -Xa.x), fma((R)-2, E.y, Xa.y));
#else
    const C2 U = cmul(tw, O);
    Xa = cadd(E, U);
    Xb = cconj(csub(E, U));
#endif
}

// inverse of split_pair, without the factor 1/2 (folded into the final 1/N):
// Z[pe] = S + i Tw, Z[q] = conj S + i conj Tw with S = A + conj B, Tw = W^-pe (A - conj B);
// both are returned with re and im swapped (the forward FFT then performs the backward transform).
template <class C2>
VOC_DEV void merge_pair(C2 A, C2 B, C2 tw, C2 &Za, C2 &Zb)
{
    using R = decltype(A.x);
    const C2 S = mk2(A.x + B.x, A.y - B.y);
    const C2 D = mk2(A.x - B.x, A.y + B.y);
#if VOC_FMA_BF
    // with Tw = D conj(tw): Za = (S.y + Tw.x, S.x - Tw.y) as fused multiply-adds, Zb = (Za.x - 2 S.y, 2 S.x - Za.y)
    Za = mk2(fma(D.x, tw.x, fma(D.y, tw.y, S.y)), fma(D.x, tw.y, fma(-D.y, tw.x, S.x)));
    Zb = mk2(fma((R)-2, S.y, Za.x), fma((R)2, S.x, -Za.y));
#else
    const C2 Tw = cmul_conj(D, tw);
    Za = mk2(S.y + Tw.x, S.x - Tw.y);
    Zb = mk2(Tw.x - S.y, S.x + Tw.y);
#endif
}

// merge_pair for A = mag (c + i s) given in parts: the products with the modulus are fused into the sums
// with B (two instructions less than forming A first)
template <class C2, class R>
VOC_DEV void merge_pair_polar(R mag, R c, R s, C2 B, C2 tw, C2 &Za, C2 &Zb)
{
    const C2 S = mk2(fma(mag, c, B.x), fma(mag, s, -B.y));
    const C2 D = mk2(fma(mag, c, -B.x), fma(mag, s, B.y));
    Za = mk2(fma(D.x, tw.x, fma(D.y, tw.y, S.y)), fma(D.x, tw.y, fma(-D.y, tw.x, S.x)));
    Zb = mk2(fma((R)-2, S.y, Za.x), fma((R)2, S.x, -Za.y));
}

#if VOC_F32_CPLX_PACKED
// split_pair_x2 and merge_pair on packed registers (float): E = P + conj Q, O = -i P + swap Q and the like are one
// instruction each; same roundings
VOC_DEV void split_pair_x2(float2 P, float2 Q, float2 tw, float2 &Xa, float2 &Xb)
{
    const f32x2 E = fma2(v2(Q), sign_pm(), v2(P));
    const f32x2 O = fma2(swap2(v2(P)), sign_pm(), swap2(v2(Q)));
    const f32x2 A = fma2(splat2(tw.x), O, fma2(splat2(tw.y) * sign_mp(), swap2(O), E));
    const f32x2 Bc = fma2(splat2(2.0f), E, -A);   // conj of 2 X[q]
    Xa = c2(A);
    Xb = make_float2(Bc.x, -Bc.y);
}
VOC_DEV void merge_pair(float2 A, float2 B, float2 tw, float2 &Za, float2 &Zb)
{
    const f32x2 S = fma2(v2(B), sign_pm(), v2(A));
    const f32x2 D = fma2(v2(B), sign_mp(), v2(A));
    // Za = (S.y, S.x) + D.x (tw.x, tw.y) + D.y (tw.y, -tw.x)
    const f32x2 a = fma2(splat2(D.x), v2(tw), fma2(splat2(D.y) * sign_pm(), swap2(v2(tw)), swap2(S)));
    // Zb = (a.x - 2 S.y, 2 S.x - a.y)
    const f32x2 b = fma2(splat2(2.0f), swap2(S), -a) * sign_mp();
    Za = c2(a);
    Zb = c2(b);
}
#endif

// Where complex element i of the transform lives in LDS: i with its low three bits XORed by the next
// three.  A pass reads 16-byte elements that are contiguous over the lanes and writes elements that are
// 8 (or 64, 512) apart over the lanes; the LDS serves a 16-byte read in groups of 16 lanes over 64
// banks and a 16-byte write in groups of 8 lanes over 32 banks.  With this placement both are free of
// bank conflicts (one spare slot per eight elements, the usual padding, leaves every read two-way
// conflicted: a quarter of all LDS cycles), nothing is wasted on gaps, and sw(i + c) = sw(i) + c for
// every c that is a multiple of 64, so offsets by multiples of the thread count stay immediates.
VOC_DEV int lds_sw(int i) { return i ^ ((i >> 3) & 7); }

// ---- the four Stockham passes, split into load / compute / store so that callers can place
//      prefetches between them.  Index algebra as in fft_passes (voc_kernels.h).
//
// Twiddles are not loaded per frame.  In the pass at stride NS thread tid needs
// exp(-2 pi i j k / (8 NS)), j = 1..7, for the one k = tid mod NS it owns in every frame: it keeps
// the j = 1 factor in registers for the life of the workgroup and forms the powers by
// multiplication (each power is at most three complex multiplications from the table value).

// Marks a register-resident value as changed without emitting an instruction.  Everything derived
// from the frame-invariant twiddle bases below is loop invariant, and the compiler would otherwise
// hoist all of it (84 registers of twiddle powers) out of the frame loop and spill it.
template <class C2>
VOC_DEV void opaque(C2 &a)
{
#ifndef VOC_EMU
    asm volatile("" : "+v"(a.x), "+v"(a.y));
#else
    (void)a;
#endif
}

#if VOC_F32_CPLX_PACKED && !defined(VOC_EMU)
// (a float pair is marked as one 64-bit register pair: marked component by component, the compiler at hand took the
//  second component of a loop-carried pair for the undefined half of a splat operand and dropped its initial copy --
//  visible in the assembly as "kill: def $vgprN killed $vgprM" -- in the one-wave analysis kernels)
VOC_DEV void opaque(float2 &a)
{
    f32x2 t = v2(a);
    asm volatile("" : "+v"(t));
    a = c2(t);
}
#endif

// Same for an integer: index arithmetic derived from an opaque copy of the thread id is redone
// every frame (a few integer instructions) instead of being hoisted as dozens of loop-invariant
// address registers.
VOC_DEV int opaque_int(int v)
{
#ifndef VOC_EMU
    asm volatile("" : "+v"(v));
#endif
    return v;
}

// Wave priorities by phase of the frame loop (s_setprio; -1: leave the priority alone).  A SIMD holds one
// wave of each of the CU's three workgroups; when two of them can issue, the one further along in its
// frame goes first, so that its workgroup reaches its next barrier and its stores leave early instead
// of all three idling together.  Measured at N = 4096: synthesis 26.0 -> 23.8 ms, analysis 27.2 -> 26.9 ms
// per 2.16 M frames (N = 2048: +2.6 % on the round trip, N = 8192: no change); the order the other way round, or one level for a whole kernel, gains nothing.
#ifndef VOC_PRIO_ANA_T
#define VOC_PRIO_ANA_T 0  // analysis: window product and transform
#define VOC_PRIO_ANA_B 3  // analysis: bin pairs
#endif
#ifndef VOC_PRIO_SYN_B
#define VOC_PRIO_SYN_B 0  // synthesis: bin pairs
#define VOC_PRIO_SYN_T 1  // synthesis: transform
#define VOC_PRIO_SYN_O 2  // synthesis: window and overlap-add
#endif
template <int PRIO, int LOG2M>
VOC_DEV void wave_prio()
{
#if !defined(VOC_EMU)
    if constexpr (PRIO >= 0 && LOG2M >= 10) __builtin_amdgcn_s_setprio(PRIO);  // one-wave workgroups (N = 1024) lose 3 % with it
#endif
}

VOC_DEV void sched_fence()
{
#ifndef VOC_EMU
    __builtin_amdgcn_sched_barrier(0);
#endif
}

// base[byte_off + imm]: a uniform base pointer, a 32-bit per-lane byte offset and a small constant --
// the shape the hardware addresses in one instruction (scalar base + vector offset + immediate), spelled
// out so that the compiler does not rebuild a 64-bit address per access
template <class R>
VOC_DEV const R &at_bytes(const R *base, unsigned byte_off, int imm)
{
    return *reinterpret_cast<const R *>(reinterpret_cast<const char *>(base) + ((size_t)byte_off + (long long)imm));
}
template <class R>
VOC_DEV R &at_bytes(R *base, unsigned byte_off, int imm)
{
    return *reinterpret_cast<R *>(reinterpret_cast<char *>(base) + ((size_t)byte_off + (long long)imm));
}

template <class C2>
VOC_DEV C2 csqr(C2 a) { return mk2(fma(a.x, a.x, -(a.y * a.y)), (a.x + a.x) * a.y); }

// (a, b) <- (t + w u, t - w u) without forming w u: the sum as two fused multiply-adds per component, the
// difference as 2 t - sum (one each).  Six instructions where the product and a plain butterfly take
// eight; the difference carries the sum's rounding error, i.e. an error relative to the larger of the
// two results -- the order every butterfly's inputs already have.
template <class C2>
VOC_DEV void bf2_tw(C2 t, C2 u, C2 w, C2 &a, C2 &b)
{
    using R = decltype(t.x);
    a = mk2(fma(w.x, u.x, fma(-w.y, u.y, t.x)), fma(w.x, u.y, fma(w.y, u.x, t.y)));
    b = mk2(fma((R)2, t.x, -a.x), fma((R)2, t.y, -a.y));
}

#if VOC_F32_CPLX_PACKED
VOC_DEV float2 csqr(float2 a) { return cmul(a, a); }
VOC_DEV void bf2_tw(float2 t, float2 u, float2 w, float2 &a, float2 &b)
{
    const f32x2 wn = splat2(w.y) * sign_mp();
    const f32x2 A = fma2(splat2(w.x), v2(u), fma2(wn, swap2(v2(u)), v2(t)));
    a = c2(A);
    b = c2(fma2(splat2(2.0f), v2(t), -A));
}
#endif

// the radix-8 butterfly after its first stage (v[j] +- v[j + 4] done): as dft_n for eight points, with the
// factor sqrt(1/2) of the eighth-turn twiddles fused into the butterfly that follows them
template <class C2>
VOC_DEV void dft8_rest(C2 (&v)[8])
{
    using R = decltype(v[0].x);
    [[maybe_unused]] const R kh = (R)kSqrtHalf;
    bf2(v[0], v[2]);
    bf2(v[1], v[3]);
    bf2_mi(v[4], v[6]);
#if VOC_FMA_MORE
    {   // v5 <- W8 v5 = kh p5, v7 <- W8^3 v7 = kh p7, then (v5, v7) <- (v5 + v7, v5 - v7)
        const C2 p5 = mk2(v[5].x + v[5].y, v[5].y - v[5].x);
        const C2 p7 = mk2(v[7].y - v[7].x, -(v[7].x + v[7].y));
        const C2 t7 = mk2(p7.x * kh, p7.y * kh);
        v[5] = mk2(fma(kh, p5.x, t7.x), fma(kh, p5.y, t7.y));
        v[7] = mk2(fma(kh, p5.x, -t7.x), fma(kh, p5.y, -t7.y));
    }
#else
    v[5] = mul_w8(v[5]);
    v[7] = mul_w8_3(v[7]);
    bf2(v[5], v[7]);
#endif
    bf2(v[0], v[1]);
    bf2_mi(v[2], v[3]);
    bf2(v[4], v[5]);
    bf2_mi(v[6], v[7]);
    C2 t;
    t = v[1]; v[1] = v[4]; v[4] = t;
    t = v[3]; v[3] = v[6]; v[6] = t;
}

// radix-8 butterfly of v[j] w^j, j = 0..7 (w^0 = 1), natural order in and out
template <class C2>
VOC_DEV void dft8_tw(C2 (&v)[8], C2 w1)
{
    const C2 w2 = csqr(w1);
    const C2 w3 = cmul(w2, w1);
    const C2 w4 = csqr(w2);
    C2 a, b;
    bf2_tw(v[0], v[4], w4, a, b);
    v[0] = a;
    v[4] = b;
    bf2_tw(cmul(v[1], w1), v[5], cmul(w4, w1), a, b);
    v[1] = a;
    v[5] = b;
    bf2_tw(cmul(v[2], w2), v[6], csqr(w3), a, b);
    v[2] = a;
    v[6] = b;
    bf2_tw(cmul(v[3], w3), v[7], cmul(w4, w3), a, b);
    v[3] = a;
    v[7] = b;
    dft8_rest(v);
}

template <class C2>
VOC_DEV void dft4_tw(C2 (&v)[4], C2 w1)
{
    const C2 w2 = csqr(w1);
    C2 a, b;
    bf2_tw(v[0], v[2], w2, a, b);
    v[0] = a;
    v[2] = b;
    bf2_tw(cmul(v[1], w1), v[3], cmul(w2, w1), a, b);
    v[1] = a;
    v[3] = b;
    bf2(v[0], v[1]);
    bf2_mi(v[2], v[3]);
    const C2 t = v[1];
    v[1] = v[2];
    v[2] = t;
}

template <class C2>
VOC_DEV void dft2_tw(C2 (&v)[2], C2 w1)
{
    C2 a, b;
    bf2_tw(v[0], v[1], w1, a, b);
    v[0] = a;
    v[1] = b;
}

template <int LOG2M, class C2>
struct FastTw {
    C2 b, c;                           // base factors of the passes at stride 8 and 64
    C2 d[FastCfg<LOG2M>::NB4];         // base factors of the fourth pass (stride 512), one per butterfly
    C2 s;                              // exp(-2 pi i tid / N): base of the split twiddles
};

// Called at the top of every frame: everything derived from the bases is then recomputed per frame
// (a few dozen multiplications) instead of living in 84 registers across the loop -- and, unlike a
// marked copy at each use, costs no instruction.
template <int LOG2M, class C2>
VOC_DEV void fast_tw_touch(FastTw<LOG2M, C2> &t)
{
    if (VOC_TW_OPAQUE_TOP) {
        opaque(t.b);
        opaque(t.c);
#pragma unroll
        for (int q = 0; q < FastCfg<LOG2M>::NB4; ++q) opaque(t.d[q]);
        opaque(t.s);
    }
}

template <int LOG2M, class R>
VOC_DEV void fast_tw_init(const ParamsT<R> &p, int tid, FastTw<LOG2M, cplx<R>> &t)
{
    constexpr int T = FastCfg<LOG2M>::T, R4 = FastCfg<LOG2M>::R4;
    t.b = p.twP[7 + (tid & 7)];
    t.c = p.twP[63 + (tid & 63)];
#pragma unroll
    for (int q = 0; q < FastCfg<LOG2M>::NB4; ++q) t.d[q] = R4 > 1 ? p.twP[511 + ((tid + q * T) & 511)] : mk2((R)1, (R)0);
    t.s = p.twS[tid];
}

// split twiddle of bin pair slot m of this thread: exp(-2 pi i (tid + m T) / N) = s * exp(-2 pi i m / 16);
// thread 0's slot 0 holds the self-paired bin M/2 instead, whose factor is exp(-i pi/2) = -i exactly.
template <class C2>
VOC_DEV C2 split_tw(C2 s, int m, bool self_pair)
{
    using R = decltype(s.x);
    if (!VOC_TW_OPAQUE_TOP) opaque(s);
    const R cr = (R)(m == 0 ? 1.0 : (m == 1 ? 0.92387953251128673848 : (m == 2 ? 0.70710678118654752440 : 0.38268343236508977173)));
    const R ci = (R)(m == 0 ? 0.0 : (m == 1 ? -0.38268343236508977173 : (m == 2 ? -0.70710678118654752440 : -0.92387953251128673848)));
    const C2 w = cmul(s, mk2(cr, ci));  // m == 0: s * (1, 0) = s exactly
    return self_pair ? mk2((R)0, (R)-1) : w;
}

template <int LOG2M, class C2>
VOC_DEV void load8(const C2 *lds, int tid, C2 (&v)[8])
{
    constexpr int T = FastCfg<LOG2M>::T;
    const C2 *src = lds + lds_sw(tid);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = src[j * T];
}

template <int LOG2M, int NS, class C2>  // store of a radix-8 pass at stride NS (1, 8 or 64)
VOC_DEV void store8(C2 *lds, int tid, const C2 (&v)[8])
{
    // element (tid - k) * 8 + k + j NS, k = tid mod NS.  The swizzle bits of that index are
    // NS = 1: tid mod 8;  NS = 8: j;  NS = 64: (k / 8) mod 8 -- only the last is the same for all j.
    const int k = tid & (NS - 1);
    if constexpr (NS == 1) {
        C2 *dst = lds + tid * 8;
        const int sw = tid & 7;
#pragma unroll
        for (int j = 0; j < 8; ++j) dst[j ^ sw] = v[j];
    } else if constexpr (NS == 8) {
        C2 *dst = lds + (tid - k) * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) dst[j * 8 + (k ^ j)] = v[j];
    } else {
        C2 *dst = lds + (tid - k) * 8 + lds_sw(k);
#pragma unroll
        for (int j = 0; j < 8; ++j) dst[j * NS] = v[j];
    }
}

// v[j] *= w^j, j = 1..7
template <class C2>
VOC_DEV void twiddle8(C2 (&v)[8], C2 w1)
{
    if (!VOC_TW_OPAQUE_TOP) opaque(w1);
    const C2 w2 = csqr(w1);
    v[1] = cmul(v[1], w1);
    v[2] = cmul(v[2], w2);
    const C2 w3 = cmul(w2, w1);
    v[3] = cmul(v[3], w3);
    const C2 w4 = csqr(w2);
    v[4] = cmul(v[4], w4);
    v[5] = cmul(v[5], cmul(w4, w1));
    v[6] = cmul(v[6], csqr(w3));
    v[7] = cmul(v[7], cmul(w4, w3));
}

// v[j] *= row[(j - 1) * stride], j = 1..7: the same factors read from a table ([j-1][k], k fastest)
template <class C2>
VOC_DEV void twiddle8_tab(C2 (&v)[8], const C2 *row, int stride)
{
#pragma unroll
    for (int j = 1; j < 8; ++j) v[j] = cmul(v[j], row[(j - 1) * stride]);
}

// fourth pass: load, twiddle, butterfly (results stay in v); pass4_store writes them
template <int LOG2M, class C2>
VOC_DEV void pass4_compute(const C2 *lds, int tid, const FastTw<LOG2M, C2> &t, C2 (&v)[8])
{
    constexpr int R4 = FastCfg<LOG2M>::R4, T = FastCfg<LOG2M>::T, M = FastCfg<LOG2M>::M;
    if constexpr (R4 > 1) {
        constexpr int NBF = M / R4;
#pragma unroll
        for (int q = 0; q < FastCfg<LOG2M>::NB4; ++q) {
            const C2 *src = lds + lds_sw(tid) + q * T;
            C2 w[R4];
#pragma unroll
            for (int j = 0; j < R4; ++j) w[j] = src[j * NBF];
            C2 pw = t.d[q];
            if (!VOC_TW_OPAQUE_TOP) opaque(pw);
            if constexpr (VOC_FMA_BF != 0) {
                if constexpr (R4 == 8) dft8_tw(w, pw);
                else if constexpr (R4 == 4) dft4_tw(w, pw);
                else dft2_tw(w, pw);
#pragma unroll
                for (int j = 0; j < R4; ++j) v[q * R4 + j] = w[j];
                continue;
            }
            w[1] = cmul(w[1], pw);
            if constexpr (R4 >= 4) {
                const C2 w2 = csqr(pw);
                w[2] = cmul(w[2], w2);
                const C2 w3 = cmul(w2, pw);
                w[3] = cmul(w[3], w3);
                if constexpr (R4 == 8) {
                    const C2 w4 = csqr(w2);
                    w[4] = cmul(w[4], w4);
                    w[5] = cmul(w[5], cmul(w4, pw));
                    w[6] = cmul(w[6], csqr(w3));
                    w[7] = cmul(w[7], cmul(w4, w3));
                }
            }
            dft<R4>(w);
#pragma unroll
            for (int j = 0; j < R4; ++j) v[q * R4 + j] = w[j];
        }
    }
}

template <int LOG2M, class C2>
VOC_DEV void pass4_store(C2 *lds, int tid, const C2 (&v)[8])
{
    constexpr int R4 = FastCfg<LOG2M>::R4, T = FastCfg<LOG2M>::T;
    if constexpr (R4 > 1) {
#pragma unroll
        for (int q = 0; q < FastCfg<LOG2M>::NB4; ++q) {
            const int i = tid + q * T;
            const int k = i & 511;
            C2 *dst = lds + (i - k) * R4 + lds_sw(k);
#pragma unroll
            for (int j = 0; j < R4; ++j) dst[j * 512] = v[q * R4 + j];
        }
    }
}

// Where the results of the last pass sit in a thread's registers.  A Stockham transform ends in natural
// order, and the last pass leaves element tid + n T (n = 0..7) of the result with thread tid: butterfly
// q of the thread (q < NB4) produces the elements tid + q T + j NB4 T, so element n is v[(n mod NB4) R4 +
// n / NB4] (v[n] where the third pass is the last).
template <int LOG2M>
VOC_DEV constexpr int nat_idx(int n)
{
    constexpr int R4 = FastCfg<LOG2M>::R4, NB4 = FastCfg<LOG2M>::NB4;
    return R4 > 1 ? (n % NB4) * R4 + n / NB4 : n;
}

// Passes 2..4 of the transform whose first-pass butterflies are already in v.  `prefetch` is
// called once, after the last butterfly and before the last store: the place for the caller's
// next-frame loads.
// KEEP_LAST: the results of the last pass are not written back: they stay in v, element tid + n T at
// v[nat_idx(n)], and the caller puts a barrier between this call and its next write to the transform's
// LDS area (other threads may still be reading the last pass' input).
// twl: optional table in LDS, the twiddle rows of the passes at stride 8 (56 entries) and 64 (448 entries).
// `early` is called once before the last pass reads its input: the place for loads that the caller uses right after the transform.
// ldsB (with KEEP_LAST and a fourth pass only): a second buffer of M elements.  The exchanges then alternate
// between the two -- first-pass results to ldsB, second-pass results to lds, third-pass results to ldsB -- and
// each needs ONE barrier, between its writes and its reads: the buffer an exchange writes was last read one
// exchange earlier, and every thread has passed the barrier in between.  The caller's own writes to lds
// before the next transform need no barrier after this call either (lds was last read before the third barrier).
struct NoHook { VOC_DEV void operator()() const {} };
template <int LOG2M, bool KEEP_LAST = false, bool TWO_BUFFERS = false, class C2, class Prefetch, class Early = NoHook>
VOC_DEV void fft_after_first(C2 *lds, const FastTw<LOG2M, C2> &t, int tid, C2 (&v)[8], Prefetch prefetch,
                             const C2 *twl = nullptr, Early early = Early(), C2 *ldsB = nullptr)
{
    constexpr int R4 = FastCfg<LOG2M>::R4;
    static_assert(!TWO_BUFFERS || (KEEP_LAST && R4 > 1), "two buffers: four passes, last results in registers");
    if constexpr (TWO_BUFFERS) {
        {
            // (a scheduling fence where the other form has a barrier: without it the compiler overlaps the passes
            //  and runs out of registers)
            sched_fence();
            store8<LOG2M, 1>(ldsB, tid, v);
            __syncthreads();
            load8<LOG2M>(ldsB, tid, v);
            if (twl) {
                twiddle8_tab(v, twl + (tid & 7), 8);
                dft<8>(v);
            } else if (VOC_FMA_BF) {
                dft8_tw(v, t.b);
            } else {
                twiddle8(v, t.b);
                dft<8>(v);
            }
            sched_fence();
            store8<LOG2M, 8>(lds, tid, v);
            __syncthreads();
            load8<LOG2M>(lds, tid, v);
            if (twl) {
                twiddle8_tab(v, twl + 56 + (tid & 63), 64);
                dft<8>(v);
            } else if (VOC_FMA_BF) {
                dft8_tw(v, t.c);
            } else {
                twiddle8(v, t.c);
                dft<8>(v);
            }
            sched_fence();
            store8<LOG2M, 64>(ldsB, tid, v);
            __syncthreads();
            early();
            pass4_compute<LOG2M>(ldsB, tid, t, v);
            prefetch();
            return;
        }
    }
    __syncthreads();
    store8<LOG2M, 1>(lds, tid, v);
    __syncthreads();
    load8<LOG2M>(lds, tid, v);
    if (twl) {
        twiddle8_tab(v, twl + (tid & 7), 8);
        dft<8>(v);
    } else if (VOC_FMA_BF) {
        dft8_tw(v, t.b);
    } else {
        twiddle8(v, t.b);
        dft<8>(v);
    }
    __syncthreads();
    store8<LOG2M, 8>(lds, tid, v);
    __syncthreads();
    if (R4 == 1) early();
    load8<LOG2M>(lds, tid, v);
    if (twl) {
        twiddle8_tab(v, twl + 56 + (tid & 63), 64);
        dft<8>(v);
    } else if (VOC_FMA_BF) {
        dft8_tw(v, t.c);
    } else {
        twiddle8(v, t.c);
        dft<8>(v);
    }
    if (R4 == 1) prefetch();
    if (R4 == 1 && KEEP_LAST) return;
    __syncthreads();
    store8<LOG2M, 64>(lds, tid, v);
    __syncthreads();
    if (R4 > 1) {
        early();
        pass4_compute<LOG2M>(lds, tid, t, v);
        prefetch();
        if (KEEP_LAST) return;
        __syncthreads();
        pass4_store<LOG2M>(lds, tid, v);
        __syncthreads();
    }
}

// VEC2: W == N and every frame starts 16-byte aligned, so a thread's two adjacent samples and
// window values are one 16-byte load each.
// SLIDE > 0 (needs VEC2 and frames that start H = SLIDE * 2T samples apart in one signal): the
// thread keeps its eight sample pairs of the current frame in registers, ordered by position in
// the frame; the next frame is the same window moved by H, i.e. by SLIDE pair slots, so only the
// SLIDE newest pairs are loaded.  Every input sample is then fetched once per workgroup instead of
// N/H times (those re-reads were missing L2: measured 3.2x the unique input bytes).
template <int LOG2M, bool VEC2, int SLIDE, class R = double>
__global__ void __launch_bounds__(FastCfg<LOG2M>::T, VOC_FAST_WAVES) analysis_fast_kernel(ParamsT<R> p, AnaArgsT<R> a)
{
    using C2 = cplx<R>;
    constexpr int M = FastCfg<LOG2M>::M, T = FastCfg<LOG2M>::T;
    constexpr int NEW = SLIDE > 0 ? SLIDE : 8;  // pair slots loaded per frame
    VOC_SMEM(C2, lds);
    R *php = reinterpret_cast<R *>(lds + FastCfg<LOG2M>::LDS_CPLX);  // previous raw phase, K bins
    const int tid = threadIdx.x;
    const long long c = blockIdx.x / a.tiles;
    const int tile = blockIdx.x % a.tiles;
    const long long f0 = (long long)tile * a.tile_len;
    const long long f1 = (f0 + a.tile_len < a.F) ? f0 + a.tile_len : a.F;
    const R *xc = a.x + c * a.chan_stride;

    // PAIRS: the previous phases of pair slot m of thread t (bins pe = t + m T -- thread 0, slot 0: M/2 --
    // and M - pe) are the two halves of the 16-byte element t + m T; element 4 T holds bins 0 and M.
    constexpr bool PAIRS = VOC_ANA_PHP_PAIRS != 0;
    C2 *php2 = reinterpret_cast<C2 *>(php);
    if (f0 == 0) {
        const R *pi = a.phase_in + c * p.K;
        if constexpr (PAIRS) {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int pe = (m == 0 && tid == 0) ? M / 2 : tid + m * T;
                php2[tid + m * T] = mk2(pi[pe], pi[M - pe]);
            }
            if (tid == 0) php2[4 * T] = mk2(pi[0], pi[M]);
        } else {
            for (int k = tid; k < p.K; k += T) php[k] = pi[k];
        }
    }
    // (ownership of php differs between this fill and the loop below; the barriers inside the
    //  FFT separate them)

    FastTw<LOG2M, C2> tw;  // frame-invariant twiddle bases of this thread
    fast_tw_init<LOG2M>(p, tid, tw);

    // Slot u holds the sample pair at frame positions 2 tid + 2 u T (+1).  The first FFT pass reads
    // the frame rewound (v[n] = padded[(n + N/2) mod N]) and even/odd packed: its element
    // tid + j T is slot (j + 4) mod 8.
    // Pair loads are a uniform base (moved by whole multiples of T pairs: scalar arithmetic), one 32-bit
    // lane offset and no more.  The lane offset comes from an opaque copy of the thread id taken where
    // the loads are issued: derived from the thread id itself, the loop-invariant addresses are hoisted
    // as one 64-bit pointer per load (16 registers for the window alone, and spills).
    C2 xs[8], xn[NEW], ws[8];
    auto sample_pair = [&](const R *fr, int u, int lt) -> C2 {
        if (VEC2) return at_bytes(reinterpret_cast<const C2 *>(fr) + u * T, (unsigned)lt * (unsigned)sizeof(C2), 0);
        const unsigned i0 = (unsigned)(2 * (lt + u * T));
        const int j0 = (int)i0 - p.hd, j1 = j0 + 1;
        return mk2((j0 >= 0 && j0 < p.W) ? fr[j0] : (R)0, (j1 >= 0 && j1 < p.W) ? fr[j1] : (R)0);
    };
    auto load_window = [&](int lt) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const unsigned i0 = (unsigned)(2 * (lt + u * T));
            if (VEC2) {
                ws[u] = at_bytes(reinterpret_cast<const C2 *>(p.window) + u * T, (unsigned)lt * (unsigned)sizeof(C2), 0);
            } else {
                const int j0 = (int)i0 - p.hd, j1 = j0 + 1;
                ws[u] = mk2((j0 >= 0 && j0 < p.W) ? p.window[j0] : (R)0, (j1 >= 0 && j1 < p.W) ? p.window[j1] : (R)0);
            }
        }
    };

    const R half_vol = (R)0.5 * a.vol;  // exact; undoes the doubling of split_pair_x2
    const long long fs = f0 > 0 ? f0 - 1 : f0;  // one frame early to recover its raw phase
    {
        const R *fr = xc + fs * a.frame_stride;
#pragma unroll
        for (int u = 0; u < 8; ++u) xs[u] = sample_pair(fr, u, tid);
        load_window(tid);
    }
    VOC_STAMP_DECL;
    for (long long f = fs; f < f1; ++f) {
        const bool emit = f >= f0;
        VOC_STAMP_FRAME;
        VOC_STAMP(0);
        fast_tw_touch<LOG2M>(tw);
        if (!(VEC2 && VOC_WS_PREFETCH) && f > fs) load_window(opaque_int(tid));  // predicated loads: not worth the registers of a prefetch
        wave_prio<VOC_PRIO_ANA_T, LOG2M>();
        C2 v[8];
        if constexpr (VOC_FMA_MORE != 0) {
            // first stage of the radix-8 butterfly on the windowed samples: the product of the upper input is
            // fused into the sum and the difference
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int u = (j + 4) & 7, u4 = j;  // inputs j and j + 4 are slots (j + 4) mod 8 and j
                const C2 pj = mk2(ws[u].x * xs[u].x, ws[u].y * xs[u].y);
                v[j] = mk2(fma(ws[u4].x, xs[u4].x, pj.x), fma(ws[u4].y, xs[u4].y, pj.y));
                v[j + 4] = mk2(fma(-ws[u4].x, xs[u4].x, pj.x), fma(-ws[u4].y, xs[u4].y, pj.y));
            }
            dft8_rest(v);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int u = (j + 4) & 7;
                v[j] = mk2(ws[u].x * xs[u].x, ws[u].y * xs[u].y);
            }
            dft<8>(v);
        }
        // The last pass leaves element tid + n T of the spectrum with thread tid, and the thread's four bin
        // pairs are (tid + m T, M - tid - m T): the first of each pair is already here (n = 0..3), the second
        // is the n = 7 - m of thread T - tid.  So only the upper four results go through LDS (and the first,
        // because thread 0's first pair is special).
        constexpr bool KEEP = VOC_LAST_IN_REGS != 0;
        fft_after_first<LOG2M, KEEP>(lds, tw, tid, v, [&]() {
            if (f + 1 < f1) {  // the pairs of the next frame that are not already in registers, and the window again
                const R *fr = xc + (f + 1) * a.frame_stride;
                const int lp = opaque_int(tid);
#pragma unroll
                for (int i = 0; i < NEW; ++i) xn[i] = sample_pair(fr, 8 - NEW + i, lp);
                if (VEC2 && VOC_WS_PREFETCH) load_window(lp);
            }
        });

        VOC_STAMP(1);
        wave_prio<VOC_PRIO_ANA_B, LOG2M>();
        const long long row = c * a.rpc + f;
        R *mrow = a.mag + row * a.ldk, *irow = a.inc + row * a.ldk;
        // Thread t owns the bin pairs (t + mT, M - t - mT), m = 0..3; thread 0's first pair is the
        // self-paired bin M/2.  The loop is unrolled so that m is a constant in every copy: the LDS
        // addresses of a pair are then one per-thread base plus an immediate (lds_sw is additive in
        // multiples of 64, and T is one, so both bases move by whole multiples of T), the split
        // twiddle's rotation is a literal, and the self-pair case exists in the first copy only.  A
        // scheduling fence after every copy keeps their live ranges apart (the register budget is
        // that of one pair).
        const int lt = opaque_int(tid);
        const bool self0 = lt == 0;
        const unsigned pe0 = self0 ? M / 2 : (unsigned)lt;      // first pair: bins pe0 and M - pe0
        PairBinsT<R> bins;
        if constexpr (SLIDE > 0) bins.frame(p, lt);
        const C2 *zlo = lds + lds_sw(lt), *zhi = lds + lds_sw(M - lt);
        if constexpr (KEEP) {
            __syncthreads();  // the last pass' input has been read by everyone
            VOC_STAMP(2);
            C2 *zw = lds + lds_sw(lt);
#pragma unroll
            for (int n = 4; n < 8; ++n) zw[n * T] = v[nat_idx<LOG2M>(n)];
            zw[0] = v[nat_idx<LOG2M>(0)];  // thread 0's first pair is the self-paired bin M/2 (its own element 4 T), and Z[0] is read by two other threads
            VOC_STAMP(3);
            __syncthreads();
            VOC_STAMP(4);
        }
        // byte offsets of the pair's bins in an STFT row and in the phase array: slot 0 has its own, slots
        // 1..3 sit at +-(m - 2) T elements around one offset each (immediates of the memory instructions)
        constexpr int SZ = (int)sizeof(R);
        const unsigned o_pe0 = pe0 * SZ, o_q0 = (unsigned)(M * SZ) - o_pe0;
        const unsigned o_pe2 = (unsigned)(lt + 2 * T) * SZ, o_q2 = (unsigned)(M * SZ) - o_pe2;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const unsigned pe = m == 0 ? pe0 : (unsigned)(lt + m * T);
            const unsigned q = M - pe;
            const unsigned ope = m == 0 ? o_pe0 : o_pe2, oq = m == 0 ? o_q0 : o_q2;
            const int ipe = m == 0 ? 0 : (m - 2) * T * SZ, iq = -ipe;
            C2 P, Q, w;
            if (m == 0) {
                P = lds[lds_sw(pe)];
                Q = lds[lds_sw(q)];
                w = split_tw(tw.s, 0, self0);
            } else {
                if constexpr (KEEP) P = v[nat_idx<LOG2M>(m)];
                else P = zlo[m * T];
                Q = zhi[-m * T];
                w = split_tw(tw.s, m, false);
            }
            C2 Xa, Xb;
            split_pair_x2(P, Q, w, Xa, Xb);
            R pa, pb;
            if constexpr (PAIRS) {
                const C2 pp = at_bytes(php2, (unsigned)lt * (unsigned)sizeof(C2), m * T * (int)sizeof(C2));
                pa = pp.x;
                pb = pp.y;
            } else {
                pa = at_bytes(php, ope, ipe);
                pb = at_bytes(php, oq, iq);
            }
            R ma, mb, pha, phb;
            polar_pair<VOC_POLAR_LEAN != 0>(Xa, Xb, half_vol, ma, pha, mb, phb);
            if (emit) {
                at_bytes(mrow, ope, ipe) = ma;
                at_bytes(mrow, oq, iq) = mb;
                if constexpr (SLIDE > 0) {  // H = SLIDE * N/8: a power of two, like N
                    R ia, ib;
                    inc_pow2_pair(p, bins, m == 0 ? (self0 ? (R)(M / 2) : bins_ft(bins)) : bins_ft(bins) + (R)(m * T), M, (int)pe, (int)q, pha - pa, phb - pb, ia, ib);
                    at_bytes(irow, ope, ipe) = ia;
                    at_bytes(irow, oq, iq) = ib;
                } else {
                    at_bytes(irow, ope, ipe) = phase_inc(p, pe, pha - pa);
                    at_bytes(irow, oq, iq) = phase_inc(p, q, phb - pb);
                }
            }
            if constexpr (PAIRS) {
                at_bytes(php2, (unsigned)lt * (unsigned)sizeof(C2), m * T * (int)sizeof(C2)) = mk2(pha, phb);
            } else {
                at_bytes(php, ope, ipe) = pha;
                at_bytes(php, oq, iq) = phb;
            }
            sched_fence();
        }
        // DC and Nyquist bins are real: phase 0 or pi, magnitude |x|.  Two different waves take one
        // each, so that no wave (and no SIMD) carries both on top of its four bin pairs.
        if (tid == 0 || tid == T / 2) {
            const C2 z0 = lds[0];
            const bool dc = tid == 0;
            const R xr = dc ? z0.x + z0.y : z0.x - z0.y;
            const int kb = dc ? 0 : M;
            const R phr = xr < (R)0 ? (R)kPi : (R)0;
            R *pk = PAIRS ? php + 8 * T + (dc ? 0 : 1) : php + kb;
            if (emit) {
                mrow[kb] = fabs(xr) * a.vol;
                irow[kb] = SLIDE > 0 ? phase_inc_pow2(p, kb, phr - *pk) : phase_inc(p, kb, phr - *pk);
            }
            *pk = phr;
        }
        VOC_STAMP(5);
        // slide the sample window by one hop
#pragma unroll
        for (int u = 0; u < 8 - NEW; ++u) xs[u] = xs[u + NEW];
#pragma unroll
        for (int i = 0; i < NEW; ++i) xs[8 - NEW + i] = xn[i];
        // the next frame's first LDS write sits behind the first barrier of fft_after_first
    }

    VOC_STAMP_FLUSH(0);
    if (f1 == a.F && f1 > f0) {
        __syncthreads();
        R *po = a.phase_out + c * p.K;
        if constexpr (PAIRS) {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int pe = (m == 0 && tid == 0) ? M / 2 : tid + m * T;
                const C2 pp = php2[tid + m * T];
                po[pe] = pp.x;
                po[M - pe] = pp.y;  // (thread 0, m = 0: the same bin and the same value twice)
            }
            if (tid == 0) {
                po[0] = php2[4 * T].x;
                po[M] = php2[4 * T].y;
            }
        } else {
            for (int k = tid; k < p.K; k += T) po[k] = php[k];
        }
    }
}

// Whether the synthesis kernel is built for two waves per SIMD (see VOC_SYN_FAT): the host's count of resident
// workgroups (synthesis_slots) follows the same rule.
template <int LOG2M, bool OLA, class R>
constexpr bool syn_fat() { return VOC_SYN_FAT != 0 && LOG2M >= 10 && OLA && sizeof(R) == 8; }

// Whether that kernel's transform alternates between two LDS buffers (fft_after_first, ldsB): one barrier per
// exchange instead of two and none at the end of the frame, for M more elements of LDS per workgroup -- which
// the two-wave build has room for from N = 4096 up (N = 4096: 2 x (2 x 32 KB + 8 KB) per CU).  Measured
// (MI355X, metric shape, same box): four barriers per frame instead of eight leave the kernel where it was,
// 27.3 against 27.1 ms -- it does not wait at barriers -- so this is off.
// The two-entry rows of the FILT = 2 build of the synthesis kernel: requested a frame ahead like the plain row (64
// registers across the transform: 21 spilled, 87 ms on the fourth configuration's shape with a factor of 1.5) or at
// the top of their frame (245 registers, none spilled, 67 ms; the chain as a pass of its own: 94-97 ms).
#ifndef VOC_FILT2_ROW_AHEAD
#define VOC_FILT2_ROW_AHEAD 0
#endif
#ifndef VOC_SYN_PINGPONG
#define VOC_SYN_PINGPONG 0
#endif
template <int LOG2M, bool OLA, class R>
constexpr bool syn_pingpong()
{
    return VOC_SYN_PINGPONG != 0 && syn_fat<LOG2M, OLA, R>() && LOG2M >= 11 && VOC_LAST_IN_REGS != 0 && VOC_SYN_PHASE_REGS != 0 &&
           VOC_SYN_SINCOS_TAB == 0 && FastCfg<LOG2M>::R4 > 1;
}
// LDS of the synthesis kernel: the transform's buffer and the K phases (or, with the phases in registers, tables
// in their place); with two transform buffers, 512 elements of tables between them.
template <int LOG2M, bool OLA, class R>
constexpr int syn_lds_bytes()
{
    return syn_pingpong<LOG2M, OLA, R>() ? (2 * FastCfg<LOG2M>::M + 512) * 2 * (int)sizeof(R) : FastCfg<LOG2M>::template lds_bytes<R>();
}

// OLA = false: frames out ([C][F][W]).
// OLA = true: overlap-add in registers.  Requires W == N and H = NHP * 2T: thread t owns the sample
// PAIRS (2(t + jT), 2(t + jT) + 1), j = 0..7, of the N-sample window -- one complex LDS element,
// one 16-byte window load and one 16-byte store per pair -- and the partial sums shift by NHP
// pairs per frame.  y rows must be 16-byte aligned.
// MAPPED: rows are looked up through a.mag_rows / a.inc_rows (a build of its own: the lookups cost
// the unmapped path 1 % when they are decided at run time).
// ALIGNED = false: the output rows are not 16-byte aligned (an odd row stride or base): the emitted sample pairs
// are stored as two 8-byte values each.  Same arithmetic: which build runs does not change a bit of the result.
// FILT (the two-wave double-precision build only): the row passes through pitchShiftInterpolate and / or one table of
// gains as it is read (SynArgsT::pitch_n, gain).  FILT = 2: each bin takes two neighbouring entries of the
// unfiltered row, a frame ahead like the plain row, and mixes them where the plain build uses the entry itself --
// the filter kernel's operations in its order, so the bits are those of a filter pass followed by the plain
// build.  FILT = 1: a whole-number n (or no pitch shift): the weight of the second entry is zero, so only entry n k
// is read (for finite rows the value of the filter pass, a zero's sign aside: the pass adds 0 times the second
// entry).
template <int LOG2M, bool OLA, int NHP, class R = double, bool MAPPED = false, bool ALIGNED = true, int FILT = 0>
__global__ void __launch_bounds__(FastCfg<LOG2M>::T, (syn_fat<LOG2M, OLA, R>() ? 2 : VOC_FAST_WAVES)) synthesis_fast_kernel(ParamsT<R> p, SynArgsT<R> a)
{
    static_assert(FILT == 0 || (syn_fat<LOG2M, OLA, R>() && !MAPPED && VOC_SYN_PHASE_REGS != 0), "filtered rows: the two-wave overlap-add build");
    using C2 = cplx<R>;
    constexpr int M = FastCfg<LOG2M>::M, T = FastCfg<LOG2M>::T, N = 2 * M;
    constexpr int NP = 8;                       // sample pairs per thread
    constexpr int NK = OLA ? NP - NHP : 0;      // pairs whose partial sums outlive the frame
    constexpr int NT = NK > 0 ? NK : 1;         // (array length: never zero)
    VOC_SMEM(C2, lds);
    R *php = reinterpret_cast<R *>(lds + FastCfg<LOG2M>::LDS_CPLX);  // accumulated phase, K bins
    const R *ldsd = reinterpret_cast<const R *>(lds);
    const int tid = threadIdx.x;
    const bool whole = (long long)blockIdx.x < a.unsplit;  // leading channels: one tile over all frames
    const long long rest = (long long)blockIdx.x - a.unsplit;
    const long long c = whole ? (long long)blockIdx.x : a.unsplit + rest / a.tiles;
    const int tile = whole ? 0 : (int)(rest % a.tiles);
    const long long f0 = (long long)tile * a.tile_len;
    const long long f1 = (whole || f0 + a.tile_len >= a.F) ? a.F : f0 + a.tile_len;
    const long long halo = OLA ? ((long long)a.halo < f0 ? (long long)a.halo : f0) : 0;
    const long long fs = f0 - halo;

    // The next frame's row can be requested during this frame's transform, at the price of its 36
    // registers staying live across the transform.  Measured (MI355X, fp64): without it the kernel
    // needs 133 instead of 166 registers and runs 5-11 % faster at N >= 2048; the one-wave workgroups
    // of N = 1024 are 4 % faster with it.
    constexpr bool SFAT = syn_fat<LOG2M, OLA, R>();
    constexpr bool PP = syn_pingpong<LOG2M, OLA, R>();
    C2 *ldsB = PP ? lds + FastCfg<LOG2M>::LDS_CPLX + 512 : nullptr;
    constexpr bool ROW_PF = SFAT || VOC_SYN_ROW_PREFETCH == 1 || (VOC_SYN_ROW_PREFETCH == 2 && LOG2M <= 9);
    constexpr bool LEAN = !ROW_PF || SFAT;  // the branch-free sine / cosine pays only with the registers the prefetch took
    // ... and so does keeping the accumulated phases of this thread's eight bins (and of DC / Nyquist
    // in thread 0) in registers instead of LDS: 4 % faster at N = 4096, 29 % slower at N = 1024.
    constexpr bool PHASE_REGS = VOC_SYN_PHASE_REGS && (!ROW_PF || SFAT);
    R phr[8] = {}, ph0 = (R)0, phm = (R)0;
    if constexpr (PHASE_REGS) {
        const R *pi = (tile == 0) ? a.phase_in + c * p.K : a.carry + ((c - a.unsplit) * a.tiles + tile) * p.K;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const unsigned pr = tid + m * T;
            const unsigned pe = pr == 0 ? M / 2 : pr;
            phr[2 * m] = pi[pe];
            phr[2 * m + 1] = pi[M - pe];
        }
        if (tid == 0) {
            ph0 = pi[0];
            phm = pi[M];
        }
    } else {
        const R *pi = (tile == 0) ? a.phase_in + c * p.K : a.carry + ((c - a.unsplit) * a.tiles + tile) * p.K;
        for (int k = tid; k < p.K; k += T) php[k] = pi[k];
        __syncthreads();
    }

    FastTw<LOG2M, C2> tw;  // frame-invariant twiddle bases of this thread
    fast_tw_init<LOG2M>(p, tid, tw);
    // with the phases in registers their LDS area is free: it can hold the twiddle rows of two passes
    constexpr bool TW_LDS = VOC_SYN_TW_LDS && PHASE_REGS && LOG2M >= 10;
    C2 *twl = reinterpret_cast<C2 *>(php);
    if constexpr (TW_LDS) {
        for (int i = tid; i < 56 + 448; i += T) twl[i] = i < 56 ? p.twP[7 + i] : p.twP[63 + (i - 56)];
        __syncthreads();
    }
    // ... or the table of the circle, sixteen copies side by side (fast_sincos_tab): 64 x 16 x 16 bytes
    constexpr bool SC_TAB = VOC_SYN_SINCOS_TAB != 0 && SFAT && !TW_LDS && FastCfg<LOG2M>::K * 8 >= 16384;
    if constexpr (SC_TAB) {
        for (int i = tid; i < 1024; i += T) twl[i] = p.cs64[i >> 4];
        __syncthreads();
    }
    C2 tail[NT];  // overlap-add partial sums that later frames still add to
#pragma unroll
    for (int j = 0; j < NT; ++j) tail[j] = mk2((R)0, (R)0);

    // STFT row of frame f: (mag, ph_inc) of this thread's eight bins, plus DC / Nyquist for thread 0
    R mg[8], ic[8], mg0 = (R)0, mgm = (R)0, ic0 = (R)0, icm = (R)0;
    // FILT: the second entry of every bin (the first is in the arrays above), and the gains of this frame's bins
    constexpr int NF = FILT == 2 ? 8 : 1, NG = FILT != 0 ? 8 : 1;
    R mg1[NF], ic1[NF], mg01 = (R)0, mgm1 = (R)0, ic01 = (R)0, icm1 = (R)0, gn[NG], gn0 = (R)1, gnm = (R)1;
    const bool has_pitch = FILT != 0 && a.pitch_n != 0.0;
    const int n_whole = FILT == 1 ? (int)a.pitch_n : 0;  // FILT = 1: the pitch factor as an integer (0: none)
    // entry read for bin k with a whole-number factor, and whether the bin is in range (n k < K - 1)
    auto whole_tap = [&](int k, bool &valid) -> int {
        const int i = n_whole * k;
        valid = n_whole == 0 || i < M;
        return n_whole == 0 ? k : (valid ? i : 0);
    };
    // bin of row slot b (slots 2m, 2m + 1 are the pair m) of the thread whose opaque id is lt
    auto slot_bin = [&](int b, int lt) -> int {
        const int m = b >> 1;
        const int pe = (m == 0 && lt == 0) ? M / 2 : lt + m * T;
        return (b & 1) ? M - pe : pe;
    };
    // Bins of pair slot m: pe = tid + m T (thread 0, slot 0: the self-paired bin M/2) and q = M - pe.  Their
    // byte offsets in a row are four per-thread values plus immediates: slot 0 has its own pair of
    // offsets, slots 1..3 sit at +-(m - 2) T elements around one offset each.
    constexpr int SZ = (int)sizeof(R);
    // (formed from the frame's opaque copy of the thread id: hoisted, they become eight 64-bit row
    // pointers per thread that live across the whole loop)
    auto load_row = [&](long long f, int lt) {
        if constexpr (FILT == 1) {
            const R *mrow = a.mag + (c * a.F + f) * a.ldk, *irow = a.inc + (c * a.F + f) * a.ldk;
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                bool valid;
                const unsigned o0 = (unsigned)whole_tap(slot_bin(b, lt), valid) * SZ;
                mg[b] = at_bytes(mrow, o0, 0);
                ic[b] = at_bytes(irow, o0, 0);
            }
            if (tid == 0) {
                bool valid;
                const int im = whole_tap(M, valid);
                mg0 = mrow[0];
                ic0 = irow[0];
                mgm = mrow[im];
                icm = irow[im];
            }
            return;
        }
        if constexpr (FILT == 2) {
            const R *mrow = a.mag + (c * a.F + f) * a.ldk, *irow = a.inc + (c * a.F + f) * a.ldk;
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const int k = slot_bin(b, lt);
                const int i = has_pitch ? pitch_tap((double)a.pitch_n, k, M + 1).i : k;
                const unsigned o0 = (unsigned)i * SZ, o1 = (unsigned)(i + (has_pitch ? 1 : 0)) * SZ;  // (no second entry without a pitch shift)
                mg[b] = at_bytes(mrow, o0, 0);
                mg1[b] = at_bytes(mrow, o1, 0);
                ic[b] = at_bytes(irow, o0, 0);
                ic1[b] = at_bytes(irow, o1, 0);
            }
            if (tid == 0) {
                const int im = has_pitch ? pitch_tap((double)a.pitch_n, M, M + 1).i : M, im1 = im + (has_pitch ? 1 : 0);
                mg0 = mrow[0];
                mg01 = mrow[1];
                ic0 = irow[0];
                ic01 = irow[1];
                mgm = mrow[im];
                mgm1 = mrow[im1];
                icm = irow[im];
                icm1 = irow[im1];
            }
            return;
        }
        const unsigned row_pe0 = (unsigned)(lt == 0 ? M / 2 : lt) * SZ, row_q0 = (unsigned)(M * SZ) - row_pe0;
        const unsigned row_pe2 = (unsigned)(lt + 2 * T) * SZ, row_q2 = (unsigned)(M * SZ) - row_pe2;
        const R *mrow = MAPPED ? mag_row_of(a, c, f) : a.mag + (c * a.F + f) * a.ldk;
        const R *irow = MAPPED ? inc_row_of(a, c, f) : a.inc + (c * a.F + f) * a.ldk;
        mg[0] = at_bytes(mrow, row_pe0, 0);
        ic[0] = at_bytes(irow, row_pe0, 0);
        mg[1] = at_bytes(mrow, row_q0, 0);
        ic[1] = at_bytes(irow, row_q0, 0);
#pragma unroll
        for (int m = 1; m < 4; ++m) {
            mg[2 * m] = at_bytes(mrow, row_pe2, (m - 2) * T * SZ);
            ic[2 * m] = at_bytes(irow, row_pe2, (m - 2) * T * SZ);
            mg[2 * m + 1] = at_bytes(mrow, row_q2, (2 - m) * T * SZ);
            ic[2 * m + 1] = at_bytes(irow, row_q2, (2 - m) * T * SZ);
        }
        if (tid == 0) {
            mg0 = mrow[0];
            ic0 = irow[0];
            mgm = mrow[M];
            icm = irow[M];
        }
    };

    constexpr bool ROW_AHEAD = VOC_FILT2_ROW_AHEAD != 0 || FILT != 2;  // (the two-entry rows of FILT = 2: a frame ahead or at use)
    if (ROW_PF && ROW_AHEAD) load_row(fs, opaque_int(tid));
    C2 wres[NP];
    if constexpr (SFAT) {
#pragma unroll
        for (int j = 0; j < NP; ++j) wres[j] = at_bytes(reinterpret_cast<const C2 *>(VOC_SYN_WIN_SCALED ? p.window_n : p.window) + j * T, (unsigned)tid * (unsigned)sizeof(C2), 0);
    }
    VOC_STAMP_DECL;
    for (long long f = fs; f < f1; ++f) {
        VOC_STAMP_FRAME;
        VOC_STAMP(0);
        const long long row = c * a.F + f;
        const int lt = opaque_int(tid);  // keeps this frame's index arithmetic inside the loop
        fast_tw_touch<LOG2M>(tw);
        wave_prio<VOC_PRIO_SYN_B, LOG2M>();
        if (!ROW_PF || !ROW_AHEAD) load_row(f, lt);
        if constexpr (FILT != 0) {
            if (a.gain) {  // (requested here, used behind the first sine / cosine: live in the bin loop only)
#pragma unroll
                for (int b = 0; b < 8; ++b) gn[b] = a.gain[slot_bin(b, lt)];
                if (tid == 0) {
                    gn0 = a.gain[0];
                    gnm = a.gain[M];
                }
            }
        }
        // the row as the filter pass would have left it: entries mixed (interpolate n), increments over n, gains
        auto filtered = [&](int k, R v0, R v1, R w0, R w1, R g, R &mo, R &io) {
            if constexpr (FILT == 1) {
                bool valid;
                (void)whole_tap(k, valid);
                mo = v0;
                io = w0;
                if (n_whole != 0) {
                    mo = valid ? v0 + (R)0 : (R)0;
                    const R iv = valid ? w0 + (R)0 : (R)0;
                    io = a.pitch_rcp != 0.0 ? iv * (R)a.pitch_rcp : iv / (R)a.pitch_n;
                }
            } else if (has_pitch) {
                const PitchTap tp = pitch_tap((double)a.pitch_n, k, M + 1);
                mo = pitch_mix(tp, v0, v1);
                const R iv = pitch_mix(tp, w0, w1);
                io = a.pitch_rcp != 0.0 ? iv * (R)a.pitch_rcp : iv / (R)a.pitch_n;
            } else {
                mo = v0;
                io = w0;
            }
            if (a.gain) mo = mo * g;
        };
        // m is a constant in every copy of the unrolled loop: the self-paired bin exists in the first
        // copy only, and the LDS addresses of the other pairs are one base each plus an immediate
        // (lds_sw is additive in multiples of 64, and T is one)
        const bool self0 = lt == 0;
        C2 *zlo = lds + lds_sw(lt), *zhi = lds + lds_sw(M - lt);
        // The first pass reads the elements tid + j T.  Those with j < 4 are this thread's own Z[pe]; the
        // others are the Z[M - pe] of thread T - tid (thread 0: its own).  Only the latter go through LDS.
        constexpr bool KEEP_IN = VOC_LAST_IN_REGS != 0;
        constexpr bool KEEP_OUT = VOC_LAST_IN_REGS != 0 && OLA;  // ... and the overlap-add wants tid + n T, which the last pass leaves here
        C2 za[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const bool self = m == 0 && self0;
            const unsigned pe = self ? M / 2 : (unsigned)(lt + m * T);
            const unsigned q = M - pe;
            C2 A, B;
            constexpr bool POLAR_FUSED = VOC_FMA_MORE != 0 && VOC_FMA_BF != 0 && PHASE_REGS;
            R sa = (R)0, ca = (R)0;
            if constexpr (POLAR_FUSED) {
                phr[2 * m] = phase_step(phr[2 * m], ic[2 * m] * p.hop_d);
                fast_sincos<LEAN>(phr[2 * m], &sa, &ca);
                A = mk2((R)0, (R)0);
                B = synthesize_bin<R, LEAN>(p, mg[2 * m + 1], ic[2 * m + 1], phr[2 * m + 1]);
            } else if constexpr (SC_TAB) {
                const C2 *col = twl + (lt & 15);
                A = synthesize_bin_tab(p, mg[2 * m], ic[2 * m], phr[2 * m], col);
                B = synthesize_bin_tab(p, mg[2 * m + 1], ic[2 * m + 1], phr[2 * m + 1], col);
            } else if constexpr (FILT != 0) {
                R ma, ia, mb, ib;
                constexpr int s1 = FILT == 2 ? 1 : 0;  // (no second entries in the one-entry build)
                filtered((int)pe, mg[2 * m], mg1[s1 * 2 * m], ic[2 * m], ic1[s1 * 2 * m], gn[2 * m], ma, ia);
                filtered((int)q, mg[2 * m + 1], mg1[s1 * (2 * m + 1)], ic[2 * m + 1], ic1[s1 * (2 * m + 1)], gn[2 * m + 1], mb, ib);
                synth_pair<LEAN>(p, ma, ia, phr[2 * m], mb, ib, phr[2 * m + 1], A, B);
            } else if constexpr (PHASE_REGS) {
                synth_pair<LEAN>(p, mg[2 * m], ic[2 * m], phr[2 * m], mg[2 * m + 1], ic[2 * m + 1], phr[2 * m + 1], A, B);
            } else {
                R pa = php[pe], pb = php[q];
                synth_pair<LEAN>(p, mg[2 * m], ic[2 * m], pa, mg[2 * m + 1], ic[2 * m + 1], pb, A, B);
                php[pe] = pa;
                php[q] = pb;
            }
            C2 Za, Zb;
            if constexpr (POLAR_FUSED) merge_pair_polar(mg[2 * m], ca, sa, B, split_tw(tw.s, m, self), Za, Zb);
            else merge_pair(A, B, split_tw(tw.s, m, self), Za, Zb);
            za[m] = Za;
            if (m == 0) {
                if constexpr (!KEEP_IN) lds[lds_sw(pe)] = Za;
                lds[lds_sw(q)] = Zb;
            } else {
                if constexpr (!KEEP_IN) zlo[m * T] = Za;
                zhi[-m * T] = Zb;
            }
#if VOC_SYN_FENCE
            sched_fence();  // one bin pair at a time: interleaving all eight sincos chains overflows the register budget
#endif
        }
        if (tid == 0) {  // DC and Nyquist: only the real parts enter the C2R transform
            if constexpr (!PHASE_REGS) {
                ph0 = php[0];
                phm = php[M];
            }
            if constexpr (FILT != 0) {
                R m0, i0, mm, im;
                filtered(0, mg0, mg01, ic0, ic01, gn0, m0, i0);
                filtered(M, mgm, mgm1, icm, icm1, gnm, mm, im);
                mg0 = m0;
                ic0 = i0;
                mgm = mm;
                icm = im;
            }
            const C2 A = synthesize_bin(p, mg0, ic0, ph0);
            const C2 B = synthesize_bin(p, mgm, icm, phm);
            if constexpr (!PHASE_REGS) {
                php[0] = ph0;
                php[M] = phm;
            }
            const C2 z0 = mk2(A.x - B.x, A.x + B.x);
            if constexpr (KEEP_IN) za[0] = z0;  // element 0; thread 0's slot 0 was bin M/2, which it reads back as element 4 T
            else lds[0] = z0;
        }
        if (SFAT && ROW_AHEAD && f + 1 < f1) load_row(f + 1, lt);
        wave_prio<VOC_PRIO_SYN_T, LOG2M>();
        VOC_STAMP(1);
        __syncthreads();
        VOC_STAMP(2);
        C2 v[8];
        if constexpr (KEEP_IN) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[j] = za[j];
                v[j + 4] = zlo[(j + 4) * T];
            }
        } else {
            load8<LOG2M>(lds, lt, v);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) bf2(v[j], v[j + 4]);
        dft8_rest(v);
        // window values of the overlap-add: uniform bases T pairs apart plus one lane offset.  Requested
        // before the last pass (they come from L2: asked for only where they are used, their latency is
        // exposed in the part of the frame that runs at the highest priority).
        constexpr int WIN_EARLY = (OLA && KEEP_OUT) ? VOC_SYN_WIN_EARLY : 0;
        C2 wv[NP];
        const unsigned lane16 = (unsigned)lt * (unsigned)sizeof(C2);
        auto load_win = [&](int j0, int j1) {
#pragma unroll
            for (int j = j0; j < j1; ++j) wv[j] = at_bytes(reinterpret_cast<const C2 *>(VOC_SYN_WIN_SCALED ? p.window_n : p.window) + j * T, lane16, 0);
        };
        fft_after_first<LOG2M, KEEP_OUT, PP>(lds, tw, lt, v, [&]() {
            if (ROW_PF && !SFAT && f + 1 < f1) load_row(f + 1, lt);
        }, TW_LDS ? twl : nullptr, [&]() {
            if constexpr (!SFAT) load_win(NP - WIN_EARLY, NP);
        }, ldsB);
        if constexpr (SFAT) {
#pragma unroll
            for (int j = 0; j < NP; ++j) wv[j] = wres[j];
        }
        // lds[pad(m)] = swapped z[m]: v[2m] = z.y / N, v[2m+1] = z.x / N.  frame[i] = w[i] * v[(i + off + N/2) mod N]
        VOC_STAMP(3);
        const bool emit = f >= f0;
        wave_prio<VOC_PRIO_SYN_O, LOG2M>();
        if (OLA) {
            // W == N (off = 0): window samples 2i, 2i+1 pair with complex element (i + M/2) mod M
            // element lt + e with e a multiple of T (of 64) sits at lds_sw(lt) + e: one LDS base plus
            // immediates; window and output rows: uniform bases T pairs apart plus one lane offset
            C2 zv[NP];
            const C2 *zb = lds + lds_sw(lt);
            if constexpr (!SFAT) load_win(0, NP - WIN_EARLY);
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const int e = (j * T + M / 2) & (M - 1);
                if constexpr (KEEP_OUT) zv[j] = v[nat_idx<LOG2M>((j + 4) & 7)];
                else zv[j] = zb[e];
            }
            C2 cur[NP];
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const C2 old = j < NK ? tail[j] : mk2((R)0, (R)0);
                // w (z / N) and (w / N) z round alike: 1/N is a power of two (a window value below 2^-1010 aside)
                if constexpr (VOC_FMA_MORE != 0 && VOC_SYN_WIN_SCALED != 0) {
                    cur[j].x = fma(wv[j].x, zv[j].y, old.x);
                    cur[j].y = fma(wv[j].y, zv[j].x, old.y);
                } else {
                    cur[j].x = old.x + (VOC_SYN_WIN_SCALED ? wv[j].x * zv[j].y : wv[j].x * (zv[j].y * p.inv_n));
                    cur[j].y = old.y + (VOC_SYN_WIN_SCALED ? wv[j].y * zv[j].x : wv[j].y * (zv[j].x * p.inv_n));
                }
            }
            // (without this the compiler moves the loads and sums of the emitted pairs into the branch
            //  below, one after the other behind the rest: two more round trips to L2 per frame)
#pragma unroll
            for (int j = 0; j < NHP; ++j) opaque(cur[j]);
            if (emit) {
                if constexpr (ALIGNED) {
                    C2 *yc = reinterpret_cast<C2 *>(a.out + c * a.out_stride + f * p.H);
#pragma unroll
                    for (int j = 0; j < NHP; ++j) at_bytes(yc + j * T, lane16, 0) = cur[j];
                } else {
                    R *ys = a.out + c * a.out_stride + f * p.H + 2 * lt;
#pragma unroll
                    for (int j = 0; j < NHP; ++j) {
                        ys[2 * j * T] = cur[j].x;
                        ys[2 * j * T + 1] = cur[j].y;
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < NK; ++j) tail[j] = cur[j + NHP];
        } else {
            R *fo = a.out + row * p.W;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int i = lt + j * T;
                const int n = (i + p.off + M) & (N - 1);
                const R s = ldsd[2 * lds_sw(n >> 1) + ((n & 1) ^ 1)] * p.inv_n;
                if (i < p.W) fo[(unsigned)i] = p.window[(unsigned)i] * s;
            }
        }
        VOC_STAMP(4);
        if constexpr (PP) sched_fence();
        else __syncthreads();  // FFT buffer is rewritten by the next frame's spectrum
        VOC_STAMP(5);
    }
    VOC_STAMP_FLUSH(1);

    if (f1 == a.F && f1 > f0) {
        R *po = a.phase_out + c * p.K;
        if constexpr (PHASE_REGS) {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const unsigned pr = tid + m * T;
                const unsigned pe = pr == 0 ? M / 2 : pr;
                po[pe] = phr[2 * m];
                po[M - pe] = phr[2 * m + 1];  // (thread 0, m = 0: the same bin and the same value twice)
            }
            if (tid == 0) {
                po[0] = ph0;
                po[M] = phm;
            }
        } else {
            for (int k = tid; k < p.K; k += T) po[k] = php[k];  // php is stable: last barrier passed
        }
        if (OLA) {
            // tail after the last frame: the remaining N - H partial sums, then the zero padding of
            // the last chunk (sumFramesE, Frames.hs:45-47)
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
