"""The filter oracle (oracle/filters_oracle.c) against independent formulations.  The reference
holds no test for any filter (parity unpinned); these checks pin the restatement to the
definitions in Filter.hs by hand-computed cases and by numpy formulas written from those
definitions a second time."""
from fractions import Fraction

import numpy as np
import pytest


def test_interpolate_hand_cases(orc):
    v = np.array([1.0, 2.0, 4.0, 8.0, 16.0])
    assert np.array_equal(orc.interpolate(1.0, v), [1, 2, 4, 8, 0])          # i+1 >= len -> 0 (Filter.hs:121)
    assert np.array_equal(orc.interpolate(0.5, v), [1, 1.5, 2, 3, 4])
    assert np.array_equal(orc.interpolate(2.0, v), [1, 4, 0, 0, 0])
    assert np.array_equal(orc.interpolate(1.5, v), [1, 3, 8, 0, 0])


def test_interpolate_matches_numpy(orc):
    rng = np.random.default_rng(0)
    v = rng.standard_normal(257)
    for n in (0.37, 0.9, 1.0, 1.1, 2.0, 3.3):
        xp = n * np.arange(v.size)
        i = np.floor(xp).astype(int)
        k = xp - i
        ok = i + 1 < v.size
        ref = np.zeros_like(v)
        ref[ok] = (1 - k[ok]) * v[i[ok]] + k[ok] * v[i[ok] + 1]
        assert np.array_equal(orc.interpolate(n, v), ref)


def test_convolution_definition(orc):
    rng = np.random.default_rng(1)
    mag = rng.uniform(0, 1, 33)
    for ker in (np.array([1.0]), np.array([1.0, 2.0, 3.0]), rng.uniform(0, 1, 8), rng.uniform(0, 1, 9)):
        h = ker.size // 2
        g = np.concatenate([np.zeros(h), mag, np.zeros(h)])
        s = 0.0
        for v in ker:
            s = s + v
        ref = np.empty_like(mag)
        for k in range(mag.size):
            acc = 0.0
            for i, v in enumerate(ker):
                acc = acc + v * g[i + k] / s
            ref[k] = acc
        assert np.array_equal(orc.convolution(ker, mag), ref)
    # a normalised box kernel of odd length is a moving average
    avg = orc.convolution(np.ones(3), mag)
    assert np.allclose(avg[1:-1], (mag[:-2] + mag[1:-1] + mag[2:]) / 3, rtol=1e-15)


def test_envelope_kernel_and_envelope(orc):
    assert np.array_equal(orc.envelope_kernel(3), [1, 1, 1])                 # boxWindow for ksize <= 3
    k = orc.envelope_kernel(32)
    x = np.arange(32) / 31.0
    assert np.allclose(k, 0.42 - 0.5 * np.cos(2 * np.pi * x) + 0.08 * np.cos(4 * np.pi * x), atol=1e-15)
    mag = np.full(64, 0.25)
    env = orc.envelope(1, mag)                                              # one-tap kernel: exp(log(m+ee)) - ee
    assert np.allclose(env, mag, rtol=1e-14)
    rng = np.random.default_rng(2)
    mag = rng.uniform(0.1, 1.0, 200)
    ee = 2.0 ** -24
    ref = np.exp(orc.convolution(orc.envelope_kernel(32), np.log(mag + ee))) - ee
    assert np.allclose(orc.envelope(32, mag), ref, rtol=1e-14)


def test_linear_filters(orc):
    k, step = 65, 48000 / 128
    x = step * np.arange(k)
    ones = np.ones((1, k))
    zeros = np.zeros((1, k))
    lp, inc = orc.apply_filter(step, [("lowpass_brickwall", 3000.0)], ones, zeros)
    assert np.array_equal(lp[0], (x <= 3000).astype(float)) and np.array_equal(inc, zeros)
    assert np.array_equal(orc.apply_filter(step, [("highpass_brickwall", 3000.0)], ones, zeros)[0][0], (x >= 3000) * 1.0)
    assert np.array_equal(orc.apply_filter(step, [("bandpass_brickwall", 750.0, 3000.0)], ones, zeros)[0][0],
                          ((x >= 750) & (x <= 3000)) * 1.0)
    assert np.array_equal(orc.apply_filter(step, [("bandstop_brickwall", 750.0, 3000.0)], ones, zeros)[0][0],
                          ((x <= 750) | (x >= 3000)) * 1.0)
    assert np.array_equal(orc.apply_filter(step, [("amplify", 0.3)], 2 * ones, zeros)[0][0], np.full(k, 0.6))
    bw = lambda n, t: 1 / np.sqrt(1 + (x / t) ** (2 * n))
    with np.errstate(divide="ignore"):
        assert np.allclose(orc.apply_filter(step, [("lowpass_butterworth", 4.0, 2000.0)], ones, zeros)[0][0], bw(4, 2000), rtol=1e-14)
        assert np.allclose(orc.apply_filter(step, [("highpass_butterworth", 4.0, 2000.0)], ones, zeros)[0][0], bw(-4, 2000), rtol=1e-14)
        assert np.allclose(orc.apply_filter(step, [("bandpass_butterworth", 2.0, 1000.0, 4000.0)], ones, zeros)[0][0],
                           bw(2, 4000) * bw(-2, 1000), rtol=1e-14)
        assert np.allclose(orc.apply_filter(step, [("bandstop_butterworth", 2.0, 1000.0, 4000.0)], ones, zeros)[0][0],
                           bw(-2, 1000) + bw(2, 4000), rtol=1e-14)
    # highpass at DC: (0/t)**(-2n) = inf -> gain 0
    assert orc.apply_filter(step, [("highpass_butterworth", 4.0, 2000.0)], ones, zeros)[0][0][0] == 0.0


def test_chain_order_and_phase_handling(orc):
    rng = np.random.default_rng(3)
    mag, inc = rng.uniform(0, 1, (3, 33)), rng.uniform(-1, 1, (3, 33))
    m, i = orc.apply_filter(10.0, [("pitch_shift", 2.0), ("lowpass_brickwall", 100.0)], mag, inc)
    for r in range(3):
        assert np.array_equal(m[r], orc.interpolate(2.0, mag[r]) * (10.0 * np.arange(33) <= 100))
        assert np.array_equal(i[r], orc.interpolate(2.0, inc[r]) / 2.0)
    m, i = orc.apply_filter(10.0, [("envelope", 5)], mag, inc)
    assert np.all(i == 0)                                                   # amplitudeFilter0 (Filter.hs:72-73)


@pytest.mark.parametrize("coeff", [Fraction(1), Fraction(2), Fraction(1, 2), Fraction(3, 2), Fraction(2, 3),
                                   Fraction(7, 5), Fraction(0.3), Fraction(1.3), Fraction(1, 1000)])
def test_play_speed_state_machine(orc, coeff):
    for f_in in (0, 1, 2, 7, 100):
        src = orc.play_speed(coeff.numerator, coeff.denominator, f_in)
        # after i inputs floor(i * coeff) frames are out; frame i is repeated floor((i+1)c) - floor(i c) times
        counts = [(i + 1) * coeff // 1 - i * coeff // 1 for i in range(f_in)]
        ref = np.repeat(np.arange(f_in), np.array(counts, dtype=np.int64)) if f_in else np.zeros(0, dtype=np.int64)
        assert np.array_equal(src, ref)
        # output frame j repeats input frame ceil((j+1)/coeff) - 1
        for j in (0, len(src) // 2, len(src) - 1):
            if 0 <= j < len(src):
                assert src[j] == -((-(j + 1) * coeff.denominator) // coeff.numerator) - 1
