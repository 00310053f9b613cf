"""Cross-checks and analytic known answers for the oracle's FFT / phase arithmetic.  The
reference holds no numeric test vector for this part (parity unpinned); these checks anchor
the restatement on pocketfft, a long-double DFT and closed-form answers."""
import math

import numpy as np
import pytest

from helpers import circ_dist


@pytest.mark.parametrize("n", [8, 16, 64, 100, 243, 512, 2048, 4096])
def test_r2c_against_pocketfft_and_long_double(orc, n):
    rng = np.random.default_rng(n)
    x = rng.standard_normal(n)
    ref = np.fft.rfft(x)
    got = orc.fft_r2c(x)
    assert np.abs(got - ref).max() <= 1e-13 * np.abs(ref).max()
    assert got[0].imag == 0.0 and (n % 2 or got[-1].imag == 0.0)
    if n <= 512:
        assert np.abs(orc.dft_r2c_ld(x) - got).max() <= 1e-13 * np.abs(ref).max()


@pytest.mark.parametrize("n", [8, 64, 100, 243, 4096])
def test_c2r_is_scaled_inverse_ignoring_dc_nyquist_imag(orc, n):
    rng = np.random.default_rng(n + 1)
    x = rng.standard_normal(n)
    spec = np.fft.rfft(x)
    assert np.abs(orc.fft_c2r(spec, n) - x).max() <= 1e-13
    dirty = spec.copy()
    dirty[0] += 5j
    if n % 2 == 0:
        dirty[-1] -= 7j
    assert np.array_equal(orc.fft_c2r(dirty, n), orc.fft_c2r(spec, n))
    assert np.abs(orc.fft_c2r(spec, n) - np.fft.irfft(spec, n=n)).max() <= 1e-13


def test_wrap_is_floored_modulus_range(orc):
    # wrap e = (e+pi) `mod'` (2 pi) - pi lands in [-pi, pi)   (Vocoder.hs:166-167)
    for e in [0.0, 1.0, -1.0, math.pi, -math.pi, 3 * math.pi, -3 * math.pi, 100.0, -100.0, 1e6, -1e6, 6.5, -6.5]:
        w = orc.wrap(e)
        assert -math.pi <= w < math.pi
        assert abs(((w - e) / (2 * math.pi)) - round((w - e) / (2 * math.pi))) < 1e-6
    assert orc.wrap(math.pi) == -math.pi and orc.wrap(-math.pi) == -math.pi


def test_wrap_exact_floor_against_rational_arithmetic(orc):
    # Data.Fixed.mod' floors the EXACT quotient: check against Python's exact Fractions
    from fractions import Fraction
    rng = np.random.default_rng(7)
    two_pi = 2 * math.pi
    for e in list(rng.uniform(-1e4, 1e4, 300)) + [k * two_pi - math.pi for k in range(-6, 7)] + \
            [np.nextafter(k * two_pi - math.pi, s) for k in range(-6, 7) for s in (-1e9, 1e9)]:
        e = float(e)
        n = e + math.pi
        f = math.floor(Fraction(n) / Fraction(two_pi))
        want = (n - float(f) * two_pi) - math.pi
        assert orc.wrap(e) == want


def test_phase_follows_ghc_atan2(orc):
    assert orc.phase(0.0, 0.0) == 0.0 and orc.phase(-0.0, -0.0) == 0.0   # phase (0:+0) = 0
    assert orc.phase(-1.0, 0.0) == math.pi and orc.phase(-1.0, -0.0) == -math.pi
    assert orc.phase(0.0, 2.0) == math.pi / 2 and orc.phase(0.0, -2.0) == -math.pi / 2
    rng = np.random.default_rng(3)
    for re, im in rng.standard_normal((200, 2)):
        assert abs(orc.phase(re, im) - math.atan2(im, re)) <= 1e-15  # pi + atan(y/x): two roundings


def test_magnitude_scaled(orc):
    assert orc.magnitude(3.0, 4.0) == 5.0
    assert abs(orc.magnitude(3e200, 4e200) / 5e200 - 1) < 1e-15     # no overflow thanks to scaleFloat
    assert abs(orc.magnitude(3e-200, 4e-200) / 5e-200 - 1) < 1e-15  # no underflow either
    assert orc.magnitude(0.0, 0.0) == 0.0


def test_volume_coeff(orc):
    w = orc.make_window("hann", 64)
    assert abs(orc.volume_coeff(16, w) - 16 / np.sum(w ** 2)) < 1e-15


def test_windows_match_closed_forms(orc):
    n = 33
    x = np.arange(n) / (n - 1)
    assert np.allclose(orc.make_window("hann", n), 0.5 * (1 - np.cos(2 * np.pi * x)), atol=1e-16)
    assert np.allclose(orc.make_window("hamming", n), 25 / 46 - 21 / 46 * np.cos(2 * np.pi * x), atol=1e-16)
    assert np.allclose(orc.make_window("blackman", n), 0.42 - 0.5 * np.cos(2 * np.pi * x) + 0.08 * np.cos(4 * np.pi * x), atol=1e-15)
    assert np.array_equal(orc.make_window("box", n), np.ones(n))
    assert np.allclose(orc.make_window("triangle", n), 1 - np.abs(2 * x - 1), atol=1e-16)
    assert np.allclose(orc.make_window("lanczos", n), np.sinc(2 * x - 1), atol=1e-15)
    # the package's host-side generators produce the same tables
    from vocoder_amd import window as Wn
    for name, fn in (("hann", Wn.hann_window), ("hamming", Wn.hamming_window), ("blackman", Wn.blackman_window),
                     ("exact_blackman", Wn.exact_blackman_window), ("box", Wn.box_window), ("triangle", Wn.triangle_window),
                     ("lanczos", Wn.lanczos_window), ("flattop", Wn.flat_top_window)):
        assert np.allclose(fn(n), orc.make_window(name, n), rtol=0, atol=2e-16), name


def test_do_fft_rewind_gives_zero_phase_window(orc):
    # a constant frame: spectrum = FFT of the centred window, which is real (Vocoder.hs:121-125)
    n = 64
    w = orc.make_window("hann", n)[:n]
    w = np.concatenate([w[: n // 2 + 1], w[1: n // 2][::-1]])  # exactly symmetric about n/2
    spec = orc.do_fft(n, w, np.ones(n))
    assert np.abs(spec.imag).max() <= 1e-13


def test_bin_centred_sine_known_answer(orc):
    # x[n] = A cos(2 pi k0 n / N + phi): mag[k0] = A sum(w)/2 (mirror-image leakage aside) and the
    # phase increment is the bin frequency 2 pi k0 / N rad/sample, constant across frames
    n, h, k0, amp = 256, 64, 19, 0.7
    w = orc.make_window("hann", n)
    f = 12
    t = np.arange((f - 1) * h + n)
    x = amp * np.cos(2 * np.pi * k0 * t / n + 0.3)
    frames = np.stack([x[j * h: j * h + n] for j in range(f)])
    ph, mag, inc = orc.analysis_stage(n, h, w, np.zeros(n // 2 + 1), frames)
    assert np.allclose(mag[:, k0], amp * w.sum() / 2, rtol=1e-5)
    assert np.allclose(inc[1:, k0], 2 * np.pi * k0 / n, atol=1e-7)


def test_synthesis_phase_telescopes_to_analysis_phase(orc):
    # with zero initial phases, phi_syn(f) = sum_j H*ph_inc_j == raw analysis phase of frame f (mod 2 pi)
    n, h = 128, 32
    rng = np.random.default_rng(11)
    w = orc.make_window("blackman", n)
    r = orc.roundtrip(n, h, w, rng.standard_normal(n + 40 * h))
    assert circ_dist(r["syn_phase"], r["ana_phase"]).max() <= 1e-9


def test_roundtrip_identity_with_cola_window(orc):
    # periodic Hann at 75 % overlap: sum_j w^2 = 1.5 N/H / ... the pipeline reproduces its input
    n, h = 256, 64
    k = np.arange(n)
    w = 0.5 * (1 - np.cos(2 * np.pi * k / n))   # periodic Hann: exact COLA for w^2 at hop N/4
    rng = np.random.default_rng(2)
    x = rng.standard_normal(n + 50 * h)
    r = orc.roundtrip(n, h, w, x)
    y = r["y"]
    assert np.abs(y[n: 40 * h] - x[n: 40 * h]).max() <= 1e-10  # grows slowly: the phase is accumulated unwrapped


def test_add_zero_padding_errors_when_window_longer_than_frame(orc):
    with pytest.raises(ValueError):
        orc.do_fft(8, np.ones(12), np.ones(12))   # Vocoder.hs:132


def test_odd_length_quirk_is_reproduced(orc):
    # doIFFT applies the same left rotation again (Vocoder.hs:196): for odd N that is a one-sample
    # circular shift, for even N the identity
    n = 9
    w = np.ones(n)
    x = np.arange(1.0, n + 1)
    back = orc.do_ifft(n, w, orc.do_fft(n, w, x))
    assert np.allclose(back, np.roll(x, -(2 * (n // 2) - n) % n), atol=1e-12) or np.allclose(back, np.roll(x, 1), atol=1e-12)
    n = 8
    x = np.arange(1.0, n + 1)
    assert np.allclose(orc.do_ifft(n, np.ones(n), orc.do_fft(n, np.ones(n), x)), x, atol=1e-12)
