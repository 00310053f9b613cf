"""Frame lengths that are not powers of two (chirp-z path) on CPU, host-emulated kernels against
the oracle, whose FFT for such lengths is a long double DFT."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import helpers as Hh  # noqa: E402

SIZES = [(3, 3, 1), (7, 7, 2), (12, 12, 3), (15, 11, 4), (100, 81, 20), (250, 250, 50), (1000, 1000, 250)]


@pytest.mark.parametrize("n,w,h", SIZES)
def test_fft_layer_any_length(emu_lib, orc, n, w, h):
    Hh.check_fft_layer(emu_lib, orc, n, w, h, np.random.default_rng(n))


@pytest.mark.parametrize("n,w,h", SIZES)
def test_stages_and_roundtrip_any_length(emu_lib, orc, n, w, h):
    rng = np.random.default_rng(n + 1)
    Hh.check_stages(emu_lib, orc, n, w, h, 7, 2, rng, ana_tile=3, syn_tile=4)
    Hh.check_roundtrip(emu_lib, orc, n, w, h, w + h * 9 + 5, 2, rng, ana_tile=4, syn_tile=3)


def test_length_above_4096_runs_through_device_memory(emu_lib, orc):
    """Above 4096 the chirp-z transform is cut into interleaved 8192-point transforms (index logic only here:
    one small case; the sizes proper are GPU tests)."""
    rng = np.random.default_rng(5)
    Hh.check_fft_layer(emu_lib, orc, 4100, 4100, 1025, rng)
    Hh.check_stages(emu_lib, orc, 4100, 3000, 1000, 3, 1, rng)


def test_filters_and_stream_on_a_non_power_of_two_length(emu_lib, orc):
    from vocoder_amd import vocoder as V, conduit as Cd, stream as S, filter as Fl
    n, h = 100, 25
    win = orc.make_window("hann", n)
    par = V.VocoderParams(n, h, win, lib=emu_lib)
    rng = np.random.default_rng(3)
    mag, inc = rng.uniform(0, 1, (2, 4, par.K)), rng.uniform(-1, 1, (2, 4, par.K))
    chain = Fl.pitch_shift_interpolate(1.5) >> Fl.lowpass_butterworth(2.0, 5000.0) >> Fl.envelope_filter(5)
    m, i = chain.bind(par, 480.0).apply(mag, inc)
    mo, io = orc.apply_filter(480.0, [("pitch_shift", 1.5), ("lowpass_butterworth", 2.0, 5000.0), ("envelope", 5)], mag, inc)
    assert np.allclose(m, mo, rtol=1e-12, atol=1e-15) and np.array_equal(i, io)
    x = rng.standard_normal((1, n + h * 13 + 2))
    y_ref = Cd.process_with(par, x)[0]
    st = S.Stream(par, 1)
    y = np.concatenate([st.push(x[:, :77]), st.push(x[:, 77:]), st.flush()], axis=1)
    assert np.array_equal(y, y_ref)


def test_length_above_16384(emu_lib, orc):
    """Frame lengths above 16384, powers of two included, take the chirp-z transform through device memory with
    eight interleaved 8192-point transforms; their STFT rows (up to 16385 bins) go through the filter kernel's
    build for long rows, which stages one array at a time in LDS."""
    from vocoder_amd import vocoder as V, filter as Fl
    rng = np.random.default_rng(6)
    Hh.check_fft_layer(emu_lib, orc, 32768, 32768, 8192, rng)
    Hh.check_stages(emu_lib, orc, 16500, 16000, 4000, 2, 1, rng)
    n = 32768
    par = V.VocoderParams(n, 8192, orc.make_window("hann", n), lib=emu_lib)
    mag, inc = rng.uniform(0, 1, (1, 2, par.K)), rng.uniform(-1, 1, (1, 2, par.K))
    chain = Fl.pitch_shift_interpolate(1.5) >> Fl.lowpass_butterworth(2.0, 5000.0) >> Fl.envelope_filter(5)
    spec = [("pitch_shift", 1.5), ("lowpass_butterworth", 2.0, 5000.0), ("envelope", 5)]
    m, i = chain.bind(par, 48000.0 / n).apply(mag, inc)
    mo, io = orc.apply_filter(48000.0 / n, spec, mag, inc)
    assert np.allclose(m, mo, rtol=1e-12, atol=1e-15) and np.array_equal(i, io)
    par.close()
    with pytest.raises(V.VocoderError) as e:
        V.VocoderParams(32770, 8192, np.ones(1000), lib=emu_lib)
    assert e.value.status == -2
