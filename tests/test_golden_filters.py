"""Committed filter / playSpeed fixtures (tests/golden/filters.npz, made by make_golden_filters.py
from the filter oracle): the oracle still reproduces them, and the library matches them -- on the
host emulation here, on the GPU under the gpu marker."""
import json
import os
from fractions import Fraction

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
FIX = np.load(os.path.join(HERE, "golden", "filters.npz"))
CHAINS = json.loads(str(FIX["chains"]))
SPEEDS = [Fraction(a, b) for a, b in json.loads(str(FIX["speeds"]))]


def _hann(n):
    return 0.5 * (1 - np.cos(2 * np.pi * np.arange(n) / n))


def _close(name, got, want):
    if "envelope" in name:          # log / exp differ between math libraries in the last bits
        return np.allclose(got, want, rtol=1e-12, atol=1e-15)
    return np.array_equal(got, want)


@pytest.mark.parametrize("n", [128, 4096])
@pytest.mark.parametrize("name", sorted(CHAINS))
def test_oracle_reproduces_filter_fixtures(orc, n, name):
    chain = [(s[0], np.array(s[1])) if s[0] == "convolution" else tuple(s) for s in CHAINS[name]]
    m, i = orc.apply_filter(48000.0 / n, chain, FIX[f"n{n}_mag"], FIX[f"n{n}_inc"])
    assert np.array_equal(m, FIX[f"n{n}_{name}_mag"]) and np.array_equal(i, FIX[f"n{n}_{name}_inc"])


def _library_case(lib, n, name):
    from vocoder_amd import vocoder as V, filter as Fl
    par = V.VocoderParams(n, n // 4, _hann(n), lib=lib)
    f = Fl.id_filter()
    for s in CHAINS[name]:
        kind, prm = s[0], s[1:]
        fn = {"pitch_shift": Fl.pitch_shift_interpolate, "convolution": Fl.convolution_filter,
              "envelope": Fl.envelope_filter}.get(kind) or getattr(Fl, kind)
        f = f >> fn(*prm)
    m, i = f.bind(par, 48000.0 / n).apply(FIX[f"n{n}_mag"], FIX[f"n{n}_inc"])
    assert _close(name, m, FIX[f"n{n}_{name}_mag"]) and np.array_equal(i, FIX[f"n{n}_{name}_inc"])
    for q in SPEEDS:
        assert np.array_equal(Fl.play_speed_map(par, q, 37), FIX[f"speed_{q.numerator}_{q.denominator}_f37"])


@pytest.mark.parametrize("n", [128, 4096])
@pytest.mark.parametrize("name", sorted(CHAINS))
def test_emulated_library_matches_filter_fixtures(emu_lib, n, name):
    _library_case(emu_lib, n, name)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [128, 4096])
@pytest.mark.parametrize("name", sorted(CHAINS))
def test_gpu_library_matches_filter_fixtures(gpu_lib, n, name):
    _library_case(gpu_lib, n, name)
