"""The C++ host mirror (include/vocoder.hpp) against the library on a GPU: the example program is
built by __graft_entry__.build(), exits 0 when the round trip reproduces its input and the
stage-by-stage path agrees with the fused one, and leaves its results for the comparison with the oracle."""
import os
import subprocess
from fractions import Fraction

import numpy as np
import pytest

import helpers as Hh

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_host_example_runs_and_matches_the_oracle(orc, tmp_path):
    exe = os.path.join(ROOT, "examples", "host_roundtrip")
    assert os.path.exists(exe), "build it first: python -c 'import __graft_entry__ as g; g.build()'"
    r = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "fused round trip" in r.stdout
    got = {k: np.fromfile(tmp_path / f"{k}.f64") for k in ("window", "x", "y_fused", "y_stages", "y_lowpass1000_amplify0.5",
                                                            "y_playspeed2")}
    n, hop, win, x = 2048, 512, got["window"], got["x"]
    assert win.size == n and x.size == 48000
    ref = orc.roundtrip(n, hop, win, x)
    for k in ("y_fused", "y_stages"):       # Vocoder.Conduit.process idFilter, fused and through the stage functions
        assert got[k].shape == ref["y"].shape and Hh.rel_l2(got[k], ref["y"]) <= Hh.TOL_L2
    step = 48000.0 / n
    flt = Hh.oracle_filtered(orc, n, hop, win, x, step, [("lowpass_brickwall", 1000.0), ("amplify", 0.5)])
    assert got["y_lowpass1000_amplify0.5"].shape == flt["y"].shape
    assert Hh.rel_l2(got["y_lowpass1000_amplify0.5"], flt["y"]) <= Hh.TOL_L2
    fast = Hh.oracle_filtered(orc, n, hop, win, x, step, [], speed=Fraction(2))
    assert got["y_playspeed2"].shape == fast["y"].shape and Hh.rel_l2(got["y_playspeed2"], fast["y"]) <= Hh.TOL_L2
