"""The C++ host mirror (include/vocoder.hpp) against the library on a GPU: the example program is
built by __graft_entry__.build() and exits 0 when the round trip reproduces its input and the
stage-by-stage path agrees with the fused one."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_host_example_runs():
    exe = os.path.join(ROOT, "examples", "host_roundtrip")
    assert os.path.exists(exe), "build it first: python -c 'import __graft_entry__ as g; g.build()'"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "fused round trip" in r.stdout
