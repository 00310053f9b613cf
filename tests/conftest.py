import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (built on demand with gcc)."""
    import oracle
    oracle.build()
    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def emu_lib():
    """Host-emulated build of the kernel source (development harness; lets the kernels' index
    logic be checked without a GPU).  Never used by the product path."""
    from vocoder_amd._ffi import Library
    d = os.path.join(ROOT, "tools", "emu")
    so = os.path.join(d, "libvocoder_emu.so")
    srcs = [os.path.join(d, "hip_emu.h"), os.path.join(ROOT, "vocoder_amd", "csrc", "voc_kernels.h"),
            os.path.join(ROOT, "vocoder_amd", "csrc", "vocoder_hip.cpp"), os.path.join(ROOT, "vocoder_amd", "csrc", "voc_kernels_fast.h"),
            os.path.join(ROOT, "vocoder_amd", "csrc", "voc_kernels_fast16.h"),
            os.path.join(ROOT, "include", "vocoder_hip.h")]
    if (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        try:
            subprocess.check_call(["make", "-C", d, "-B"], stdout=subprocess.DEVNULL)
        except Exception as e:  # pragma: no cover
            pytest.skip(f"cannot build the host emulation harness: {e}")
    return Library(so)


@pytest.fixture(scope="session")
def gpu_lib():
    """The product library on a real GPU.  Fails (does not skip) if the library is missing."""
    import torch
    from vocoder_amd._ffi import default_library
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return default_library()
