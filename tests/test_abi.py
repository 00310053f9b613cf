"""The C-ABI library loads on a machine without a GPU and exports every symbol that
include/vocoder_hip.h declares (no compute calls here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "vocoder_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(voc_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_surface():
    syms = declared_symbols()
    for must in ("voc_ctx_create", "voc_ctx_destroy", "voc_r2c", "voc_c2r", "voc_do_fft", "voc_do_ifft",
                 "voc_analysis_step", "voc_synthesis_step", "voc_analysis_stage", "voc_synthesis_stage",
                 "voc_analysis_stage_dev", "voc_synthesis_stage_dev", "voc_roundtrip", "voc_roundtrip_dev",
                 "voc_frames_of", "voc_sum_frames", "voc_strerror", "voc_analysis_step_hn", "voc_synthesis_step_h",
                 "voc_frame_from_complex", "voc_frame_to_complex", "voc_add_frames_n"):
        assert must in syms


def test_product_library_exports_every_declared_symbol():
    from vocoder_amd._ffi import DEFAULT_LIB, SIGNATURES
    assert os.path.exists(DEFAULT_LIB), "build the library first: make -C vocoder_amd/csrc"
    dll = ctypes.CDLL(DEFAULT_LIB)
    for name in declared_symbols():
        assert hasattr(dll, name), f"{name} is declared in vocoder_hip.h but not exported"
    assert sorted(SIGNATURES) == declared_symbols(), "python binding and header disagree"


def test_status_strings_and_argument_checks_without_a_gpu():
    from vocoder_amd._ffi import Library
    lib = Library()
    assert lib.dll.voc_strerror(0) == b"ok"
    assert b"2 to 32768" in lib.dll.voc_strerror(-2)
    h = ctypes.c_void_p()
    w = (ctypes.c_double * 16)(*([1.0] * 16))
    # argument errors are reported before any device is touched
    assert lib.dll.voc_ctx_create(8, 2, w, 16, 0, 0, ctypes.byref(h)) == -1      # W > N (Vocoder.hs:132)
    assert lib.dll.voc_ctx_create(40000, 2, w, 16, 0, 0, ctypes.byref(h)) == -2  # above 32768
    assert lib.dll.voc_ctx_create(65536, 2, w, 16, 0, 0, ctypes.byref(h)) == -2  # power of two above 32768
    assert lib.dll.voc_ctx_create(32768, 2, w, 16, 0, 1, ctypes.byref(h)) == -2  # single precision: up to 16384
    assert lib.dll.voc_ctx_create(12, 2, w, 12, 0, 1, ctypes.byref(h)) == -2     # single precision: powers of two only
    assert lib.dll.voc_ctx_create(16, 0, w, 16, 0, 0, ctypes.byref(h)) == -1     # hop < 1
    assert lib.dll.voc_ctx_create(16, 4, w, 16, 0, 7, ctypes.byref(h)) == -2     # unknown precision
    assert lib.dll.voc_ctx_create(16, 4, None, 16, 0, 0, ctypes.byref(h)) == -1
    assert h.value is None


def test_missing_library_fails_loudly(tmp_path):
    from vocoder_amd._ffi import Library
    with pytest.raises(ImportError):
        Library(str(tmp_path / "nope.so"))


def test_product_package_does_not_reach_into_the_oracle():
    # the oracle is test infrastructure: nothing under vocoder_amd/ may import or link it
    pkg = os.path.join(ROOT, "vocoder_amd")
    for base, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip", "Makefile")):
                text = open(os.path.join(base, f), errors="ignore").read()
                assert "oracle" not in text.lower(), f"{f} mentions the oracle"
                assert "hip_emu" not in text and "libvocoder_emu" not in text, f"{f} mentions the emulation harness"
