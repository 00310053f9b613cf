"""ctypes binding of the C ABI declared in include/vocoder_hip.h.

The product path loads ``vocoder_amd/libvocoder_hip.so`` (built by hipcc for gfx950) and
raises if it is missing: there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, "libvocoder_hip.so")

_dp = C.POINTER(C.c_double)
_vp = C.c_void_p
_i32, _i64 = C.c_int32, C.c_int64

# name -> (restype, argtypes); mirrors include/vocoder_hip.h one to one
SIGNATURES = {
    "voc_strerror": (C.c_char_p, [_i32]),
    "voc_last_hip_error": (C.c_char_p, [_vp]),
    "voc_ctx_create": (_i32, [_i32, _i32, _vp, _i32, _i32, _i32, C.POINTER(_vp)]),
    "voc_ctx_destroy": (_i32, [_vp]),
    "voc_frame_length": (_i32, [_vp]),
    "voc_input_frame_length": (_i32, [_vp]),
    "voc_freq_frame_length": (_i32, [_vp]),
    "voc_hop_size": (_i32, [_vp]),
    "voc_window": (_i32, [_vp, _vp]),
    "voc_volume_coeff": (C.c_double, [_vp]),
    "voc_r2c": (_i32, [_vp, _vp, _vp, _i64]),
    "voc_c2r": (_i32, [_vp, _vp, _vp, _i64]),
    "voc_do_fft": (_i32, [_vp, _vp, _vp, _i64]),
    "voc_do_ifft": (_i32, [_vp, _vp, _vp, _i64]),
    "voc_analysis_step": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64]),
    "voc_synthesis_step": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64]),
    "voc_add_frames": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64]),
    "voc_add_frames_dev": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "voc_analysis_step_hn": (_i32, [_i32, _i32, _i64, _vp, _vp, _vp, _vp, _vp, _i64]),
    "voc_synthesis_step_h": (_i32, [_i32, _i64, _vp, _vp, _vp, _vp, _vp, _i64]),
    "voc_frame_from_complex": (_i32, [_vp, _vp, _vp, _i64]),
    "voc_frame_to_complex": (_i32, [_vp, _vp, _vp, _i64]),
    "voc_add_frames_n": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64]),
    "voc_analysis_stage": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i64]),
    "voc_synthesis_stage": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i64]),
    "voc_analysis_stage_dev": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _vp]),
    "voc_synthesis_stage_dev": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _vp]),
    "voc_num_frames": (_i64, [_vp, _i64]),
    "voc_leftover_length": (_i64, [_vp, _i64]),
    "voc_ola_length": (_i64, [_vp, _i64, _i64]),
    "voc_frames_of": (_i32, [_vp, _vp, _i64, _i64, _i64, _vp]),
    "voc_sum_frames": (_i32, [_vp, _vp, _i64, _i64, _i64, _vp, _i64]),
    "voc_roundtrip": (_i32, [_vp, _vp, _i64, _i64, _i64, _i64, _vp, _i64, _vp, _vp, _vp, _i64]),
    "voc_roundtrip_workspace_bytes": (_i64, [_vp, _i64, _i64]),
    "voc_analysis_signal_dev": (_i32, [_vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _i64, _i32, _vp]),
    "voc_synthesis_signal_dev": (_i32, [_vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _i64, _i64, _vp]),
    "voc_filter_apply_f32_dev": (_i32, [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _vp]),
    "voc_filter_apply_f32": (_i32, [_vp, _vp, _vp, _i64, _i64]),
    "voc_synthesis_mapped_f32_dev": (_i32, [_vp, _vp, _i64, _vp, _vp, _i64, _vp, _i64, _i64, _i64, _vp, _vp, _i64, _i64, _vp]),
    "voc_analysis_signal_f32_dev": (_i32, [_vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _i64, _i32, _vp]),
    "voc_synthesis_signal_f32_dev": (_i32, [_vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _i64, _i64, _vp]),
    "voc_r2c_f32": (_i32, [_vp, _vp, _vp, _i64]),
    "voc_c2r_f32": (_i32, [_vp, _vp, _vp, _i64]),
    "voc_do_fft_f32": (_i32, [_vp, _vp, _vp, _i64]),
    "voc_do_ifft_f32": (_i32, [_vp, _vp, _vp, _i64]),
    "voc_analysis_step_f32": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64]),
    "voc_synthesis_step_f32": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64]),
    "voc_add_frames_f32": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64]),
    "voc_analysis_stage_f32": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i64]),
    "voc_synthesis_stage_f32": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i64]),
    "voc_analysis_stage_f32_dev": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _vp]),
    "voc_synthesis_stage_f32_dev": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _vp]),
    "voc_phase_advance_dev": (_i32, [_vp, _vp, _i64, _i64, _i64, _vp, _vp]),
    "voc_filter_create": (_i32, [_vp, C.c_double, C.POINTER(_vp)]),
    "voc_filter_destroy": (_i32, [_vp]),
    "voc_filter_length": (_i32, [_vp]),
    "voc_filter_append": (_i32, [_vp, _i32, _vp, _i32]),
    "voc_filter_apply": (_i32, [_vp, _vp, _vp, _i64, _i64]),
    "voc_filter_apply_dev": (_i32, [_vp, _vp, _vp, _i64, _i64, _i64, _vp]),
    "voc_synthesis_filtered_dev": (_i32, [_vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _i64, _i64, _vp]),
    "voc_filter_apply_at_dev": (_i32, [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _vp]),
    "voc_binary_frames_dev": (_i32, [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "voc_roundtrip_static_dev": (_i32, [_vp, _vp, _i64, _i64, _i64, _i64, _vp, _i64, _vp, _vp, _vp, _i64, _vp]),
    "voc_play_speed_frames": (_i64, [_i64, _i64, _i64]),
    "voc_play_speed_map": (_i32, [_i64, _i64, _i64, _vp, _i64]),
    "voc_synthesis_mapped_dev": (_i32, [_vp, _vp, _i64, _vp, _vp, _i64, _vp, _i64, _i64, _i64, _vp, _vp, _i64, _i64, _vp]),
    "voc_filter_touches": (_i32, [_vp, _vp]),
    "voc_gather_frames_dev": (_i32, [_vp, _vp, _vp, _i64, _i64, _vp, _i64, _vp, _vp, _i64, _vp]),
    "voc_deinterleave_dev": (_i32, [_vp, _vp, _i64, _i64, _vp, _i64, _vp]),
    "voc_interleave_dev": (_i32, [_vp, _vp, _i64, _i64, _i64, _vp, _vp]),
    "voc_pcm16_to_planar_dev": (_i32, [_vp, _vp, _i64, _i64, _vp, _i64, _vp]),
    "voc_planar_to_pcm16_dev": (_i32, [_vp, _vp, _i64, _i64, _i64, _i32, _vp, _vp]),
    "voc_stream_create": (_i32, [_vp, _i64, _i64, _vp, C.POINTER(_vp)]),
    "voc_stream_create_mode": (_i32, [_vp, _i64, _i64, _vp, _i32, C.POINTER(_vp)]),
    "voc_stream_set_filter": (_i32, [_vp, _vp]),
    "voc_stream_set_play_speed": (_i32, [_vp, _i64, _i64, _vp]),
    "voc_stream_set_phases_f32": (_i32, [_vp, _vp, _vp]),
    "voc_stream_get_phases_f32": (_i32, [_vp, _vp, _vp]),
    "voc_stream_leftover_f32": (_i64, [_vp, _vp, _i64]),
    "voc_stream_push_f32": (_i32, [_vp, _vp, _i64, _i64, _vp, _i64, _i64, C.POINTER(_i64)]),
    "voc_stream_push_f32_dev": (_i32, [_vp, _vp, _i64, _i64, _vp, _i64, _i64, C.POINTER(_i64), _vp]),
    "voc_stream_flush_f32": (_i32, [_vp, _vp, _i64, _i64, C.POINTER(_i64)]),
    "voc_stream_flush_f32_dev": (_i32, [_vp, _vp, _i64, _i64, C.POINTER(_i64), _vp]),
    "voc_stream_destroy": (_i32, [_vp]),
    "voc_stream_set_phases": (_i32, [_vp, _vp, _vp]),
    "voc_stream_get_phases": (_i32, [_vp, _vp, _vp]),
    "voc_stream_leftover": (_i64, [_vp, _vp, _i64]),
    "voc_stream_push_length": (_i64, [_vp, _i64]),
    "voc_stream_flush_length": (_i64, [_vp]),
    "voc_stream_push": (_i32, [_vp, _vp, _i64, _i64, _vp, _i64, _i64, C.POINTER(_i64)]),
    "voc_stream_push_dev": (_i32, [_vp, _vp, _i64, _i64, _vp, _i64, _i64, C.POINTER(_i64), _vp]),
    "voc_stream_flush": (_i32, [_vp, _vp, _i64, _i64, C.POINTER(_i64)]),
    "voc_stream_flush_dev": (_i32, [_vp, _vp, _i64, _i64, C.POINTER(_i64), _vp]),
    "voc_roundtrip_f32": (_i32, [_vp, _vp, _i64, _i64, _i64, _i64, _vp, _i64, _vp, _vp, _vp, _i64]),
    "voc_roundtrip_f32_dev": (_i32, [_vp, _vp, _i64, _i64, _i64, _i64, _vp, _i64, _vp, _vp, _vp, _i64, _vp, _i64, _vp]),
    "voc_profile_enable": (_i32, [_vp, _i32]),
    "voc_profile_read": (_i32, [_vp, _vp, _vp]),
    "voc_roundtrip_dev": (_i32, [_vp, _vp, _i64, _i64, _i64, _i64, _vp, _i64, _vp, _vp, _vp, _i64, _vp, _i64, _vp]),
}


class VocoderError(RuntimeError):
    """A C-ABI call returned a non-zero status (the reference raises `error` here)."""

    def __init__(self, status: int, text: str):
        super().__init__(f"vocoder_hip: {text} (status {status})")
        self.status = status


class Library:
    """One loaded copy of the C-ABI library."""

    def __init__(self, path: str | None = None):
        path = path or DEFAULT_LIB
        if not os.path.exists(path):
            raise ImportError(
                f"{path} is missing: build it with `make -C vocoder_amd/csrc` (hipcc, gfx950). "
                "vocoder_amd has no CPU fallback.")
        if path == DEFAULT_LIB:
            # share the HIP runtime PyTorch already loaded (same soname), when PyTorch is in use
            import sys
            if "torch" not in sys.modules:
                try:
                    import torch  # noqa: F401
                except Exception:  # torch is plumbing only; the library also runs without it
                    pass
        self.path = path
        self.dll = C.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(self.dll, name)  # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args

    def check(self, status: int, ctx=None):
        if status != 0:
            text = self.dll.voc_strerror(status).decode()
            if ctx is not None and status == -3:
                text += ": " + self.dll.voc_last_hip_error(ctx).decode()
            raise VocoderError(status, text)


_default: Library | None = None


def default_library() -> Library:
    global _default
    if _default is None:
        _default = Library()
    return _default
