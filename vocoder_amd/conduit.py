"""Batch forms of the reference's streaming wrappers (vocoder-conduit/src/Vocoder/Conduit.hs and
Conduit/Frames.hs) over the C ABI.

The reference threads three pieces of state between chunks / sources: the samples genFramesOfE
has not yet consumed, and the analysis and synthesis phases (Conduit.hs:46-53,
vocoder-audio/src/Vocoder/Audio.hs:31-33).  `process_with` takes and returns exactly that
triple, so a long stream can be fed segment by segment; `process` is the reference's
`process` (empty leftover, zero phases, results of the state discarded).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from .vocoder import VocoderParams, _f64, _p, zero_phase

__all__ = ["num_frames", "leftover_length", "ola_length", "frames_of_e", "sum_frames_e", "process_with",
           "process", "process_dev", "process_static_dev", "binary_frames_dev", "roundtrip_workspace_bytes", "process_filter_dev", "deinterleave_dev", "interleave_dev", "pcm16_to_planar_dev", "planar_to_pcm16_dev"]


def num_frames(par: VocoderParams, length: int) -> int:
    return par.lib.dll.voc_num_frames(par.handle, int(length))


def leftover_length(par: VocoderParams, length: int) -> int:
    return par.lib.dll.voc_leftover_length(par.handle, int(length))


def ola_length(par: VocoderParams, nframes: int, chunk: int) -> int:
    return par.lib.dll.voc_ola_length(par.handle, int(nframes), int(chunk))


def frames_of_e(par: VocoderParams, x):
    """genFramesOfE W H on one whole buffer per channel (Frames.hs:26-38).
    x: (L,) or (C, L) -> (frames (.., F, W), leftover (.., L - F*H))."""
    x = _f64(x)
    multi = x.ndim == 2
    x2 = x if multi else x[None]
    c, length = x2.shape
    f = num_frames(par, length)
    frames = np.empty((c, f, par.W))
    par._call("voc_frames_of", _p(x2), length, length, c, _p(frames))
    keep = leftover_length(par, length)
    left = x2[:, length - keep:].copy()
    return (frames, left) if multi else (frames[0], left[0])


def sum_frames_e(par: VocoderParams, frames, chunk: int | None = None):
    """sumFramesE chunk H (Frames.hs:41-58). frames: (F, W) or (C, F, W) -> (.., out_len)."""
    frames = _f64(frames)
    multi = frames.ndim == 3
    fr = frames if multi else frames[None]
    c, f, _ = fr.shape
    chunk = par.H if chunk is None else chunk
    n = ola_length(par, f, chunk)
    y = np.zeros((c, n))
    par._call("voc_sum_frames", _p(fr), f, c, chunk, _p(y), n)
    return y if multi else y[0]


def process_with(par: VocoderParams, x, leftover=None, phases=None, chunk: int | None = None):
    """One `processWith q phs idFilter` step (Conduit.hs:53): returns (y, leftover', (ana_ph', syn_ph')).
    x: (L,) or (C, L) new samples; leftover: (.., r) samples kept from the previous call."""
    f32 = getattr(par, "precision", "f64") == "f32"
    dt = np.float32 if f32 else np.float64
    x = np.ascontiguousarray(x, dtype=dt)
    multi = x.ndim == 2
    x2 = x if multi else x[None]
    c = x2.shape[0]
    if leftover is not None and np.size(leftover):
        x2 = np.concatenate([np.asarray(leftover, dtype=dt).reshape(c, -1), x2], axis=1)
    x2 = np.ascontiguousarray(x2)
    length = x2.shape[1]
    chunk = par.H if chunk is None else chunk
    f = num_frames(par, length)
    n = ola_length(par, f, chunk)
    keep = leftover_length(par, length)
    if phases is None:
        aph, sph = zero_phase(par, c).astype(dt), zero_phase(par, c).astype(dt)
    else:
        aph = np.array(phases[0], dtype=dt, copy=True).reshape(c, par.K)
        sph = np.array(phases[1], dtype=dt, copy=True).reshape(c, par.K)
    y = np.empty((c, n), dtype=dt)
    left = np.empty((c, keep), dtype=dt)
    par._call("voc_roundtrip_f32" if f32 else "voc_roundtrip", _p(x2), length, length, c, chunk, _p(y), n, _p(aph), _p(sph), _p(left), keep)
    if not multi:
        return y[0], left[0], (aph[0], sph[0])
    return y, left, (aph, sph)


def process(par: VocoderParams, x):
    """Vocoder.Conduit.process par idFilter on a whole buffer (Conduit.hs:50-53)."""
    return process_with(par, x)[0]


def roundtrip_workspace_bytes(par: VocoderParams, length: int, channels: int) -> int:
    return par.lib.dll.voc_roundtrip_workspace_bytes(par.handle, int(length), int(channels))


def process_dev(par: VocoderParams, x, y, ana_phase, syn_phase, length: int | None = None, chunk: int | None = None,
                leftover=None, workspace=None, stream=None):
    """Device-resident `process_with`: x (C, >=length), y (C, >=ola_length), phases (C, K) and the
    optional leftover / workspace are torch CUDA float64 / uint8 tensors; enqueues on `stream`
    (default: torch's current stream) and returns without synchronising."""
    import torch

    f32 = getattr(par, "precision", "f64") == "f32"
    dt = torch.float32 if f32 else torch.float64
    assert x.is_cuda and y.is_cuda and x.dtype == dt and y.dtype == dt
    assert x.stride(-1) == 1 and y.stride(-1) == 1 and ana_phase.is_contiguous() and syn_phase.is_contiguous()
    c = x.shape[0]
    length = x.shape[1] if length is None else length
    chunk = par.H if chunk is None else chunk
    st = torch.cuda.current_stream(x.device).cuda_stream if stream is None else stream
    lp, ls = (None, 0) if leftover is None else (C.c_void_p(leftover.data_ptr()), leftover.stride(0))
    wp, wb = (None, 0) if workspace is None else (C.c_void_p(workspace.data_ptr()), workspace.numel() * workspace.element_size())
    par._call("voc_roundtrip_f32_dev" if f32 else "voc_roundtrip_dev", C.c_void_p(x.data_ptr()), x.stride(0), length, c, chunk,
              C.c_void_p(y.data_ptr()), y.stride(0), C.c_void_p(ana_phase.data_ptr()),
              C.c_void_p(syn_phase.data_ptr()), lp, ls, wp, wb, C.c_void_p(st))


def process_static_dev(par: VocoderParams, x, y, ana_phase, syn_phase, length: int | None = None, chunk: int | None = None,
                       leftover=None, stream=None):
    """`process_dev` through the entry point that allocates nothing (voc_roundtrip_static_dev): the
    STFT of the call lives in the library's static scratch array, so one channel of the input must
    fit VOC_STATIC_SCRATCH_BYTES, and calls that overlap on one GPU race."""
    import torch

    assert x.is_cuda and y.is_cuda and x.dtype == torch.float64 and y.dtype == torch.float64
    assert x.stride(-1) == 1 and y.stride(-1) == 1 and ana_phase.is_contiguous() and syn_phase.is_contiguous()
    c = x.shape[0]
    length = x.shape[1] if length is None else length
    chunk = par.H if chunk is None else chunk
    st = torch.cuda.current_stream(x.device).cuda_stream if stream is None else stream
    lp, ls = (None, 0) if leftover is None else (C.c_void_p(leftover.data_ptr()), leftover.stride(0))
    par._call("voc_roundtrip_static_dev", C.c_void_p(x.data_ptr()), x.stride(0), length, c, chunk,
              C.c_void_p(y.data_ptr()), y.stride(0), C.c_void_p(ana_phase.data_ptr()),
              C.c_void_p(syn_phase.data_ptr()), lp, ls, C.c_void_p(st))


def binary_frames_dev(par: VocoderParams, op: str, stft1, stft2):
    """Two STFT streams combined element by element the way vocoder-jack's `--multiply`, `--divide`
    and `--add` combine them (vocoder-dunai/example/VocoderJack.hs:107-110,127-135).  stft1, stft2:
    (mag, ph_inc) pairs of equally shaped contiguous float64 device tensors; returns a new pair."""
    import torch

    code = {"multiply": 0, "divide": 1, "add": 2}[op]
    (m1, i1), (m2, i2) = stft1, stft2
    assert m1.shape == i1.shape == m2.shape == i2.shape and all(t.is_contiguous() for t in (m1, i1, m2, i2))
    mo, io = torch.empty_like(m1), torch.empty_like(i1)
    st = C.c_void_p(torch.cuda.current_stream(m1.device).cuda_stream) if m1.is_cuda else None
    if m1.numel():
        par._call("voc_binary_frames_dev", code, C.c_void_p(m1.data_ptr()), C.c_void_p(i1.data_ptr()),
                  C.c_void_p(m2.data_ptr()), C.c_void_p(i2.data_ptr()), C.c_void_p(mo.data_ptr()),
                  C.c_void_p(io.data_ptr()), m1.numel(), st)
    return mo, io


def process_filter_dev(par: VocoderParams, filt, x, length: int | None = None, chunk: int | None = None,
                       ana_phase=None, syn_phase=None):
    """`processWith` with a frequency-domain filter (Conduit.hs:46-47): framing .| analysis .|
    volumeFix .| filter .| synthesis .| overlap-add, all in device memory.

    filt: a `vocoder_amd.filter.BoundFilter` (or None for idFilter).  x (C, >= length) is a torch
    float64 tensor (float32 on a single-precision context) on the GPU (or in host memory with the host build of the kernels used by the
    CPU tests); phases (C, K) are updated in place when given.  Returns (y, ana_phase, syn_phase)
    with y (C, ola_length(frames after the filter))."""
    import torch

    f32 = getattr(par, "precision", "f64") == "f32"
    dt = torch.float32 if f32 else torch.float64
    sfx = "_f32_dev" if f32 else "_dev"
    assert x.dtype == dt and x.stride(-1) == 1
    c = x.shape[0]
    length = x.shape[1] if length is None else length
    chunk = par.H if chunk is None else chunk
    dev = x.device
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream) if x.is_cuda else None
    aph = torch.zeros((c, par.K), dtype=dt, device=dev) if ana_phase is None else ana_phase
    sph = torch.zeros((c, par.K), dtype=dt, device=dev) if syn_phase is None else syn_phase
    f = num_frames(par, length)
    ldk = (par.K + 1) & ~1
    mag = torch.empty((c, f, ldk), dtype=dt, device=dev)
    inc = torch.empty_like(mag)
    if f:
        par._call("voc_analysis_signal" + sfx, C.c_void_p(x.data_ptr()), x.stride(0), f, c, C.c_void_p(aph.data_ptr()),
                  C.c_void_p(mag.data_ptr()), C.c_void_p(inc.data_ptr()), ldk, 1, st)
    mrows = irows = None
    fo = f
    if (filt is not None and not f32 and f and len(filt.parts) == 1 and filt.parts[0][0] == "chain"
            and not os.environ.get("VOC_NO_FILTERED_SYNTHESIS")):
        # one chain of per-frame filters and nothing else: filter .| synthesis in one call (the library folds a
        # `pitch shift >> gain` chain into synthesis' row loads and applies any other chain first)
        n = ola_length(par, f, chunk)
        y = torch.empty((c, n), dtype=dt, device=dev)
        par.lib.check(par.lib.dll.voc_synthesis_filtered_dev(filt.parts[0][1]._h, C.c_void_p(mag.data_ptr()), C.c_void_p(inc.data_ptr()),
                                                             ldk, f, c, C.c_void_p(sph.data_ptr()), C.c_void_p(y.data_ptr()),
                                                             y.stride(0), n, st), par.handle)
        return y, aph, sph
    if filt is not None:
        mag, mrows, inc, irows, fo = filt.apply_dev_lazy(mag, inc)
    n = ola_length(par, fo, chunk)
    y = torch.empty((c, n), dtype=dt, device=dev)
    if fo and mrows is None and irows is None:
        par._call("voc_synthesis_signal" + sfx, C.c_void_p(mag.data_ptr()), C.c_void_p(inc.data_ptr()), ldk, fo, c,
                  C.c_void_p(sph.data_ptr()), C.c_void_p(y.data_ptr()), y.stride(0), n, st)
    elif fo:   # play_speed: synthesis reads the rows through the maps, nothing is copied
        par._call("voc_synthesis_mapped" + sfx, C.c_void_p(mag.data_ptr()), mag.shape[1],
                  C.c_void_p(mrows.data_ptr()) if mrows is not None else None,
                  C.c_void_p(inc.data_ptr()), inc.shape[1], C.c_void_p(irows.data_ptr()) if irows is not None else None,
                  ldk, fo, c, C.c_void_p(sph.data_ptr()), C.c_void_p(y.data_ptr()), y.stride(0), n, st)
    return y, aph, sph


def deinterleave_dev(par: VocoderParams, inter, channels: int):
    """Interleaved samples (n*channels,) -> planar (channels, n), device tensors (Audio.hs:51-55 splits per frame;
    here once per buffer)."""
    import torch
    n = inter.numel() // channels
    out = torch.empty((channels, n), dtype=torch.float64, device=inter.device)
    st = C.c_void_p(torch.cuda.current_stream(inter.device).cuda_stream) if inter.is_cuda else None
    par._call("voc_deinterleave_dev", C.c_void_p(inter.data_ptr()), n, channels, C.c_void_p(out.data_ptr()), n, st)
    return out


def interleave_dev(par: VocoderParams, planar):
    """Planar (channels, n) -> interleaved (n*channels,), device tensors."""
    import torch
    c, n = planar.shape
    out = torch.empty(n * c, dtype=torch.float64, device=planar.device)
    st = C.c_void_p(torch.cuda.current_stream(planar.device).cuda_stream) if planar.is_cuda else None
    par._call("voc_interleave_dev", C.c_void_p(planar.data_ptr()), planar.stride(0), n, c, C.c_void_p(out.data_ptr()), st)
    return out


def pcm16_to_planar_dev(par: VocoderParams, inter, channels: int):
    """Interleaved int16 samples (n*channels,) -> planar float64 (channels, n) the way the reference's
    sourceSnd hands them over (libsndfile: s / 32768; VocoderFile.hs:172)."""
    import torch
    assert inter.dtype == torch.int16 and inter.is_contiguous()
    n = inter.numel() // channels
    out = torch.empty((channels, n), dtype=torch.float64, device=inter.device)
    st = C.c_void_p(torch.cuda.current_stream(inter.device).cuda_stream) if inter.is_cuda else None
    par._call("voc_pcm16_to_planar_dev", C.c_void_p(inter.data_ptr()), n, channels, C.c_void_p(out.data_ptr()), n, st)
    return out


def planar_to_pcm16_dev(par: VocoderParams, planar, clip: bool = False):
    """Planar float64 (channels, n) -> interleaved int16 (n*channels,) the way sinkSnd writes PCM16
    (VocoderFile.hs:158,173; libsndfile: lrint(x * 32767), or with clip lrint(x * 32768) saturated)."""
    import torch
    assert planar.dtype == torch.float64 and planar.stride(-1) == 1
    c, n = planar.shape
    out = torch.empty(n * c, dtype=torch.int16, device=planar.device)
    st = C.c_void_p(torch.cuda.current_stream(planar.device).cuda_stream) if planar.is_cuda else None
    par._call("voc_planar_to_pcm16_dev", C.c_void_p(planar.data_ptr()), planar.stride(0), n, c, 1 if clip else 0,
              C.c_void_p(out.data_ptr()), st)
    return out
