"""ctypes front-end of the CPU oracle (``libvocoder_oracle.so``).

TEST INFRASTRUCTURE ONLY: this package is the checker the HIP path is compared
with.  Nothing under ``vocoder_amd/`` imports it; only ``tests/``, the smoke
check and the benchmark's CPU-baseline leg do.

Pinning status: framing / overlap-add are pinned by the reference's list models
(vocoder-conduit/test/main.hs:18-28, vocoder-dunai/test/main.hs:28-38); the FFT
and phase arithmetic are PARITY UNPINNED (the reference holds no numeric test
vector for them) and are cross-checked against numpy.fft and analytic answers.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libvocoder_oracle.so")

WINDOW_KINDS = {
    "box": 0, "triangle": 1, "hamming": 2, "hann": 3, "generalized_blackman": 4,
    "blackman": 5, "exact_blackman": 6, "lanczos": 7, "flattop": 8,
}


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (see oracle/Makefile)."""
    src = [os.path.join(_HERE, f) for f in ("vocoder_oracle.c", "frames_generic.inc", "filters_oracle.c", "Makefile")]
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in src)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "libvocoder_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int64)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_wrap.restype = C.c_double
        _lib.orc_wrap.argtypes = [C.c_double]
        _lib.orc_calc_phase_inc.restype = C.c_double
        _lib.orc_calc_phase_inc.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double]
        _lib.orc_volume_coeff.restype = C.c_double
        _lib.orc_magnitude.restype = C.c_double
        _lib.orc_magnitude.argtypes = [C.c_double, C.c_double]
        _lib.orc_phase.restype = C.c_double
        _lib.orc_phase.argtypes = [C.c_double, C.c_double]
        _lib.orc_make_window.argtypes = [C.c_int, C.c_double, C.c_int, _dp]
        _lib.orc_num_frames.restype = C.c_int64
        _lib.orc_filter_linear.argtypes = [C.c_int, _dp, C.c_double, C.c_int, _dp]
        _lib.orc_interpolate.argtypes = [C.c_double, _dp, C.c_int, _dp]
        _lib.orc_filter_pitch_shift.argtypes = [C.c_double, C.c_int, _dp, _dp]
        _lib.orc_convolution.argtypes = [_dp, C.c_int, _dp, C.c_int, _dp]
        _lib.orc_filter_convolution.argtypes = [_dp, C.c_int, C.c_int, _dp, _dp]
        _lib.orc_envelope_kernel.argtypes = [C.c_int, _dp]
        _lib.orc_envelope.argtypes = [C.c_int, _dp, C.c_int, _dp]
        _lib.orc_filter_envelope.argtypes = [C.c_int, C.c_int, _dp, _dp]
        _lib.orc_play_speed.restype = C.c_int64
        _lib.orc_play_speed.argtypes = [C.c_int64, C.c_int64, C.c_int64, _ip, C.c_int64]
        _lib.orc_num_frames.argtypes = [C.c_int64, C.c_int, C.c_int]
        _lib.orc_ola_length.restype = C.c_int64
        _lib.orc_ola_length.argtypes = [C.c_int64, C.c_int, C.c_int, C.c_int64]
        for suf in ("f64", "i64"):
            for name in ("orc_gen_frames_of_e", "orc_sum_frames_e", "orc_frames_of_s"):
                getattr(_lib, f"{name}_{suf}").restype = C.c_int64
    return _lib


def _d(a):
    return a.ctypes.data_as(_dp)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


# --------------------------------------------------------------------------- windows
def make_window(kind: str, n: int, alpha: float = 0.16) -> np.ndarray:
    out = np.empty(n, dtype=np.float64)
    lib().orc_make_window(WINDOW_KINDS[kind], float(alpha), int(n), _d(out))
    return out


# --------------------------------------------------------------------------- FFT
def fft_r2c(x) -> np.ndarray:
    x = _f64(x)
    n = x.shape[-1]
    out = np.empty(2 * (n // 2 + 1), dtype=np.float64)
    lib().orc_fft_r2c(C.c_int(n), _d(x), _d(out))
    return out.view(np.complex128)


def dft_r2c_ld(x) -> np.ndarray:
    x = _f64(x)
    n = x.shape[-1]
    out = np.empty(2 * (n // 2 + 1), dtype=np.float64)
    lib().orc_dft_r2c_ld(C.c_int(n), _d(x), _d(out))
    return out.view(np.complex128)


def fft_c2r(spec, n: int) -> np.ndarray:
    spec = np.ascontiguousarray(spec, dtype=np.complex128)
    assert spec.shape[-1] == n // 2 + 1
    out = np.empty(n, dtype=np.float64)
    lib().orc_fft_c2r(C.c_int(n), _d(spec.view(np.float64)), _d(out))
    return out


# --------------------------------------------------------------------------- Vocoder.hs
def wrap(e: float) -> float:
    return lib().orc_wrap(float(e))


def calc_phase_inc(n: int, hop: int, k: int, ph_diff: float) -> float:
    return lib().orc_calc_phase_inc(n, hop, k, float(ph_diff))


def magnitude(re: float, im: float) -> float:
    return lib().orc_magnitude(float(re), float(im))


def phase(re: float, im: float) -> float:
    return lib().orc_phase(float(re), float(im))


def volume_coeff(hop: int, window) -> float:
    window = _f64(window)
    return lib().orc_volume_coeff(C.c_int(hop), C.c_int(window.size), _d(window))


def do_fft(n: int, window, frame) -> np.ndarray:
    window, frame = _f64(window), _f64(frame)
    out = np.empty(2 * (n // 2 + 1), dtype=np.float64)
    rc = lib().orc_do_fft(C.c_int(n), C.c_int(window.size), _d(window), _d(frame), _d(out))
    if rc:
        raise ValueError("addZeroPadding: input is longer than target length")
    return out.view(np.complex128)


def do_ifft(n: int, window, spec) -> np.ndarray:
    window = _f64(window)
    spec = np.ascontiguousarray(spec, dtype=np.complex128)
    out = np.empty(window.size, dtype=np.float64)
    lib().orc_do_ifft(C.c_int(n), C.c_int(window.size), _d(window), _d(spec.view(np.float64)), _d(out))
    return out


def analysis_step(hop: int, n: int, prev_ph, spec):
    prev_ph = _f64(prev_ph)
    spec = np.ascontiguousarray(spec, dtype=np.complex128)
    k = spec.size
    ph, mag, inc = (np.empty(k) for _ in range(3))
    lib().orc_analysis_step(C.c_int(hop), C.c_int(n), C.c_int(k), _d(prev_ph), _d(spec.view(np.float64)),
                            _d(ph), _d(mag), _d(inc))
    return ph, mag, inc


def synthesis_step(hop: int, ph, mag, ph_inc):
    ph, mag, ph_inc = _f64(ph), _f64(mag), _f64(ph_inc)
    k = ph.size
    ph_out = np.empty(k)
    spec = np.empty(2 * k)
    lib().orc_synthesis_step(C.c_int(hop), C.c_int(k), _d(ph), _d(mag), _d(ph_inc), _d(ph_out), _d(spec))
    return ph_out, spec.view(np.complex128)


def analysis_stage(n: int, hop: int, window, phase, frames):
    """frames: (F, W).  Returns (final phase (K,), mag (F,K), ph_inc (F,K))."""
    window, frames = _f64(window), _f64(frames)
    f, w = frames.shape
    assert w == window.size
    k = n // 2 + 1
    ph = _f64(phase).copy()
    mag = np.empty((f, k))
    inc = np.empty((f, k))
    rc = lib().orc_analysis_stage(C.c_int(n), C.c_int(w), C.c_int(hop), _d(window), _d(ph), _d(frames),
                                  C.c_int64(f), _d(mag), _d(inc))
    if rc:
        raise ValueError("addZeroPadding: input is longer than target length")
    return ph, mag, inc


def synthesis_stage(n: int, hop: int, window, phase, mag, ph_inc):
    """mag, ph_inc: (F, K).  Returns (final phase (K,), frames (F, W))."""
    window, mag, ph_inc = _f64(window), _f64(mag), _f64(ph_inc)
    f, k = mag.shape
    assert k == n // 2 + 1
    ph = _f64(phase).copy()
    frames = np.empty((f, window.size))
    lib().orc_synthesis_stage(C.c_int(n), C.c_int(window.size), C.c_int(hop), _d(window), _d(ph), _d(mag),
                              _d(ph_inc), C.c_int64(f), _d(frames))
    return ph, frames


def frame_from_complex(spec):
    """frameFromComplex (Vocoder.hs:217-218) -> (mag, phase)."""
    spec = np.ascontiguousarray(spec, dtype=np.complex128).ravel()
    mag, ph = np.empty(spec.size), np.empty(spec.size)
    lib().orc_frame_from_complex(C.c_int(spec.size), _d(spec.view(np.float64)), _d(mag), _d(ph))
    return mag, ph


def frame_to_complex(mag, ph):
    """frameToComplex (Vocoder.hs:213-214)."""
    mag, ph = _f64(mag).ravel(), _f64(ph).ravel()
    spec = np.empty(2 * mag.size)
    lib().orc_frame_to_complex(C.c_int(mag.size), _d(mag), _d(ph), _d(spec))
    return spec.view(np.complex128)


def add_frames(mag1, inc1, mag2, inc2):
    mag1, inc1, mag2, inc2 = map(_f64, (mag1, inc1, mag2, inc2))
    k = mag1.size
    mo, io = np.empty(k), np.empty(k)
    lib().orc_add_frames(C.c_int(k), _d(mag1), _d(inc1), _d(mag2), _d(inc2), _d(mo), _d(io))
    return mo, io


def binary_frames(op: str, mag1, inc1, mag2, inc2):
    """mkBinary op1 op2 (vocoder-dunai/example/VocoderJack.hs:107-110) as vocoder-jack instantiates it:
    `--multiply` = mkBinary (*) (+) (:127-129), `--divide` = mkBinary (/) (-) (:130-132); zipWith over
    the moduli and over the phase increments, one IEEE operation per element."""
    m1, i1, m2, i2 = (_f64(a) for a in (mag1, inc1, mag2, inc2))
    if op == "multiply":
        return m1 * m2, i1 + i2
    if op == "divide":
        with np.errstate(divide="ignore", invalid="ignore"):
            return m1 / m2, i1 - i2
    raise ValueError(op)


def num_frames(length: int, w: int, hop: int) -> int:
    return lib().orc_num_frames(int(length), int(w), int(hop))


def ola_length(nframes: int, w: int, hop: int, chunk: int) -> int:
    return lib().orc_ola_length(int(nframes), int(w), int(hop), int(chunk))


def roundtrip(n: int, hop: int, window, x, chunk: int | None = None, ana_phase=None, syn_phase=None,
              want_stft: bool = False):
    """Vocoder.Conduit.process with the identity filter on one channel.

    Returns dict(y, ana_phase, syn_phase, leftover[, mag, ph_inc])."""
    window, x = _f64(window), _f64(x)
    w, k = window.size, n // 2 + 1
    chunk = hop if chunk is None else chunk
    f = num_frames(x.size, w, hop)
    y = np.empty(ola_length(f, w, hop, chunk))
    aph = np.zeros(k) if ana_phase is None else _f64(ana_phase).copy()
    sph = np.zeros(k) if syn_phase is None else _f64(syn_phase).copy()
    left = np.empty(x.size)
    llen = C.c_int64(0)
    mag = np.empty((f, k)) if want_stft else None
    inc = np.empty((f, k)) if want_stft else None
    rc = lib().orc_roundtrip(C.c_int(n), C.c_int(w), C.c_int(hop), _d(window), _d(x), C.c_int64(x.size),
                             C.c_int64(chunk), _d(aph), _d(sph), _d(y), _d(left), C.byref(llen),
                             _d(mag) if want_stft else None, _d(inc) if want_stft else None)
    if rc:
        raise ValueError("addZeroPadding: input is longer than target length")
    out = dict(y=y, ana_phase=aph, syn_phase=sph, leftover=left[: llen.value].copy(), nframes=f)
    if want_stft:
        out.update(mag=mag, ph_inc=inc)
    return out


# --------------------------------------------------------------------------- framing / OLA
def _typed(dtype):
    if np.dtype(dtype) == np.int64:
        return "i64", np.int64, C.POINTER(C.c_int64)
    return "f64", np.float64, _dp


def gen_frames_of_e(chunk_size: int, hop: int, chunks, q0=(), dtype=np.float64):
    """Returns (frames (F, chunk_size), leftover)."""
    suf, npdt, ptr = _typed(dtype)
    chunks = [np.ascontiguousarray(c, dtype=npdt).ravel() for c in chunks]
    data = np.concatenate(chunks) if chunks else np.empty(0, dtype=npdt)
    lens = np.array([c.size for c in chunks], dtype=np.int64)
    q0 = np.ascontiguousarray(q0, dtype=npdt).ravel()
    total = data.size + q0.size
    maxf = max(0, (total - chunk_size) // hop + 1) + len(chunks) + 1
    frames = np.empty((maxf, chunk_size), dtype=npdt)
    left = np.empty(total + 1, dtype=npdt)
    llen = C.c_int64(0)
    fn = getattr(lib(), f"orc_gen_frames_of_e_{suf}")
    nf = fn(C.c_int64(chunk_size), C.c_int64(hop), q0.ctypes.data_as(ptr), C.c_int64(q0.size),
            data.ctypes.data_as(ptr), lens.ctypes.data_as(_ip), C.c_int64(len(chunks)),
            frames.ctypes.data_as(ptr), C.c_int64(maxf), left.ctypes.data_as(ptr), C.byref(llen))
    assert nf <= maxf
    return frames[:nf].copy(), left[: llen.value].copy()


def sum_frames_e(chunk_size: int, hop: int, frames, dtype=np.float64):
    """frames: list of 1-D arrays (any lengths).  Returns chunks (nchunks, chunk_size)."""
    suf, npdt, ptr = _typed(dtype)
    frames = [np.ascontiguousarray(f, dtype=npdt).ravel() for f in frames]
    data = np.concatenate(frames) if frames else np.empty(0, dtype=npdt)
    lens = np.array([f.size for f in frames], dtype=np.int64)
    last = max([-1] + [k * hop + f.size - 1 for k, f in enumerate(frames)])
    maxc = last // chunk_size + 2 + len(frames)
    out = np.empty((maxc, chunk_size), dtype=npdt)
    fn = getattr(lib(), f"orc_sum_frames_e_{suf}")
    nc = fn(C.c_int64(chunk_size), C.c_int64(hop), data.ctypes.data_as(ptr), lens.ctypes.data_as(_ip),
            C.c_int64(len(frames)), out.ctypes.data_as(ptr), C.c_int64(maxc))
    assert nc <= maxc
    return out[:nc].copy()


def sum_frames_e_schedule(chunk_size: int, hop: int, nframes: int) -> np.ndarray:
    """published[f] = chunks sumFramesE has yielded once frame f has been awaited (Frames.hs:49-53)."""
    out = np.zeros(max(nframes, 0), dtype=np.int64)
    lib().orc_sum_frames_e_schedule(C.c_int64(chunk_size), C.c_int64(hop), C.c_int64(nframes), out.ctypes.data_as(_ip))
    return out


def frames_of_s(chunk_size: int, hop: int, ticks, dtype=np.float64):
    """ticks: list of 1-D arrays.  Returns list (per tick) of arrays (nframes_t, chunk_size)."""
    suf, npdt, ptr = _typed(dtype)
    ticks = [np.ascontiguousarray(t, dtype=npdt).ravel() for t in ticks]
    data = np.concatenate(ticks) if ticks else np.empty(0, dtype=npdt)
    lens = np.array([t.size for t in ticks], dtype=np.int64)
    maxf = int(sum(t.size // hop for t in ticks)) + 1
    frames = np.empty((maxf, chunk_size), dtype=npdt)
    per = np.zeros(max(1, len(ticks)), dtype=np.int64)
    fn = getattr(lib(), f"orc_frames_of_s_{suf}")
    nf = fn(C.c_int64(chunk_size), C.c_int64(hop), data.ctypes.data_as(ptr), lens.ctypes.data_as(_ip),
            C.c_int64(len(ticks)), frames.ctypes.data_as(ptr), C.c_int64(maxf), per.ctypes.data_as(_ip))
    assert nf >= 0, "framesOfS produced a short frame"
    out, pos = [], 0
    for t in range(len(ticks)):
        out.append(frames[pos: pos + per[t]].copy())
        pos += per[t]
    return out


def sum_frames_s(hop: int, ticks, out_lens, dtype=np.float64):
    """ticks: list (per tick) of 2-D arrays (nframes_t, frame_len).  Returns list of 1-D outputs."""
    suf, npdt, ptr = _typed(dtype)
    ticks = [np.ascontiguousarray(t, dtype=npdt) for t in ticks]
    flen = max([t.shape[1] for t in ticks if t.ndim == 2 and t.shape[0] > 0] + [1])
    ticks = [t.reshape(-1, flen) if t.size else np.empty((0, flen), dtype=npdt) for t in ticks]
    data = np.concatenate(ticks) if ticks else np.empty((0, flen), dtype=npdt)
    per = np.array([t.shape[0] for t in ticks], dtype=np.int64)
    olens = np.array(out_lens, dtype=np.int64)
    out = np.empty(int(olens.sum()) + 1, dtype=npdt)
    fn = getattr(lib(), f"orc_sum_frames_s_{suf}")
    fn(C.c_int64(hop), C.c_int64(flen), data.ctypes.data_as(ptr), per.ctypes.data_as(_ip),
       olens.ctypes.data_as(_ip), C.c_int64(len(ticks)), out.ctypes.data_as(ptr))
    res, pos = [], 0
    for n in olens:
        res.append(out[pos: pos + n].copy())
        pos += int(n)
    return res


# --------------------------------------------------------------------------- filters (Vocoder.Filter)
LINEAR_FILTERS = {
    "amplify": 0, "lowpass_brickwall": 1, "highpass_brickwall": 2, "bandpass_brickwall": 3,
    "bandstop_brickwall": 4, "lowpass_butterworth": 5, "highpass_butterworth": 6,
    "bandpass_butterworth": 7, "bandstop_butterworth": 8,
}


def apply_filter(step: float, chain, mag, ph_inc):
    """Runs a chain [(name, params...), ...] in composeFilters order (Filter.hs:54-55) on every
    frame of mag / ph_inc (.., K).  Names: the LINEAR_FILTERS, "gain_table" (g,), "pitch_shift"
    (n,), "convolution" (ker,), "envelope" (ksize,).  Returns new (mag, ph_inc)."""
    mag, inc = _f64(mag).copy(), _f64(ph_inc).copy()
    k = mag.shape[-1]
    m2, i2 = mag.reshape(-1, k), inc.reshape(-1, k)
    L = lib()
    for name, *prm in chain:
        for r in range(m2.shape[0]):
            mr, ir = m2[r], i2[r]
            if name in LINEAR_FILTERS:
                q = _f64(prm)
                L.orc_filter_linear(LINEAR_FILTERS[name], _d(q), float(step), k, _d(mr))
            elif name == "gain_table":
                mr *= _f64(prm[0])                      # linearAmplitudeFilter with a tabulated f
            elif name == "pitch_shift":
                L.orc_filter_pitch_shift(float(prm[0]), k, _d(mr), _d(ir))
            elif name == "convolution":
                ker = _f64(prm[0])
                L.orc_filter_convolution(_d(ker), ker.size, k, _d(mr), _d(ir))
            elif name == "envelope":
                L.orc_filter_envelope(int(prm[0]), k, _d(mr), _d(ir))
            else:
                raise ValueError(name)
    return mag, inc


def interpolate(n: float, v) -> np.ndarray:
    v = _f64(v)
    out = np.empty_like(v)
    lib().orc_interpolate(float(n), _d(v), v.size, _d(out))
    return out


def convolution(ker, mag) -> np.ndarray:
    ker, mag = _f64(ker), _f64(mag)
    out = np.empty_like(mag)
    lib().orc_convolution(_d(ker), ker.size, _d(mag), mag.size, _d(out))
    return out


def envelope_kernel(ksize: int) -> np.ndarray:
    out = np.empty(ksize)
    lib().orc_envelope_kernel(int(ksize), _d(out))
    return out


def envelope(ksize: int, mag) -> np.ndarray:
    mag = _f64(mag)
    out = np.empty_like(mag)
    lib().orc_envelope(int(ksize), _d(mag), mag.size, _d(out))
    return out


def play_speed(num: int, den: int, nframes_in: int) -> np.ndarray:
    """Input frame index repeated by each frame playSpeed (num/den) emits (Conduit/Filter.hs:122-136)."""
    cap = nframes_in * num // den + 2
    src = np.empty(cap, dtype=np.int64)
    n = lib().orc_play_speed(int(num), int(den), int(nframes_in), src.ctypes.data_as(_ip), cap)
    if n < 0:
        raise RuntimeError("play_speed: capacity")
    return src[:n].copy()
