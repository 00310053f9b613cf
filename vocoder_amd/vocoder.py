"""Host-side mirror of the reference's `Vocoder` module (vocoder/src/Vocoder.hs:12-42) over the
C ABI.  Names follow the Haskell exports in snake_case; arguments keep their meaning.  Every
function here is a thin marshalling layer: the arithmetic runs in the HIP library.

Shapes: a `Frame` is (W,), a `Phase` / `Moduli` / `PhaseInc` is (K,), an `FFTOutput` is (K,)
complex128.  Wherever the reference takes one of these, a leading batch axis is accepted too;
stage functions additionally take a leading channel axis (C independent folds, as
`analysisF` / `synthesisF` do, vocoder-conduit/src/Vocoder/Conduit.hs:63-68).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._ffi import Library, VocoderError, default_library

__all__ = [
    "VocoderParams", "vocoder_params", "voc_frame_length", "voc_input_frame_length", "voc_freq_frame_length",
    "voc_hop_size", "voc_window", "do_fft", "do_ifft", "analysis_block", "analysis_step", "analysis_stage",
    "synthesis_block", "synthesis_step", "synthesis_stage", "zero_phase", "volume_coeff", "add_frames",
    "fft_execute_r2c", "fft_execute_c2r", "frame_from_complex", "frame_to_complex", "VocoderError",
]


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class VocoderParams:
    """`VocoderParams` (Vocoder.hs:89-99): the two FFT plans, hop size and window, held by the
    library as one opaque context."""

    def __init__(self, frame_length: int, hop_size: int, window, device: int = 0, lib: Library | None = None,
                 precision: str = "f64"):
        """precision "f32" makes a context of the single-precision variant: the stage functions here and the
        round trips of `conduit` (`process_with`, `process_dev`, `process_filter_dev` without a filter) run on it."""
        self.lib = lib or default_library()
        self.precision = precision
        if precision not in ("f64", "f32"):
            raise ValueError("precision must be 'f64' or 'f32'")
        window = _f64(window).ravel()
        h = C.c_void_p()
        st = self.lib.dll.voc_ctx_create(int(frame_length), int(hop_size), _p(window), int(window.size), int(device),
                                         1 if precision == "f32" else 0, C.byref(h))
        if st == -1 and window.size > frame_length:
            # addZeroPadding's `error` (Vocoder.hs:132)
            raise VocoderError(st, f"addZeroPadding: input is {window.size - frame_length} samples longer than target length")
        self.lib.check(st)
        self._h = h
        self.N = self.lib.dll.voc_frame_length(h)
        self.W = self.lib.dll.voc_input_frame_length(h)
        self.K = self.lib.dll.voc_freq_frame_length(h)
        self.H = self.lib.dll.voc_hop_size(h)
        self.device = device

    @property
    def handle(self):
        return self._h

    def close(self):
        if getattr(self, "_h", None):
            self.lib.dll.voc_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _call(self, name, *args):
        self.lib.check(getattr(self.lib.dll, name)(self._h, *args), self._h)


def vocoder_params(length: int, hop_size: int, window, **kw) -> VocoderParams:
    """vocoderParams len hs wnd (Vocoder.hs:114-115)."""
    return VocoderParams(length, hop_size, window, **kw)


def voc_frame_length(par):  # Vocoder.hs:106-107
    return par.N


def voc_input_frame_length(par):  # Vocoder.hs:110-111
    return par.W


def voc_freq_frame_length(par):  # Vocoder.hs:102-103
    return par.K


def voc_hop_size(par):  # Vocoder.hs:95
    return par.H


def voc_window(par) -> np.ndarray:  # Vocoder.hs:97
    out = np.empty(par.W)
    par._call("voc_window", _p(out))
    return out


def volume_coeff(par) -> float:
    """volumeCoeff (Vocoder.hs:209-210)."""
    return par.lib.dll.voc_volume_coeff(par.handle)


def zero_phase(par, channels: int | None = None) -> np.ndarray:
    """zeroPhase (Vocoder.hs:204-205)."""
    return np.zeros(par.K if channels is None else (channels, par.K))


def _rc(par):
    """(real dtype, complex dtype, entry-point suffix) of a context: single-precision contexts take float arrays
    through the `_f32` forms of the same entry points."""
    return (np.float32, np.complex64, "_f32") if getattr(par, "precision", "f64") == "f32" else (np.float64, np.complex128, "")


def _batched(a, inner, dtype=np.float64):
    a = np.ascontiguousarray(a, dtype=dtype)
    if a.shape[-1] != inner:
        # vector-fftw's `execute` raises on a size mismatch
        raise VocoderError(-5, f"size mismatch: expected last axis {inner}, got {a.shape[-1]}")
    lead = a.shape[:-1]
    return a.reshape(-1, inner), lead


def fft_execute_r2c(par, x) -> np.ndarray:
    """`FFT.execute (plan dftR2C N)`: N reals -> K complex, unnormalised (Vocoder.hs:145)."""
    rt, ct, sfx = _rc(par)
    x, lead = _batched(x, par.N, rt)
    out = np.empty((x.shape[0], par.K), dtype=ct)
    par._call("voc_r2c" + sfx, _p(x), _p(out), x.shape[0])
    return out.reshape(lead + (par.K,))


def fft_execute_c2r(par, spec) -> np.ndarray:
    """`FFT.execute (plan dftC2R N)`: K complex -> N reals, scaled by 1/N (Vocoder.hs:196)."""
    rt, ct, sfx = _rc(par)
    spec, lead = _batched(spec, par.K, ct)
    out = np.empty((spec.shape[0], par.N), dtype=rt)
    par._call("voc_c2r" + sfx, _p(spec), _p(out), spec.shape[0])
    return out.reshape(lead + (par.N,))


def do_fft(par, frame) -> np.ndarray:
    """doFFT (Vocoder.hs:143-145)."""
    rt, ct, sfx = _rc(par)
    frame, lead = _batched(frame, par.W, rt)
    out = np.empty((frame.shape[0], par.K), dtype=ct)
    par._call("voc_do_fft" + sfx, _p(frame), _p(out), frame.shape[0])
    return out.reshape(lead + (par.K,))


def do_ifft(par, spec) -> np.ndarray:
    """doIFFT (Vocoder.hs:194-196)."""
    rt, ct, sfx = _rc(par)
    spec, lead = _batched(spec, par.K, ct)
    out = np.empty((spec.shape[0], par.W), dtype=rt)
    par._call("voc_do_ifft" + sfx, _p(spec), _p(out), spec.shape[0])
    return out.reshape(lead + (par.W,))


def _free_call(lib, name, *args):
    lib = lib or default_library()
    lib.check(getattr(lib.dll, name)(*args))


def analysis_step(par_or_hop, *args, lib: Library | None = None):
    """analysisStep h eN prev_ph vec -> (ph, (mag, ph_inc)) (Vocoder.hs:158-163).

    Either the reference's own argument list, `analysis_step(h, eN, prev_ph, vec)`, or
    `analysis_step(par, prev_ph, vec)` with h and eN taken from `par`."""
    if isinstance(par_or_hop, VocoderParams):
        par = par_or_hop
        prev_ph, spec = args
        rt, ct, sfx = _rc(par)
        spec, lead = _batched(spec, par.K, ct)
        prev_ph, _ = _batched(prev_ph, par.K, rt)
        b, k = spec.shape[0], par.K
        ph, mag, inc = np.empty((b, k), dtype=rt), np.empty((b, k), dtype=rt), np.empty((b, k), dtype=rt)
        par._call("voc_analysis_step" + sfx, _p(prev_ph), _p(spec), _p(ph), _p(mag), _p(inc), b)
    else:
        hop, n, prev_ph, spec = (par_or_hop,) + args
        spec = np.ascontiguousarray(spec, dtype=np.complex128)
        k = spec.shape[-1]
        spec, lead = _batched(spec, k, np.complex128)
        prev_ph, _ = _batched(prev_ph, k)
        b = spec.shape[0]
        ph, mag, inc = np.empty((b, k)), np.empty((b, k)), np.empty((b, k))
        _free_call(lib, "voc_analysis_step_hn", int(hop), int(n), k, _p(prev_ph), _p(spec), _p(ph), _p(mag), _p(inc), b)
    sh = lead + (k,)
    return ph.reshape(sh), (mag.reshape(sh), inc.reshape(sh))


def synthesis_step(par_or_hop, ph, stft, lib: Library | None = None):
    """synthesisStep hop ph (mag, ph_inc) -> (new_ph, spectrum) (Vocoder.hs:186-190).  The first
    argument is the hop size, as in the reference, or a `VocoderParams` whose hop size is used."""
    mag, inc = stft
    rt, ct, sfx = _rc(par_or_hop) if isinstance(par_or_hop, VocoderParams) else (np.float64, np.complex128, "")
    mag = np.ascontiguousarray(mag, dtype=rt)
    k = par_or_hop.K if isinstance(par_or_hop, VocoderParams) else mag.shape[-1]
    mag, lead = _batched(mag, k, rt)
    inc, _ = _batched(inc, k, rt)
    ph, _ = _batched(ph, k, rt)
    b = mag.shape[0]
    out, spec = np.empty((b, k), dtype=rt), np.empty((b, k), dtype=ct)
    if isinstance(par_or_hop, VocoderParams):
        par_or_hop._call("voc_synthesis_step" + sfx, _p(ph), _p(mag), _p(inc), _p(out), _p(spec), b)
    else:
        _free_call(lib, "voc_synthesis_step_h", int(par_or_hop), k, _p(ph), _p(mag), _p(inc), _p(out), _p(spec), b)
    sh = lead + (k,)
    return out.reshape(sh), spec.reshape(sh)


def frame_from_complex(spec, lib: Library | None = None):
    """frameFromComplex (Vocoder.hs:217-218): complex spectrum -> (mag, phase)."""
    spec = np.ascontiguousarray(spec, dtype=np.complex128)
    mag, ph = np.empty(spec.shape), np.empty(spec.shape)
    _free_call(lib, "voc_frame_from_complex", _p(spec), _p(mag), _p(ph), spec.size)
    return mag, ph


def frame_to_complex(stft, lib: Library | None = None):
    """frameToComplex (Vocoder.hs:213-214): (mag, phase) -> complex spectrum."""
    mag, ph = _f64(stft[0]), _f64(stft[1])
    if mag.shape != ph.shape:
        raise VocoderError(-5, "size mismatch between moduli and phases")
    spec = np.empty(mag.shape, dtype=np.complex128)
    _free_call(lib, "voc_frame_to_complex", _p(mag), _p(ph), _p(spec), mag.size)
    return spec


def add_frames(*args, lib: Library | None = None):
    """addFrames f1 f2 (Vocoder.hs:220-222); `add_frames(par, f1, f2)` is accepted too."""
    if len(args) == 3:
        par, f1, f2 = args
        rt, _, sfx = _rc(par)
        m1, lead = _batched(f1[0], par.K, rt)
        i1, _ = _batched(f1[1], par.K, rt)
        m2, _ = _batched(f2[0], par.K, rt)
        i2, _ = _batched(f2[1], par.K, rt)
        b = m1.shape[0]
        mo, io = np.empty((b, par.K), dtype=rt), np.empty((b, par.K), dtype=rt)
        par._call("voc_add_frames" + sfx, _p(m1), _p(i1), _p(m2), _p(i2), _p(mo), _p(io), b)
        sh = lead + (par.K,)
        return mo.reshape(sh), io.reshape(sh)
    f1, f2 = args
    arrs = [_f64(a) for a in (f1[0], f1[1], f2[0], f2[1])]
    if any(a.shape != arrs[0].shape for a in arrs):
        raise VocoderError(-5, "size mismatch between the frames")
    mo, io = np.empty(arrs[0].shape), np.empty(arrs[0].shape)
    _free_call(lib, "voc_add_frames_n", *[_p(a) for a in arrs], _p(mo), _p(io), arrs[0].size)
    return mo, io


def _stage_shapes(phase, data, inner, dtype=np.float64):
    data = np.ascontiguousarray(data, dtype=dtype)
    if data.shape[-1] != inner:
        raise VocoderError(-5, f"size mismatch: expected last axis {inner}, got {data.shape[-1]}")
    multi = data.ndim == 3
    if not multi:
        data = data[None]
    c, f = data.shape[0], data.shape[1]
    phase = np.array(phase, dtype=dtype, copy=True).reshape(c, -1)
    return phase, data, c, f, multi


def _dtype_of(par):
    return np.float32 if par.precision == "f32" else np.float64


def analysis_stage(par, ph, frames):
    """analysisStage par ph frames -> (ph', [(mag, ph_inc)]) (Vocoder.hs:149-150).
    frames: (F, W) or (C, F, W); returns (phase, (mag, ph_inc)) with mag / ph_inc (.., F, K)."""
    dt = _dtype_of(par)
    phase, frames, c, f, multi = _stage_shapes(ph, frames, par.W, dt)
    mag, inc = np.empty((c, f, par.K), dtype=dt), np.empty((c, f, par.K), dtype=dt)
    par._call("voc_analysis_stage_f32" if dt is np.float32 else "voc_analysis_stage", _p(phase), _p(frames), _p(mag), _p(inc), c, f)
    if not multi:
        return phase[0], (mag[0], inc[0])
    return phase, (mag, inc)


def synthesis_stage(par, ph, stft):
    """synthesisStage par ph stft -> (ph', frames) (Vocoder.hs:177-178)."""
    mag, inc = stft
    dt = _dtype_of(par)
    phase, mag, c, f, multi = _stage_shapes(ph, mag, par.K, dt)
    inc = np.ascontiguousarray(inc, dtype=dt).reshape(c, f, par.K)
    frames = np.empty((c, f, par.W), dtype=dt)
    par._call("voc_synthesis_stage_f32" if dt is np.float32 else "voc_synthesis_stage", _p(phase), _p(mag), _p(inc), _p(frames), c, f)
    if not multi:
        return phase[0], frames[0]
    return phase, frames


def analysis_block(par, prev_ph, frame):
    """analysisBlock par prev_ph vec -> (ph, (mag, ph_inc)) (Vocoder.hs:153-154)."""
    ph, (mag, inc) = analysis_stage(par, prev_ph, np.asarray(frame)[None, :])
    return ph, (mag[0], inc[0])


def synthesis_block(par, ph, stft):
    """synthesisBlock par ph (mag, ph_inc) -> (ph', frame) (Vocoder.hs:181-182)."""
    mag, inc = stft
    ph2, frames = synthesis_stage(par, ph, (np.asarray(mag)[None, :], np.asarray(inc)[None, :]))
    return ph2, frames[0]
