"""Host-side mirror of `Vocoder.Filter` (vocoder/src/Vocoder/Filter.hs) and of `playSpeed`
(vocoder-conduit/src/Vocoder/Conduit/Filter.hs:122-136) over the C ABI.

A `Filter` here is a description — a tuple of stages in composeFilters order — that is bound to a
`VocoderParams` and a frequency step when it runs, like the reference's `Filter m = FreqStep ->
STFTFrame -> m STFTFrame`.  Per-frame filters become one device chain (`voc_filter_*`); `play_speed`
changes the number of frames and therefore splits a description into chains with frame gathers
between them.  The arithmetic runs in the HIP library on the STFT where it already rests in
device memory between analysis and synthesis.
"""
from __future__ import annotations

import ctypes as C
from fractions import Fraction

import numpy as np

from .vocoder import VocoderParams, _f64, _p

__all__ = [
    "Filter", "id_filter", "compose_filters", "add_filters", "amplify", "linear_amplitude_filter", "lowpass_brickwall",
    "highpass_brickwall", "bandpass_brickwall", "bandstop_brickwall", "lowpass_butterworth",
    "highpass_butterworth", "bandpass_butterworth", "bandstop_butterworth", "pitch_shift_interpolate",
    "convolution_filter", "envelope_filter", "random_phase_filter", "play_speed", "BoundFilter",
    "play_speed_frames", "play_speed_map",
]

_KINDS = {
    "amplify": 0, "lowpass_brickwall": 1, "highpass_brickwall": 2, "bandpass_brickwall": 3,
    "bandstop_brickwall": 4, "lowpass_butterworth": 5, "highpass_butterworth": 6, "bandpass_butterworth": 7,
    "bandstop_butterworth": 8, "gain_table": 9, "pitch_shift": 10, "convolution": 11, "envelope": 12,
    "random_phase": 13,
}


class Filter:
    """A filter description: stages (name, params) in application order."""

    def __init__(self, stages=()):
        self.stages = tuple(stages)

    def __rshift__(self, other: "Filter") -> "Filter":
        return compose_filters(self, other)

    def bind(self, par: VocoderParams, freq_step: float) -> "BoundFilter":
        return BoundFilter(par, freq_step, self)


def id_filter() -> Filter:
    """idFilter (Filter.hs:62-63)."""
    return Filter()


def compose_filters(f1: Filter, f2: Filter) -> Filter:
    """composeFilters f1 f2: f1 first, then f2 (Filter.hs:54-55)."""
    return Filter(f1.stages + f2.stages)


def add_filters(f1: Filter, f2: Filter) -> Filter:
    """addFilters f1 f2 (Filter.hs:58-59): both filters see the same frame; their outputs are
    added as complex numbers (addFrames, Vocoder.hs:220-222).  Both must produce the same number
    of frames."""
    return Filter((("add", (f1, f2)),))


def _one(name, *params):
    return Filter(((name, tuple(params)),))


def amplify(k: float) -> Filter:  # Filter.hs:80-81
    return _one("amplify", float(k))


def linear_amplitude_filter(fn) -> Filter:
    """linearAmplitudeFilter f (Filter.hs:76-77); f is evaluated on the host at step * k when bound."""
    return _one("linear_fn", fn)


def lowpass_brickwall(t):  # Filter.hs:84-85
    return _one("lowpass_brickwall", float(t))


def highpass_brickwall(t):  # Filter.hs:88-89
    return _one("highpass_brickwall", float(t))


def bandpass_brickwall(t, u):  # Filter.hs:92-93
    return _one("bandpass_brickwall", float(t), float(u))


def bandstop_brickwall(t, u):  # Filter.hs:96-97
    return _one("bandstop_brickwall", float(t), float(u))


def lowpass_butterworth(n, t):  # Filter.hs:103-104
    return _one("lowpass_butterworth", float(n), float(t))


def highpass_butterworth(n, t):  # Filter.hs:107-108
    return _one("highpass_butterworth", float(n), float(t))


def bandpass_butterworth(n, t, u):  # Filter.hs:111-112
    return _one("bandpass_butterworth", float(n), float(t), float(u))


def bandstop_butterworth(n, t, u):  # Filter.hs:115-116
    return _one("bandstop_butterworth", float(n), float(t), float(u))


def pitch_shift_interpolate(n):  # Filter.hs:128-129
    return _one("pitch_shift", float(n))


def convolution_filter(ker):  # Filter.hs:140-141
    return _one("convolution", *[float(v) for v in np.asarray(ker, dtype=np.float64).ravel()])


def envelope_filter(ksize: int):  # Filter.hs:151-152
    return _one("envelope", float(int(ksize)))


def random_phase_filter(seed: int = 0):
    """randomPhaseFilter (Filter.hs:156-157).  The reference uses the process-global generator;
    here the increments are a reproducible function of `seed`."""
    return _one("random_phase", float(int(seed)))


def play_speed(coeff) -> Filter:
    """playSpeed coeff (Conduit/Filter.hs:122-136); coeff is taken exactly (a float converts the
    way `toRational` does, vocoder-audio/example/VocoderFile.hs:119)."""
    return _one("play_speed", Fraction(coeff))


def play_speed_frames(par: VocoderParams, coeff, nframes_in: int) -> int:
    q = Fraction(coeff)
    return par.lib.dll.voc_play_speed_frames(q.numerator, q.denominator, int(nframes_in))


def play_speed_map(par: VocoderParams, coeff, nframes_in: int) -> np.ndarray:
    q = Fraction(coeff)
    n = play_speed_frames(par, q, nframes_in)
    src = np.empty(max(n, 1), dtype=np.int64)
    par.lib.check(par.lib.dll.voc_play_speed_map(q.numerator, q.denominator, int(nframes_in), _p(src), n))
    return src[:n]


class _Chain:
    """One device chain (voc_filter)."""

    def __init__(self, par: VocoderParams, step: float, stages):
        self.par = par
        h = C.c_void_p()
        par.lib.check(par.lib.dll.voc_filter_create(par.handle, float(step), C.byref(h)))
        self._h = h
        for name, params in stages:
            if name == "linear_fn":
                q = _f64([params[0](step * float(k)) for k in range(par.K)])
                kind = _KINDS["gain_table"]
            else:
                q = _f64(params)
                kind = _KINDS[name]
            par.lib.check(par.lib.dll.voc_filter_append(h, kind, _p(q), int(q.size)), par.handle)

    def close(self):
        if getattr(self, "_h", None):
            self.par.lib.dll.voc_filter_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BoundFilter:
    """A filter description bound to parameters and a frequency step: chains of per-frame
    operations separated by play_speed stages."""

    def __init__(self, par: VocoderParams, freq_step: float, filt: Filter):
        self.par, self.step = par, float(freq_step)
        self.parts = []            # ("chain", _Chain) | ("speed", Fraction)
        # True when the chain leaves every phase increment at zero: amplitudeFilter0 (Filter.hs:73, under
        # convolutionFilter and envelopeFilter) zeroes them, gains keep them, interpolating zeros gives zeros
        self.zero_inc = False
        run = []
        for st in filt.stages:
            if st[0] in ("convolution", "envelope"):
                self.zero_inc = True
            elif st[0] in ("random_phase", "add"):      # addFrames returns phases as increments
                self.zero_inc = False
            if st[0] == "add":
                if run:
                    self._add_chains(run)
                    run = []
                self.parts.append(("add", tuple(BoundFilter(par, freq_step, f) for f in st[1])))
            elif st[0] == "play_speed":
                if run:
                    self._add_chains(run)
                    run = []
                self.parts.append(("speed", st[1][0]))
            else:
                run.append(st)
        if run:
            self._add_chains(run)

    def _add_chains(self, stages, limit: int = 12):
        for i in range(0, len(stages), limit):
            self.parts.append(("chain", _Chain(self.par, self.step, stages[i:i + limit])))

    @property
    def changes_frame_count(self) -> bool:
        """True when a play_speed stage occurs anywhere in the description."""
        return any(kind == "speed" or (kind == "add" and any(b.changes_frame_count for b in v))
                   for kind, v in self.parts)

    def frames_out(self, nframes_in: int) -> int:
        f = nframes_in
        for kind, v in self.parts:
            if kind == "speed":
                f = play_speed_frames(self.par, v, f)
            elif kind == "add":
                fa, fb = (b.frames_out(f) for b in v)
                if fa != fb:
                    raise ValueError(f"add_filters: the two filters produce {fa} and {fb} frames")
                f = fa
        return f

    def apply(self, mag, ph_inc):
        """Host arrays (C, F, K) or (F, K): returns the filtered (mag, ph_inc) (new arrays)."""
        f32 = getattr(self.par, "precision", "f64") == "f32"
        dt = np.float32 if f32 else np.float64
        mag, inc = np.array(mag, dtype=dt, order="C"), np.array(ph_inc, dtype=dt, order="C")
        multi = mag.ndim == 3
        m3, i3 = (mag, inc) if multi else (mag[None], inc[None])
        for kind, v in self.parts:
            if kind == "chain":
                c, f, _ = m3.shape
                fn = self.par.lib.dll.voc_filter_apply_f32 if f32 else self.par.lib.dll.voc_filter_apply
                self.par.lib.check(fn(v._h, _p(m3), _p(i3), f, c), self.par.handle)
            elif kind == "add":
                if f32:
                    raise ValueError("add_filters runs in double precision only")
                self.frames_out(m3.shape[1])
                (ma, ia), (mb, ib) = (b.apply(m3, i3) for b in v)
                m3, i3 = np.empty_like(ma), np.empty_like(ia)
                self.par._call("voc_add_frames", _p(ma), _p(ia), _p(mb), _p(ib), _p(m3), _p(i3),
                               ma.shape[0] * ma.shape[1])
            else:
                src = play_speed_map(self.par, v, m3.shape[1])
                m3, i3 = np.ascontiguousarray(m3[:, src]), np.ascontiguousarray(i3[:, src])
        return (m3, i3) if multi else (m3[0], i3[0])

    def _run_chain(self, chain, mag, ph_inc, ldk, f, c, st, first_frame, first_channel):
        import torch
        # a float STFT (single-precision context) runs the same chain, widened on load and rounded once on store
        fn = self.par.lib.dll.voc_filter_apply_f32_dev if mag.dtype == torch.float32 else self.par.lib.dll.voc_filter_apply_at_dev
        self.par.lib.check(fn(
            chain._h, C.c_void_p(mag.data_ptr()), C.c_void_p(ph_inc.data_ptr()), ldk, f, c, int(first_frame),
            int(first_channel), st), self.par.handle)

    def apply_dev_lazy(self, mag, ph_inc, first_frame: int = 0, first_channel: int = 0):
        """Like apply_dev, but a play_speed stage only composes a row map; an array is gathered
        only when a later chain reads it (a chain that overwrites ph_inc without reading it, like
        random_phase, gets a fresh array instead).  Returns (mag, mag_rows, ph_inc, inc_rows,
        frames_out): maps are int64 device tensors or None, to be handed to
        voc_synthesis_mapped_dev."""
        import torch

        st = C.c_void_p(torch.cuda.current_stream(mag.device).cuda_stream) if mag.is_cuda else None
        mrows = irows = None
        c, f, ldk = mag.shape
        for kind, v in self.parts:
            if kind == "speed":
                if first_frame:
                    raise ValueError("play_speed renumbers the frames: it cannot follow a frame offset")
                src = torch.from_numpy(play_speed_map(self.par, v, f)).to(mag.device)
                mrows = src if mrows is None else mrows[src]
                irows = src if irows is None else irows[src]
                f = src.numel()
                continue
            if kind == "add":      # both branches read everything: gather what is still mapped, then as in apply_dev
                if mrows is not None:
                    mag, mrows = mag.index_select(1, mrows), None
                if irows is not None:
                    ph_inc, irows = ph_inc.index_select(1, irows), None
                mag, ph_inc = self._apply_add(v, mag, ph_inc, st, first_frame, first_channel)
                f = mag.shape[1]
                continue
            flags = (C.c_int32 * 4)()
            self.par.lib.check(self.par.lib.dll.voc_filter_touches(v._h, flags))
            load_mag, store_mag, load_inc, store_inc = (int(x) for x in flags)
            if (load_mag or store_mag) and mrows is not None:
                mag, mrows = mag.index_select(1, mrows), None
            if irows is not None:
                if load_inc:
                    ph_inc, irows = ph_inc.index_select(1, irows), None
                elif store_inc:
                    ph_inc, irows = torch.empty((c, f, ldk), dtype=ph_inc.dtype, device=ph_inc.device), None
            if f and (load_mag or store_mag or load_inc or store_inc):
                self._run_chain(v, mag, ph_inc, ldk, f, c, st, first_frame, first_channel)
        return mag, mrows, ph_inc, irows, f

    def _apply_add(self, pair, mag, ph_inc, st, first_frame=0, first_channel=0):
        import torch
        if mag.dtype == torch.float32:
            raise ValueError("add_filters runs in double precision only")
        self.frames_out(mag.shape[1])
        ma, ia = pair[0].apply_dev(mag.clone(), ph_inc.clone(), first_frame, first_channel)
        mag, ph_inc = pair[1].apply_dev(mag, ph_inc, first_frame, first_channel)
        if mag.numel():
            self.par._call("voc_add_frames_dev", C.c_void_p(ma.data_ptr()), C.c_void_p(ia.data_ptr()),
                           C.c_void_p(mag.data_ptr()), C.c_void_p(ph_inc.data_ptr()), C.c_void_p(mag.data_ptr()),
                           C.c_void_p(ph_inc.data_ptr()), mag.numel(), st)
        return mag, ph_inc

    def apply_dev(self, mag, ph_inc, first_frame: int = 0, first_channel: int = 0):
        """torch tensors (C, F, ldk) in device memory: per-frame chains run in place; a play_speed
        stage gathers into new tensors.  Returns the (mag, ph_inc) to hand to synthesis.
        first_frame / first_channel: where row 0 sits in the whole signal when the arrays hold a piece
        of it (a run of a time-sharded segment, a block of channels); random_phase draws by position."""
        import torch

        st = C.c_void_p(torch.cuda.current_stream(mag.device).cuda_stream) if mag.is_cuda else None
        for kind, v in self.parts:
            c, f, ldk = mag.shape
            if kind == "chain":
                self._run_chain(v, mag, ph_inc, ldk, f, c, st, first_frame, first_channel)
            elif kind == "add":
                mag, ph_inc = self._apply_add(v, mag, ph_inc, st, first_frame, first_channel)
            else:
                if first_frame:
                    raise ValueError("play_speed renumbers the frames: it cannot follow a frame offset")
                if mag.dtype == torch.float32:
                    raise ValueError("single precision: play_speed goes through apply_dev_lazy (row maps)")
                src = torch.from_numpy(play_speed_map(self.par, v, f)).to(mag.device)
                fo = src.numel()
                mo = torch.empty((c, fo, ldk), dtype=mag.dtype, device=mag.device)
                io = torch.empty_like(mo)
                if fo:
                    self.par._call("voc_gather_frames_dev", C.c_void_p(mag.data_ptr()), C.c_void_p(ph_inc.data_ptr()),
                                   ldk, f, C.c_void_p(src.data_ptr()), fo, C.c_void_p(mo.data_ptr()),
                                   C.c_void_p(io.data_ptr()), c, st)
                mag, ph_inc = mo, io
        return mag, ph_inc
