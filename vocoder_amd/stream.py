"""`Vocoder.Conduit.process` as an incremental object (vocoder-conduit/src/Vocoder/Conduit.hs:46-53):
push chunks of samples of any sizes, collect the chunks sumFramesE publishes
(Conduit/Frames.hs:41-58), flush at end of input.  The state lives in the HIP library
(`voc_stream_*`); this class only marshals."""
from __future__ import annotations

import ctypes as C

import numpy as np

from .vocoder import VocoderParams, _f64, _p

__all__ = ["Stream"]


class Stream:
    def __init__(self, par: VocoderParams, channels: int = 1, chunk: int | None = None, filt=None,
                 ana_phase=None, syn_phase=None, dunai: bool = False):
        """filt: a `vocoder_amd.filter.BoundFilter`, or None: one chain of per-frame stages, optionally with
        one play_speed stage anywhere in it (`chain >> play_speed(r) >> chain'`).
        dunai=True: the conventions of Vocoder.Dunai.process (vocoder-dunai/src/Vocoder/Dunai.hs:46-48):
        zero pre-pad, pushes in multiples of the hop, as many samples out as in."""
        self.par, self.channels = par, int(channels)
        self.chunk = par.H if chunk is None else int(chunk)
        self.f32 = getattr(par, "precision", "f64") == "f32"      # a single-precision context: float state and samples
        self.dtype = np.float32 if self.f32 else np.float64
        self._sfx = "_f32" if self.f32 else ""
        fh, speed, after = None, None, None
        if filt is not None:
            kinds = [k for k, _ in filt.parts]
            if kinds not in ([], ["chain"], ["speed"], ["chain", "speed"], ["speed", "chain"], ["chain", "speed", "chain"]):
                raise ValueError("a stream takes chains of at most 12 per-frame filters around at most one play_speed")
            for i, (kind, v) in enumerate(filt.parts):
                if kind == "speed":
                    speed = v
                elif speed is None:
                    fh = v._h
                else:
                    after = v._h
        self._filt = filt                         # keeps the device chains alive
        h = C.c_void_p()
        par.lib.check(par.lib.dll.voc_stream_create_mode(par.handle, self.channels, self.chunk, fh, 1 if dunai else 0,
                                                         C.byref(h)), par.handle)
        self._h = h
        if speed is not None:
            par.lib.check(par.lib.dll.voc_stream_set_play_speed(h, speed.numerator, speed.denominator, after), par.handle)
        if ana_phase is not None or syn_phase is not None:
            zero = np.zeros((self.channels, par.K), dtype=self.dtype)
            a = zero if ana_phase is None else np.ascontiguousarray(ana_phase, dtype=self.dtype).reshape(self.channels, par.K)
            s = zero if syn_phase is None else np.ascontiguousarray(syn_phase, dtype=self.dtype).reshape(self.channels, par.K)
            par.lib.check(self._fn("voc_stream_set_phases")(h, _p(a), _p(s)), par.handle)

    def _fn(self, name, dev: bool = False):
        return getattr(self.par.lib.dll, name + self._sfx + ("_dev" if dev else ""))

    def set_filter(self, filt):
        """Swaps the per-frame filter chain between pushes (a new source of a concatenation)."""
        fh = None
        if filt is not None:
            kinds = [k for k, _ in filt.parts]
            if kinds not in ([], ["chain"]):
                raise ValueError("a stream takes one chain of per-frame filters")
            fh = filt.parts[0][1]._h if kinds else None
        self.par.lib.check(self.par.lib.dll.voc_stream_set_filter(self._h, fh), self.par.handle)
        self._filt = filt

    def push(self, x) -> np.ndarray:
        """x: (n,) for one channel or (C, n).  Returns the samples published by this push, (C, m)."""
        x2 = np.ascontiguousarray(x, dtype=self.dtype).reshape(self.channels, -1)
        n = x2.shape[1]
        m = self.par.lib.dll.voc_stream_push_length(self._h, n)
        y = np.empty((self.channels, max(m, 0)), dtype=self.dtype)
        got = C.c_int64(0)
        self.par.lib.check(self._fn("voc_stream_push")(self._h, _p(x2), n, n, _p(y), y.shape[1], y.shape[1],
                                                       C.byref(got)), self.par.handle)
        assert got.value == m
        return y

    def flush(self) -> np.ndarray:
        m = self.par.lib.dll.voc_stream_flush_length(self._h)
        y = np.empty((self.channels, max(m, 0)), dtype=self.dtype)
        got = C.c_int64(0)
        self.par.lib.check(self._fn("voc_stream_flush")(self._h, _p(y), y.shape[1], y.shape[1], C.byref(got)),
                           self.par.handle)
        return y

    def push_dev(self, x, n: int | None = None):
        """torch tensors in device memory: x (C, >= n); returns a new tensor (C, m) on the same device."""
        import torch
        n = x.shape[1] if n is None else n
        m = self.par.lib.dll.voc_stream_push_length(self._h, n)
        y = torch.empty((self.channels, max(m, 0)), dtype=torch.float32 if self.f32 else torch.float64, device=x.device)
        st = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream) if x.is_cuda else None
        got = C.c_int64(0)
        self.par.lib.check(self._fn("voc_stream_push", dev=True)(
            self._h, C.c_void_p(x.data_ptr()), x.stride(0), n, C.c_void_p(y.data_ptr()), max(y.stride(0), y.shape[1]),
            y.shape[1], C.byref(got), st), self.par.handle)
        return y

    def phases(self):
        a = np.empty((self.channels, self.par.K), dtype=self.dtype)
        s = np.empty_like(a)
        self.par.lib.check(self._fn("voc_stream_get_phases")(self._h, _p(a), _p(s)), self.par.handle)
        return a, s

    def leftover(self) -> np.ndarray:
        n = self._fn("voc_stream_leftover")(self._h, None, 0)
        out = np.empty((self.channels, n), dtype=self.dtype)
        if n:
            self._fn("voc_stream_leftover")(self._h, _p(out), n)
        return out

    def close(self):
        if getattr(self, "_h", None):
            self.par.lib.dll.voc_stream_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
