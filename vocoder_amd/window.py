"""Window tables, mirroring the reference's `Vocoder.Window` (vocoder/src/Vocoder/Window.hs).

A window is a W-entry table computed once on the host and uploaded when the context is
created; it is not part of the per-frame path.  `make_window f n` samples f at k/(n-1)
(Window.hs:28-29).
"""
from __future__ import annotations

import math

import numpy as np

__all__ = ["make_window", "box_window", "triangle_window", "hamming_window", "hann_window",
           "generalized_blackman_window", "blackman_window", "exact_blackman_window", "lanczos_window",
           "flat_top_window"]


def make_window(f, n: int) -> np.ndarray:
    """makeWindow (Window.hs:28-29)."""
    return np.array([f(k / (float(n) - 1)) for k in range(n)], dtype=np.float64)


def box_window(n):  # Window.hs:32-33
    return make_window(lambda x: 1.0, n)


def triangle_window(n):  # Window.hs:36-37
    return make_window(lambda x: 2 * (0.5 - abs(x - 0.5)), n)


def hamming_window(n):  # Window.hs:40-44
    alpha, beta = 25.0 / 46.0, 21.0 / 46.0
    return make_window(lambda x: alpha - beta * math.cos(2 * math.pi * x), n)


def hann_window(n):  # Window.hs:47-48
    return make_window(lambda x: 0.5 * (1 - math.cos(2 * math.pi * x)), n)


def generalized_blackman_window(a, n):  # Window.hs:51-56
    a0, a1, a2 = (1 - a) / 2, 0.5, a / 2

    def f(x):
        p = 2 * math.pi * x
        return a0 - a1 * math.cos(p) + a2 * math.cos(2 * p)
    return make_window(f, n)


def blackman_window(n):  # Window.hs:59-60
    return generalized_blackman_window(0.16, n)


def exact_blackman_window(n):  # Window.hs:63-68
    a0, a1, a2 = 7938.0 / 18608.0, 9240.0 / 18608.0, 1430.0 / 18608.0

    def f(x):
        p = 2 * math.pi * x
        return a0 - a1 * math.cos(p) + a2 * math.cos(2 * p)
    return make_window(f, n)


def lanczos_window(n):  # Window.hs:71-75
    def f(x):
        s = 2 * x - 1
        return 1.0 if s == 0 else math.sin(math.pi * s) / (math.pi * s)
    return make_window(f, n)


def flat_top_window(n):  # Window.hs:78-85
    a0, a1, a2, a3, a4 = 0.21557895, 0.41663158, 0.277263158, 0.083578947, 0.006947368
    return make_window(lambda x: a0 - a1 * math.cos(2 * math.pi * x) + a2 * math.cos(4 * math.pi * x)
                       - a3 * math.cos(6 * math.pi * x) + a4 * math.cos(8 * math.pi * x), n)
