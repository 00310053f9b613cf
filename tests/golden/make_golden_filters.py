"""Regenerates filters.npz FROM THE FILTER ORACLE (oracle/filters_oracle.c): the reference ships no
vectors for Vocoder.Filter or playSpeed.  Freezes the oracle's answers and gives the CPU and GPU
tests fixed inputs.  Run from the repo root:  python tests/golden/make_golden_filters.py"""
import json
import os
import sys
from fractions import Fraction

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle as o  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
CHAINS = {
    "pitch2_lowpass1000": [("pitch_shift", 2.0), ("lowpass_brickwall", 1000.0)],
    "envelope32": [("envelope", 32)],
    "butterworth_band": [("bandpass_butterworth", 2.0, 300.0, 3400.0), ("amplify", 1.5)],
    "pitch0p8_highpass": [("pitch_shift", 0.8), ("highpass_brickwall", 200.0)],
    "convolution7": [("convolution", [0.1, 0.35, 0.85, 1.1, 0.85, 0.35, 0.1])],
    "bandstop_envelope5": [("bandstop_butterworth", 3.0, 1000.0, 4000.0), ("envelope", 5)],
}
SPEEDS = [Fraction(2), Fraction(1, 2), Fraction(3, 2), Fraction(1.3), Fraction(0.3)]


def main():
    rng = np.random.default_rng(20240612)
    out = {"chains": json.dumps(CHAINS), "speeds": json.dumps([[q.numerator, q.denominator] for q in SPEEDS])}
    for n in (128, 4096):
        k = n // 2 + 1
        step = 48000.0 / n
        mag, inc = rng.uniform(0, 1, (3, k)), rng.uniform(-1, 1, (3, k))
        out[f"n{n}_mag"], out[f"n{n}_inc"] = mag, inc
        for name, chain in CHAINS.items():
            ch = [(s[0], np.array(s[1])) if s[0] == "convolution" else tuple(s) for s in chain]
            m, i = o.apply_filter(step, ch, mag, inc)
            out[f"n{n}_{name}_mag"], out[f"n{n}_{name}_inc"] = m, i
    for q in SPEEDS:
        out[f"speed_{q.numerator}_{q.denominator}_f37"] = o.play_speed(q.numerator, q.denominator, 37)
    np.savez_compressed(os.path.join(HERE, "filters.npz"), **out)
    print("wrote filters.npz", os.path.getsize(os.path.join(HERE, "filters.npz")))


if __name__ == "__main__":
    main()
