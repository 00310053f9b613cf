"""Regenerates the golden fixtures in this directory FROM THE ORACLE (not from the reference:
neither GHC nor FFTW exists in the build environment, and the reference ships no vectors of its
own).  They freeze the oracle's answers so that later edits to it are caught, and give the GPU
tests fixed inputs.  Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle as o  # noqa: E402
from listmodels import list_frames_of_e, list_sum_frames_e  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = [  # name, N, W, H, frames, window
    ("n8", 8, 8, 2, 6, "hann"),
    ("n16_pad", 16, 12, 4, 8, "blackman"),
    ("n64", 64, 64, 16, 16, "blackman"),
    ("n64_oddhop", 64, 50, 13, 11, "hamming"),
    ("n2048", 2048, 2048, 512, 6, "blackman"),
    ("n4096", 4096, 4096, 1024, 8, "hann"),
]


def main():
    rng = np.random.default_rng(20240611)
    for name, n, w, h, f, wk in CASES:
        win = o.make_window(wk, w)
        length = (f - 1) * h + w + h // 2
        t = np.arange(length)
        x = 0.5 * np.sin(2 * np.pi * 440.0 * t / 48000.0 + 0.1) + 1e-2 * rng.uniform(-1, 1, length)
        k = n // 2 + 1
        ph0, sp0 = rng.uniform(-3, 3, k), rng.uniform(-3, 3, k)
        r = o.roundtrip(n, h, win, x, ana_phase=ph0, syn_phase=sp0, want_stft=True)
        np.savez_compressed(os.path.join(HERE, f"roundtrip_{name}.npz"), N=n, W=w, H=h, window=win, x=x,
                            ana_phase_in=ph0, syn_phase_in=sp0, y=r["y"], ana_phase_out=r["ana_phase"],
                            syn_phase_out=r["syn_phase"], leftover=r["leftover"], mag=r["mag"][:3], ph_inc=r["ph_inc"][:3])
    # integer framing / overlap-add vectors, from the reference's list models
    irng = np.random.default_rng(7)
    chunks = [irng.integers(-99, 99, int(irng.integers(0, 30))) for _ in range(9)]
    data = np.concatenate(chunks)
    np.savez_compressed(os.path.join(HERE, "framing_int.npz"),
                        data=data, chunk_lens=np.array([c.size for c in chunks]),
                        frames_w10_h3=np.array(list_frames_of_e(10, 3, [c.tolist() for c in chunks])),
                        frames_w7_h7=np.array(list_frames_of_e(7, 7, [c.tolist() for c in chunks])),
                        ola_c4_h3=np.array(list_sum_frames_e(4, 3, list_frames_of_e(10, 3, [data.tolist()]))),
                        ola_c16_h3=np.array(list_sum_frames_e(16, 3, list_frames_of_e(10, 3, [data.tolist()]))))
    print("wrote", sorted(p for p in os.listdir(HERE) if p.endswith(".npz")))


if __name__ == "__main__":
    main()
