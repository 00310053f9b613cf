"""PCIe-inclusive rate of the host-pointer entry point voc_roundtrip (what a Haskell caller with
host vectors sees).  usage: host_bench.py [streams]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vocoder_amd import vocoder as V, conduit as Cd, window as Wn
C = int(sys.argv[1]) if len(sys.argv) > 1 else 128
N, H, L = 4096, 1024, 2_880_000
par = V.VocoderParams(N, H, Wn.hann_window(N))
x = np.random.default_rng(0).uniform(-0.5, 0.5, (C, L))
F = Cd.num_frames(par, L)
for it in range(3):
    t0 = time.perf_counter()
    y, left, ph = Cd.process_with(par, x)
    dt = time.perf_counter() - t0
    gb = (x.nbytes + y.nbytes) / 1e9
    print(f"host entry: C={C} {dt*1e3:.0f} ms  {C*F/dt/1e6:.2f} M frames/s  ({gb:.1f} GB over PCIe, {gb/dt:.1f} GB/s inclusive)", flush=True)
