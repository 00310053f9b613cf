"""What a caller with HOST vectors sees (the Haskell bindings): the PCIe-inclusive rate of
voc_roundtrip on a batch, and the call rate of the per-frame entry points that
Vocoder.Conduit.analysis / synthesis hit once per frame (Conduit.hs:39,43).
usage: host_bench.py [streams]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vocoder_amd import vocoder as V, conduit as Cd, window as Wn
C = int(sys.argv[1]) if len(sys.argv) > 1 else 128
N, H, L = 4096, 1024, 2_880_000
par = V.VocoderParams(N, H, Wn.hann_window(N))
rng = np.random.default_rng(0)

# per-frame calls: analysisBlock then synthesisBlock, one frame each (F = 1, C = 1)
fr = rng.standard_normal(N)
aph, sph = V.zero_phase(par), V.zero_phase(par)
for reps in (20, 300):
    t0 = time.perf_counter()
    for _ in range(reps):
        aph, stft = V.analysis_block(par, aph, fr)
        sph, out = V.synthesis_block(par, sph, stft)
    dt = time.perf_counter() - t0
print(f"per-frame calls (analysisBlock + synthesisBlock, F = 1): {dt / reps * 1e6:.0f} us per frame round trip, "
      f"{reps / dt:.0f} frames/s", flush=True)

# the same two calls without the Python marshalling around them (what a compiled host pays per call)
dll, hnd = par.lib.dll, par.handle
K = par.K
ph_a, ph_s = np.zeros(K), np.zeros(K)
mag, inc, out = np.empty(K), np.empty(K), np.empty(N)
p = lambda a: a.ctypes.data
for reps in (20, 1000):
    t0 = time.perf_counter()
    for _ in range(reps):
        dll.voc_analysis_stage(hnd, p(ph_a), p(fr), p(mag), p(inc), 1, 1)
        dll.voc_synthesis_stage(hnd, p(ph_s), p(mag), p(inc), p(out), 1, 1)
    dt = time.perf_counter() - t0
print(f"  the two C calls alone: {dt / reps * 1e6:.0f} us per frame round trip, {reps / dt:.0f} frames/s", flush=True)

# stage calls on a chunk of frames: 8 channels x 64 frames
frames = rng.standard_normal((8, 64, N))
aphs, sphs = V.zero_phase(par, 8), V.zero_phase(par, 8)
for reps in (3, 30):
    t0 = time.perf_counter()
    for _ in range(reps):
        aphs, stft = V.analysis_stage(par, aphs, frames)
        sphs, out = V.synthesis_stage(par, sphs, stft)
    dt = time.perf_counter() - t0
print(f"stage calls (8 channels x 64 frames): {dt / reps * 1e3:.2f} ms per call pair, {reps * 512 / dt / 1e3:.0f} k frames/s", flush=True)

x = rng.uniform(-0.5, 0.5, (C, L))
F = Cd.num_frames(par, L)
for it in range(3):
    t0 = time.perf_counter()
    y, left, ph = Cd.process_with(par, x)
    dt = time.perf_counter() - t0
    gb = (x.nbytes + y.nbytes) / 1e9
    print(f"host entry: C={C} {dt*1e3:.0f} ms  {C*F/dt/1e6:.2f} M frames/s  ({gb:.1f} GB over PCIe, {gb/dt:.1f} GB/s inclusive)", flush=True)
