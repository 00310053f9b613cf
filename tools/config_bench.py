"""Times the filtered benchmark configurations on one GPU (development aid; bench.py measures the
identity round trip only).  usage: config_bench.py {3|4} [streams]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fractions import Fraction
import numpy as np, torch
from vocoder_amd import vocoder as V, conduit as Cd, filter as Fl, window as Wn

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 4
L = 2_880_000
if cfg == 3:     # --playSpeed 2 --randomPhase, N=8192 hop=2048, 256 stereo streams
    N, H, C = 8192, 2048, 512
    filt = Fl.play_speed(Fraction(2)) >> Fl.random_phase_filter(7)
else:            # --pitchShiftInterpolate 2 --lowpassBrickwall 1000, N=4096 hop=1024, 512 stereo streams
    N, H, C = 4096, 1024, 1024
    filt = Fl.pitch_shift_interpolate(2.0) >> Fl.lowpass_brickwall(1000.0)
if len(sys.argv) > 2:
    C = int(sys.argv[2])
par = V.VocoderParams(N, H, Wn.hann_window(N))
bound = filt.bind(par, 48000.0 / N)
x = torch.rand(C, L, dtype=torch.float64, device="cuda") - 0.5
F = Cd.num_frames(par, L)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    y, aph, sph = Cd.process_filter_dev(par, bound, x)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    fo = bound.frames_out(F)
    print(f"config {cfg}: C={C} N={N} frames in {C*F} out {C*fo}: {dt*1e3:.1f} ms  {C*F/dt/1e6:.2f} M input frames/s  {C*(F+fo)/2/dt/1e6:.2f} M frames/s (mean of in/out)", flush=True)
    del y
