"""Times the filter-chain kernel and the playSpeed gather on an STFT of benchmark size in device
memory.  usage: filter_bench.py [streams] [N]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vocoder_amd import vocoder as V, filter as Fl

C_ = int(sys.argv[1]) if len(sys.argv) > 1 else 512
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
H = N // 4
F = (2880000 - N) // H + 1
win = 0.5 * (1 - np.cos(2 * np.pi * np.arange(N) / N))
from vocoder_amd._ffi import Library
par = V.VocoderParams(N, H, win, lib=Library(os.environ['VOC_LIB']) if os.environ.get('VOC_LIB') else None)   # development: time an experimental build
ldk = (par.K + 1) & ~1
mag = torch.rand((C_, F, ldk), dtype=torch.float64, device="cuda")
inc = torch.rand_like(mag)
step = 48000.0 / N
rows = C_ * F
cases = {
    "pitchShift2+lowpass1000": Fl.pitch_shift_interpolate(2.0) >> Fl.lowpass_brickwall(1000.0),
    "envelope32": Fl.envelope_filter(32),
    "amplify": Fl.amplify(0.5),
    "randomPhase": Fl.random_phase_filter(1),
    "butterworth x4": Fl.lowpass_butterworth(2, 3000.) >> Fl.highpass_butterworth(2, 100.) >> Fl.amplify(2.) >> Fl.bandstop_brickwall(400., 500.),
}
for name, f in cases.items():
    b = f.bind(par, step)
    for it in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        b.apply_dev(mag, inc)
        e1.record()
        torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    gb = rows * 32 * par.K / 1e9
    print(f"{name:28s} {ms:8.2f} ms  {rows / ms / 1e3:8.2f} M rows/s  {gb / ms:7.2f} TB/s algorithmic (32K B/row)")
    mag.uniform_(0, 1)
