"""Development aid: the fourth configuration's shape with a pitch shift that is not a whole number (the two-entry
build of the filtered synthesis kernel) -- folded into synthesis against a pass of its own."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vocoder_amd import vocoder as V, conduit as Cd, filter as Fl, window as Wn
from vocoder_amd._ffi import Library
lib = Library(os.environ['VOC_LIB']) if os.environ.get('VOC_LIB') else None
N, H, C, L = 4096, 1024, 1024, 2_880_000
par = V.VocoderParams(N, H, Wn.hann_window(N), lib=lib)
filt = (Fl.pitch_shift_interpolate(float(sys.argv[1]) if len(sys.argv) > 1 else 1.5) >> Fl.lowpass_brickwall(4000.0)).bind(par, 48000.0 / N)
x = torch.rand(C, L, dtype=torch.float64, device="cuda") - 0.5
aph = torch.zeros(C, par.K, dtype=torch.float64, device="cuda"); sph = torch.zeros_like(aph)
for mode in ("folded", "pass"):
    if mode == "pass":
        os.environ["VOC_NO_FILTERED_SYNTHESIS"] = "1"
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        y = Cd.process_filter_dev(par, filt, x, ana_phase=aph, syn_phase=sph)[0]
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        del y
    print(f"{mode}: {dt*1e3:.1f} ms", flush=True)
