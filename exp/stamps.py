import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vocoder_amd import vocoder as V, conduit as Cd, window as Wn
from vocoder_amd._ffi import Library
lib = Library(os.environ['VOC_LIB'])
C = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
N, H, L = 4096, 1024, 2_880_000
par = V.VocoderParams(N, H, Wn.hann_window(N), lib=lib)
dev = torch.device('cuda:0')
x = torch.rand(C, L, dtype=torch.float64, device=dev) - 0.5
F = Cd.num_frames(par, L); n = Cd.ola_length(par, F, H)
y = torch.empty(C, n, dtype=torch.float64, device=dev)
aph = torch.zeros(C, par.K, dtype=torch.float64, device=dev); sph = torch.zeros_like(aph)
ws = torch.empty(Cd.roundtrip_workspace_bytes(par, L, C), dtype=torch.uint8, device=dev)
Cd.process_dev(par, x, y, aph, sph, workspace=ws); torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 8)()
lib.dll.voc_debug_stamps(out, 1)
Cd.process_dev(par, x, y, aph, sph, workspace=ws); torch.cuda.synchronize()
lib.dll.voc_debug_stamps(out, 1)
fft, binl, tot, frames = out[0], out[1], out[2], out[3]
print("fft detail per wave-frame: barriers %.0f stores %.0f loads %.0f compute %.0f" % tuple(out[i]/frames for i in (4,5,6,7)))
print(f"per wave-frame cycles: fft section {fft/frames:.0f}, bin section {binl/frames:.0f}, whole kernel {tot/frames:.0f} (wave-frames {frames})")
