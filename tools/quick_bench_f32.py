"""Timing of the single-precision fused round trip at the metric shape (development aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vocoder_amd import vocoder as V, conduit as Cd, window as Wn
C = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = W = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
H = N // 4
L = 2_880_000
from vocoder_amd._ffi import Library
lib = Library(os.environ['VOC_LIB']) if os.environ.get('VOC_LIB') else None   # development: time an experimental build
par = V.VocoderParams(N, H, Wn.hann_window(W), precision="f32", lib=lib)
x = torch.rand(C, L, dtype=torch.float32, device="cuda") - 0.5
F = Cd.num_frames(par, L); n = Cd.ola_length(par, F, H)
y = torch.empty(C, n, dtype=torch.float32, device="cuda")
aph = torch.zeros(C, par.K, dtype=torch.float32, device="cuda"); sph = torch.zeros_like(aph)
ws = torch.empty(Cd.roundtrip_workspace_bytes(par, L, C), dtype=torch.uint8, device="cuda")
for it in range(2):
    Cd.process_dev(par, x, y, aph, sph, workspace=ws)
torch.cuda.synchronize()
par.lib.dll.voc_profile_enable(par.handle, 1)
t0 = time.perf_counter()
for it in range(3):
    Cd.process_dev(par, x, y, aph, sph, workspace=ws)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
ms = np.zeros(4); cnt = np.zeros(4, dtype=np.int64)
par.lib.dll.voc_profile_read(par.handle, ms.ctypes.data, cnt.ctypes.data)
bpf = (16 * H + 32 * (N // 2 + 1)) // 2
print(f'f32 N={N} C={C} frames={C*F} {dt*1e3:.1f} ms/step {C*F/dt/1e6:.2f} Mframes/s roofline_frac(f32 bytes)={C*F/dt*bpf/8e12:.3f} | analysis {ms[0]/max(cnt[0],1):.1f} carry {ms[1]/max(cnt[1],1):.1f} synthesis {ms[2]/max(cnt[2],1):.1f} ms')
aph.zero_(); sph.zero_()
Cd.process_dev(par, x, y, aph, sph, workspace=ws); torch.cuda.synchronize()
print('identity err', (y[:8, N:L - N] - x[:8, N:L - N]).abs().max().item())
