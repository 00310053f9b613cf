"""Quick on-GPU sanity run: parity of every entry-point family against the oracle, then a first
timing of the fused round trip.  Development aid; the judged checks live in tests/ and bench.py."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import oracle as o
from vocoder_amd import vocoder as V, conduit as Cd

rng = np.random.default_rng(0)
def circ(a, b):
    d = (a - b + np.pi) % (2 * np.pi) - np.pi
    return np.abs(d).max()

for (N, W, H, F, C, at, st) in ((64, 64, 16, 9, 2, 0, 0), (64, 48, 16, 23, 2, 4, 5), (256, 256, 64, 12, 3, 5, 0),
                                (2048, 2048, 512, 10, 2, 0, 4), (4096, 4096, 1024, 12, 2, 5, 6), (8192, 8192, 2048, 6, 1, 0, 0),
                                (16384, 10000, 3000, 5, 1, 0, 0)):
    os.environ['VOC_ANA_TILE_LEN'] = str(at); os.environ['VOC_SYN_TILE_LEN'] = str(st)
    w = o.make_window('blackman', W) + 0.01
    par = V.VocoderParams(N, H, w)
    K = N // 2 + 1
    frames = rng.standard_normal((C, F, W))
    ph0 = rng.uniform(-3, 3, (C, K))
    ph, (mag, inc) = V.analysis_stage(par, ph0, frames)
    res = [o.analysis_stage(N, H, w, ph0[c], frames[c]) for c in range(C)]
    pho = np.stack([r[0] for r in res]); mago = np.stack([r[1] for r in res]); inco = np.stack([r[2] for r in res])
    print('ana', N, W, H, 'ph', circ(ph, pho), 'mag', np.abs(mag - mago).max() / mago.max(), 'inc', circ(inc * H, inco * H), flush=True)
    sp0 = rng.uniform(-3, 3, (C, K))
    sph, fr = V.synthesis_stage(par, sp0, (mago, inco))
    res = [o.synthesis_stage(N, H, w, sp0[c], mago[c], inco[c]) for c in range(C)]
    spho = np.stack([r[0] for r in res]); fro = np.stack([r[1] for r in res])
    print('  syn phase bitexact', np.array_equal(sph, spho), 'frames', np.abs(fr - fro).max() / np.abs(fro).max(), flush=True)
    L = (F - 1) * H + W + 5
    x = rng.standard_normal((C, L))
    y, left, (a2, s2) = Cd.process_with(par, x, phases=(ph0, sp0))
    for c in range(C):
        r = o.roundtrip(N, H, w, x[c], ana_phase=ph0[c], syn_phase=sp0[c])
        print('  rt c', c, 'y', np.linalg.norm(y[c] - r['y']) / np.linalg.norm(r['y']), 'left', np.array_equal(left[c], r['leftover']),
              'aph', circ(a2[c], r['ana_phase']), 'sph', circ(s2[c], r['syn_phase']), flush=True)
    par.close()

os.environ['VOC_ANA_TILE_LEN'] = '0'; os.environ['VOC_SYN_TILE_LEN'] = '0'
# ---- first timing: config-2 shape at a reduced channel count
N = W = 4096; H = 1024; L = 2_880_000
from vocoder_amd.window import hann_window
par = V.VocoderParams(N, H, hann_window(W))
for C in (64, 256):
    dev = torch.device('cuda:0')
    x = (torch.rand(C, L, dtype=torch.float64, device=dev) - 0.5)
    F = Cd.num_frames(par, L); n = Cd.ola_length(par, F, H)
    y = torch.empty(C, n, dtype=torch.float64, device=dev)
    aph = torch.zeros(C, par.K, dtype=torch.float64, device=dev); sph = torch.zeros_like(aph)
    ws = torch.empty(Cd.roundtrip_workspace_bytes(par, L, C), dtype=torch.uint8, device=dev)
    for it in range(3):
        aph.zero_(); sph.zero_()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        Cd.process_dev(par, x, y, aph, sph, workspace=ws)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f'C={C} F={F} frames={C*F} time {dt*1e3:.1f} ms  {C*F/dt/1e6:.2f} Mframes/s', flush=True)
    err = (y[:, 4096:L - 4096] - x[:, 4096:L - 4096]).abs().max().item()
    print('identity err (hann COLA ripple expected ~1e-7)', err, flush=True)
    del x, y, ws
