"""Throughput of the chunk-at-a-time entry point (voc_stream_push_dev) against the one-call path:
C streams fed in pieces of `piece` seconds.  usage: stream_bench.py [streams] [piece_seconds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vocoder_amd import vocoder as V, conduit as Cd, stream as S, window as Wn
C = int(sys.argv[1]) if len(sys.argv) > 1 else 256
piece = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
N, H, L = 4096, 1024, 2_880_000 // 4          # 15 s per stream
par = V.VocoderParams(N, H, Wn.hann_window(N))
x = torch.rand(C, L, dtype=torch.float64, device="cuda") - 0.5
m = int(piece * 48000)
for it in range(2):
    st = S.Stream(par, C)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    outs = [st.push_dev(x[:, p:p + m], min(m, L - p)) for p in range(0, L, m)]
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    frames = C * Cd.num_frames(par, L)
    print(f"stream: C={C} pieces of {m} samples ({len(outs)} pushes): {dt*1e3:.1f} ms, {frames/dt/1e6:.2f} M frames/s", flush=True)
    st.close()
