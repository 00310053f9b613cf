"""Experiment (development aid): the identity round trip of the metric shape with the channels cut into P pieces whose
analysis and synthesis kernels are enqueued on two HIP streams, so that the synthesis of piece i runs beside the
analysis of piece i + 1 (one context per stream: a context orders its own workspace between streams).
usage: overlap_exp.py [pieces ...]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vocoder_amd import vocoder as V, conduit as Cd, window as Wn

N, H, CH, L = 4096, 1024, 1024, 2_880_000
pars = [V.VocoderParams(N, H, Wn.hann_window(N)) for _ in range(2)]
par = pars[0]
F = Cd.num_frames(par, L); n = Cd.ola_length(par, F, H); ldk = (par.K + 1) & ~1
x = torch.rand(CH, L, dtype=torch.float64, device="cuda") - 0.5
y = torch.empty(CH, n, dtype=torch.float64, device="cuda")
mag = torch.empty(CH, F, ldk, dtype=torch.float64, device="cuda"); inc = torch.empty_like(mag)
aph = torch.zeros(CH, par.K, dtype=torch.float64, device="cuda"); sph = torch.zeros_like(aph)
p = lambda t: C.c_void_p(t.data_ptr())
streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def ana(pr, sl, st):
    pr._call("voc_analysis_signal_dev", p(x[sl]), x.stride(0), F, sl.stop - sl.start, p(aph[sl]), p(mag[sl]), p(inc[sl]), ldk, 1,
             C.c_void_p(st.cuda_stream))


def syn(pr, sl, st):
    pr._call("voc_synthesis_signal_dev", p(mag[sl]), p(inc[sl]), ldk, F, sl.stop - sl.start, p(sph[sl]), p(y[sl]), y.stride(0), n,
             C.c_void_p(st.cuda_stream))


def step(pieces):
    per = CH // pieces
    prev_done = None
    for i in range(pieces):
        sl = slice(i * per, (i + 1) * per)
        st, pr = streams[i % 2], pars[i % 2]
        if prev_done is not None:
            st.wait_event(prev_done)          # analysis of piece i starts when the analysis of piece i - 1 is done
        ana(pr, sl, st)
        prev_done = torch.cuda.Event(); prev_done.record(st)
        syn(pr, sl, st)


for pieces in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]:
    for it in range(2):
        step(pieces)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for it in range(3):
        step(pieces)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    print(f"pieces {pieces}: {dt*1e3:.1f} ms per step, {CH*F/dt/1e6:.2f} M frames/s", flush=True)
