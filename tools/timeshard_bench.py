"""The shape of a long multichannel recording cut by time over several GPUs, on the one GPU at hand:
this process plays one rank of `--world` and walks its segment in runs of frames, the way
vocoder_amd.timeshard.process_pieces does under torch.distributed (the all-gather is replaced by
its single-rank stand-in, so the walk for the phase advance is forced with --two-walks).

    python tools/timeshard_bench.py --channels 64 --minutes 10 --piece 8192 --filter envelope

The signal is generated on the device run by run (a ten-hour recording of 64 channels is 885 GB),
the output is reduced to a checksum as it leaves.
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--channels", type=int, default=64)
    ap.add_argument("--minutes", type=float, default=10.0, help="length of this rank's segment")
    ap.add_argument("--piece", type=int, default=8192, help="frames per run")
    ap.add_argument("--filter", choices=["identity", "envelope"], default="envelope")
    ap.add_argument("--two-walks", action="store_true", help="also time the walk that only accumulates the phase advance")
    a = ap.parse_args()

    import numpy as np
    import torch
    import ctypes as C
    from vocoder_amd import vocoder as V, timeshard as T, filter as Fl
    n, hop = 4096, 1024
    win = 0.5 * (1 - np.cos(2 * np.pi * np.arange(n) / n))
    par = V.VocoderParams(n, hop, win)
    length = int(a.minutes * 60 * 48000)
    dev = torch.device("cuda:0")
    freqs = (110.0 * 2.0 ** ((torch.arange(a.channels, device=dev) % 72) / 12.0)).double()[:, None]

    def source(lo, hi):
        t = torch.arange(lo, hi, device=dev, dtype=torch.float64)[None, :]
        return 0.5 * torch.sin((2 * np.pi / 48000.0) * freqs * t)

    filt = Fl.envelope_filter(32).bind(par, 48000.0 / n) if a.filter == "envelope" else None
    acc = torch.zeros((), dtype=torch.float64, device=dev)
    count = [0]

    def sink(lo, y):
        acc.add_(y.square().sum())
        count[0] += y.shape[1]

    frames = par.lib.dll.voc_num_frames(par.handle, length) * a.channels
    for rep in range(2):                       # the first repetition warms the allocator and the library buffers
        acc.zero_()
        count[0] = 0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if a.two_walks:                        # what ranks above 0 add when increments are not all zero
            sp = T.plan(par, length, 1, 0)
            carried, total = None, torch.zeros((a.channels, par.K), dtype=torch.float64, device=dev)
            for i, (lo, hi) in enumerate(T.pieces_of(par, sp, a.piece)):
                pp = T._range_plan(par, sp.frames_total, lo, hi, False, False, hop)
                mag, inc, carried = T.analyse(par, pp, source(pp.in_lo, pp.in_hi), carried)
                mag, inc = T._filtered(filt, mag, inc, pp.f_lo)
                par._call("voc_phase_advance_dev", C.c_void_p(inc.data_ptr()), inc.shape[2], inc.shape[1], inc.shape[0],
                          C.c_void_p(total.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream))
                del mag, inc
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        T.process_pieces(par, source, length, a.piece, sink=sink, filt=filt)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    print(f"channels={a.channels} segment={a.minutes} min frames={frames} piece={a.piece} filter={a.filter} "
          f"advance walk {1e3 * (t1 - t0):.1f} ms, output walk {1e3 * (t2 - t1):.1f} ms: "
          f"{frames / (t2 - t0) / 1e6:.2f} Mframes/s, samples out {count[0]}, energy {acc.item():.6e}, "
          f"peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")


if __name__ == "__main__":
    main()
