"""GPU: the two halves of the fused path as separate entry points, and the time-sharded round trip
(ranks run one after the other in this process; the exchanges are the same tensor operations the
distributed driver performs)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from helpers import TOL_L2  # noqa: E402

pytestmark = pytest.mark.gpu


def _hann(n):
    return 0.5 * (1 - np.cos(2 * np.pi * np.arange(n) / n))


def _signal(c, length, seed=11):
    rng = np.random.default_rng(seed)
    t = np.arange(length)
    return np.stack([0.5 * np.sin(2 * np.pi * (0.01 + 0.003 * i) * t + i) for i in range(c)]) + \
        1e-2 * rng.uniform(-1, 1, (c, length))


def _wrap_err(a, b):
    d = np.abs(a - b) % (2 * np.pi)
    return np.minimum(d, 2 * np.pi - d).max()


@pytest.mark.parametrize("n,w,hop", [(4096, 4096, 1024), (256, 200, 64), (16384, 16384, 4096)])
def test_split_entry_points_equal_fused_roundtrip(gpu_lib, n, w, hop):
    import ctypes as C
    import torch
    from vocoder_amd import vocoder as V, conduit as Cd
    par = V.VocoderParams(n, hop, _hann(w), lib=gpu_lib)
    c, frames = 3, 21
    length = w + hop * (frames - 1) + 5
    x = torch.from_numpy(_signal(c, length)).cuda()
    olen = Cd.ola_length(par, frames, hop)
    y1 = torch.empty((c, olen), dtype=torch.float64, device="cuda")
    a1 = torch.zeros((c, par.K), dtype=torch.float64, device="cuda")
    s1 = torch.zeros_like(a1)
    Cd.process_dev(par, x, y1, a1, s1)

    ldk = (par.K + 1) & ~1
    mag = torch.empty((c, frames, ldk), dtype=torch.float64, device="cuda")
    inc = torch.empty_like(mag)
    a2, s2, adv = torch.zeros_like(a1), torch.zeros_like(a1), torch.zeros_like(a1)
    y2 = torch.empty_like(y1)
    p = lambda t: C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    par._call("voc_analysis_signal_dev", p(x), x.stride(0), frames, c, p(a2), p(mag), p(inc), ldk, 1, st)
    par._call("voc_phase_advance_dev", p(inc), ldk, frames, c, p(adv), st)
    par._call("voc_synthesis_signal_dev", p(mag), p(inc), ldk, frames, c, p(s2), p(y2), y2.stride(0), olen, st)
    torch.cuda.synchronize()
    assert torch.equal(y1, y2) and torch.equal(a1, a2) and torch.equal(s1, s2)
    assert torch.equal(adv, s2)          # the chain alone ends on the same bits as the chain inside synthesis


def test_split_entry_points_reject_bad_sizes(gpu_lib):
    import ctypes as C
    import torch
    from vocoder_amd import vocoder as V
    par = V.VocoderParams(64, 16, _hann(64), lib=gpu_lib)
    t = torch.zeros(4096, dtype=torch.float64, device="cuda")
    p = C.c_void_p(t.data_ptr())
    d = gpu_lib.dll
    assert d.voc_analysis_signal_dev(par.handle, p, 64 + 16, 3, 1, p, p, p, 34, 1, None) == -5     # row too short
    assert d.voc_analysis_signal_dev(par.handle, p, 200, 3, 1, p, p, p, 32, 1, None) == -5         # ldk < K
    assert d.voc_synthesis_signal_dev(par.handle, p, p, 34, 3, 1, p, p, 200, 64 + 31, None) == -5  # y_len too short
    assert d.voc_phase_advance_dev(par.handle, None, 34, 3, 1, p, None) == -1
    assert d.voc_analysis_signal_dev(par.handle, p, 200, 0, 1, p, p, p, 34, 1, None) == 0          # no frames: no-op


@pytest.mark.parametrize("n,hop,world,frames_per_rank", [(4096, 1024, 4, 12), (1024, 256, 3, 9)])
def test_time_sharded_matches_sequential(gpu_lib, orc, n, hop, world, frames_per_rank):
    import torch
    from vocoder_amd import vocoder as V, conduit as Cd, timeshard as T
    win = _hann(n)
    par = V.VocoderParams(n, hop, win, lib=gpu_lib)
    c = 4
    length = n + hop * (frames_per_rank * world + 2) + 9
    x = _signal(c, length)
    xt = torch.from_numpy(x).cuda()
    plans = [T.plan(par, length, world, r) for r in range(world)]
    ana = [T.analyse(par, sp, xt[:, sp.in_lo:sp.in_hi].contiguous()) for sp in plans]
    totals = [T.local_advance(par, inc) for _, inc, _ in ana]
    outs = [T.synthesise(par, sp, m, i, T.start_phase(totals, sp.rank)) for sp, (m, i, _) in zip(plans, ana)]
    ys = []
    for r, (sp, (y, _)) in enumerate(zip(plans, outs)):
        if r > 0:
            pv = plans[r - 1]
            T.add_incoming_tail(y, outs[r - 1][0][:, pv.out_hi - pv.out_lo:].clone())
        ys.append(y[:, :sp.out_hi - sp.out_lo])
    y = torch.cat(ys, dim=1)

    # against the unsharded run on the same GPU
    f = Cd.num_frames(par, length)
    y_seq = torch.empty((c, Cd.ola_length(par, f, hop)), dtype=torch.float64, device="cuda")
    aph = torch.zeros((c, par.K), dtype=torch.float64, device="cuda")
    sph = torch.zeros_like(aph)
    Cd.process_dev(par, xt, y_seq, aph, sph)
    torch.cuda.synchronize()
    assert y.shape == y_seq.shape
    rel = (torch.linalg.norm(y - y_seq) / torch.linalg.norm(y_seq)).item()
    assert rel <= 1e-12, rel                                # differs only by re-association at rank boundaries
    assert torch.equal(ana[-1][2], aph)                     # raw phase of the last frame: no exchange involved
    assert _wrap_err(outs[-1][1].cpu().numpy(), sph.cpu().numpy()) <= 1e-9

    # and against the sequential oracle
    ref = orc.roundtrip(n, hop, win, x[0], hop, np.zeros(par.K), np.zeros(par.K))
    yy = y[0].cpu().numpy()
    assert np.linalg.norm(yy - ref["y"]) / np.linalg.norm(ref["y"]) <= TOL_L2


@pytest.mark.parametrize("chain", ["identity", "envelope", "pitch+lowpass"])
def test_one_rank_in_pieces_equals_whole_signal(gpu_lib, orc, chain):
    """A long signal walked in runs of frames: phases carried run to run reproduce the unpieced
    run's phases bit for bit; samples differ only at run boundaries, by the re-associated overlap-add."""
    import torch
    from vocoder_amd import vocoder as V, conduit as Cd, timeshard as T, filter as Fl
    n, hop = 4096, 1024
    win = _hann(n)
    par = V.VocoderParams(n, hop, win, lib=gpu_lib)
    c, frames = 5, 75
    length = n + hop * (frames - 1) + 17
    x = _signal(c, length)
    xt = torch.from_numpy(x).cuda()
    descr = {"identity": None, "envelope": Fl.envelope_filter(32),
             "pitch+lowpass": Fl.pitch_shift_interpolate(2.0) >> Fl.lowpass_brickwall(1000.0)}[chain]
    filt = descr.bind(par, 48000.0 / n) if descr is not None else None
    reads = []

    def source(lo, hi):
        reads.append(hi - lo)
        return xt[:, lo:hi].contiguous()

    y, (aph, sph) = T.process_pieces(par, source, length, 16, filt=filt)
    y_seq, aph_seq, sph_seq = Cd.process_filter_dev(par, filt, xt)
    torch.cuda.synchronize()
    assert len(reads) == 5 and max(reads) <= 15 * hop + n           # 75 frames in 5 runs of 15, one walk on one rank
    assert y.shape == y_seq.shape
    rel = (torch.linalg.norm(y - y_seq) / torch.linalg.norm(y_seq)).item()
    assert rel <= 1e-14, rel
    assert torch.equal(aph, aph_seq) and torch.equal(sph, sph_seq)
    if chain == "identity":
        ref = orc.roundtrip(n, hop, win, x[1], hop, np.zeros(par.K), np.zeros(par.K))
        assert np.linalg.norm(y[1].cpu().numpy() - ref["y"]) / np.linalg.norm(ref["y"]) <= TOL_L2


@pytest.mark.parametrize("n,hop", [(1024, 256), (2048, 512), (4096, 1024), (8192, 2048)])
def test_fused_analysis_accuracy_is_as_stated(gpu_lib, orc, n, hop):
    """The fused analysis kernels use the shorter polar conversion (no residual correction of quotient and square
    root, degree-8 arctangent).  Its stated accuracy -- magnitudes within 1e-13 of the frame maximum, raw phases
    within 1e-12 rad, per-hop phase advance within 1e-11 rad (it is formed at a few thousand radians, where
    one ulp is 5e-13) -- is four to five orders inside the 1e-9 bars of the path, and is held here against the
    CPU restatement of Vocoder.hs:158-173,217-218.  Phases are compared on bins of at least a hundredth of the
    frame's maximum: below that the two transforms' own rounding (1e-16 of the maximum) already moves a phase by
    more than the bar."""
    import ctypes as C
    import torch
    from vocoder_amd import vocoder as V
    win = _hann(n)
    par = V.VocoderParams(n, hop, win, lib=gpu_lib)
    c, frames = 2, 9
    length = n + hop * (frames - 1)
    xs = _signal(c, length)
    x = torch.from_numpy(xs).cuda()
    ldk = (par.K + 1) & ~1
    mag = torch.empty((c, frames, ldk), dtype=torch.float64, device="cuda")
    inc = torch.empty_like(mag)
    rng = np.random.default_rng(3)
    ph0 = rng.uniform(-3, 3, (c, par.K))
    aph = torch.from_numpy(ph0.copy()).cuda()
    p = lambda t: C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    par._call("voc_analysis_signal_dev", p(x), x.stride(0), frames, c, p(aph), p(mag), p(inc), ldk, 0, st)
    torch.cuda.synchronize()
    mag, inc, aph = mag.cpu().numpy()[:, :, :par.K], inc.cpu().numpy()[:, :, :par.K], aph.cpu().numpy()
    for ch in range(c):
        fr = np.stack([xs[ch, j * hop: j * hop + n] for j in range(frames)])
        pho, mago, inco = orc.analysis_stage(n, hop, win, ph0[ch], fr)
        top = mago.max(axis=-1, keepdims=True)
        assert (np.abs(mag[ch] - mago) <= 1e-13 * top).all()
        big = mago >= 1e-2 * top
        d = np.abs(inc[ch] * hop - inco * hop) % (2 * np.pi)
        assert np.minimum(d, 2 * np.pi - d)[big].max() <= 1e-11
        d = np.abs(aph[ch] - pho) % (2 * np.pi)
        assert np.minimum(d, 2 * np.pi - d)[big[-1]].max() <= 1e-12
    par.close()


class _ThreadComm:
    """The exchange step of `process_pieces` between ranks that are threads of one process (one GPU)."""

    def __init__(self, shared, rank):
        self.s, self.rank, self.world = shared, rank, shared["world"]

    def all_gather(self, total):
        self.s["totals"][self.rank] = total.clone()
        self.s["bar"].wait()
        out = [t.clone() for t in self.s["totals"]]
        self.s["bar"].wait()
        return out

    def neighbour_exchange(self, send, recv):
        if send is not None and self.rank + 1 < self.world:
            self.s["tails"][self.rank + 1] = send.clone()
        self.s["bar"].wait()
        if recv is not None:
            recv.copy_(self.s["tails"][self.rank])
        self.s["bar"].wait()

    def broadcast_from_last(self, tensors):
        if self.rank == self.world - 1:
            self.s["last"] = [t.clone() for t in tensors]
        self.s["bar"].wait()
        for t, v in zip(tensors, self.s["last"]):
            t.copy_(v)
        self.s["bar"].wait()


@pytest.mark.parametrize("chain", [[("envelope", 32)], [("pitch_shift", 2.0), ("lowpass_brickwall", 1000.0)], []],
                         ids=["envelope32", "pitch2+lowpass1000", "identity"])
def test_four_time_segments_with_a_filter_chain_match_the_oracle(gpu_lib, orc, chain):
    """`--envelope 32` (and `--pitchShiftInterpolate 2 --lowpassBrickwall 1000`) on a signal cut by time over four
    ranks, N = 4096, hop 1024, every rank walking its segment in runs of frames (`process_pieces`: the walk for the
    phase advance, the all-gather, the output walk, the tail exchange -- one walk for `envelope`, whose increments
    are zero, Filter.hs:73,144-152), against the sequential reference pipeline with that filter.  The ranks are
    four threads on the one GPU; the exchange goes through `_ThreadComm`."""
    import threading
    import torch
    from helpers import oracle_filtered
    from vocoder_amd import vocoder as V, timeshard as T, filter as Fl
    n, hop, world, piece, c = 4096, 1024, 4, 5, 2
    win = _hann(n)
    par = V.VocoderParams(n, hop, win, lib=gpu_lib)
    length = n + hop * (13 * world + 1) + 9
    x = _signal(c, length)
    xt = torch.from_numpy(x).cuda()
    f = Fl.id_filter()
    for name, *prm in chain:
        fn = {"pitch_shift": Fl.pitch_shift_interpolate, "envelope": Fl.envelope_filter}.get(name) or getattr(Fl, name)
        f = Fl.compose_filters(f, fn(*prm))
    filt = f.bind(par, 48000.0 / n) if chain else None
    shared = {"world": world, "bar": threading.Barrier(world), "totals": [None] * world, "tails": [None] * world, "last": None}
    results, reads, errors = [None] * world, [[] for _ in range(world)], []

    def rank_main(r):
        try:
            def source(lo, hi):
                reads[r].append((lo, hi))
                return xt[:, lo:hi].contiguous()
            results[r] = T.process_pieces(par, source, length, piece, filt=filt, comm=_ThreadComm(shared, r))
        except BaseException as e:      # a failed rank must not leave the others waiting at a barrier
            errors.append(e)
            shared["bar"].abort()

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    torch.cuda.synchronize()
    y = torch.cat([res[0] for res in results], dim=1).cpu().numpy()
    walks = 1 if any(s[0] == "envelope" for s in chain) else 2
    for r in range(world):
        sp = T.plan(par, length, world, r)
        assert len(reads[r]) == walks * len(T.pieces_of(par, sp, piece))
        assert torch.equal(results[r][1][0], results[0][1][0]) and torch.equal(results[r][1][1], results[0][1][1])
    for ch in range(c):
        ref = oracle_filtered(orc, n, hop, win, x[ch], 48000.0 / n, chain)
        assert y.shape[1] == ref["y"].size
        assert np.linalg.norm(y[ch] - ref["y"]) / np.linalg.norm(ref["y"]) <= TOL_L2
        assert _wrap_err(results[0][1][0][ch].cpu().numpy(), ref["ana_phase"]) <= 1e-9
        assert _wrap_err(results[0][1][1][ch].cpu().numpy(), ref["syn_phase"]) <= 1e-9
    par.close()
