"""Time-sharded round trip on CPU: segment planning, the three per-rank phases run rank after rank
in one process, and the same thing over two gloo ranks with the real all-gather / neighbour
exchange.  Arithmetic runs through the host-emulated build of the kernel source; the checker is
the sequential oracle over the whole signal."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from helpers import TOL_L2  # noqa: E402


def _hann(n):
    return 0.5 * (1 - np.cos(2 * np.pi * np.arange(n) / n))


def _signal(c, length, seed=7):
    rng = np.random.default_rng(seed)
    t = np.arange(length)
    return np.stack([0.5 * np.sin(2 * np.pi * (0.01 + 0.003 * i) * t + i) for i in range(c)]) + \
        1e-2 * rng.uniform(-1, 1, (c, length))


def _wrap_err(a, b):
    d = np.abs(a - b) % (2 * np.pi)
    return np.minimum(d, 2 * np.pi - d).max()


def test_plan_covers_signal_once(emu_lib):
    from vocoder_amd import vocoder as V, timeshard as T
    par = V.VocoderParams(64, 16, _hann(64), lib=emu_lib)
    length = 64 + 16 * 40 + 5
    for world in (1, 2, 3, 4):
        plans = [T.plan(par, length, world, r) for r in range(world)]
        assert plans[0].out_lo == 0 and plans[0].in_lo == 0 and not plans[0].has_prev
        for a, b in zip(plans, plans[1:]):
            assert a.f_hi == b.f_lo and a.out_hi == b.out_lo
            assert b.in_lo == b.f_lo * 16 - 16 and b.has_prev
            assert a.tail == 48 and a.local_len == (a.frames - 1) * 16 + 64
        assert plans[-1].f_hi == 41 and plans[-1].tail == 0
        assert plans[-1].out_hi == emu_lib.dll.voc_ola_length(par.handle, 41, 16)
        assert plans[-1].in_hi == 40 * 16 + 64
    with pytest.raises(ValueError):
        T.plan(par, 64 + 16 * 3, 4, 0)      # one frame per rank: shorter than the 48-sample overlap


@pytest.mark.parametrize("n,w,hop,world", [(64, 64, 16, 3), (64, 48, 16, 2), (32, 32, 32, 2), (1024, 1024, 256, 2)])
def test_segments_in_one_process_match_sequential_oracle(emu_lib, orc, n, w, hop, world):
    import torch
    from vocoder_amd import vocoder as V, timeshard as T
    win = _hann(w)
    par = V.VocoderParams(n, hop, win, lib=emu_lib)
    c, length = 2, w + hop * (9 * world + 1) + 3
    x = _signal(c, length)
    rng = np.random.default_rng(1)
    aph0, sph0 = rng.uniform(-3, 3, (c, par.K)), rng.uniform(-3, 3, (c, par.K))
    plans = [T.plan(par, length, world, r) for r in range(world)]
    xt = torch.from_numpy(x)
    ana = [T.analyse(par, sp, xt[:, sp.in_lo:sp.in_hi].contiguous(), torch.from_numpy(aph0)) for sp in plans]
    totals = [T.local_advance(par, inc) for _, inc, _ in ana]
    outs = [T.synthesise(par, sp, m, i, T.start_phase(totals, sp.rank, torch.from_numpy(sph0)))
            for sp, (m, i, _) in zip(plans, ana)]
    ys = []
    for r, (sp, (y, _)) in enumerate(zip(plans, outs)):
        if r > 0 and plans[r - 1].tail:
            prev_y = outs[r - 1][0]
            T.add_incoming_tail(y, prev_y[:, plans[r - 1].out_hi - plans[r - 1].out_lo:].clone())
        ys.append(y[:, :sp.out_hi - sp.out_lo].numpy())
    y = np.concatenate(ys, axis=1)

    for ch in range(c):
        ref = orc.roundtrip(n, hop, win, x[ch], hop, aph0[ch], sph0[ch])
        assert y.shape[1] == ref["y"].size
        err = np.linalg.norm(y[ch] - ref["y"]) / np.linalg.norm(ref["y"])
        assert err <= TOL_L2, err
        assert _wrap_err(ana[-1][2][ch].numpy(), ref["ana_phase"]) <= 1e-9
        # accumulated phase: re-associated once per rank boundary, compared modulo 2 pi
        assert _wrap_err(outs[-1][1][ch].numpy(), ref["syn_phase"]) <= 1e-9


def _worker(rank, world_size, port, emu_path, out_dir, n, hop, length):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world_size),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import torch
    from vocoder_amd import distributed as D, vocoder as V, timeshard as T
    from vocoder_amd._ffi import Library
    assert D.init("gloo") == world_size
    par = V.VocoderParams(n, hop, _hann(n), lib=Library(emu_path))
    x = torch.from_numpy(_signal(3, length))
    sp = T.plan(par, length, world_size, rank)
    y, (aph, sph) = T.process(par, x[:, sp.in_lo:sp.in_hi].contiguous(), length)
    np.savez(os.path.join(out_dir, f"ts{rank}.npz"), y=y.numpy(), aph=aph.numpy(), sph=sph.numpy(), lo=sp.out_lo)
    D.finish()


def test_two_gloo_ranks_shard_time(emu_lib, orc, tmp_path):
    import torch.multiprocessing as mp
    n, hop = 64, 16
    length = 64 + 16 * 30 + 7
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, emu_lib.path, str(tmp_path), n, hop, length), nprocs=2, join=True)
    parts = [np.load(tmp_path / f"ts{r}.npz") for r in range(2)]
    assert int(parts[0]["lo"]) == 0 and int(parts[1]["lo"]) == parts[0]["y"].shape[1]
    y = np.concatenate([p["y"] for p in parts], axis=1)
    x = _signal(3, length)
    for ch in range(3):
        ref = orc.roundtrip(n, hop, _hann(n), x[ch], hop, np.zeros(33), np.zeros(33))
        assert y.shape[1] == ref["y"].size
        assert np.linalg.norm(y[ch] - ref["y"]) / np.linalg.norm(ref["y"]) <= TOL_L2
        for p in parts:                                   # the final phases reached both ranks
            assert _wrap_err(p["aph"][ch], ref["ana_phase"]) <= 1e-9
            assert _wrap_err(p["sph"][ch], ref["syn_phase"]) <= 1e-9
    assert np.array_equal(parts[0]["sph"], parts[1]["sph"]) and np.array_equal(parts[0]["aph"], parts[1]["aph"])


def _bound(par, chain, step):
    from vocoder_amd import filter as Fl
    f = Fl.id_filter()
    for name, *prm in chain:
        fn = {"pitch_shift": Fl.pitch_shift_interpolate, "envelope": Fl.envelope_filter}.get(name) or getattr(Fl, name)
        f = Fl.compose_filters(f, fn(*prm))
    return f.bind(par, step)


PIECE_CHAINS = [[], [("pitch_shift", 1.5), ("lowpass_brickwall", 9000.0)], [("envelope", 5)]]


def test_pieces_cover_a_segment(emu_lib):
    from vocoder_amd import vocoder as V, timeshard as T
    par = V.VocoderParams(64, 16, _hann(64), lib=emu_lib)
    sp = T.plan(par, 64 + 16 * 40, 2, 1)
    for want in (1, 3, 4, 7, 1000):
        runs = T.pieces_of(par, sp, want)
        assert runs[0][0] == sp.f_lo and runs[-1][1] == sp.f_hi
        assert all(a[1] == b[0] for a, b in zip(runs, runs[1:]))
        assert all((hi - lo) * 16 >= 48 or len(runs) == 1 for lo, hi in runs)     # a run holds its predecessor's overlap
    assert len(T.pieces_of(par, sp, 1)) == sp.frames // 3                          # runs of one frame are too short: 3 each


@pytest.mark.parametrize("chain", PIECE_CHAINS, ids=lambda c: "+".join(s[0] for s in c) or "identity")
@pytest.mark.parametrize("n,w,hop,piece", [(64, 64, 16, 5), (64, 48, 16, 4), (32, 32, 32, 1), (64, 24, 32, 2)])
def test_one_rank_in_pieces_matches_oracle(emu_lib, orc, n, w, hop, piece, chain):
    """A single rank walking its signal in runs of a few frames: analysis and synthesis phases are
    carried exactly; only the overlap-add across a run boundary is re-associated."""
    import torch
    from helpers import oracle_filtered
    from vocoder_amd import vocoder as V, timeshard as T
    win = _hann(w)
    par = V.VocoderParams(n, hop, win, lib=emu_lib)
    c, length = 2, w + hop * 23 + 3
    x = _signal(c, length)
    xt = torch.from_numpy(x)
    step = 48000.0 / n
    filt = _bound(par, chain, step) if chain else None
    got = []
    y, (aph, sph) = T.process_pieces(par, lambda lo, hi: xt[:, lo:hi].contiguous(), length, piece, filt=filt)
    _, (aph2, sph2) = T.process_pieces(par, lambda lo, hi: xt[:, lo:hi].contiguous(), length, piece, filt=filt,
                                       sink=lambda lo, t: got.append((lo, t.clone())))
    pos = 0
    for lo, t in got:                                   # one rank: runs arrive in order, nothing is held back
        assert lo == pos
        pos += t.shape[1]
    assert torch.equal(torch.cat([t for _, t in got], dim=1), y) and torch.equal(sph, sph2)
    for ch in range(c):
        ref = oracle_filtered(orc, n, hop, win, x[ch], step, chain)
        assert y.shape[1] == ref["y"].size
        assert np.linalg.norm(y[ch].numpy() - ref["y"]) / np.linalg.norm(ref["y"]) <= TOL_L2
        assert _wrap_err(aph[ch].numpy(), ref["ana_phase"]) <= 1e-9
        assert _wrap_err(sph[ch].numpy(), ref["syn_phase"]) <= 1e-9


def test_too_short_for_a_frame_in_pieces(emu_lib):
    import torch
    from vocoder_amd import vocoder as V, timeshard as T
    par = V.VocoderParams(64, 16, _hann(64), lib=emu_lib)
    xt = torch.zeros((2, 10), dtype=torch.float64)
    y, (aph, sph) = T.process_pieces(par, lambda lo, hi: xt[:, lo:hi], 10, 4)
    assert y.shape == (2, 0) and aph.shape == (2, 33) and not sph.any()


def _piece_worker(rank, world_size, port, emu_path, out_dir, n, hop, length, chain, piece):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world_size),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    from vocoder_amd import distributed as D, vocoder as V, timeshard as T
    from vocoder_amd._ffi import Library
    assert D.init("gloo") == world_size
    par = V.VocoderParams(n, hop, _hann(n), lib=Library(emu_path))
    x = torch.from_numpy(_signal(2, length))
    filt = _bound(par, chain, 48000.0 / n) if chain else None
    reads, got = [], []

    def source(lo, hi):
        reads.append((lo, hi))
        return x[:, lo:hi].contiguous()

    sp = T.plan(par, length, world_size, rank)
    _, (aph, sph) = T.process_pieces(par, source, length, piece, sink=lambda lo, t: got.append((lo, t.clone())),
                                     filt=filt)
    y = torch.zeros((2, sp.out_hi - sp.out_lo), dtype=torch.float64)
    seen = torch.zeros(sp.out_hi - sp.out_lo, dtype=torch.int64)
    for lo, t in got:
        y[:, lo - sp.out_lo:lo - sp.out_lo + t.shape[1]] = t
        seen[lo - sp.out_lo:lo - sp.out_lo + t.shape[1]] += 1
    assert bool((seen == 1).all())                      # every owned sample delivered exactly once
    np.savez(os.path.join(out_dir, f"tp{rank}.npz"), y=y.numpy(), aph=aph.numpy(), sph=sph.numpy(), lo=sp.out_lo,
             reads=np.array(reads), span=max(hi - lo for lo, hi in reads))
    D.finish()


@pytest.mark.parametrize("chain", PIECE_CHAINS, ids=lambda c: "+".join(s[0] for s in c) or "identity")
def test_two_gloo_ranks_walk_their_segments_in_pieces(emu_lib, orc, tmp_path, chain):
    import torch.multiprocessing as mp
    from helpers import oracle_filtered
    n, hop, piece = 64, 16, 6
    length = 64 + 16 * 45 + 7
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_piece_worker, args=(2, port, emu_lib.path, str(tmp_path), n, hop, length, chain, piece), nprocs=2,
             join=True)
    parts = [np.load(tmp_path / f"tp{r}.npz") for r in range(2)]
    y = np.concatenate([p["y"] for p in parts], axis=1)
    x = _signal(2, length)
    for p in parts:
        assert int(p["span"]) <= (piece + 1) * hop + n                 # never more than one run (+ the halo hop) resident
        walks = 1 if any(s[0] == "envelope" for s in chain) else 2     # zero increments: no pass for the advance
        assert len(p["reads"]) == walks * 4                            # 23 frames in runs of <= 6: 4 runs per walk
    for ch in range(2):
        ref = oracle_filtered(orc, n, hop, _hann(n), x[ch], 48000.0 / n, chain)
        assert y.shape[1] == ref["y"].size
        assert np.linalg.norm(y[ch] - ref["y"]) / np.linalg.norm(ref["y"]) <= TOL_L2
        for p in parts:
            assert _wrap_err(p["aph"][ch], ref["ana_phase"]) <= 1e-9
            assert _wrap_err(p["sph"][ch], ref["syn_phase"]) <= 1e-9
