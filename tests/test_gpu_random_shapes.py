"""GPU: a seeded sweep over frame length, window length, hop, channel count, signal length, output chunk and
tile cuts -- the combinations nobody wrote down by hand.  Every case runs the fused round trip and the stage
level through the C ABI against the oracle, to the bars of tests/helpers.py (framing, leftover and overlap-add
bit exact)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import helpers as Hh  # noqa: E402

pytestmark = pytest.mark.gpu


def _case(seed):
    rng = np.random.default_rng(1000 + seed)
    kind = seed % 4
    if kind == 0:      # the specialised kernels: W = N, hop a multiple of N/16 (sliding window, register overlap-add)
        n = int(rng.choice([1024, 2048, 4096, 8192, 16384]))
        w, h = n, int(n // 16 * rng.choice([1, 2, 4, 8]))
    elif kind == 1:    # the same lengths with a short window and an arbitrary hop (scalar loads, frames in memory)
        n = int(rng.choice([1024, 2048, 4096, 8192]))
        w = int(rng.integers(n // 2, n + 1))
        h = int(rng.integers(1, w + 1))
    elif kind == 2:    # small powers of two: the generic kernels
        n = int(rng.choice([8, 16, 32, 64, 128, 256, 512]))
        w = int(rng.integers(min(3, n), n + 1))  # (Hann windows of one or two points are all zero: volumeCoeff would be H / 0)
        h = int(rng.integers(1, w + 1))
    else:              # any other length: chirp-z in LDS
        n = int(rng.integers(2, 3000))
        w = int(rng.integers(min(3, n), n + 1))
        h = int(rng.integers(1, w + 1))
    c = int(rng.integers(1, 4))
    frames = int(rng.integers(0, 14))
    length = max(0, w + h * (frames - 1) + int(rng.integers(0, h + 1))) if frames else int(rng.integers(0, w))
    chunk = int(rng.choice([h, 1, 7, w, 1000]))
    at, st = (0, 0) if rng.integers(0, 2) else (int(rng.integers(1, 9)), int(rng.integers(1, 9)))
    return n, w, h, c, length, chunk, at, st


@pytest.mark.parametrize("seed", range(120))
def test_random_shapes_against_the_oracle(gpu_lib, orc, seed):
    n, w, h, c, length, chunk, at, st = _case(seed)
    rng = np.random.default_rng(seed)
    Hh.check_roundtrip(gpu_lib, orc, n, w, h, length, c, rng, ana_tile=at, syn_tile=st, chunk=chunk,
                       kind="tone" if seed % 3 else "noise")
    f = (length - w) // h + 1 if length >= w else 0
    if f > 0:
        Hh.check_stages(gpu_lib, orc, n, w, h, min(f, 6), c, rng, ana_tile=at, syn_tile=st)


@pytest.mark.parametrize("seed", range(40))
def test_random_shapes_single_precision(gpu_lib, orc, seed):
    """The same sweep on single-precision contexts (powers of two only), to the variant's bar against the
    double-precision oracle; leftover samples bit exact."""
    from vocoder_amd import vocoder as V, conduit as Cd
    rng = np.random.default_rng(5000 + seed)
    n = int(rng.choice([64, 256, 1024, 2048, 4096, 8192, 16384]))
    if seed % 2:
        w, h = n, int(max(1, n // 16) * rng.choice([1, 2, 4, 8]))
    else:
        w = int(rng.integers(max(3, n // 2), n + 1))
        h = int(rng.integers(max(1, w // 8), w + 1))
    c = int(rng.integers(1, 4))
    frames = int(rng.integers(1, 14))
    length = w + h * (frames - 1) + int(rng.integers(0, h + 1))
    at, st = (0, 0) if rng.integers(0, 2) else (int(rng.integers(1, 9)), int(rng.integers(1, 9)))
    Hh.set_tiles(at, st)
    try:
        win = orc.make_window("hann", w)
        par = V.VocoderParams(n, h, win, lib=gpu_lib, precision="f32")
        t = np.arange(length)
        x = (np.stack([0.5 * np.sin(2 * np.pi * 0.0173 * (i + 1) * t) for i in range(c)]) +
             0.01 * rng.uniform(-1, 1, (c, length))).astype(np.float32)
        y, left, _ = Cd.process_with(par, x)
        for ch in range(c):
            ref = orc.roundtrip(n, h, win, x[ch].astype(np.float64), h)
            assert y.shape[1] == ref["y"].size and np.array_equal(left[ch], ref["leftover"].astype(np.float32))
            assert np.linalg.norm(y[ch] - ref["y"]) <= 2e-4 * np.linalg.norm(ref["y"])
        par.close()
    finally:
        Hh.set_tiles()


@pytest.mark.parametrize("seed", range(40))
def test_random_streams_equal_the_batch_path(gpu_lib, seed):
    """A stream pushed in random pieces (empty ones included), with initial phases and -- every other case -- a
    filter chain with a speed change in it, gives the bits of the one-call path on the same context."""
    import torch
    from fractions import Fraction
    from vocoder_amd import vocoder as V, conduit as Cd, stream as S, filter as Fl
    n, w, h, c, length, chunk, _, _ = _case(seed)
    if length < w:
        length = w + 3 * h + 1
    rng = np.random.default_rng(9000 + seed)
    length += int(rng.integers(0, 6)) * h
    par = V.VocoderParams(n, h, np.hanning(w + 2)[1:-1], lib=gpu_lib)
    x = rng.standard_normal((c, length))
    cuts = np.concatenate([[0], np.sort(rng.integers(0, length + 1, int(rng.integers(0, 9)))), [length]])
    if seed % 2 == 0:
        a0, s0 = rng.uniform(-3, 3, (c, par.K)), rng.uniform(-3, 3, (c, par.K))
        y_ref, left_ref, (a_ref, s_ref) = Cd.process_with(par, x, phases=(a0, s0), chunk=chunk)
        st = S.Stream(par, c, chunk, ana_phase=a0, syn_phase=s0)
        y = np.concatenate([st.push(x[:, a:b]) for a, b in zip(cuts[:-1], cuts[1:])] + [st.flush()], axis=1)
        assert y.shape == y_ref.shape and np.array_equal(y, y_ref)
        a, s = st.phases()
        assert np.array_equal(a, a_ref) and np.array_equal(s, s_ref) and np.array_equal(st.leftover(), left_ref)
    else:
        speed = Fraction(int(rng.integers(1, 4)), int(rng.integers(1, 4)))
        descr = Fl.pitch_shift_interpolate(float(rng.uniform(0.5, 2.0))) >> Fl.play_speed(speed) >> Fl.random_phase_filter(seed)
        filt = descr.bind(par, 48000.0 / n)
        y_ref, _, s_ref = Cd.process_filter_dev(par, filt, torch.from_numpy(x).cuda())
        st = S.Stream(par, c, filt=filt)
        y = np.concatenate([st.push(x[:, a:b]) for a, b in zip(cuts[:-1], cuts[1:])] + [st.flush()], axis=1)
        assert np.array_equal(y, y_ref.cpu().numpy())
        assert np.array_equal(st.phases()[1], s_ref.cpu().numpy())
    par.close()


@pytest.mark.parametrize("precision", ["f64", "f32"])
@pytest.mark.parametrize("seed", range(24))
def test_random_device_layouts_give_the_bits_of_the_host_call(gpu_lib, seed, precision):
    """Device-resident calls whose rows start at odd element offsets, with odd row strides and a caller workspace
    too small for all channels at once, give the bits of the host-pointer call: which kernel a context runs depends
    on the context alone, never on a call's alignment, strides or grouping."""
    import torch
    from vocoder_amd import vocoder as V, conduit as Cd
    rng = np.random.default_rng(7000 + seed)
    n = int(rng.choice([256, 1024, 2048, 4096, 8192, 16384]))
    if seed % 3:
        w, h = n, int(max(1, n // 16) * rng.choice([1, 2, 4, 8]))
    else:
        w = int(rng.integers(n // 2, n + 1))
        h = int(rng.integers(max(1, w // 6), w + 1))
    c = int(rng.integers(2, 6))
    frames = int(rng.integers(1, 12))
    length = w + h * (frames - 1) + int(rng.integers(0, h + 1))
    chunk = int(rng.choice([h, 1, 7, 1000]))
    par = V.VocoderParams(n, h, np.hanning(w + 2)[1:-1], lib=gpu_lib, precision=precision)
    npdt, tdt = (np.float32, torch.float32) if precision == "f32" else (np.float64, torch.float64)
    x = rng.standard_normal((c, length)).astype(npdt)
    a0, s0 = rng.uniform(-3, 3, (c, par.K)).astype(npdt), rng.uniform(-3, 3, (c, par.K)).astype(npdt)
    y_ref, left_ref, (a_ref, s_ref) = Cd.process_with(par, x, phases=(a0, s0), chunk=chunk)
    # rows at an odd element offset inside a wider buffer with an odd stride
    xs, ys = length + 1 + 2 * int(rng.integers(0, 3)), y_ref.shape[1] + 1 + 2 * int(rng.integers(0, 3))
    xb = torch.zeros(c * xs + 1, dtype=tdt, device="cuda")
    yb = torch.full((c * ys + 1,), 7.0, dtype=tdt, device="cuda")
    xv = xb[1:].view(c, xs)[:, :length]
    yv = yb[1:].view(c, ys)[:, :y_ref.shape[1]]
    xv.copy_(torch.from_numpy(x))
    aph, sph = torch.from_numpy(a0).cuda(), torch.from_numpy(s0).cuda()
    left = torch.empty((c, max(left_ref.shape[1], 1)), dtype=tdt, device="cuda")
    full = Cd.roundtrip_workspace_bytes(par, length, c)
    ws = torch.empty(max(full * 2 // c, 1024) if seed % 2 else full, dtype=torch.uint8, device="cuda")   # odd seeds: room for two channels
    Cd.process_dev(par, xv, yv, aph, sph, length=length, chunk=chunk, leftover=left, workspace=ws)
    torch.cuda.synchronize()
    assert np.array_equal(yv.cpu().numpy(), y_ref)
    assert np.array_equal(aph.cpu().numpy(), a_ref) and np.array_equal(sph.cpu().numpy(), s_ref)
    assert np.array_equal(left.cpu().numpy()[:, :left_ref.shape[1]], left_ref)
    assert float(yb[0]) == 7.0 and bool((yb[1:].view(c, ys)[:, y_ref.shape[1]:] == 7.0).all())   # nothing outside the rows is written
    par.close()


@pytest.mark.parametrize("seed", range(40))
def test_random_stage_level_equals_the_fused_call(gpu_lib, seed):
    """framesOfE -> analysisStage -> volumeFix -> synthesisStage -> sumFramesE through the stage-level entry points
    gives the bits of the fused call, whatever kernels each of them runs (frames in memory against frames cut in
    place, overlap-add in memory against overlap-add in registers)."""
    from vocoder_amd import vocoder as V, conduit as Cd
    n, w, h, c, length, chunk, at, st = _case(seed)
    if length < w:
        length = w + 2 * h
    rng = np.random.default_rng(3000 + seed)
    Hh.set_tiles(at, st)
    try:
        par = V.VocoderParams(n, h, np.hanning(w + 2)[1:-1], lib=gpu_lib)
        x = rng.standard_normal((c, length))
        a0, s0 = rng.uniform(-3, 3, (c, par.K)), rng.uniform(-3, 3, (c, par.K))
        y_ref, left_ref, (a_ref, s_ref) = Cd.process_with(par, x, phases=(a0, s0), chunk=chunk)
        frames, left = Cd.frames_of_e(par, x)
        aph, (mag, inc) = V.analysis_stage(par, a0, frames)
        mag = mag * par.lib.dll.voc_volume_coeff(par.handle)               # volumeFix (Conduit.hs:34-35)
        sph, out = V.synthesis_stage(par, s0, (mag, inc))
        y = Cd.sum_frames_e(par, out, chunk)
        assert np.array_equal(left, left_ref) and np.array_equal(aph, a_ref) and np.array_equal(sph, s_ref)
        assert y.shape == y_ref.shape and np.array_equal(y, y_ref)
        par.close()
    finally:
        Hh.set_tiles()


def _random_chain(rng):
    ops = []
    for _ in range(int(rng.integers(1, 6))):
        kind = int(rng.integers(0, 9))
        lo, hi = sorted(rng.uniform(50.0, 20000.0, 2))
        if kind == 0:
            ops.append(("amplify", float(rng.uniform(0.1, 3.0))))
        elif kind == 1:
            ops.append((str(rng.choice(["lowpass_brickwall", "highpass_brickwall"])), float(lo)))
        elif kind == 2:
            ops.append((str(rng.choice(["bandpass_brickwall", "bandstop_brickwall"])), float(lo), float(hi)))
        elif kind == 3:
            ops.append((str(rng.choice(["lowpass_butterworth", "highpass_butterworth"])), float(rng.integers(1, 5)), float(lo)))
        elif kind == 4:
            ops.append((str(rng.choice(["bandpass_butterworth", "bandstop_butterworth"])), float(rng.integers(1, 4)), float(lo), float(hi)))
        elif kind == 5:
            ops.append(("pitch_shift", float(rng.uniform(0.4, 2.5))))
        elif kind == 6:
            ops.append(("convolution", tuple(rng.uniform(0.1, 1.0, int(rng.integers(1, 12))))))
        elif kind == 7:
            ops.append(("envelope", int(rng.integers(1, 40))))
        else:
            ops.append(("pitch_shift", float(rng.choice([0.5, 2.0]))))
    return ops


@pytest.mark.parametrize("seed", range(40))
def test_random_filter_chains_against_the_filter_oracle(gpu_lib, orc, seed):
    """Random chains of the per-frame filters (Filter.hs) on STFTs of random frame lengths, channel and frame counts,
    rows padded to an even length: bit exact against the filter oracle, except behind an envelope stage (log / exp)."""
    import torch
    from vocoder_amd import vocoder as V, filter as Fl
    rng = np.random.default_rng(11000 + seed)
    n = int(rng.choice([16, 64, 100, 1000, 1024, 3000, 4096, 8192, 16384, 20000])) if seed % 8 else 32768
    par = V.VocoderParams(n, max(1, n // 4), np.hanning(n + 2)[1:-1], lib=gpu_lib)
    chain = _random_chain(rng)
    f = Fl.id_filter()
    for name, *prm in chain:
        fn = {"pitch_shift": Fl.pitch_shift_interpolate, "convolution": Fl.convolution_filter,
              "envelope": Fl.envelope_filter}.get(name) or getattr(Fl, name)
        f = f >> fn(*prm)
    step = 48000.0 / n
    c, nf = int(rng.integers(1, 4)), int(rng.integers(1, 6))
    ldk = (par.K + 1) & ~1
    mag, inc = rng.uniform(0, 1, (c, nf, ldk)), rng.uniform(-1, 1, (c, nf, ldk))
    m, i = f.bind(par, step).apply_dev(torch.from_numpy(mag).cuda(), torch.from_numpy(inc).cuda())
    torch.cuda.synchronize()
    m, i = m.cpu().numpy(), i.cpu().numpy()
    chain_o = [(nm, np.array(p[0])) if nm == "convolution" else (nm, *p) for nm, *p in chain]
    mo, io = orc.apply_filter(step, chain_o, mag[..., :par.K], inc[..., :par.K])
    if any(s[0] == "envelope" for s in chain):
        assert np.allclose(m[..., :par.K], mo, rtol=1e-11, atol=1e-14)
    else:
        assert np.array_equal(m[..., :par.K], mo)
    assert np.array_equal(i[..., :par.K], io)
    assert np.array_equal(m[..., par.K:], mag[..., par.K:]) and np.array_equal(i[..., par.K:], inc[..., par.K:])
    par.close()


@pytest.mark.parametrize("seed", range(24))
def test_random_streams_single_precision(gpu_lib, seed):
    """The stream sweep on single-precision contexts: float state, the bits of the float one-call path."""
    import torch
    from fractions import Fraction
    from vocoder_amd import vocoder as V, conduit as Cd, stream as S, filter as Fl
    rng = np.random.default_rng(13000 + seed)
    n = int(rng.choice([64, 1024, 4096, 8192, 16384]))
    if seed % 3:
        w, h = n, int(max(1, n // 16) * rng.choice([1, 2, 4, 8]))
    else:
        w = int(rng.integers(max(3, n // 2), n + 1))
        h = int(rng.integers(max(1, w // 6), w + 1))
    c = int(rng.integers(1, 4))
    length = w + h * int(rng.integers(2, 12)) + int(rng.integers(0, h + 1))
    chunk = int(rng.choice([h, 1, 7, 1000]))
    par = V.VocoderParams(n, h, np.hanning(w + 2)[1:-1], lib=gpu_lib, precision="f32")
    x = rng.standard_normal((c, length)).astype(np.float32)
    cuts = np.concatenate([[0], np.sort(rng.integers(0, length + 1, int(rng.integers(0, 9)))), [length]])
    if seed % 2 == 0:
        a0 = rng.uniform(-3, 3, (c, par.K)).astype(np.float32)
        s0 = rng.uniform(-3, 3, (c, par.K)).astype(np.float32)
        y_ref, left_ref, (a_ref, s_ref) = Cd.process_with(par, x, phases=(a0, s0), chunk=chunk)
        st = S.Stream(par, c, chunk, ana_phase=a0, syn_phase=s0)
        y = np.concatenate([st.push(x[:, a:b]) for a, b in zip(cuts[:-1], cuts[1:])] + [st.flush()], axis=1)
        assert y.dtype == np.float32 and y.shape == y_ref.shape and np.array_equal(y, y_ref)
        a, s = st.phases()
        assert np.array_equal(a, a_ref) and np.array_equal(s, s_ref) and np.array_equal(st.leftover(), left_ref)
    else:
        speed = Fraction(int(rng.integers(1, 4)), int(rng.integers(1, 4)))
        descr = Fl.lowpass_butterworth(2.0, 4000.0) >> Fl.play_speed(speed) >> Fl.random_phase_filter(seed)
        filt = descr.bind(par, 48000.0 / n)
        y_ref, _, s_ref = Cd.process_filter_dev(par, filt, torch.from_numpy(x).cuda())
        st = S.Stream(par, c, filt=filt)
        y = np.concatenate([st.push(x[:, a:b]) for a, b in zip(cuts[:-1], cuts[1:])] + [st.flush()], axis=1)
        assert np.array_equal(y, y_ref.cpu().numpy())
        assert np.array_equal(st.phases()[1], s_ref.cpu().numpy())
    par.close()


@pytest.mark.parametrize("seed", range(24))
def test_random_time_shards_against_the_unsharded_run(gpu_lib, seed):
    """A signal cut by time over a random number of ranks (run one after the other here: halo frame, advances
    summed in order, tails added) against the unsharded call: equal up to the re-association at the rank
    boundaries, the last raw analysis phase bit for bit."""
    import torch
    from vocoder_amd import vocoder as V, conduit as Cd, timeshard as T
    rng = np.random.default_rng(15000 + seed)
    n = int(rng.choice([256, 1024, 2048, 4096, 8192, 16384]))
    if seed % 3:
        w, h = n, int(max(1, n // 16) * rng.choice([1, 2, 4, 8]))
    else:
        w = int(rng.integers(n // 2, n + 1))
        h = int(rng.integers(max(1, w // 6), w + 1))
    world = int(rng.integers(2, 6))
    c = int(rng.integers(1, 4))
    per_rank = -(-w // h) + 1                       # a rank's segment is at least as long as the overlap W - H
    frames = int(rng.integers(world * per_rank, world * (per_rank + 6)))
    length = w + h * (frames - 1) + int(rng.integers(0, h + 1))
    par = V.VocoderParams(n, h, np.hanning(w + 2)[1:-1], lib=gpu_lib)
    t = np.arange(length)
    x = np.stack([0.5 * np.sin(2 * np.pi * (0.01 + 0.003 * i) * t + i) for i in range(c)]) + 1e-2 * rng.uniform(-1, 1, (c, length))
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
    f = Cd.num_frames(par, length)
    y_seq = torch.empty((c, Cd.ola_length(par, f, h)), dtype=torch.float64, device="cuda")
    aph = torch.zeros((c, par.K), dtype=torch.float64, device="cuda")
    sph = torch.zeros_like(aph)
    Cd.process_dev(par, xt, y_seq, aph, sph)
    torch.cuda.synchronize()
    assert y.shape == y_seq.shape
    assert (torch.linalg.norm(y - y_seq) / torch.linalg.norm(y_seq)).item() <= 1e-11
    assert torch.equal(ana[-1][2], aph)
    d = np.abs(outs[-1][1].cpu().numpy() - sph.cpu().numpy()) % (2 * np.pi)
    assert np.minimum(d, 2 * np.pi - d).max() <= 1e-8
    par.close()


@pytest.mark.parametrize("seed", range(12))
def test_random_long_lengths_stream_stage_and_fused_agree(gpu_lib, seed):
    """Frame lengths between 4097 and 32768 that take the chirp-z transform through device memory: the fused
    call, the stage-level chain and a stream pushed in pieces give the same bits, on rows at odd offsets too."""
    import torch
    from vocoder_amd import vocoder as V, conduit as Cd, stream as S
    rng = np.random.default_rng(19000 + seed)
    n = int(rng.integers(4097, 32769)) if seed % 4 else int(rng.choice([32768, 16385, 24576, 5000]))
    w = n if seed % 2 else int(rng.integers(n // 2, n + 1))
    h = int(rng.integers(max(1, w // 5), w + 1))
    c = int(rng.integers(1, 3))
    length = w + h * int(rng.integers(1, 6)) + int(rng.integers(0, h + 1))
    chunk = int(rng.choice([h, 7, 1000]))
    par = V.VocoderParams(n, h, np.hanning(w + 2)[1:-1], lib=gpu_lib)
    x = rng.standard_normal((c, length))
    a0, s0 = rng.uniform(-3, 3, (c, par.K)), rng.uniform(-3, 3, (c, par.K))
    y_ref, left_ref, (a_ref, s_ref) = Cd.process_with(par, x, phases=(a0, s0), chunk=chunk)
    cuts = np.concatenate([[0], np.sort(rng.integers(0, length + 1, 3)), [length]])
    st = S.Stream(par, c, chunk, ana_phase=a0, syn_phase=s0)
    y = np.concatenate([st.push(x[:, a:b]) for a, b in zip(cuts[:-1], cuts[1:])] + [st.flush()], axis=1)
    assert np.array_equal(y, y_ref) and np.array_equal(st.leftover(), left_ref)
    a, s = st.phases()
    assert np.array_equal(a, a_ref) and np.array_equal(s, s_ref)
    frames, left = Cd.frames_of_e(par, x)
    aph, (mag, inc) = V.analysis_stage(par, a0, frames)
    sph, out = V.synthesis_stage(par, s0, (mag * par.lib.dll.voc_volume_coeff(par.handle), inc))
    assert np.array_equal(aph, a_ref) and np.array_equal(sph, s_ref)
    assert np.array_equal(Cd.sum_frames_e(par, out, chunk), y_ref)
    xb = torch.zeros(c * (length + 3) + 1, dtype=torch.float64, device="cuda")
    xv = xb[1:].view(c, length + 3)[:, :length]
    xv.copy_(torch.from_numpy(x))
    yb = torch.zeros(c * (y_ref.shape[1] + 1) + 1, dtype=torch.float64, device="cuda")
    yv = yb[1:].view(c, y_ref.shape[1] + 1)[:, :y_ref.shape[1]]
    ad, sd = torch.from_numpy(a0).cuda(), torch.from_numpy(s0).cuda()
    Cd.process_dev(par, xv, yv, ad, sd, length=length, chunk=chunk)
    torch.cuda.synchronize()
    assert np.array_equal(yv.cpu().numpy(), y_ref) and np.array_equal(sd.cpu().numpy(), s_ref)
    par.close()


def test_host_threads_share_one_context(gpu_lib):
    """Several host threads on ONE context at once (the reference leaves this open, Vocoder.hs:98; vocoder-jack runs
    its processing in a forked thread): calls serialise inside the library and every thread gets the bits it would
    have got alone -- host-pointer round trips, single frames through analysisBlock / synthesisBlock, a stream
    each, and a filter chain."""
    import threading
    from vocoder_amd import vocoder as V, conduit as Cd, stream as S, filter as Fl
    n, h = 4096, 1024
    par = V.VocoderParams(n, h, np.hanning(n + 2)[1:-1], lib=gpu_lib)
    filt = (Fl.pitch_shift_interpolate(1.25) >> Fl.lowpass_butterworth(2.0, 3000.0)).bind(par, 48000.0 / n)
    rng = np.random.default_rng(23)
    jobs = [rng.standard_normal((2, n + h * int(rng.integers(3, 30)) + 5)) for _ in range(8)]

    def work(x):
        y, left, (a, s) = Cd.process_with(par, x)
        ph, (mag, inc) = V.analysis_block(par, np.zeros(par.K), x[0, :n])
        m2, i2 = filt.apply(mag[None, None], inc[None, None])
        sp, fr = V.synthesis_block(par, np.zeros(par.K), (m2[0, 0], i2[0, 0]))
        st = S.Stream(par, 2)
        ys = np.concatenate([st.push(x[:, :777]), st.push(x[:, 777:]), st.flush()], axis=1)
        return y, a, s, ph, fr, ys

    alone = [work(x) for x in jobs]
    got = [None] * len(jobs)
    errs = []

    def run(i):
        try:
            for _ in range(3):
                got[i] = work(jobs[i])
        except Exception as e:       # noqa: BLE001
            errs.append(e)

    threads = [threading.Thread(target=run, args=(i,)) for i in range(len(jobs))]
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert not errs, errs
    for a, g in zip(alone, got):
        assert all(np.array_equal(u, v) for u, v in zip(a, g))
    par.close()


@pytest.mark.parametrize("seed", range(32))
def test_random_chains_folded_into_synthesis(gpu_lib, monkeypatch, seed):
    """voc_synthesis_filtered_dev: `pitch shift >> gain` chains are applied by synthesis to the rows it reads, every
    other chain first and in place -- either way the bits of the filter pass followed by plain synthesis, on
    contexts that have the specialised kernels and on those that do not, with few channels (tiles and the carry
    pass on pitch-shifted increments) and many."""
    import torch
    from vocoder_amd import vocoder as V, conduit as Cd, filter as Fl
    rng = np.random.default_rng(21000 + seed)
    n = int(rng.choice([1024, 2048, 4096, 8192, 16384, 1000]))
    if seed % 4:
        w, h = n, max(1, n // int(rng.choice([2, 4, 8])))
    else:
        w = int(rng.integers(max(3, n // 2), n + 1))
        h = int(rng.integers(max(1, w // 6), w + 1))
    kind = seed % 5
    if kind == 0:
        descr = Fl.pitch_shift_interpolate(2.0) >> Fl.lowpass_brickwall(1000.0)
    elif kind == 1:
        descr = Fl.pitch_shift_interpolate(float(rng.uniform(0.4, 2.5)))
    elif kind == 2:
        descr = Fl.bandpass_butterworth(2.0, 300.0, 3400.0)
    elif kind == 3:
        descr = Fl.pitch_shift_interpolate(0.5) >> Fl.amplify(1.7)
    else:
        descr = Fl.lowpass_brickwall(4000.0) >> Fl.pitch_shift_interpolate(1.5) >> Fl.amplify(0.5)   # not foldable: applied first
    c = int(rng.choice([1, 3, 40]))
    frames = int(rng.integers(2, 40))
    length = w + h * (frames - 1) + int(rng.integers(0, h + 1))
    if seed % 2:
        monkeypatch.setenv("VOC_SYN_SLOTS", "16")       # few slots: 40 channels leave a partial round, cut into tiles
    par = V.VocoderParams(n, h, np.hanning(w + 2)[1:-1], lib=gpu_lib)
    filt = descr.bind(par, 48000.0 / n)
    x = torch.from_numpy(rng.standard_normal((c, length))).cuda()
    s0 = torch.from_numpy(rng.uniform(-3, 3, (c, par.K))).cuda()
    y, _, sph = Cd.process_filter_dev(par, filt, x, syn_phase=s0.clone())
    monkeypatch.setenv("VOC_NO_FILTERED_SYNTHESIS", "1")
    y_ref, _, sph_ref = Cd.process_filter_dev(par, filt, x, syn_phase=s0.clone())
    torch.cuda.synchronize()
    assert torch.equal(y, y_ref) and torch.equal(sph, sph_ref)
    par.close()
