"""Chunk-at-a-time streaming through the C ABI on CPU (host-emulated kernels): pushing a signal in
pieces gives, concatenated, the bits of one call over the whole signal; chunks are published when
sumFramesE's state machine (Frames.hs:49-58) publishes them; the state that remains is the
reference's (leftover, phases) triple."""
import os
import sys

import numpy as np
import pytest

import helpers as Hh

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from helpers import TOL_L2  # noqa: E402


def _hann(n):
    return 0.5 * (1 - np.cos(2 * np.pi * np.arange(n) / n))


def published_after(frames, chunk, hop):
    """Chunks sumFramesE has yielded once `frames` frames were awaited (Frames.hs:49-58)."""
    sofar = pub = 0
    for _ in range(frames):
        while sofar >= chunk:          # process2: publish while a whole chunk lies before the new frame
            pub += 1
            sofar -= chunk
        sofar += hop
    return pub


def _pieces(rng, total, kind):
    if kind == "one":
        return [total]
    if kind == "tiny":
        cuts = np.sort(rng.integers(0, total + 1, 40))
    elif kind == "hop":
        cuts = np.arange(16, total, 16)
    else:
        cuts = np.sort(rng.integers(0, total + 1, 6))
    edges = np.concatenate([[0], cuts, [total]])
    return list(np.diff(edges))


@pytest.mark.parametrize("n,w,hop,chunk", [(64, 64, 16, 16), (64, 48, 16, 40), (64, 64, 16, 200), (32, 32, 32, 32),
                                           (32, 16, 24, 7), (1024, 1024, 256, 256)])
@pytest.mark.parametrize("kind", ["one", "few", "tiny", "hop"])
def test_pushes_concatenate_to_the_batch_result(emu_lib, orc, n, w, hop, chunk, kind):
    from vocoder_amd import vocoder as V, conduit as Cd, stream as S
    par = V.VocoderParams(n, hop, _hann(w), lib=emu_lib)
    rng = np.random.default_rng(n + w + hop + chunk)
    c, total = 2, w + hop * 23 + 5
    x = rng.standard_normal((c, total))
    a0, s0 = rng.uniform(-3, 3, (c, par.K)), rng.uniform(-3, 3, (c, par.K))
    y_ref, left_ref, (a_ref, s_ref) = Cd.process_with(par, x, phases=(a0, s0), chunk=chunk)

    st = S.Stream(par, c, chunk, ana_phase=a0, syn_phase=s0)
    outs, pos, frames = [], 0, 0
    buffered = 0
    pieces = _pieces(rng, total, kind)
    for m in pieces:
        y = st.push(x[:, pos:pos + m])
        pos += m
        buffered += m
        f_new = 0 if buffered < w else (buffered - w) // hop + 1
        frames += f_new
        if f_new:
            buffered = max(buffered - f_new * hop, 0)       # Seq.drop past the end leaves nothing (Frames.hs:36-37)
        assert y.shape[1] % chunk == 0
        outs.append(y)
        assert sum(o.shape[1] for o in outs) == published_after(frames, chunk, hop) * chunk
    outs.append(st.flush())
    y = np.concatenate(outs, axis=1)
    # the framing state machine of the oracle, fed the same pieces, cuts the same frames
    fr, left = orc.gen_frames_of_e(w, hop, np.split(x[0], np.cumsum(pieces)[:-1]))
    assert fr.shape[0] == frames and np.array_equal(st.leftover()[0], left)
    if hop > w and kind != "one":
        # a hop longer than the window drops samples between frames; when a piece ends inside such
        # a gap the reference restarts its hop grid at the next piece, so the result legitimately
        # depends on the pieces and is not the one-piece result
        assert y.shape[1] == -(-((frames - 1) * hop + w) // chunk) * chunk
        return
    assert y.shape == y_ref.shape
    assert np.array_equal(y, y_ref)
    a, s = st.phases()
    assert np.array_equal(a, a_ref) and np.array_equal(s, s_ref)
    assert np.array_equal(st.leftover(), left_ref)


def test_stream_with_filter_matches_filtered_pipeline(emu_lib):
    import torch
    from vocoder_amd import vocoder as V, conduit as Cd, stream as S, filter as Fl
    par = V.VocoderParams(128, 32, _hann(128), lib=emu_lib)
    filt = (Fl.pitch_shift_interpolate(1.5) >> Fl.lowpass_butterworth(2.0, 4000.0)).bind(par, 48000.0 / 128)
    rng = np.random.default_rng(5)
    x = rng.standard_normal((1, 128 + 32 * 30 + 3))
    y_ref, _, s_ref = Cd.process_filter_dev(par, filt, torch.from_numpy(x))
    st = S.Stream(par, 1, filt=filt)
    outs = [st.push(x[:, :200]), st.push(x[:, 200:777]), st.push(x[:, 777:]), st.flush()]
    assert np.array_equal(np.concatenate(outs, axis=1), y_ref.numpy())
    assert np.array_equal(st.phases()[1], s_ref.numpy())


def test_stream_argument_errors(emu_lib):
    import ctypes as C
    from vocoder_amd import vocoder as V, stream as S, filter as Fl
    par = V.VocoderParams(64, 16, _hann(64), lib=emu_lib)
    d = emu_lib.dll
    h = C.c_void_p()
    assert d.voc_stream_create(par.handle, 0, 16, None, C.byref(h)) == -1
    assert d.voc_stream_create(par.handle, 1, 0, None, C.byref(h)) == -1
    st = S.Stream(par, 1)
    x = np.zeros((1, 200))
    got = C.c_int64(0)
    need = d.voc_stream_push_length(st._h, 200)
    assert need == ((200 - 64) // 16) * 16
    y = np.zeros((1, 8))
    assert d.voc_stream_push(st._h, x.ctypes.data_as(C.c_void_p), 200, 200, y.ctypes.data_as(C.c_void_p), 8, 8,
                             C.byref(got)) == -5          # output buffer too small: nothing consumed
    assert got.value == need and d.voc_stream_push_length(st._h, 200) == need
    assert st.push(x).shape == (1, need)
    st.flush()
    with pytest.raises(V.VocoderError):
        st.push(x)                                        # a flushed stream takes no more input
    with pytest.raises(ValueError):
        S.Stream(par, 1, filt=(Fl.play_speed(2) >> Fl.play_speed(2)).bind(par, 1.0))   # one play_speed at most
    with pytest.raises(V.VocoderError):
        S.Stream(par, 1, filt=Fl.play_speed(2).bind(par, 1.0), dunai=True)            # conduit conventions only
    sp = S.Stream(par, 1, filt=Fl.play_speed(2).bind(par, 1.0))
    sp.push(np.zeros((1, 100)))
    assert d.voc_stream_set_play_speed(sp._h, 3, 2, None) == -1                       # only before the first frame


def _oracle_dunai(orc, n, hop, window, ticks):
    """Vocoder.Dunai.process par (arr id) on one channel, from oracle pieces (Dunai.hs:46-48):
    framesOfS >>> analysis >>> synthesis, sumFramesWithLengthS, volumeFix on the samples."""
    w = len(window)
    k = n // 2 + 1
    per_tick = orc.frames_of_s(w, hop, ticks)
    aph, sph = np.zeros(k), np.zeros(k)
    out_frames = []
    for fr in per_tick:
        if fr.shape[0] == 0:
            out_frames.append(np.empty((0, w)))
            continue
        aph, mag, inc = orc.analysis_stage(n, hop, window, aph, fr)
        sph, res = orc.synthesis_stage(n, hop, window, sph, mag, inc)
        out_frames.append(res)
    outs = orc.sum_frames_s(hop, out_frames, [t.size for t in ticks])
    vc = orc.volume_coeff(hop, window)
    return [o * vc for o in outs], aph, sph


@pytest.mark.parametrize("n,w,hop", [(64, 64, 16), (64, 40, 16), (32, 32, 32), (1024, 1024, 256)])
def test_dunai_mode_matches_oracle_ticks(emu_lib, orc, n, w, hop):
    from vocoder_amd import vocoder as V, stream as S
    win = _hann(w)
    par = V.VocoderParams(n, hop, win, lib=emu_lib)
    rng = np.random.default_rng(n + w)
    sizes = [hop * m for m in (3, 1, 0, 5, 2, 8)]
    ticks = [rng.standard_normal(m) for m in sizes]
    ref, aph, sph = _oracle_dunai(orc, n, hop, win, ticks)
    st = S.Stream(par, 1, dunai=True)
    for t, r in zip(ticks, ref):
        y = st.push(t)
        assert y.shape == (1, t.size)                      # as many samples out as in
        if t.size:
            scale = max(np.abs(np.concatenate(ref)).max(), 1e-300)
            assert np.abs(y[0] - r).max() <= 1e-12 * scale
    a, s = st.phases()
    assert np.abs(a[0] - aph).max() <= 1e-9
    d = np.abs(s[0] - sph) % (2 * np.pi)
    assert np.minimum(d, 2 * np.pi - d).max() <= 1e-9
    with pytest.raises(V.VocoderError):
        st.push(np.zeros(hop + 1))                        # not a multiple of the hop (Dunai.hs:45)


def test_concatenated_sources_with_their_own_filters(emu_lib, orc):
    """concatenateV (Audio.hs:60-69): the leftover and phases of source 1 flow into source 2, each
    source runs its own filter, all frames feed one overlap-add."""
    from vocoder_amd import vocoder as V, stream as S, filter as Fl
    n, hop = 128, 32
    win = _hann(n)
    par = V.VocoderParams(n, hop, win, lib=emu_lib)
    step = 48000.0 / n
    rng = np.random.default_rng(21)
    x1, x2 = rng.standard_normal(n + hop * 9 + 7), rng.standard_normal(hop * 11 + 3)
    f1 = Fl.lowpass_brickwall(6000.0).bind(par, step)
    f2 = Fl.pitch_shift_interpolate(1.25).bind(par, step)
    st = S.Stream(par, 1, filt=f1)
    outs = [st.push(x1)]
    st.set_filter(f2)
    outs += [st.push(x2), st.flush()]
    y = np.concatenate(outs, axis=1)[0]

    # oracle: frames of (x1), then frames of (leftover1 ++ x2); one phase chain; one overlap-add
    k = n // 2 + 1
    fr1, left1 = orc.gen_frames_of_e(n, hop, [x1])
    fr2, _ = orc.gen_frames_of_e(n, hop, [x2], q0=left1)
    vc = orc.volume_coeff(hop, win)
    aph, m1, i1 = orc.analysis_stage(n, hop, win, np.zeros(k), fr1)
    aph, m2, i2 = orc.analysis_stage(n, hop, win, aph, fr2)
    m1, i1 = orc.apply_filter(step, [("lowpass_brickwall", 6000.0)], m1 * vc, i1)
    m2, i2 = orc.apply_filter(step, [("pitch_shift", 1.25)], m2 * vc, i2)
    _, frames = orc.synthesis_stage(n, hop, win, np.zeros(k), np.concatenate([m1, m2]), np.concatenate([i1, i2]))
    ref = orc.sum_frames_e(hop, hop, list(frames)).ravel()
    assert y.size == ref.size
    assert np.linalg.norm(y - ref) / np.linalg.norm(ref) <= TOL_L2


def test_random_phase_does_not_depend_on_the_chunking(emu_lib):
    """randomPhaseFilter draws fresh increments for every frame (Filter.hs:156-157), however the
    frames arrive: a stream pushed hop by hop gives the one-call result, and consecutive frames of a
    channel get different increments."""
    import torch
    from helpers import random_phase_expected
    from vocoder_amd import vocoder as V, conduit as Cd, stream as S, filter as Fl
    n, hop, c = 128, 32, 2
    par = V.VocoderParams(n, hop, _hann(n), lib=emu_lib)
    filt = Fl.random_phase_filter(9).bind(par, 375.0)
    rng = np.random.default_rng(2)
    x = rng.standard_normal((c, n + hop * 12 + 5))
    y_ref, _, s_ref = Cd.process_filter_dev(par, filt, torch.from_numpy(x))
    st = S.Stream(par, c, filt=filt)
    outs, pos = [], 0
    for m in [n] + [hop] * 12 + [5]:
        outs.append(st.push(x[:, pos:pos + m]))
        pos += m
    outs.append(st.flush())
    assert np.array_equal(np.concatenate(outs, axis=1), y_ref.numpy())
    assert np.array_equal(st.phases()[1], s_ref.numpy())
    # the increments themselves: a chain applied to frames 5.. of channels 3.. of a signal
    ldk = (par.K + 1) & ~1
    mag, inc = torch.zeros((c, 4, ldk), dtype=torch.float64), torch.zeros((c, 4, ldk), dtype=torch.float64)
    _, got = filt.apply_dev(mag, inc, first_frame=5, first_channel=3)
    exp = random_phase_expected(9, c, 4, par.K, position=0, first_frame=5, first_channel=3)
    assert np.array_equal(got.numpy()[..., :par.K], exp)
    assert not np.array_equal(exp[0, 0], exp[0, 1]) and not np.array_equal(exp[0, 0], exp[1, 0])


@pytest.mark.parametrize("speed,cuts", [
    ("2", (200, 777)), ("1/2", (64, 500, 501)), ("3/2", (333,)), ("2/3", (100, 150, 900)), ("1/5", (400,)), ("0", (300,)),
])
def test_play_speed_inside_a_stream_equals_the_batch_path(emu_lib, orc, speed, cuts):
    """chain .| playSpeed r .| chain' as one stream (Conduit/Filter.hs:122-136): pushes of any sizes give
    the batch path's samples and phases bit for bit, and the oracle's within the bars."""
    import torch
    from fractions import Fraction
    from vocoder_amd import vocoder as V, conduit as Cd, stream as S, filter as Fl
    n, hop = 128, 32
    win = _hann(n)
    par = V.VocoderParams(n, hop, win, lib=emu_lib)
    step = 48000.0 / n
    r = Fraction(speed)
    desc = Fl.lowpass_butterworth(2.0, 6000.0) >> Fl.play_speed(r) >> Fl.amplify(0.5)
    filt = desc.bind(par, step)
    x = np.random.default_rng(11).standard_normal((2, n + hop * 30 + 7))
    y_ref, a_ref, s_ref = Cd.process_filter_dev(par, filt, torch.from_numpy(x))
    st = S.Stream(par, 2, filt=filt)
    edges = (0,) + tuple(cuts) + (x.shape[1],)
    outs = [st.push(x[:, a:b]) for a, b in zip(edges[:-1], edges[1:])] + [st.flush()]
    y = np.concatenate(outs, axis=1)
    assert y.shape == tuple(y_ref.shape) and np.array_equal(y, y_ref.numpy())
    a, sph = st.phases()
    assert np.array_equal(a, a_ref.numpy()) and np.array_equal(sph, s_ref.numpy())
    ref = Hh.oracle_filtered(orc, n, hop, win, x[1], step, [("lowpass_butterworth", (2.0, 6000.0))], speed=r)
    mag2, inc2 = orc.apply_filter(step, [("amplify", (0.5,))], ref["mag"], ref["ph_inc"]) if ref["mag"].shape[0] else (ref["mag"], ref["ph_inc"])
    assert np.array_equal(inc2, ref["ph_inc"])            # a gain leaves the increments alone: ref["y"] scales by 0.5
    assert y.shape[1] == ref["y"].size
    if ref["y"].size:
        assert Hh.rel_l2(y[1], 0.5 * ref["y"]) <= Hh.TOL_L2


def test_random_phase_after_play_speed_in_a_stream(emu_lib):
    """`--playSpeed 2 --randomPhase` chunk by chunk: the random increments are keyed by the position of the
    yielded frame in the stream, so the chunking does not change them."""
    import torch
    from vocoder_amd import vocoder as V, conduit as Cd, stream as S, filter as Fl
    n, hop = 128, 32
    par = V.VocoderParams(n, hop, _hann(n), lib=emu_lib)
    filt = (Fl.play_speed(2) >> Fl.random_phase_filter(9)).bind(par, 1.0)
    x = np.random.default_rng(12).standard_normal((2, n + hop * 20 + 5))
    y_ref, _, s_ref = Cd.process_filter_dev(par, filt, torch.from_numpy(x))
    st = S.Stream(par, 2, filt=filt)
    outs = [st.push(x[:, :130]), st.push(x[:, 130:131]), st.push(x[:, 131:600]), st.push(x[:, 600:]), st.flush()]
    assert np.array_equal(np.concatenate(outs, axis=1), y_ref.numpy())
    assert np.array_equal(st.phases()[1], s_ref.numpy())
