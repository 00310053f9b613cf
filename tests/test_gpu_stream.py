"""GPU: chunk-at-a-time streaming equals the one-call result bit for bit (host and device entry
points), with and without a filter chain."""
import os
import sys

import numpy as np
import pytest

import helpers as Hh

pytestmark = pytest.mark.gpu


def _hann(n):
    return 0.5 * (1 - np.cos(2 * np.pi * np.arange(n) / n))


@pytest.mark.parametrize("n,w,hop,chunk", [(4096, 4096, 1024, 1024), (4096, 3000, 1000, 4096), (256, 256, 64, 100),
                                           (16384, 16384, 4096, 4096)])
def test_stream_pushes_equal_one_call(gpu_lib, n, w, hop, chunk):
    from vocoder_amd import vocoder as V, conduit as Cd, stream as S
    par = V.VocoderParams(n, hop, _hann(w), lib=gpu_lib)
    rng = np.random.default_rng(n + chunk)
    c, total = 3, w + hop * 37 + 11
    x = rng.standard_normal((c, total))
    y_ref, left_ref, (a_ref, s_ref) = Cd.process_with(par, x, chunk=chunk)
    st = S.Stream(par, c, chunk)
    cuts = np.concatenate([[0], np.sort(rng.integers(0, total + 1, 7)), [total]])
    outs = [st.push(x[:, a:b]) for a, b in zip(cuts[:-1], cuts[1:])]
    outs.append(st.flush())
    y = np.concatenate(outs, axis=1)
    assert y.shape == y_ref.shape and np.array_equal(y, y_ref)
    a, s = st.phases()
    assert np.array_equal(a, a_ref) and np.array_equal(s, s_ref)
    assert np.array_equal(st.leftover(), left_ref)


def test_stream_device_entry_with_filter(gpu_lib):
    import torch
    from vocoder_amd import vocoder as V, conduit as Cd, stream as S, filter as Fl
    par = V.VocoderParams(4096, 1024, _hann(4096), lib=gpu_lib)
    filt = (Fl.pitch_shift_interpolate(2.0) >> Fl.lowpass_brickwall(1000.0)).bind(par, 48000.0 / 4096)
    c, total = 4, 4096 + 1024 * 50 + 77
    x = torch.randn((c, total), dtype=torch.float64, device="cuda")
    y_ref, _, s_ref = Cd.process_filter_dev(par, filt, x)
    st = S.Stream(par, c, filt=filt)
    outs, pos = [], 0
    for m in (5000, 1, 20000, 0, total - 25001):
        outs.append(st.push_dev(x[:, pos:pos + m].contiguous() if m else x[:, :0], m))
        pos += m
    torch.cuda.synchronize()
    y = np.concatenate([o.cpu().numpy() for o in outs] + [st.flush()], axis=1)
    assert np.array_equal(y, y_ref.cpu().numpy())
    assert np.array_equal(st.phases()[1], s_ref.cpu().numpy())


def test_dunai_mode_on_device_matches_oracle(gpu_lib, orc):
    """Vocoder.Dunai.process conventions (Dunai.hs:46-48) at N=4096: zero pre-pad, ticks in
    multiples of the hop, as many samples out as in, volume correction on the samples."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_stream_cpu import _oracle_dunai
    from vocoder_amd import vocoder as V, stream as S
    n, hop = 4096, 1024
    win = _hann(n)
    par = V.VocoderParams(n, hop, win, lib=gpu_lib)
    rng = np.random.default_rng(8)
    ticks = [rng.standard_normal(hop * m) for m in (4, 1, 7, 0, 12)]
    ref, aph, sph = _oracle_dunai(orc, n, hop, win, ticks)
    st = S.Stream(par, 1, dunai=True)
    got = [st.push(t)[0] for t in ticks]
    y, r = np.concatenate(got), np.concatenate(ref)
    assert y.shape == r.shape
    assert np.linalg.norm(y - r) / np.linalg.norm(r) <= 1e-9


def test_concatenated_sources_with_their_own_filters_on_device(gpu_lib, orc):
    """concatenateV (Audio.hs:60-69) at N=4096: leftover and phases of source 1 flow into source 2,
    each source runs its own filter, all frames feed one overlap-add; against the oracle."""
    from helpers import TOL_L2
    from vocoder_amd import vocoder as V, stream as S, filter as Fl
    n, hop = 4096, 1024
    win = _hann(n)
    par = V.VocoderParams(n, hop, win, lib=gpu_lib)
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
    k = n // 2 + 1
    fr1, left1 = orc.gen_frames_of_e(n, hop, [x1])
    fr2, left2 = orc.gen_frames_of_e(n, hop, [x2], q0=left1)
    vc = orc.volume_coeff(hop, win)
    aph, m1, i1 = orc.analysis_stage(n, hop, win, np.zeros(k), fr1)
    aph, m2, i2 = orc.analysis_stage(n, hop, win, aph, fr2)
    m1, i1 = orc.apply_filter(step, [("lowpass_brickwall", 6000.0)], m1 * vc, i1)
    m2, i2 = orc.apply_filter(step, [("pitch_shift", 1.25)], m2 * vc, i2)
    _, frames = orc.synthesis_stage(n, hop, win, np.zeros(k), np.concatenate([m1, m2]), np.concatenate([i1, i2]))
    ref = orc.sum_frames_e(hop, hop, list(frames)).ravel()
    assert y.size == ref.size
    assert np.linalg.norm(y - ref) / np.linalg.norm(ref) <= TOL_L2
    assert np.array_equal(st.leftover()[0], left2)


@pytest.mark.parametrize("n,w,hop,chunk", [(4096, 4096, 1024, 1024), (4096, 3000, 1000, 4096), (256, 256, 64, 100)])
def test_stream_publishes_when_sum_frames_e_would(gpu_lib, orc, n, w, hop, chunk):
    """After every push the stream has published exactly the chunks the oracle's sumFramesE state
    machine (Frames.hs:49-53) has yielded for the frames cut so far, and the samples are the oracle's."""
    from helpers import TOL_L2
    from vocoder_amd import vocoder as V, stream as S
    win = _hann(w)
    par = V.VocoderParams(n, hop, win, lib=gpu_lib)
    rng = np.random.default_rng(n + w)
    total = w + hop * 29 + 13
    x = rng.standard_normal(total)
    cuts = np.concatenate([[0], np.sort(rng.integers(0, total + 1, 9)), [total]])
    pieces = [x[a:b] for a, b in zip(cuts[:-1], cuts[1:])]
    st = S.Stream(par, 1, chunk)
    got, frames_so_far = [], 0
    for i, piece in enumerate(pieces):
        got.append(st.push(piece)[0])
        fr, _ = orc.gen_frames_of_e(w, hop, pieces[: i + 1])          # the reference's framing of the pieces so far
        frames_so_far = fr.shape[0]
        sched = orc.sum_frames_e_schedule(chunk, hop, frames_so_far)
        assert sum(g.size for g in got) == (sched[-1] if frames_so_far else 0) * chunk
    got.append(st.flush()[0])
    y = np.concatenate(got)
    r = orc.roundtrip(n, hop, win, x, chunk=chunk)
    assert y.shape == r["y"].shape and np.linalg.norm(y - r["y"]) / np.linalg.norm(r["y"]) <= TOL_L2


def test_random_phase_stream_hop_by_hop_on_device(gpu_lib):
    """A stream pushed one hop at a time with randomPhase equals the one-call result (the random
    increments are keyed by the frame's position in the signal, not in the push)."""
    import torch
    from vocoder_amd import vocoder as V, conduit as Cd, stream as S, filter as Fl
    n, hop, c = 4096, 1024, 2
    par = V.VocoderParams(n, hop, _hann(n), lib=gpu_lib)
    filt = Fl.random_phase_filter(9).bind(par, 48000.0 / n)
    x = torch.randn((c, n + hop * 10 + 5), dtype=torch.float64, device="cuda")
    y_ref, _, s_ref = Cd.process_filter_dev(par, filt, x)
    st = S.Stream(par, c, filt=filt)
    outs, pos = [], 0
    for m in [n] + [hop] * 10 + [5]:
        outs.append(st.push_dev(x[:, pos:pos + m].contiguous(), m))
        pos += m
    torch.cuda.synchronize()
    y = np.concatenate([o.cpu().numpy() for o in outs] + [st.flush()], axis=1)
    assert np.array_equal(y, y_ref.cpu().numpy())
    assert np.array_equal(st.phases()[1], s_ref.cpu().numpy())


@pytest.mark.parametrize("speed", ["2", "2/3", "1/3"])
def test_play_speed_inside_a_stream_on_device(gpu_lib, orc, speed):
    """chain .| playSpeed r .| chain' chunk by chunk (Conduit/Filter.hs:122-136): the batch path's bits,
    the oracle's samples within the bar; random increments keyed by the yielded frame's position."""
    import torch
    from fractions import Fraction
    from vocoder_amd import vocoder as V, conduit as Cd, stream as S, filter as Fl
    n, hop = 4096, 1024
    win = 0.5 * (1 - np.cos(2 * np.pi * np.arange(n) / n))
    par = V.VocoderParams(n, hop, win, lib=gpu_lib)
    step = 48000.0 / n
    r = Fraction(speed)
    x = np.random.default_rng(21).standard_normal((3, n + hop * 40 + 77))
    xd = torch.from_numpy(x).cuda()
    filt = (Fl.lowpass_butterworth(2.0, 6000.0) >> Fl.play_speed(r)).bind(par, step)
    y_ref, a_ref, s_ref = Cd.process_filter_dev(par, filt, xd)
    st = S.Stream(par, 3, filt=filt)
    cuts = (0, 5000, 5001, 20000, x.shape[1])
    outs = [st.push_dev(xd[:, a:b].contiguous()) for a, b in zip(cuts[:-1], cuts[1:])]
    torch.cuda.synchronize()
    y = np.concatenate([o.cpu().numpy() for o in outs] + [st.flush()], axis=1)
    assert y.shape == tuple(y_ref.shape) and np.array_equal(y, y_ref.cpu().numpy())
    a, sph = st.phases()
    assert np.array_equal(a, a_ref.cpu().numpy()) and np.array_equal(sph, s_ref.cpu().numpy())
    ref = Hh.oracle_filtered(orc, n, hop, win, x[2], step, [("lowpass_butterworth", (2.0, 6000.0))], speed=r)
    assert y.shape[1] == ref["y"].size and Hh.rel_l2(y[2], ref["y"]) <= Hh.TOL_L2
    # the same with a random-phase chain behind the speed change, host pushes
    filt2 = (Fl.play_speed(r) >> Fl.random_phase_filter(5)).bind(par, step)
    y2_ref, _, s2_ref = Cd.process_filter_dev(par, filt2, xd)
    st2 = S.Stream(par, 3, filt=filt2)
    outs2 = [st2.push(x[:, a:b]) for a, b in zip(cuts[:-1], cuts[1:])] + [st2.flush()]
    assert np.array_equal(np.concatenate(outs2, axis=1), y2_ref.cpu().numpy())
    assert np.array_equal(st2.phases()[1], s2_ref.cpu().numpy())
