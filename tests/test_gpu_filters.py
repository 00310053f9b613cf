"""GPU: filter chains and playSpeed on the STFT in device memory against the filter oracle,
at the frame lengths of the benchmark configurations (row in LDS at N=4096, row in the global
scratch area at N=8192 and 16384), and the filtered pipelines of those configurations end to end."""
import os
import sys
from fractions import Fraction

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from helpers import TOL_L2, oracle_filtered, random_phase_expected  # noqa: E402

pytestmark = pytest.mark.gpu


def _hann(n):
    return 0.5 * (1 - np.cos(2 * np.pi * np.arange(n) / n))


def _signal(c, length, seed=3):
    rng = np.random.default_rng(seed)
    t = np.arange(length)
    return np.stack([0.5 * np.sin(2 * np.pi * (0.004 + 0.003 * i) * t + i) for i in range(c)]) + \
        1e-2 * rng.uniform(-1, 1, (c, length))


def _build(chain):
    from vocoder_amd import filter as Fl
    f = Fl.id_filter()
    for name, *prm in chain:
        fn = {"pitch_shift": Fl.pitch_shift_interpolate, "convolution": Fl.convolution_filter,
              "envelope": Fl.envelope_filter}.get(name) or getattr(Fl, name)
        f = f >> fn(*prm)
    return f


CHAINS = [
    [("pitch_shift", 2.0), ("lowpass_brickwall", 1000.0)],                      # vocoder-file --pitchShiftInterpolate 2 --lowpassBrickwall 1000
    [("envelope", 32)],
    [("bandpass_butterworth", 2.0, 300.0, 3400.0), ("amplify", 1.5), ("pitch_shift", 0.8)],
    [("convolution", tuple(np.hanning(7) + 0.1))],
]


@pytest.mark.parametrize("n", [4096, 8192, 16384, 64])
@pytest.mark.parametrize("chain", CHAINS, ids=lambda c: "+".join(s[0] for s in c))
def test_chain_on_device_matches_oracle(gpu_lib, orc, n, chain):
    import torch
    from vocoder_amd import vocoder as V
    par = V.VocoderParams(n, n // 4, _hann(n), lib=gpu_lib)
    step = 48000.0 / n
    rng = np.random.default_rng(n)
    c, f = 3, 7
    ldk = (par.K + 1) & ~1
    mag = rng.uniform(0, 1, (c, f, ldk))
    inc = rng.uniform(-1, 1, (c, f, ldk))
    tm, ti = torch.from_numpy(mag).cuda(), torch.from_numpy(inc).cuda()
    bound = _build(chain).bind(par, step)
    m, i = bound.apply_dev(tm, ti)
    torch.cuda.synchronize()
    m, i = m.cpu().numpy(), i.cpu().numpy()
    chain_o = [(nm, np.array(p[0])) if nm == "convolution" else (nm, *p) for nm, *p in chain]
    mo, io = orc.apply_filter(step, chain_o, mag[..., :par.K], inc[..., :par.K])
    if any(s[0] == "envelope" for s in chain):        # device log / exp vs glibc: not bit identical
        assert np.allclose(m[..., :par.K], mo, rtol=1e-12, atol=1e-15)
    else:
        assert np.array_equal(m[..., :par.K], mo)
    assert np.array_equal(i[..., :par.K], io)
    assert np.array_equal(m[..., par.K:], mag[..., par.K:])                     # row padding untouched


def test_random_phase_stream_on_device(gpu_lib):
    import torch
    from vocoder_amd import vocoder as V, filter as Fl
    par = V.VocoderParams(8192, 2048, _hann(8192), lib=gpu_lib)
    c, f, ldk = 2, 5, par.K + 1
    mag = torch.rand((c, f, ldk), dtype=torch.float64, device="cuda")
    inc = torch.zeros_like(mag)
    keep = mag.clone()
    m, i = Fl.random_phase_filter(77).bind(par, 1.0).apply_dev(mag, inc)
    torch.cuda.synchronize()
    assert torch.equal(m, keep)
    exp = random_phase_expected(77, c, f, par.K, position=0)
    assert np.array_equal(i.cpu().numpy()[..., :par.K], exp)


def test_play_speed_gather_on_device(gpu_lib, orc):
    import torch
    from vocoder_amd import vocoder as V, filter as Fl
    par = V.VocoderParams(1024, 256, _hann(1024), lib=gpu_lib)
    c, f, ldk = 3, 40, par.K + 1
    mag = torch.rand((c, f, ldk), dtype=torch.float64, device="cuda")
    inc = torch.rand_like(mag)
    for coeff in (Fraction(2), Fraction(1, 2), Fraction(1.3), Fraction(7, 5)):
        m, i = Fl.play_speed(coeff).bind(par, 1.0).apply_dev(mag, inc)
        src = torch.from_numpy(orc.play_speed(coeff.numerator, coeff.denominator, f)).cuda()
        assert torch.equal(m, mag[:, src]) and torch.equal(i, inc[:, src])


PIPELINES = [
    # N, hop, chain, playSpeed -- the filtered benchmark configurations at test size
    (4096, 1024, [("pitch_shift", 2.0), ("lowpass_brickwall", 1000.0)], None),
    (4096, 1024, [("envelope", 32)], None),
    (8192, 2048, [], Fraction(2)),
    (1024, 256, [("highpass_butterworth", 2.0, 500.0)], Fraction(2, 3)),
]


@pytest.mark.parametrize("n,hop,chain,speed", PIPELINES)
def test_filtered_pipeline_matches_oracle(gpu_lib, orc, n, hop, chain, speed):
    import torch
    from vocoder_amd import vocoder as V, conduit as Cd, filter as Fl
    par = V.VocoderParams(n, hop, _hann(n), lib=gpu_lib)
    step = 48000.0 / n
    c = 2
    length = n + hop * 17 + 11
    x = _signal(c, length)
    f = _build(chain)
    if speed is not None:
        f = f >> Fl.play_speed(speed)
    y, aph, sph = Cd.process_filter_dev(par, f.bind(par, step), torch.from_numpy(x).cuda())
    torch.cuda.synchronize()
    y = y.cpu().numpy()
    for ch in range(c):
        ref = oracle_filtered(orc, n, hop, _hann(n), x[ch], step, [(nm, *p) for nm, *p in chain], speed)
        assert y.shape[1] == ref["y"].size
        err = np.linalg.norm(y[ch] - ref["y"]) / np.linalg.norm(ref["y"])
        assert err <= TOL_L2, err


def test_pcm16_edge_on_device(gpu_lib):
    """16-bit PCM <-> planar doubles on the GPU against the numpy restatement of libsndfile's
    conversions (parity unpinned: libsndfile is outside the reference tree)."""
    import torch
    from vocoder_amd import vocoder as V, conduit as Cd
    from test_filters_cpu import _pcm16_expected
    par = V.VocoderParams(256, 64, _hann(256), lib=gpu_lib)
    rng = np.random.default_rng(8)
    c, n = 2, 100003
    pcm = rng.integers(-32768, 32768, n * c, dtype=np.int16)
    planar = Cd.pcm16_to_planar_dev(par, torch.from_numpy(pcm).cuda(), c)
    assert np.array_equal(planar.cpu().numpy(), pcm.reshape(n, c).T.astype(np.float64) / 32768.0)
    x = rng.uniform(-1.1, 1.1, (c, n))
    x[0, :6] = [0.5 / 32767, 1.5 / 32767, -2.5 / 32767, 1.0, -1.0, 1.00002]
    x[1, :4] = [0.5 / 32768, 1.5 / 32768, 32766.5 / 32768, -32767.5 / 32768]
    for clip in (False, True):
        got = Cd.planar_to_pcm16_dev(par, torch.from_numpy(x).cuda(), clip).cpu().numpy()
        assert np.array_equal(got.reshape(n, c).T, _pcm16_expected(x, clip)), clip


@pytest.mark.parametrize("n,hop,c,f_in,coeff,tile", [(4096, 1024, 5, 40, Fraction(2), 0), (4096, 1024, 5, 40, Fraction(2, 3), 7),
                                                    (8192, 2048, 3, 21, Fraction(2), 0), (256, 64, 4, 33, Fraction(5, 3), 6),
                                                    (16384, 4096, 2, 9, Fraction(2), 3)])
def test_mapped_synthesis_equals_gather_then_synthesis(gpu_lib, monkeypatch, n, hop, c, f_in, coeff, tile):
    """playSpeed through row maps (no copy) against the gathered STFT, bit for bit, in the
    specialised, the generic and the frames-in-memory synthesis kernels, whole and tiled."""
    from test_filters_cpu import _mapped_vs_gathered
    _mapped_vs_gathered(gpu_lib, n, hop, c, f_in, coeff, monkeypatch, tile, None, device="cuda")


def test_add_filters_on_device(gpu_lib, orc):
    """addFilters (Filter.hs:58-59) on device arrays against the oracle's chains and addFrames."""
    import torch
    from vocoder_amd import vocoder as V, filter as Fl
    par = V.VocoderParams(4096, 1024, _hann(4096), lib=gpu_lib)
    step = 48000.0 / 4096
    rng = np.random.default_rng(21)
    c, f = 2, 5
    ldk = (par.K + 1) & ~1
    mag, inc = rng.uniform(0, 1, (c, f, ldk)), rng.uniform(-1, 1, (c, f, ldk))
    low, high = [("lowpass_brickwall", 1000.0)], [("highpass_butterworth", 2.0, 4000.0), ("amplify", 0.5)]
    b = Fl.add_filters(_build(low), _build(high)).bind(par, step)
    md, idv = b.apply_dev(torch.from_numpy(mag).cuda(), torch.from_numpy(inc).cuda())
    md, idv = md.cpu().numpy()[..., :par.K], idv.cpu().numpy()[..., :par.K]
    m0, i0 = mag[..., :par.K], inc[..., :par.K]
    (ma, ia), (mb, ib) = orc.apply_filter(step, low, m0, i0), orc.apply_filter(step, high, m0, i0)
    for ch in range(c):
        for fr in range(f):
            me, ie = orc.add_frames(ma[ch, fr], ia[ch, fr], mb[ch, fr], ib[ch, fr])
            assert np.allclose(md[ch, fr], me, rtol=1e-12, atol=1e-15)
            live = me > 1e-9
            assert np.abs(np.angle(np.exp(1j * (idv[ch, fr] - ie))))[live].max() <= 1e-9


def _oracle_speed_then_random_phase(orc, n, hop, win, x, speed, seed, channel):
    """`--playSpeed s --randomPhase` (VocoderFile.hs:72 folds the filters left to right) on one channel
    from oracle pieces: analysis .| volumeFix, playSpeed's frame map, increments replaced by the
    library's documented random stream (the reference's own generator is unseeded), synthesisStage,
    sumFramesE."""
    k = n // 2 + 1
    rt = orc.roundtrip(n, hop, win, x, want_stft=True)
    src = orc.play_speed(speed.numerator, speed.denominator, rt["mag"].shape[0])
    mag = rt["mag"][src]
    inc = random_phase_expected(seed, 1, mag.shape[0], k, position=0, first_channel=channel)[0]
    sph, frames = orc.synthesis_stage(n, hop, win, np.zeros(k), mag, inc)
    y = np.zeros(orc.ola_length(mag.shape[0], len(win), hop, hop))
    for j in range(mag.shape[0]):
        y[j * hop:j * hop + len(win)] = y[j * hop:j * hop + len(win)] + frames[j]
    return y, sph


@pytest.mark.parametrize("n,hop,speed", [(8192, 2048, Fraction(2)), (1024, 256, Fraction(3, 2))])
def test_play_speed_then_random_phase_matches_oracle(gpu_lib, orc, n, hop, speed):
    """The time-stretch configuration as written: playSpeed feeding a write-only random-phase pass,
    synthesis reading the moduli through the row map and the fresh increments directly."""
    import torch
    from vocoder_amd import vocoder as V, conduit as Cd, filter as Fl
    win = _hann(n)
    par = V.VocoderParams(n, hop, win, lib=gpu_lib)
    c, seed = 3, 1234
    x = _signal(c, n + hop * 13 + 5)
    filt = (Fl.play_speed(speed) >> Fl.random_phase_filter(seed)).bind(par, 48000.0 / n)
    sph = torch.zeros((c, par.K), dtype=torch.float64, device="cuda")
    y, _, sph = Cd.process_filter_dev(par, filt, torch.from_numpy(x).cuda(), syn_phase=sph)
    torch.cuda.synchronize()
    y, sph = y.cpu().numpy(), sph.cpu().numpy()
    for ch in range(c):
        yo, spo = _oracle_speed_then_random_phase(orc, n, hop, win, x[ch], speed, seed, ch)
        assert y.shape[1] == yo.size
        assert np.linalg.norm(y[ch] - yo) / np.linalg.norm(yo) <= TOL_L2
        assert np.array_equal(sph[ch], spo)          # identical increments in, identical phase chain out


def test_binary_stream_operations_on_device(gpu_lib, orc):
    """`--multiply` / `--divide` (VocoderJack.hs:107-110,127-132) on device arrays against the oracle."""
    import torch
    from vocoder_amd import vocoder as V, conduit as Cd
    par = V.VocoderParams(4096, 1024, _hann(4096), lib=gpu_lib)
    rng = np.random.default_rng(5)
    shape = (3, 11, par.K + 1)
    m1, m2 = rng.uniform(0, 2, shape), rng.uniform(0.1, 2, shape)
    i1, i2 = rng.uniform(-1, 1, shape), rng.uniform(-1, 1, shape)
    m2[0, 0, :2] = 0.0
    m1[0, 0, 1] = 0.0
    t = [torch.from_numpy(a).cuda() for a in (m1, i1, m2, i2)]
    for op in ("multiply", "divide"):
        mo, io = Cd.binary_frames_dev(par, op, (t[0], t[1]), (t[2], t[3]))
        me, ie = orc.binary_frames(op, m1, i1, m2, i2)
        assert np.array_equal(mo.cpu().numpy(), me, equal_nan=True) and np.array_equal(io.cpu().numpy(), ie)
