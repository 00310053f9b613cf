"""Parity of the HIP path against the oracle, through the C ABI, on a real MI355X.

Bars (fp64): relative L2 <= 1e-9 on output samples, <= 1e-9 rad on wrapped phases (bins with
magnitude >= 1e-5 of the frame's maximum; the phase of a noise-floor bin is not reproducible
across FFT implementations), magnitudes <= 1e-9 relative.  Given identical STFT input the
synthesis phase chain is required to match the oracle BIT FOR BIT.
"""
import glob
import os

import numpy as np
import pytest

import helpers as Hh

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("n,w,h", [(8, 8, 2), (16, 12, 4), (64, 64, 16), (256, 200, 64), (1024, 1024, 256),
                                   (2048, 2048, 512), (4096, 4096, 1024), (4096, 3000, 1000), (8192, 8192, 2048),
                                   (16384, 16384, 4096)])
def test_fft_layer(gpu_lib, orc, n, w, h):
    Hh.check_fft_layer(gpu_lib, orc, n, w, h, np.random.default_rng(n))


@pytest.mark.parametrize("n,w,h,f,c,at,st,kind", [
    (64, 64, 16, 9, 2, 0, 0, "noise"),
    (64, 48, 16, 23, 2, 4, 5, "noise"),
    (128, 100, 30, 17, 1, 3, 7, "noise"),
    (64, 64, 64, 7, 1, 2, 3, "noise"),
    (2048, 2048, 512, 40, 2, 0, 0, "tone"),        # config 1 shape
    (4096, 4096, 1024, 48, 3, 0, 0, "tone"),       # metric shape, default tiling
    (4096, 4096, 1024, 48, 2, 7, 11, "tone"),      # metric shape, forced multi-tile + carries
    (8192, 8192, 2048, 12, 2, 0, 5, "tone"),       # config 3 shape
    (16384, 12000, 3000, 6, 1, 0, 0, "noise"),
])
def test_stages(gpu_lib, orc, n, w, h, f, c, at, st, kind):
    Hh.check_stages(gpu_lib, orc, n, w, h, f, c, np.random.default_rng(1), ana_tile=at, syn_tile=st, kind=kind)


@pytest.mark.parametrize("n,w,h,length,c,at,st,chunk,kind", [
    (64, 64, 16, 64 + 16 * 12 + 5, 2, 0, 0, None, "noise"),
    (64, 48, 16, 48 + 16 * 22 + 3, 2, 4, 5, 48, "noise"),
    (64, 32, 40, 32 + 40 * 8, 1, 2, 4, None, "noise"),         # hop > window
    (128, 128, 32, 100, 2, 0, 0, None, "noise"),               # no full frame: leftover only
    (128, 128, 32, 128, 1, 0, 0, 50, "noise"),                 # exactly one frame, ragged chunk
    (256, 256, 1, 300, 1, 0, 0, None, "noise"),                # hop 1 (halo 255)
    (2048, 2048, 512, 48000, 1, 0, 0, None, "tone"),           # config 1 shape, 1 s
    (4096, 4096, 1024, 96000, 4, 0, 0, None, "tone"),          # metric shape, 2 s, 4 streams
    (4096, 4096, 1024, 60000, 2, 9, 13, 1024, "tone"),         # metric shape, tiles + halo + carries
    (4096, 2500, 700, 40000, 2, 0, 16, 4096, "tone"),          # padded frames, chunk = N
    (8192, 8192, 2048, 60000, 2, 0, 0, None, "tone"),
    (16384, 16384, 4096, 80000, 1, 0, 0, None, "noise"),       # overlap-add ring too large for LDS
])
def test_roundtrip(gpu_lib, orc, n, w, h, length, c, at, st, chunk, kind):
    Hh.check_roundtrip(gpu_lib, orc, n, w, h, length, c, np.random.default_rng(2), ana_tile=at, syn_tile=st,
                       chunk=chunk, kind=kind)


ROUNDTRIPS = sorted(glob.glob(os.path.join(HERE, "golden", "roundtrip_*.npz")))


@pytest.mark.parametrize("path", ROUNDTRIPS, ids=[os.path.basename(p)[10:-4] for p in ROUNDTRIPS])
def test_golden_fixtures(gpu_lib, path):
    from vocoder_amd import vocoder as V, conduit as Cd
    g = np.load(path)
    par = V.VocoderParams(int(g["N"]), int(g["H"]), g["window"], lib=gpu_lib)
    y, left, (aph, sph) = Cd.process_with(par, g["x"], phases=(g["ana_phase_in"], g["syn_phase_in"]))
    assert y.shape == g["y"].shape and np.array_equal(left, g["leftover"])
    assert Hh.rel_l2(y, g["y"]) <= Hh.TOL_L2
    big = g["mag"][-1] >= 0  # fixtures are tone + noise: every bin is above the noise floor gate
    assert Hh.circ_dist(aph, g["ana_phase_out"])[big].max() <= Hh.TOL_PHASE
    # accumulated, unwrapped phase: a few ulp of its magnitude per frame, never looser than that (helpers.check_roundtrip)
    frames = g["mag"].shape[-2]
    tol = max(Hh.TOL_PHASE, 8 * frames ** 0.5 * np.spacing(np.abs(g["syn_phase_out"]).max()))
    assert Hh.circ_dist(sph, g["syn_phase_out"]).max() <= tol
    par.close()


@pytest.mark.parametrize("ana16,syn16", [(0, 0), (1, 0), (0, 1), (1, 1)])
@pytest.mark.parametrize("n,h,nf,c,at,st", [
    (4096, 1024, 40, 3, 0, 0),     # metric shape: one tile per channel
    (4096, 1024, 33, 2, 7, 11),    # tiles: the frame before a tile is analysed again, halo and carries in synthesis
    (4096, 256, 70, 1, 9, 16),     # hop N/16: one new sample-pair slot per frame
    (4096, 512, 21, 2, 0, 0),
    (4096, 2048, 9, 1, 3, 3),      # hop N/2
    (8192, 2048, 12, 2, 0, 5),     # config 3 length
    (8192, 1024, 19, 1, 4, 4),
    (16384, 4096, 8, 2, 0, 0),     # no eight-point kernels at this length: (0, 0) is the generic path through frames
    (16384, 2048, 11, 1, 3, 4),
])
def test_roundtrip_with_eight_and_sixteen_points_per_thread(gpu_lib, orc, monkeypatch, n, h, nf, c, at, st, ana16, syn16):
    """N = 4096 / 8192 have two builds of the specialised kernels: eight and sixteen complex points per thread
    (voc_kernels_fast.h, voc_kernels_fast16.h).  The library takes the sixteen-point analysis by default; every
    combination is held to the same bars against the oracle."""
    monkeypatch.setenv("VOC_FAST16_ANA", str(ana16))
    monkeypatch.setenv("VOC_FAST16_SYN", str(syn16))
    Hh.check_roundtrip(gpu_lib, orc, n, n, h, n + h * (nf - 1) + 6, c, np.random.default_rng(n + h + nf), ana_tile=at,
                       syn_tile=st, kind="tone")


def test_steps_framing_and_ola(gpu_lib, orc):
    from vocoder_amd import vocoder as V, conduit as Cd
    rng = np.random.default_rng(3)
    n, w, h = 1024, 800, 200
    win = orc.make_window("hamming", w)
    par = V.VocoderParams(n, h, win, lib=gpu_lib)
    k = par.K
    spec = rng.standard_normal((5, k)) + 1j * rng.standard_normal((5, k))
    spec[0, 3] = 0.0                      # phase (0 :+ 0) = 0
    spec[1, 0] = -2.0 + 0.0j              # atan2 (+0) (-2) = pi
    prev = rng.uniform(-3, 3, (5, k))
    ph, (mag, inc) = V.analysis_step(par, prev, spec)
    for b in range(5):
        rp, rm, ri = orc.analysis_step(h, n, prev[b], spec[b])
        assert np.abs(mag[b] - rm).max() <= 1e-15 * rm.max()
        assert Hh.circ_dist(ph[b], rp).max() <= Hh.TOL_PHASE and Hh.circ_dist(inc[b] * h, ri * h).max() <= Hh.TOL_PHASE
    assert ph[0, 3] == 0.0 and ph[1, 0] == np.pi
    nph, sp = V.synthesis_step(par, prev, (mag, inc))
    for b in range(5):
        rp, rs = orc.synthesis_step(h, prev[b], mag[b], inc[b])
        assert np.array_equal(nph[b], rp) and Hh.rel_l2(sp[b], rs) <= Hh.TOL_L2
    m2, i2 = V.add_frames(par, (mag, inc), (mag[::-1], inc[::-1]))
    for b in range(5):
        rm, ri = orc.add_frames(mag[b], inc[b], mag[4 - b], inc[4 - b])
        assert Hh.rel_l2(m2[b], rm) <= Hh.TOL_L2
        assert Hh.circ_dist(i2[b], ri)[rm > 1e-9 * rm.max()].max() <= Hh.TOL_PHASE
    x = rng.standard_normal((3, 5000))
    frames, left = Cd.frames_of_e(par, x)
    for ch in range(3):
        rf, rl = orc.gen_frames_of_e(w, h, [x[ch]])
        assert np.array_equal(frames[ch], rf) and np.array_equal(left[ch], rl)     # bit exact: pure data movement
    for chunk in (h, 7, 1000):
        y = Cd.sum_frames_e(par, frames, chunk)
        for ch in range(3):
            assert np.array_equal(y[ch], orc.sum_frames_e(chunk, h, list(frames[ch])).ravel())  # same summation order
    assert V.volume_coeff(par) == orc.volume_coeff(h, win)
    assert np.array_equal(V.voc_window(par), win)
    par.close()


def test_functions_with_the_reference_argument_lists(gpu_lib, orc):
    """analysisStep h eN, synthesisStep hop, frameFromComplex, frameToComplex and addFrames take no
    VocoderParams in the reference (Vocoder.hs:158,186,213,217,221): the forms that need no context."""
    Hh.check_context_free_functions(gpu_lib, orc, np.random.default_rng(8), k=2049, hop=1024, n=4096)
    Hh.check_context_free_functions(gpu_lib, orc, np.random.default_rng(9), k=401, hop=200, n=800)
    # many small calls in a row reuse one staging buffer
    for i in range(50):
        Hh.check_context_free_functions(gpu_lib, orc, np.random.default_rng(100 + i), k=17 + i, hop=4, n=32 + 2 * i, batch=1)


def test_integer_framing_fixture_through_the_library(gpu_lib):
    """Framing and overlap-add move / add values without rounding when they are small integers:
    the library must reproduce the reference's list-model vectors exactly."""
    from vocoder_amd import vocoder as V, conduit as Cd
    g = np.load(os.path.join(HERE, "golden", "framing_int.npz"))
    data = g["data"].astype(np.float64)
    par = V.VocoderParams(16, 3, np.ones(10), lib=gpu_lib)     # W = 10, H = 3
    frames, left = Cd.frames_of_e(par, data)
    assert np.array_equal(frames, g["frames_w10_h3"].astype(np.float64))
    assert np.array_equal(left, data[frames.shape[0] * 3:])
    assert np.array_equal(Cd.sum_frames_e(par, frames, 4), g["ola_c4_h3"].astype(np.float64).ravel())
    assert np.array_equal(Cd.sum_frames_e(par, frames, 16), g["ola_c16_h3"].astype(np.float64).ravel())
    par.close()


def test_empty_and_degenerate_inputs(gpu_lib, orc):
    from vocoder_amd import vocoder as V, conduit as Cd
    win = orc.make_window("hann", 64)
    par = V.VocoderParams(64, 16, win, lib=gpu_lib)
    ph = np.arange(33.0)
    p2, (mag, inc) = V.analysis_stage(par, ph, np.empty((0, 64)))       # mapAccumL over no frames
    assert np.array_equal(p2, ph) and mag.shape == (0, 33)
    y, left, (a, s) = Cd.process_with(par, np.empty(0))
    assert y.size == 0 and left.size == 0
    y, left, _ = Cd.process_with(par, np.ones(63))
    assert y.size == 0 and np.array_equal(left, np.ones(63))
    with pytest.raises(V.VocoderError):
        V.VocoderParams(64, 16, np.ones(65), lib=gpu_lib)               # addZeroPadding error (Vocoder.hs:132)
    with pytest.raises(V.VocoderError):
        V.do_fft(par, np.ones(60))                                      # size mismatch
    par.close()


def test_streaming_state_equals_one_shot(gpu_lib, orc):
    """Feeding a stream in pieces through (leftover, phases) gives the one-shot result: the carry
    triple is the reference's own resume state (Conduit.hs:46-53, Audio.hs:31-33)."""
    from vocoder_amd import vocoder as V, conduit as Cd
    rng = np.random.default_rng(9)
    n, h = 512, 128
    win = orc.make_window("hann", n)
    par = V.VocoderParams(n, h, win, lib=gpu_lib)
    x = rng.standard_normal((2, 20000))
    y_all, left_all, (a_all, s_all) = Cd.process_with(par, x)
    left, phases, pieces = None, None, []
    for lo, hi in ((0, 3000), (3000, 3100), (3100, 11111), (11111, 20000)):
        y, left, phases = Cd.process_with(par, x[:, lo:hi], leftover=left, phases=phases)
        pieces.append(y)
    # each call's overlap-add output restarts at its own first frame: overlap the pieces at hop multiples
    total = np.zeros_like(y_all)
    pos = 0
    for lo, piece in zip((0, 3000, 3100, 11111), pieces):
        total[:, pos:pos + piece.shape[1]] += piece[:, : total.shape[1] - pos]
        nf = piece.shape[1] // h - (n // h - 1) if piece.shape[1] else 0
        pos += max(nf, 0) * h
    assert Hh.rel_l2(total, y_all) <= Hh.TOL_L2
    assert np.array_equal(left, left_all)
    assert Hh.circ_dist(phases[0], a_all).max() <= Hh.TOL_PHASE
    par.close()


def test_device_entry_matches_host_entry_and_runs_the_native_library(gpu_lib, orc):
    import torch
    from vocoder_amd import vocoder as V, conduit as Cd
    rng = np.random.default_rng(4)
    n, h, c, length = 4096, 1024, 5, 70000
    win = orc.make_window("hann", n)
    par = V.VocoderParams(n, h, win, lib=gpu_lib)
    x = Hh.signal(rng, c, length, "tone")
    y_host, left_host, (a_host, s_host) = Cd.process_with(par, x)
    dev = torch.device("cuda:0")
    xd = torch.from_numpy(x).to(dev)
    f = Cd.num_frames(par, length)
    yd = torch.full((c, Cd.ola_length(par, f, h) + 4), 7.0, dtype=torch.float64, device=dev)   # padded rows
    ad = torch.zeros(c, par.K, dtype=torch.float64, device=dev)
    sd = torch.zeros_like(ad)
    ld = torch.empty(c, Cd.leftover_length(par, length), dtype=torch.float64, device=dev)
    ws = torch.empty(Cd.roundtrip_workspace_bytes(par, length, c), dtype=torch.uint8, device=dev)
    Cd.process_dev(par, xd, yd, ad, sd, leftover=ld, workspace=ws)
    torch.cuda.synchronize()
    assert np.array_equal(yd[:, :-4].cpu().numpy(), y_host)        # same kernels, same bits
    assert torch.all(yd[:, -4:] == 7.0)                            # bytes between rows untouched
    assert np.array_equal(ld.cpu().numpy(), left_host)
    assert np.array_equal(ad.cpu().numpy(), a_host) and np.array_equal(sd.cpu().numpy(), s_host)
    # small workspace: the library walks the channels in groups and gets the same answer
    small = torch.empty(Cd.roundtrip_workspace_bytes(par, length, 2), dtype=torch.uint8, device=dev)
    yd2 = torch.zeros_like(yd)
    ad.zero_(); sd.zero_()
    Cd.process_dev(par, xd, yd2, ad, sd, workspace=small)
    torch.cuda.synchronize()
    assert np.array_equal(yd2[:, :-4].cpu().numpy(), y_host)
    tiny = torch.empty(64, dtype=torch.uint8, device=dev)       # less than one channel needs: refused, nothing run
    with pytest.raises(V.VocoderError) as e:
        Cd.process_dev(par, xd, yd2, ad, sd, workspace=tiny)
    assert e.value.status == -5
    # rows that are not 16-byte aligned take the generic overlap-add kernel: same result to rounding
    yd3 = torch.full((c, Cd.ola_length(par, f, h) + 3), 7.0, dtype=torch.float64, device=dev)
    ad.zero_(); sd.zero_()
    Cd.process_dev(par, xd, yd3, ad, sd, workspace=ws)
    torch.cuda.synchronize()
    assert Hh.rel_l2(yd3[:, :-3].cpu().numpy(), y_host) <= 1e-13 and torch.all(yd3[:, -3:] == 7.0)
    par.close()
    loaded = open("/proc/self/maps").read()
    assert "libvocoder_hip.so" in loaded and "libvocoder_emu" not in loaded and "libvocoder_oracle" in loaded


@pytest.mark.parametrize("n,w,h", [(1000, 1000, 250), (1500, 1200, 300), (3000, 3000, 750), (4093, 4093, 1023),
                                   (4095, 3001, 1000), (12, 12, 3), (7, 5, 2)])
def test_lengths_that_are_not_powers_of_two(gpu_lib, orc, n, w, h):
    """The chirp-z path in LDS (any frame length up to 4096, as FFTW accepts) against the oracle's long
    double DFT."""
    rng = np.random.default_rng(n)
    Hh.check_fft_layer(gpu_lib, orc, n, w, h, rng)
    Hh.check_stages(gpu_lib, orc, n, w, h, 5, 2, rng)
    Hh.check_roundtrip(gpu_lib, orc, n, w, h, w + h * 7 + 3, 2, rng)


@pytest.mark.parametrize("n,w,h,f,c", [(5000, 5000, 1250, 4, 2), (6000, 4800, 1500, 3, 2), (8191, 8191, 2048, 3, 1),
                                       (9001, 7000, 2000, 3, 2), (12000, 12000, 3000, 2, 1), (16383, 16383, 4096, 2, 1)])
def test_lengths_above_4096_that_are_not_powers_of_two(gpu_lib, orc, n, w, h, f, c):
    """The chirp-z transform through device memory (2 or 4 interleaved 8192-point transforms per frame) against
    the oracle's long double DFT: FFT layer, stages, fused round trip, a stream and a filter chain."""
    from vocoder_amd import vocoder as V, conduit as Cd, stream as S, filter as Fl
    rng = np.random.default_rng(n)
    Hh.check_fft_layer(gpu_lib, orc, n, w, h, rng)
    Hh.check_stages(gpu_lib, orc, n, w, h, f, c, rng)
    Hh.check_roundtrip(gpu_lib, orc, n, w, h, w + h * f + 3, c, rng)
    if n == 9001:
        win = orc.make_window("hann", w)
        par = V.VocoderParams(n, h, win, lib=gpu_lib)
        mag, inc = rng.uniform(0, 1, (2, 3, par.K)), rng.uniform(-1, 1, (2, 3, par.K))
        chain = Fl.pitch_shift_interpolate(1.5) >> Fl.lowpass_butterworth(2.0, 5000.0)
        m, i = chain.bind(par, 48000.0 / n).apply(mag, inc)
        mo, io = orc.apply_filter(48000.0 / n, [("pitch_shift", 1.5), ("lowpass_butterworth", 2.0, 5000.0)], mag, inc)
        assert np.allclose(m, mo, rtol=1e-12, atol=1e-15) and np.array_equal(i, io)
        x = rng.standard_normal((2, w + h * 9 + 2))
        y_ref = Cd.process_with(par, x)[0]
        st = S.Stream(par, 2)
        y = np.concatenate([st.push(x[:, :7777]), st.push(x[:, 7777:]), st.flush()], axis=1)
        assert np.array_equal(y, y_ref)
        par.close()


@pytest.mark.parametrize("n,w,h,f,c", [(32768, 32768, 8192, 3, 2), (20000, 18000, 5000, 2, 1), (32767, 32767, 8191, 1, 1)])
def test_lengths_above_16384(gpu_lib, orc, n, w, h, f, c):
    """Frame lengths above 16384 (powers of two included) run as chirp-z transforms of eight interleaved 8192-point
    transforms through device memory: FFT layer, stages, fused round trip, a filter chain on rows of up to 16385
    bins (one staging array at a time in LDS) and a stream, against the oracle."""
    from vocoder_amd import vocoder as V, conduit as Cd, stream as S, filter as Fl
    rng = np.random.default_rng(n)
    Hh.check_fft_layer(gpu_lib, orc, n, w, h, rng)
    Hh.check_stages(gpu_lib, orc, n, w, h, f, c, rng)
    Hh.check_roundtrip(gpu_lib, orc, n, w, h, w + h * f + 3, c, rng)
    if n in (32768, 20000):
        win = orc.make_window("hann", w)
        par = V.VocoderParams(n, h, win, lib=gpu_lib)
        mag, inc = rng.uniform(0, 1, (2, 3, par.K)), rng.uniform(-1, 1, (2, 3, par.K))
        chain = Fl.pitch_shift_interpolate(1.5) >> Fl.lowpass_butterworth(2.0, 5000.0) >> Fl.envelope_filter(32)
        m, i = chain.bind(par, 48000.0 / n).apply(mag, inc)
        mo, io = orc.apply_filter(48000.0 / n, [("pitch_shift", 1.5), ("lowpass_butterworth", 2.0, 5000.0), ("envelope", 32)], mag, inc)
        assert np.allclose(m, mo, rtol=1e-10, atol=1e-15) and np.array_equal(i, io)
        filt = (Fl.pitch_shift_interpolate(0.75) >> Fl.random_phase_filter(5) >> Fl.play_speed(2)).bind(par, 48000.0 / n)
        x = rng.standard_normal((2, w + h * 6 + 2))
        y_ref = Cd.process_with(par, x)[0]
        st = S.Stream(par, 2)
        y = np.concatenate([st.push(x[:, :7777]), st.push(x[:, 7777:]), st.flush()], axis=1)
        assert np.array_equal(y, y_ref)
        import torch
        yf = Cd.process_filter_dev(par, filt, torch.from_numpy(x).cuda())[0]
        sf = S.Stream(par, 2, filt=filt)
        y2 = np.concatenate([sf.push(x[:, :50000]), sf.push(x[:, 50000:]), sf.flush()], axis=1)
        assert np.array_equal(y2, yf.cpu().numpy())
        par.close()


def test_interleaved_layout_on_device(gpu_lib, orc):
    """vocoder-audio cuts frames of W*C interleaved samples at hops of H*C and splits each frame per
    channel (Audio.hs:51-55): de-interleaving once and framing each planar channel at (W, H) gives the
    same frames; interleave inverts de-interleave; bit exact (data movement only)."""
    import torch
    from vocoder_amd import vocoder as V, conduit as Cd
    w, hop, c = 4096, 1024, 3
    par = V.VocoderParams(w, hop, np.ones(w), lib=gpu_lib)
    rng = np.random.default_rng(13)
    n = w + hop * 6 + 5
    planar = rng.standard_normal((c, n))
    inter = planar.T.reshape(-1).copy()
    got = Cd.deinterleave_dev(par, torch.from_numpy(inter).cuda(), c)
    assert np.array_equal(got.cpu().numpy(), planar)
    assert np.array_equal(Cd.interleave_dev(par, got).cpu().numpy(), inter)
    fr_inter, left = orc.gen_frames_of_e(w * c, hop * c, [inter])          # the reference's interleaved framing
    frames, leftover = Cd.frames_of_e(par, got.cpu().numpy())
    for ch in range(c):
        assert np.array_equal(fr_inter.reshape(-1, w, c)[:, :, ch], frames[ch])
        assert np.array_equal(left.reshape(-1, c)[:, ch], leftover[ch])
    par.close()


def test_calls_on_two_hip_streams_share_the_context_workspace(gpu_lib, orc):
    """Stage calls on one context from two different HIP streams: the library orders its own
    workspace between them, so both results are those of the calls made one after the other."""
    import ctypes as C
    import torch
    from vocoder_amd import vocoder as V
    n, hop, c, f = 4096, 1024, 48, 40
    win = orc.make_window("hann", n)
    par = V.VocoderParams(n, hop, win, lib=gpu_lib)
    k = par.K
    rng = np.random.default_rng(77)
    sets = []
    for i in range(2):
        mag = torch.from_numpy(rng.uniform(0, 1, (c, f, k))).cuda()
        inc = torch.from_numpy(rng.uniform(-0.05, 0.05, (c, f, k))).cuda()
        ph = torch.from_numpy(rng.uniform(-3, 3, (c, k))).cuda()
        sets.append((mag, inc, ph))
    p = lambda t: C.c_void_p(t.data_ptr())

    def run(streams):
        outs = []
        for (mag, inc, ph), st in zip(sets, streams):
            phc, fr = ph.clone(), torch.empty((c, f, n), dtype=torch.float64, device="cuda")
            with torch.cuda.stream(st):
                for _ in range(3):                     # several calls back to back on each stream
                    phc.copy_(ph)
                    par._call("voc_synthesis_stage_dev", p(phc), p(mag), p(inc), p(fr), c, f, C.c_void_p(st.cuda_stream))
            outs.append((phc, fr))
        torch.cuda.synchronize()
        return outs

    torch.cuda.synchronize()
    one = torch.cuda.Stream()
    ref = run([one, one])
    got = run([torch.cuda.Stream(), torch.cuda.Stream()])
    for (pa, fa), (pb, fb) in zip(ref, got):
        assert torch.equal(pa, pb) and torch.equal(fa, fb)
    par.close()


def test_chirp_z_calls_on_two_hip_streams_with_caller_workspaces(gpu_lib, orc):
    """Frame lengths above 4096 that are not powers of two run their transforms through one buffer of the
    context, whoever brings the workspace of the call: two streams must still be ordered for it."""
    import torch
    from vocoder_amd import vocoder as V, conduit as Cd
    n, hop, c = 5000, 1250, 3
    par = V.VocoderParams(n, hop, orc.make_window("hann", n), lib=gpu_lib)
    length = n + hop * 9 + 6
    f = Cd.num_frames(par, length)
    rng = np.random.default_rng(5)
    xs = [torch.from_numpy(rng.standard_normal((c, length))).cuda() for _ in range(2)]
    wss = [torch.empty(Cd.roundtrip_workspace_bytes(par, length, c), dtype=torch.uint8, device="cuda") for _ in range(2)]

    def run(streams):
        outs = []
        for x, ws, st in zip(xs, wss, streams):
            y = torch.empty((c, Cd.ola_length(par, f, hop)), dtype=torch.float64, device="cuda")
            aph = torch.zeros((c, par.K), dtype=torch.float64, device="cuda")
            sph = torch.zeros_like(aph)
            with torch.cuda.stream(st):
                for _ in range(3):
                    aph.zero_(); sph.zero_()
                    Cd.process_dev(par, x, y, aph, sph, workspace=ws, stream=st.cuda_stream)
            outs.append((y, aph, sph))
        torch.cuda.synchronize()
        return outs

    torch.cuda.synchronize()
    one = torch.cuda.Stream()
    ref = run([one, one])
    got = run([torch.cuda.Stream(), torch.cuda.Stream()])
    for a, b in zip(ref, got):
        assert all(torch.equal(u, v) for u, v in zip(a, b))
    r = orc.roundtrip(n, hop, orc.make_window("hann", n), xs[0][0].cpu().numpy())
    assert Hh.rel_l2(got[0][0][0].cpu().numpy(), r["y"]) <= Hh.TOL_L2
    par.close()


def test_analysis_step_equals_analysis_stage_on_a_length_that_is_not_a_power_of_two(gpu_lib, orc):
    """analysisStep and analysisBlock of one context divide by the same N (Vocoder.hs:173): same bits."""
    from vocoder_amd import vocoder as V
    n, hop = 800, 200
    win = orc.make_window("hann", n)
    par = V.VocoderParams(n, hop, win, lib=gpu_lib)
    rng = np.random.default_rng(11)
    frames = rng.standard_normal((1, 1, n))
    prev = rng.uniform(-3, 3, (1, par.K))
    ph_stage, (mag_stage, inc_stage) = V.analysis_stage(par, prev, frames)
    spec = V.do_fft(par, frames[0])
    ph_step, (mag_step, inc_step) = V.analysis_step(par, prev, spec)
    assert np.array_equal(ph_step[0], ph_stage[0]) and np.array_equal(inc_step[0], inc_stage[0, 0])
    ph_hn, (mag_hn, inc_hn) = V.analysis_step(hop, n, prev, spec, lib=gpu_lib)
    assert np.array_equal(inc_hn, inc_step)
    rp, rm, ri = orc.analysis_step(hop, n, prev[0], spec[0])
    assert Hh.circ_dist(inc_step[0] * hop, ri * hop).max() <= Hh.TOL_PHASE
    par.close()
