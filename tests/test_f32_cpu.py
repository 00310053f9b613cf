"""Single-precision variant of the fused path on CPU (host-emulated kernels) against the
double-precision oracle, to the variant's own bar: relative L2 <= 2e-4 on output samples."""
import numpy as np
import pytest

TOL_F32_L2 = 2e-4


def _hann(n):
    return 0.5 * (1 - np.cos(2 * np.pi * np.arange(n) / n))


@pytest.mark.parametrize("n,w,hop", [(64, 64, 16), (256, 200, 64), (1024, 1024, 256)])
def test_f32_roundtrip_against_double_oracle(emu_lib, orc, n, w, hop):
    from vocoder_amd import vocoder as V, conduit as Cd
    win = _hann(w)
    par = V.VocoderParams(n, hop, win, lib=emu_lib, precision="f32")
    rng = np.random.default_rng(n)
    c, length = 2, w + hop * 40 + 3
    t = np.arange(length)
    x = np.stack([0.5 * np.sin(2 * np.pi * 0.013 * (i + 1) * t) for i in range(c)]) + 0.01 * rng.uniform(-1, 1, (c, length))
    y, left, (aph, sph) = Cd.process_with(par, x)
    assert y.dtype == np.float32 and aph.dtype == np.float32
    for ch in range(c):
        ref = orc.roundtrip(n, hop, win, x[ch].astype(np.float32).astype(np.float64), hop)
        assert y.shape[1] == ref["y"].size
        err = np.linalg.norm(y[ch] - ref["y"]) / np.linalg.norm(ref["y"])
        assert err <= TOL_F32_L2, err
        assert np.array_equal(left[ch], x[ch].astype(np.float32)[length - left.shape[1]:])
        assert np.all(np.abs(sph[ch]) <= np.pi + 1e-6)                   # the float phase chain stays wrapped


def test_precisions_do_not_mix(emu_lib):
    from vocoder_amd import vocoder as V, conduit as Cd
    p32 = V.VocoderParams(64, 16, _hann(64), lib=emu_lib, precision="f32")
    p64 = V.VocoderParams(64, 16, _hann(64), lib=emu_lib)
    x = np.zeros((1, 200))
    fr, sp = np.zeros(64), np.zeros(33, dtype=np.complex128)
    assert emu_lib.dll.voc_do_fft(p32.handle, fr.ctypes.data, sp.ctypes.data, 1) == -2     # double form, float context
    assert emu_lib.dll.voc_do_fft_f32(p64.handle, fr.ctypes.data, sp.ctypes.data, 1) == -2
    ph = np.zeros(33)
    assert emu_lib.dll.voc_analysis_stage(p32.handle, ph.ctypes.data, x.ctypes.data, x.ctypes.data, x.ctypes.data, 1, 1) == -2
    assert emu_lib.dll.voc_analysis_stage_f32(p64.handle, ph.ctypes.data, x.ctypes.data, x.ctypes.data, x.ctypes.data, 1, 1) == -2
    y = np.empty((1, 200), dtype=np.float32)
    ph = np.zeros((1, 33), dtype=np.float32)
    xf = x.astype(np.float32)
    st = emu_lib.dll.voc_roundtrip_f32(p64.handle, xf.ctypes.data, 200, 200, 1, 16, y.ctypes.data, 200,
                                       ph.ctypes.data, ph.ctypes.data, None, 0)
    assert st == -2
    assert Cd.roundtrip_workspace_bytes(p32, 4096, 3) * 2 == Cd.roundtrip_workspace_bytes(p64, 4096, 3)


@pytest.mark.parametrize("n,w,hop", [(64, 64, 16), (256, 200, 64), (1024, 1024, 256)])
def test_f32_stage_and_signal_forms(emu_lib, orc, n, w, hop):
    """Single-precision forms of analysisStage / synthesisStage and of the two halves of the fused call: the stages
    against the double-precision oracle to the variant's bar, the halves against the fused call bit for bit."""
    import torch
    from vocoder_amd import vocoder as V, conduit as Cd
    win = _hann(w)
    par = V.VocoderParams(n, hop, win, lib=emu_lib, precision="f32")
    rng = np.random.default_rng(n + 7)
    c, f = 2, 9
    t = np.arange(w + hop * (f - 1))
    x = np.stack([0.5 * np.sin(2 * np.pi * 0.021 * (i + 1) * t) for i in range(c)]) + 0.01 * rng.uniform(-1, 1, (c, t.size))
    x32 = x.astype(np.float32)
    frames = np.stack([[x32[ch, j * hop:j * hop + w] for j in range(f)] for ch in range(c)])
    aph, (mag, inc) = V.analysis_stage(par, np.zeros((c, par.K), dtype=np.float32), frames)
    assert mag.dtype == np.float32 and aph.dtype == np.float32
    sph, out = V.synthesis_stage(par, np.zeros((c, par.K), dtype=np.float32), (mag, inc))
    for ch in range(c):
        _, rmag, rinc = orc.analysis_stage(n, hop, win, np.zeros(par.K), frames[ch].astype(np.float64))
        assert np.linalg.norm(mag[ch] - rmag) <= 1e-5 * np.linalg.norm(rmag)
        _, rframes = orc.synthesis_stage(n, hop, win, np.zeros(par.K), rmag, rinc)
        assert np.linalg.norm(out[ch] - rframes) <= TOL_F32_L2 * np.linalg.norm(rframes)
    # one frame at a time (analysisBlock / synthesisBlock) continues the same chain
    a1, (m1, i1) = V.analysis_block(par, np.zeros(par.K, dtype=np.float32), frames[0, 0])
    assert np.array_equal(m1, mag[0, 0]) and np.array_equal(i1, inc[0, 0])
    # the two halves of the fused call
    xt = torch.from_numpy(x32)
    y_ref, _, (a_ref, s_ref) = Cd.process_with(par, x32)
    y, a2, s2 = Cd.process_filter_dev(par, None, xt)
    assert np.array_equal(y.numpy(), y_ref) and np.array_equal(a2.numpy(), a_ref) and np.array_equal(s2.numpy(), s_ref)


def test_f32_filter_chain_on_a_float_stft(emu_lib, orc):
    """A chain on a float STFT computes in double and rounds once: equal to the double chain on the widened rows,
    rounded; the filtered single-precision pipeline (play_speed and random phase included) runs end to end."""
    import torch
    from fractions import Fraction
    from vocoder_amd import vocoder as V, conduit as Cd, filter as Fl
    n, hop = 128, 32
    win = _hann(n)
    p32 = V.VocoderParams(n, hop, win, lib=emu_lib, precision="f32")
    p64 = V.VocoderParams(n, hop, win, lib=emu_lib)
    rng = np.random.default_rng(4)
    mag = rng.uniform(0, 1, (2, 5, p32.K)).astype(np.float32)
    inc = rng.uniform(-1, 1, (2, 5, p32.K)).astype(np.float32)
    desc = Fl.pitch_shift_interpolate(1.5) >> Fl.lowpass_butterworth(2.0, 5000.0) >> Fl.amplify(0.7)
    m32, i32 = desc.bind(p32, 375.0).apply(mag, inc)
    m64, i64 = desc.bind(p64, 375.0).apply(mag.astype(np.float64), inc.astype(np.float64))
    assert m32.dtype == np.float32
    assert np.array_equal(m32, m64.astype(np.float32)) and np.array_equal(i32, i64.astype(np.float32))
    # device path with a speed change and a random-phase stage behind it
    x = (0.5 * np.sin(2 * np.pi * 0.03 * np.arange(n + hop * 20)) + 0.01 * rng.uniform(-1, 1, n + hop * 20)).astype(np.float32).reshape(1, -1).copy()
    filt = (Fl.lowpass_brickwall(6000.0) >> Fl.play_speed(Fraction(3, 2)) >> Fl.random_phase_filter(3)).bind(p32, 375.0)
    y, _, sph = Cd.process_filter_dev(p32, filt, torch.from_numpy(x))
    ref = Cd.process_filter_dev(p64, (Fl.lowpass_brickwall(6000.0) >> Fl.play_speed(Fraction(3, 2)) >>
                                      Fl.random_phase_filter(3)).bind(p64, 375.0), torch.from_numpy(x.astype(np.float64)))[0]
    assert y.dtype == torch.float32 and y.shape == ref.shape
    assert np.linalg.norm(y.numpy() - ref.numpy()) <= 5e-3 * np.linalg.norm(ref.numpy())   # random phases: every bin carries energy


@pytest.mark.parametrize("dunai", [False, True])
def test_f32_stream_equals_the_f32_batch_path(emu_lib, dunai):
    """A stream on a single-precision context holds float state: pushes of any sizes give the bits of the fused
    single-precision call (conduit conventions), and a filtered stream those of the filtered batch path."""
    import torch
    from vocoder_amd import vocoder as V, conduit as Cd, stream as S, filter as Fl
    n, hop = 128, 32
    par = V.VocoderParams(n, hop, _hann(n), lib=emu_lib, precision="f32")
    rng = np.random.default_rng(31)
    x = rng.uniform(-1, 1, (2, n + hop * 24 + (0 if dunai else 5))).astype(np.float32)
    if dunai:
        x = x[:, :hop * 27]
        st = S.Stream(par, 2, dunai=True)
        outs = [st.push(x[:, :hop * 5]), st.push(x[:, hop * 5:hop * 6]), st.push(x[:, hop * 6:])]
        y = np.concatenate(outs, axis=1)
        assert y.dtype == np.float32 and y.shape == x.shape
        lag = ((n - 1) // hop) * hop                       # the zero pre-pad delays the signal (Dunai.hs:59-62)
        err = np.linalg.norm(y[:, lag + n:] - x[:, n:x.shape[1] - lag]) / np.linalg.norm(x[:, n:x.shape[1] - lag])
        assert err <= TOL_F32_L2, err
        return
    y_ref, left_ref, (a_ref, s_ref) = Cd.process_with(par, x)
    st = S.Stream(par, 2)
    outs = [st.push(x[:, :100]), st.push(x[:, 100:101]), st.push(x[:, 101:700]), st.push(x[:, 700:]), st.flush()]
    y = np.concatenate(outs, axis=1)
    assert y.dtype == np.float32 and np.array_equal(y, y_ref)
    a, sp = st.phases()
    assert np.array_equal(a, a_ref) and np.array_equal(sp, s_ref) and np.array_equal(st.leftover(), left_ref)
    filt = (Fl.lowpass_butterworth(2.0, 6000.0) >> Fl.play_speed(2)).bind(par, 375.0)
    yf_ref = Cd.process_filter_dev(par, filt, torch.from_numpy(x))[0].numpy()
    sf = S.Stream(par, 2, filt=filt)
    yf = np.concatenate([sf.push(x[:, :333]), sf.push(x[:, 333:]), sf.flush()], axis=1)
    assert np.array_equal(yf, yf_ref)


@pytest.mark.parametrize("n,w,hop", [(256, 256, 64), (64, 48, 16), (1024, 1024, 256)])
def test_f32_fft_layer_steps_and_add_frames(emu_lib, orc, n, w, hop):
    """The FFT-compatible layer, the step level and addFrames on float data (voc_r2c_f32 ... voc_add_frames_f32)
    against the double-precision oracle, to the float variant's accuracy."""
    from vocoder_amd import vocoder as V
    win = _hann(w)
    p32 = V.VocoderParams(n, hop, win, lib=emu_lib, precision="f32")
    k = p32.K
    rng = np.random.default_rng(n + hop)
    x = rng.standard_normal((3, n)).astype(np.float32)
    spec = V.fft_execute_r2c(p32, x)
    assert spec.dtype == np.complex64
    ref = np.stack([orc.fft_r2c(r.astype(np.float64)) for r in x])
    assert np.linalg.norm(spec - ref) / np.linalg.norm(ref) <= 2e-6
    back = V.fft_execute_c2r(p32, spec)
    assert back.dtype == np.float32 and np.abs(back - x).max() <= 2e-5
    fr = rng.standard_normal((3, w)).astype(np.float32)
    s2 = V.do_fft(p32, fr)
    ref2 = np.stack([orc.do_fft(n, win, f.astype(np.float64)) for f in fr])
    assert np.linalg.norm(s2 - ref2) / np.linalg.norm(ref2) <= 2e-6
    f2 = V.do_ifft(p32, s2)
    reff = np.stack([orc.do_ifft(n, win, s.astype(np.complex128)) for s in s2])
    assert np.linalg.norm(f2 - reff) / np.linalg.norm(reff) <= 2e-6
    prev = rng.uniform(-3, 3, (3, k)).astype(np.float32)
    ph, (mag, inc) = V.analysis_step(p32, prev, s2)
    assert ph.dtype == np.float32
    for b in range(3):
        rp, rm, ri = orc.analysis_step(hop, n, prev[b].astype(np.float64), s2[b].astype(np.complex128))
        assert np.abs(mag[b] - rm).max() <= 1e-6 * rm.max()
        big = rm >= 1e-3 * rm.max()
        d = np.abs(ph[b] - rp) % (2 * np.pi)
        assert np.minimum(d, 2 * np.pi - d)[big].max() <= 1e-4
        # the per-hop advance: a float increment carries 6e-8 of a value of up to a few thousand radians
        d = np.abs(inc[b].astype(np.float64) * hop - ri * hop) % (2 * np.pi)
        assert np.minimum(d, 2 * np.pi - d)[big].max() <= 1e-3
    nph, sp = V.synthesis_step(p32, prev, (mag, inc))
    assert np.abs(nph).max() <= np.pi + 1e-6 and sp.dtype == np.complex64
    for b in range(3):
        rp, rs = orc.synthesis_step(hop, prev[b].astype(np.float64), mag[b].astype(np.float64), inc[b].astype(np.float64))
        d = np.abs(nph[b] - rp) % (2 * np.pi)
        assert np.minimum(d, 2 * np.pi - d).max() <= 1e-3
        assert np.linalg.norm(sp[b] - rs) / np.linalg.norm(rs) <= 1e-3
    m2, i2 = V.add_frames(p32, (mag, prev), (mag[::-1].copy(), prev[::-1].copy()))
    for b in range(3):
        rm, ri = orc.add_frames(mag[b].astype(np.float64), prev[b].astype(np.float64), mag[2 - b].astype(np.float64),
                                prev[2 - b].astype(np.float64))
        assert np.linalg.norm(m2[b] - rm) / np.linalg.norm(rm) <= 1e-5
    # a double-precision context refuses the float forms and the other way round
    p64 = V.VocoderParams(n, hop, win, lib=emu_lib)
    assert emu_lib.dll.voc_r2c_f32(p64.handle, x.ctypes.data, spec.ctypes.data, 1) == -2
    assert emu_lib.dll.voc_r2c(p32.handle, x.ctypes.data, spec.ctypes.data, 1) == -2
    p32.close()
    p64.close()


@pytest.mark.parametrize("ana16,syn16", [(0, 0), (1, 0), (1, 1)])
@pytest.mark.parametrize("n,hop,nf,at,st", [(4096, 1024, 9, 0, 0), (4096, 512, 14, 4, 5), (8192, 2048, 6, 3, 3)])
def test_f32_eight_and_sixteen_points_per_thread(emu_lib, orc, monkeypatch, n, hop, nf, at, st, ana16, syn16):
    """Single-precision contexts take the sixteen-point analysis kernel at N >= 4096 (voc_kernels_fast16.h); every
    combination of the two builds is held to the variant's bar against the double-precision oracle."""
    from vocoder_amd import vocoder as V, conduit as Cd
    monkeypatch.setenv("VOC_FAST16_F32_ANA", str(ana16))
    monkeypatch.setenv("VOC_FAST16_F32_SYN", str(syn16))
    if at:
        monkeypatch.setenv("VOC_ANA_TILE_LEN", str(at))
        monkeypatch.setenv("VOC_SYN_TILE_LEN", str(st))
    win = _hann(n)
    par = V.VocoderParams(n, hop, win, lib=emu_lib, precision="f32")
    rng = np.random.default_rng(n + hop + nf)
    c, length = 2, n + hop * (nf - 1) + 6
    t = np.arange(length)
    x = (np.stack([0.5 * np.sin(2 * np.pi * 0.0131 * (i + 1) * t) for i in range(c)]) + 0.01 * rng.uniform(-1, 1, (c, length))).astype(np.float32)
    y, left, (aph, sph) = Cd.process_with(par, x)
    for ch in range(c):
        ref = orc.roundtrip(n, hop, win, x[ch].astype(np.float64), hop)
        assert y.shape[1] == ref["y"].size
        err = np.linalg.norm(y[ch] - ref["y"]) / np.linalg.norm(ref["y"])
        assert err <= TOL_F32_L2, err
    par.close()
