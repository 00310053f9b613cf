"""Filter chains and playSpeed through the C ABI on CPU (host-emulated build of the kernel
source): index logic, chain order, error behaviour.  Checker: the filter oracle."""
import ctypes as C
import os
import sys
from fractions import Fraction

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from helpers import TOL_L2, oracle_filtered, random_phase_expected  # noqa: E402


def _hann(n):
    return 0.5 * (1 - np.cos(2 * np.pi * np.arange(n) / n))


@pytest.fixture
def par(emu_lib):
    from vocoder_amd import vocoder as V
    return V.VocoderParams(128, 32, _hann(128), lib=emu_lib)


def _stft(par, c=2, f=5, seed=0):
    rng = np.random.default_rng(seed)
    return rng.uniform(0, 1, (c, f, par.K)), rng.uniform(-1, 1, (c, f, par.K))


CHAINS = [
    [("amplify", 0.3)],
    [("lowpass_brickwall", 3000.0)],
    [("highpass_brickwall", 3000.0)],
    [("bandpass_brickwall", 1000.0, 5000.0)],
    [("bandstop_brickwall", 1000.0, 5000.0)],
    [("lowpass_butterworth", 4.0, 2000.0)],
    [("highpass_butterworth", 4.0, 2000.0)],
    [("bandpass_butterworth", 2.0, 1000.0, 4000.0)],
    [("bandstop_butterworth", 2.0, 1000.0, 4000.0)],
    [("pitch_shift", 2.0)],
    [("pitch_shift", 0.77)],
    [("convolution", (0.25, 0.5, 1.0, 0.5))],
    [("envelope", 3)],
    [("envelope", 9)],
    [("pitch_shift", 2.0), ("lowpass_brickwall", 1000.0)],
    [("lowpass_butterworth", 2.0, 3000.0), ("pitch_shift", 1.5), ("amplify", 2.0), ("envelope", 5), ("amplify", 0.5)],
]


def _describe(chain):
    from vocoder_amd import filter as Fl
    f = Fl.id_filter()
    for name, *prm in chain:
        fn = {"pitch_shift": Fl.pitch_shift_interpolate, "convolution": lambda k: Fl.convolution_filter(k),
              "envelope": Fl.envelope_filter}.get(name) or getattr(Fl, name)
        f = Fl.compose_filters(f, fn(*prm))
    return f


@pytest.mark.parametrize("chain", CHAINS, ids=lambda c: "+".join(s[0] for s in c))
def test_chain_matches_oracle(par, orc, chain):
    step = 48000.0 / par.N
    mag, inc = _stft(par)
    m, i = _describe(chain).bind(par, step).apply(mag, inc)
    chain_o = [(n, np.array(p[0])) if n == "convolution" else (n, *p) for n, *p in chain]
    mo, io = orc.apply_filter(step, chain_o, mag, inc)
    has_env = any(s[0] == "envelope" for s in chain)
    if has_env:   # log / exp come from different math libraries
        assert np.allclose(m, mo, rtol=1e-12, atol=1e-15)
    else:
        assert np.array_equal(m, mo)
    assert np.array_equal(i, io)


def test_gain_table_and_function(par, orc):
    from vocoder_amd import filter as Fl
    step = 375.0
    mag, inc = _stft(par, seed=4)
    fn = lambda x: 1.0 / (1.0 + x / 1000.0)
    m, i = Fl.linear_amplitude_filter(fn).bind(par, step).apply(mag, inc)
    g = np.array([fn(step * k) for k in range(par.K)])
    assert np.array_equal(m, mag * g) and np.array_equal(i, inc)


def test_identity_and_long_chain(par):
    from vocoder_amd import filter as Fl
    mag, inc = _stft(par, seed=5)
    m, i = Fl.id_filter().bind(par, 1.0).apply(mag, inc)
    assert np.array_equal(m, mag) and np.array_equal(i, inc)
    f = Fl.id_filter()
    for _ in range(15):                                   # more than one device chain holds: split in two
        f = f >> Fl.amplify(1.25)
    b = f.bind(par, 1.0)
    assert [k for k, _ in b.parts] == ["chain", "chain"]
    m, _ = b.apply(mag, inc)
    ref = mag.copy()
    for _ in range(15):
        ref = ref * 1.25
    assert np.array_equal(m, ref)


def test_random_phase_is_the_documented_philox_stream(par):
    from vocoder_amd import filter as Fl
    mag, inc = _stft(par, c=2, f=3, seed=6)
    seed = 0x1234_5678_9ABC
    m, i = (Fl.amplify(1.0) >> Fl.random_phase_filter(seed)).bind(par, 1.0).apply(mag, inc)
    assert np.array_equal(m, mag)
    exp = random_phase_expected(seed, 2, 3, par.K, position=1)
    assert np.array_equal(i, exp)
    assert i.min() >= 0 and i.max() < 2 * np.pi and abs(i.mean() - np.pi) < 0.2


def test_append_errors(par, emu_lib):
    d = emu_lib.dll
    h = C.c_void_p()
    assert d.voc_filter_create(par.handle, 375.0, C.byref(h)) == 0
    q = (C.c_double * 3)(1.0, 2.0, 3.0)
    assert d.voc_filter_append(h, 1, q, 2) == -1          # lowpassBrickwall takes one parameter
    assert d.voc_filter_append(h, 99, q, 1) == -1
    assert d.voc_filter_append(h, 9, q, 3) == -5          # gain table must have K entries
    q[0] = 0.0
    assert d.voc_filter_append(h, 10, q, 1) == -1         # pitch coefficient must be positive
    assert d.voc_filter_append(h, 12, q, 1) == -1         # envelope size >= 1
    assert d.voc_filter_length(h) == 0
    q[0] = 2.0
    for _ in range(12):
        assert d.voc_filter_append(h, 0, q, 1) == 0
    assert d.voc_filter_append(h, 0, q, 1) == -5          # chain full
    assert d.voc_filter_destroy(h) == 0
    assert d.voc_filter_apply_dev(None, None, None, 0, 0, 0, None) == -1


@pytest.mark.parametrize("coeff", [Fraction(1), Fraction(2), Fraction(1, 2), Fraction(3, 2), Fraction(0.3), Fraction(1.3)])
def test_play_speed_map_matches_state_machine(par, orc, coeff):
    from vocoder_amd import filter as Fl
    for f_in in (0, 1, 9, 100):
        assert np.array_equal(Fl.play_speed_map(par, coeff, f_in), orc.play_speed(coeff.numerator, coeff.denominator, f_in))
    assert par.lib.dll.voc_play_speed_map(1, 0, 4, None, 0) == -1
    buf = np.empty(2, dtype=np.int64)
    assert par.lib.dll.voc_play_speed_map(2, 1, 4, buf.ctypes.data_as(C.c_void_p), 2) == -5


@pytest.mark.parametrize("chain,speed", [
    ([("pitch_shift", 2.0), ("lowpass_brickwall", 3000.0)], None),
    ([("envelope", 8)], None),
    ([], Fraction(2)),
    ([("amplify", 0.5)], Fraction(2, 3)),
])
def test_filtered_pipeline_matches_oracle(par, orc, chain, speed):
    import torch
    from vocoder_amd import conduit as Cd, filter as Fl
    step = 48000.0 / par.N
    rng = np.random.default_rng(9)
    length = 128 + 32 * 14 + 5
    t = np.arange(length)
    x = np.stack([0.5 * np.sin(2 * np.pi * 0.03 * (c + 1) * t) + 0.01 * rng.uniform(-1, 1, length) for c in range(2)])
    f = _describe(chain)
    if speed is not None:
        f = f >> Fl.play_speed(speed)
    y, aph, sph = Cd.process_filter_dev(par, f.bind(par, step), torch.from_numpy(x))
    for c in range(2):
        ref = oracle_filtered(orc, par.N, par.H, _hann(128), x[c], step,
                              [(n, *p) for n, *p in chain], speed)
        assert y.shape[1] == ref["y"].size
        err = np.linalg.norm(y[c].numpy() - ref["y"]) / max(np.linalg.norm(ref["y"]), 1e-300)
        assert err <= TOL_L2, err


def test_interleaved_framing_equals_planar_per_channel(par, orc):
    """vocoder-audio cuts frames of W*C interleaved samples at hops of H*C and splits each frame
    per channel (Audio.hs:51-55): the same frames as cutting each planar channel at (W, H)."""
    import torch
    from vocoder_amd import conduit as Cd
    rng = np.random.default_rng(13)
    c, n = 3, 128 + 32 * 6 + 5
    planar = rng.integers(-1000, 1000, (c, n)).astype(np.float64)
    inter = planar.T.reshape(-1).copy()
    got = Cd.deinterleave_dev(par, torch.from_numpy(inter), c)
    assert np.array_equal(got.numpy(), planar)
    assert np.array_equal(Cd.interleave_dev(par, got).numpy(), inter)
    fr_inter, left = orc.gen_frames_of_e(128 * c, 32 * c, [inter])          # the reference's framing
    for ch in range(c):
        fr_ch, left_ch = orc.gen_frames_of_e(128, 32, [planar[ch]])
        assert np.array_equal(fr_inter.reshape(-1, 128, c)[:, :, ch], fr_ch)
        assert np.array_equal(left.reshape(-1, c)[:, ch], left_ch)


def _pcm16_expected(x, clip):
    """libsndfile's double -> short conversions, restated with numpy (rint rounds halves to even)."""
    if clip:
        v = x * 32768.0
        r = np.rint(np.clip(v, -32768.0, 32767.0))
        return np.where(v >= 32767.0, 32767, np.where(v <= -32768.0, -32768, r)).astype(np.int64).astype(np.int16)
    return (np.rint(x * 32767.0).astype(np.int64) & 0xffff).astype(np.uint16).view(np.int16)


def test_pcm16_edge_matches_the_libsndfile_convention(par):
    """File-edge conversions (parity unpinned: libsndfile is not part of the reference tree; the
    expected values restate its documented conversions)."""
    import torch
    from vocoder_amd import conduit as Cd
    rng = np.random.default_rng(3)
    c, n = 3, 1000
    pcm = rng.integers(-32768, 32768, n * c, dtype=np.int16)
    pcm[:4] = [-32768, 32767, 0, -1]
    planar = Cd.pcm16_to_planar_dev(par, torch.from_numpy(pcm), c).numpy()
    assert np.array_equal(planar, pcm.reshape(n, c).T.astype(np.float64) / 32768.0)
    x = rng.uniform(-1.2, 1.2, (c, n))
    x[0, :8] = [0.5 / 32767, 1.5 / 32767, 2.5 / 32767, -0.5 / 32767, 1.0, -1.0, 1.00002, -1.00004]   # ties, full scale, overflow
    x[1, :4] = [0.5 / 32768, 1.5 / 32768, 32766.5 / 32768, -32767.5 / 32768]
    for clip in (False, True):
        got = Cd.planar_to_pcm16_dev(par, torch.from_numpy(x), clip).numpy()
        assert np.array_equal(got.reshape(n, c).T, _pcm16_expected(x, clip)), clip
    # a file read and written back without processing: s/32768 * 32767 rounds back to s except towards full scale
    back = Cd.planar_to_pcm16_dev(par, torch.from_numpy(planar), False).numpy()
    assert np.abs(back.astype(np.int32) - pcm.astype(np.int32)).max() <= 1


def _mapped_vs_gathered(lib, n, hop, c, f_in, coeff, monkeypatch, syn_tile=0, slots=None, device=None):
    """voc_synthesis_mapped_dev against voc_gather_frames_dev + voc_synthesis_signal_dev: same bits."""
    import torch
    from vocoder_amd import vocoder as V, filter as Fl
    monkeypatch.setenv("VOC_SYN_TILE_LEN", str(syn_tile))
    if slots:
        monkeypatch.setenv("VOC_SYN_SLOTS", str(slots))
    else:
        monkeypatch.delenv("VOC_SYN_SLOTS", raising=False)
    par = V.VocoderParams(n, hop, _hann(n), lib=lib)
    ldk = (par.K + 1) & ~1
    rng = np.random.default_rng(12)
    dev = device or "cpu"
    mag = torch.from_numpy(rng.uniform(0, 1, (c, f_in, ldk))).to(dev)
    inc = torch.from_numpy(rng.uniform(-0.02, 0.02, (c, f_in, ldk))).to(dev)
    src = torch.from_numpy(Fl.play_speed_map(par, coeff, f_in)).to(dev)
    fo = src.numel()
    ylen = (fo - 1) * hop + n
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream) if dev != "cpu" else None
    ph0 = torch.from_numpy(rng.uniform(-3, 3, (c, par.K))).to(dev)

    def run(mag_t, mrows, inc_t, irows):
        ph, y = ph0.clone(), torch.empty((c, ylen), dtype=torch.float64, device=dev)
        par._call("voc_synthesis_mapped_dev", p(mag_t), mag_t.shape[1], p(mrows), p(inc_t), inc_t.shape[1], p(irows), ldk,
                  fo, c, p(ph), p(y), ylen, ylen, st)
        return y, ph

    mg, ig = mag[:, src].contiguous(), inc[:, src].contiguous()
    ph, y = ph0.clone(), torch.empty((c, ylen), dtype=torch.float64, device=dev)
    par._call("voc_synthesis_signal_dev", p(mg), p(ig), ldk, fo, c, p(ph), p(y), ylen, ylen, st)
    for args in ((mag, src, inc, src), (mg, None, inc, src), (mag, src, ig, None), (mg, None, ig, None)):
        y2, ph2 = run(*args)
        assert torch.equal(y2, y) and torch.equal(ph2, ph), [a is None for a in args]
    # entries outside the array are clamped into it, never followed
    wild = src.clone()
    wild[0], wild[-1] = -7, 10 ** 12
    y3, _ = run(mag, wild, inc, wild)
    y4, _ = run(mag, wild.clamp(0, f_in - 1), inc, wild.clamp(0, f_in - 1))
    assert torch.equal(y3, y4)
    d = par.lib.dll
    assert d.voc_synthesis_mapped_dev(par.handle, p(mag), f_in, None, p(inc), f_in, p(src), ldk, fo, c, p(ph), p(y), ylen,
                                      ylen, st) == (-5 if fo != f_in else 0)      # no map: the array must hold nframes rows
    assert d.voc_synthesis_mapped_dev(par.handle, p(mag), 0, p(src), p(inc), f_in, p(src), ldk, fo, c, p(ph), p(y), ylen,
                                      ylen, st) == -5


@pytest.mark.parametrize("n,hop,tile,slots", [(128, 32, 0, None), (128, 32, 5, None), (1024, 256, 4, 2), (1024, 256, 0, 2)])
def test_mapped_synthesis_equals_gather_then_synthesis(emu_lib, monkeypatch, n, hop, tile, slots):
    _mapped_vs_gathered(emu_lib, n, hop, 3, 9, Fraction(2), monkeypatch, tile, slots)
    _mapped_vs_gathered(emu_lib, n, hop, 3, 11, Fraction(2, 3), monkeypatch, tile, slots)


def test_play_speed_then_filters_lazy_equals_eager(par, emu_lib):
    """A chain after playSpeed gathers only the array it reads: random_phase gets a fresh ph_inc
    and leaves the moduli mapped; a gain gathers the moduli and leaves ph_inc mapped."""
    import torch
    from vocoder_amd import filter as Fl
    mag, inc = (torch.from_numpy(a) for a in _stft(par, c=2, f=7, seed=4))
    ldk = (par.K + 1) & ~1
    pad = lambda t: torch.nn.functional.pad(t, (0, ldk - par.K)).contiguous()
    mag, inc = pad(mag), pad(inc)
    for descr, lazy_mag, lazy_inc in (
            (Fl.play_speed(2) >> Fl.random_phase_filter(5), True, False),
            (Fl.play_speed(Fraction(3, 2)) >> Fl.amplify(0.25), False, True),
            (Fl.amplify(2.0) >> Fl.play_speed(2) >> Fl.play_speed(Fraction(1, 3)), True, True),
            (Fl.play_speed(2) >> Fl.pitch_shift_interpolate(1.5), False, False)):
        b = descr.bind(par, 375.0)
        me, ie = b.apply_dev(mag.clone(), inc.clone())
        ml, mr, il, ir, fo = b.apply_dev_lazy(mag.clone(), inc.clone())
        assert (mr is not None) == lazy_mag and (ir is not None) == lazy_inc and fo == me.shape[1]
        k = par.K                                   # (the pad column of a fresh ph_inc is never written)
        assert torch.equal((ml if mr is None else ml[:, mr])[..., :k], me[..., :k])
        assert torch.equal((il if ir is None else il[:, ir])[..., :k], ie[..., :k])


def test_add_filters_matches_oracle(par, orc):
    """addFilters f1 f2 (Filter.hs:58-59): both chains on the same frames, outputs added as
    complex numbers; host arrays, device arrays and inside a longer description."""
    import torch
    from vocoder_amd import filter as Fl
    step = 375.0
    mag, inc = _stft(par, c=2, f=4, seed=11)
    low = [("lowpass_butterworth", 2.0, 3000.0)]
    high = [("highpass_brickwall", 6000.0), ("amplify", 0.5)]
    descr = Fl.amplify(2.0) >> Fl.add_filters(_describe(low), _describe(high)) >> Fl.amplify(0.25)
    b = descr.bind(par, step)
    assert not b.changes_frame_count and b.frames_out(4) == 4

    m0, i0 = orc.apply_filter(step, [("amplify", 2.0)], mag, inc)
    (ma, ia), (mb, ib) = orc.apply_filter(step, low, m0, i0), orc.apply_filter(step, high, m0, i0)
    ms, is_ = np.empty_like(ma), np.empty_like(ia)
    for c in range(2):
        for f in range(4):
            ms[c, f], is_[c, f] = orc.add_frames(ma[c, f], ia[c, f], mb[c, f], ib[c, f])
    me, ie = orc.apply_filter(step, [("amplify", 0.25)], ms, is_)

    m, i = b.apply(mag, inc)
    assert np.allclose(m, me, rtol=1e-12, atol=1e-15)
    assert np.abs(np.angle(np.exp(1j * (i - ie)))).max() <= 1e-9
    ldk = (par.K + 1) & ~1
    pad = lambda a: torch.nn.functional.pad(torch.from_numpy(a), (0, ldk - par.K)).contiguous()
    md, idv = b.apply_dev(pad(mag), pad(inc))
    assert np.array_equal(md[..., :par.K].numpy(), m) and np.array_equal(idv[..., :par.K].numpy(), i)
    ml, mr, il, ir, fo = b.apply_dev_lazy(pad(mag), pad(inc))
    assert mr is None and ir is None and fo == 4 and torch.equal(ml[..., :par.K], md[..., :par.K])
    with pytest.raises(ValueError):
        Fl.add_filters(Fl.play_speed(2), Fl.id_filter()).bind(par, step).frames_out(4)


def test_binary_stream_operations_match_the_reference_definition(par, orc):
    """`--multiply` = mkBinary (*) (+), `--divide` = mkBinary (/) (-) (VocoderJack.hs:107-110,127-132);
    `add` is addFrames."""
    import torch
    from vocoder_amd import conduit as Cd, vocoder as V
    rng = np.random.default_rng(31)
    shape = (2, 3, par.K)
    m1, m2 = rng.uniform(0, 2, shape), rng.uniform(0.1, 2, shape)
    i1, i2 = rng.uniform(-1, 1, shape), rng.uniform(-1, 1, shape)
    m2[0, 0, 0] = 0.0                                        # division by a zero modulus: IEEE infinity / NaN, as in Haskell
    m1[0, 0, 1], m2[0, 0, 1] = 0.0, 0.0
    t = [torch.from_numpy(a.copy()) for a in (m1, i1, m2, i2)]
    for op in ("multiply", "divide"):
        mo, io = Cd.binary_frames_dev(par, op, (t[0], t[1]), (t[2], t[3]))
        me, ie = orc.binary_frames(op, m1, i1, m2, i2)
        assert np.array_equal(mo.numpy(), me, equal_nan=True) and np.array_equal(io.numpy(), ie)
    mo, io = Cd.binary_frames_dev(par, "add", (t[0], t[1]), (t[2], t[3]))
    ma, ia = V.add_frames(par, (m1, i1), (m2, i2))
    assert np.array_equal(mo.numpy(), ma, equal_nan=True) and np.array_equal(io.numpy(), ia, equal_nan=True)
    assert par.lib.dll.voc_binary_frames_dev(par.handle, 9, None, None, None, None, None, None, 0, None) == -1


@pytest.mark.parametrize("n,hop,chain", [
    (1024, 256, [("pitch_shift", 2.0), ("lowpass_brickwall", 1000.0)]),
    (1024, 128, [("pitch_shift", 1.3)]),
    (2048, 512, [("bandpass_butterworth", 2.0, 300.0, 3400.0)]),
    (1024, 256, [("lowpass_brickwall", 4000.0), ("pitch_shift", 1.5)]),      # not foldable: applied first
    (256, 64, [("pitch_shift", 2.0)]),                                        # no specialised kernel: applied first
])
def test_chain_folded_into_synthesis_equals_the_pass(emu_lib, monkeypatch, n, hop, chain):
    """voc_synthesis_filtered_dev on the host build of the kernels (index logic): the bits of the filter pass
    followed by plain synthesis, with few channels on few slots (tiles and the carry pass on mixed increments)."""
    import torch
    from vocoder_amd import vocoder as V, conduit as Cd
    monkeypatch.setenv("VOC_SYN_SLOTS", "4")
    par = V.VocoderParams(n, hop, np.hanning(n + 2)[1:-1], lib=emu_lib)
    filt = _describe(chain).bind(par, 48000.0 / n)
    rng = np.random.default_rng(n + hop)
    x = torch.from_numpy(rng.standard_normal((3, n + hop * 21 + 5)))
    s0 = torch.from_numpy(rng.uniform(-3, 3, (3, par.K)))
    y, _, sph = Cd.process_filter_dev(par, filt, x, syn_phase=s0.clone())
    monkeypatch.setenv("VOC_NO_FILTERED_SYNTHESIS", "1")
    y_ref, _, sph_ref = Cd.process_filter_dev(par, filt, x, syn_phase=s0.clone())
    assert torch.equal(y, y_ref) and torch.equal(sph, sph_ref)
    par.close()
