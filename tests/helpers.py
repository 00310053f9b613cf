"""Shared comparison logic: the same parity checks run against the product library on the GPU
(-m gpu) and against the host-emulated build of the kernel source (CPU, development harness)."""
import os

import numpy as np

# fp64 parity bars (north star): relative L2 <= 1e-9 on samples, <= 1e-9 rad on wrapped phases
TOL_L2 = 1e-9
TOL_PHASE = 1e-9
TOL_MAG = 1e-9
GATE = 1e-5  # phases are compared on bins with mag >= GATE * max(mag) of the frame


def circ_dist(a, b):
    """|a - b| modulo 2 pi."""
    return np.abs((np.asarray(a) - np.asarray(b) + np.pi) % (2 * np.pi) - np.pi)


def rel_l2(a, b):
    n = np.linalg.norm(b)
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / (n if n > 0 else 1.0)


def set_tiles(ana=0, syn=0):
    os.environ["VOC_ANA_TILE_LEN"] = str(ana)
    os.environ["VOC_SYN_TILE_LEN"] = str(syn)


def signal(rng, c, length, kind="noise"):
    if kind == "noise":
        return rng.standard_normal((c, length))
    n = np.arange(length)
    out = np.empty((c, length))
    for s in range(c):
        f = 110.0 * 2 ** ((s % 72) / 12.0)
        out[s] = 0.5 * np.sin(2 * np.pi * f * n / 48000.0 + rng.uniform(0, 2 * np.pi)) + 1e-2 * rng.uniform(-1, 1, length)
    return out


def check_fft_layer(lib, orc, n, w, h, rng, batch=3):
    from vocoder_amd import vocoder as V
    win = orc.make_window("blackman", w) + 0.01
    par = V.VocoderParams(n, h, win, lib=lib)
    x = rng.standard_normal((batch, n))
    spec = V.fft_execute_r2c(par, x)
    ref = np.stack([orc.fft_r2c(r) for r in x])
    assert rel_l2(spec, ref) <= TOL_L2
    assert np.all(spec[:, 0].imag == 0) and (n % 2 == 1 or np.all(spec[:, -1].imag == 0))
    back = V.fft_execute_c2r(par, ref)
    assert rel_l2(back, np.stack([orc.fft_c2r(s, n) for s in ref])) <= TOL_L2
    # imaginary parts of DC / Nyquist are ignored by C2R
    dirty = ref.copy()
    dirty[:, 0] += 3j
    if n % 2 == 0:          # an odd length has no Nyquist bin: its last bin is an ordinary complex one
        dirty[:, -1] -= 2j
    assert rel_l2(V.fft_execute_c2r(par, dirty), back) <= 1e-15
    fr = rng.standard_normal((batch, w))
    s2 = V.do_fft(par, fr)
    ref2 = np.stack([orc.do_fft(n, win, f) for f in fr])
    assert rel_l2(s2, ref2) <= TOL_L2
    f2 = V.do_ifft(par, ref2)
    assert rel_l2(f2, np.stack([orc.do_ifft(n, win, s) for s in ref2])) <= TOL_L2
    par.close()


def check_stages(lib, orc, n, w, h, f, c, rng, ana_tile=0, syn_tile=0, kind="noise", window="blackman"):
    from vocoder_amd import vocoder as V
    set_tiles(ana_tile, syn_tile)
    try:
        win = orc.make_window(window, w)
        par = V.VocoderParams(n, h, win, lib=lib)
        k = n // 2 + 1
        length = (f - 1) * h + w
        x = signal(rng, c, length, kind)
        frames = np.stack([np.stack([x[ch, j * h: j * h + w] for j in range(f)]) for ch in range(c)])
        ph0 = rng.uniform(-3, 3, (c, k))
        ph, (mag, inc) = V.analysis_stage(par, ph0, frames)
        ref = [orc.analysis_stage(n, h, win, ph0[ch], frames[ch]) for ch in range(c)]
        pho, mago, inco = (np.stack([r[i] for r in ref]) for i in range(3))
        scale = mago.max(axis=-1, keepdims=True)
        assert np.abs(mag - mago).max() <= TOL_MAG * mago.max()
        big = mago >= GATE * scale
        # phase increments are radians per sample: compare the per-hop advance modulo 2 pi
        assert circ_dist(inc * h, inco * h)[big].max() <= TOL_PHASE
        assert circ_dist(ph, pho)[big[:, -1, :]].max() <= TOL_PHASE
        # synthesis on identical STFT input: the phase chain is bit exact, frames to 1e-9
        sp0 = rng.uniform(-3, 3, (c, k))
        sph, fr = V.synthesis_stage(par, sp0, (mago, inco))
        sref = [orc.synthesis_stage(n, h, win, sp0[ch], mago[ch], inco[ch]) for ch in range(c)]
        spho, fro = (np.stack([r[i] for r in sref]) for i in range(2))
        assert np.array_equal(sph, spho), "synthesis phase accumulation must match the reference bit for bit"
        assert rel_l2(fr, fro) <= TOL_L2
        par.close()
    finally:
        set_tiles()


def check_roundtrip(lib, orc, n, w, h, length, c, rng, ana_tile=0, syn_tile=0, chunk=None, kind="noise",
                    window="hann", nonzero_phase=True):
    from vocoder_amd import vocoder as V, conduit as Cd
    set_tiles(ana_tile, syn_tile)
    try:
        win = orc.make_window(window, w)
        par = V.VocoderParams(n, h, win, lib=lib)
        k = n // 2 + 1
        x = signal(rng, c, length, kind)
        ph0 = rng.uniform(-3, 3, (c, k)) if nonzero_phase else np.zeros((c, k))
        sp0 = rng.uniform(-3, 3, (c, k)) if nonzero_phase else np.zeros((c, k))
        y, left, (aph, sph) = Cd.process_with(par, x, phases=(ph0, sp0), chunk=chunk)
        for ch in range(c):
            r = orc.roundtrip(n, h, win, x[ch], chunk=chunk, ana_phase=ph0[ch], syn_phase=sp0[ch], want_stft=True)
            assert y[ch].shape == r["y"].shape
            assert np.array_equal(left[ch], r["leftover"])
            if r["nframes"] == 0:
                assert np.array_equal(aph[ch], ph0[ch]) and np.array_equal(sph[ch], sp0[ch])
                continue
            assert rel_l2(y[ch], r["y"]) <= TOL_L2
            big = r["mag"][-1] >= GATE * r["mag"][-1].max()
            assert circ_dist(aph[ch], r["ana_phase"])[big].max() <= TOL_PHASE
            # the accumulated synthesis phase is unwrapped and grows with the frame count; two
            # implementations agree modulo 2 pi to a few ulp of its magnitude per frame
            tol = max(TOL_PHASE, 8 * r["nframes"] ** 0.5 * np.spacing(np.abs(r["syn_phase"]).max()))
            assert circ_dist(sph[ch], r["syn_phase"])[big].max() <= tol
        par.close()
        return y
    finally:
        set_tiles()


# ---------------------------------------------------------------------------------- filters
def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Philox-4x32-10 on numpy uint32 arrays (written from the published round function)."""
    import numpy as np
    c = [np.asarray(v, dtype=np.uint64) for v in np.broadcast_arrays(c0, c1, c2, c3)]
    k0, k1 = int(k0), int(k1)
    mask = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = np.uint64(0xD2511F53) * c[0]
        p1 = np.uint64(0xCD9E8D57) * c[2]
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & mask, p1 >> np.uint64(32), p1 & mask
        c = [hi1 ^ c[1] ^ np.uint64(k0), lo1, hi0 ^ c[3] ^ np.uint64(k1), lo0]
        k0 = (k0 + 0x9E3779B9) & 0xFFFFFFFF
        k1 = (k1 + 0xBB67AE85) & 0xFFFFFFFF
    return c


def random_phase_expected(seed, channels, frames, k, position, first_frame=0, first_channel=0):
    """ph_inc the library's random-phase operation writes, shape (channels, frames, k): 53 bits of
    Philox words 0,1 scaled to [0, 2 pi); counter (bin | position in the chain << 16, channel,
    frame low, frame high) with channel and frame counted in the whole signal, key = seed."""
    import numpy as np
    bins = (np.arange(k, dtype=np.uint64) | (np.uint64(position) << np.uint64(16)))[None, None, :]
    ch = (np.arange(channels, dtype=np.uint64) + np.uint64(first_channel))[:, None, None]
    fr = (np.arange(frames, dtype=np.uint64) + np.uint64(first_frame))[None, :, None]
    r = philox4x32_10(bins, ch, fr & np.uint64(0xFFFFFFFF), fr >> np.uint64(32), seed & 0xFFFFFFFF, seed >> 32)
    bits = ((r[0] << np.uint64(32)) | r[1]) >> np.uint64(11)
    return (bits.astype(np.float64) * 2.0 ** -53) * (2 * np.pi)


def oracle_filtered(orc, n, hop, window, x, step, chain, speed=None, chunk=None):
    """The reference pipeline with a filter on one channel, from oracle pieces: analysis and
    volumeFix (orc.roundtrip's STFT), the filter chain, playSpeed, synthesisStage, sumFramesE."""
    import numpy as np
    w = len(window)
    k = n // 2 + 1
    chunk = hop if chunk is None else chunk
    rt = orc.roundtrip(n, hop, window, x, chunk, want_stft=True)
    mag, inc = orc.apply_filter(step, chain, rt["mag"], rt["ph_inc"]) if chain else (rt["mag"], rt["ph_inc"])
    if speed is not None:
        src = orc.play_speed(speed.numerator, speed.denominator, mag.shape[0])
        mag, inc = mag[src], inc[src]
    f = mag.shape[0]
    y = np.zeros(orc.ola_length(f, w, hop, chunk))
    if f:
        sph, frames = orc.synthesis_stage(n, hop, window, np.zeros(k), mag, inc)
        for j in range(f):
            y[j * hop:j * hop + w] = y[j * hop:j * hop + w] + frames[j]
    else:
        sph = np.zeros(k)
    return dict(y=y, mag=mag, ph_inc=inc, syn_phase=sph, ana_phase=rt["ana_phase"])


def check_context_free_functions(lib, orc, rng, k, hop, n, batch=3):
    """The forms of analysisStep / synthesisStep / frameFromComplex / frameToComplex / addFrames that
    take the reference's own arguments (Vocoder.hs:158-163,186-190,213-222) against the oracle."""
    from vocoder_amd import vocoder as V
    spec = rng.standard_normal((batch, k)) + 1j * rng.standard_normal((batch, k))
    spec[0, 1] = 0.0
    spec[0, 0] = -2.0 + 0.0j
    prev = rng.uniform(-3, 3, (batch, k))
    ph, (mag, inc) = V.analysis_step(hop, n, prev, spec, lib=lib)
    for b in range(batch):
        rp, rm, ri = orc.analysis_step(hop, n, prev[b], spec[b])
        assert np.abs(mag[b] - rm).max() <= 4e-16 * rm.max()
        assert circ_dist(ph[b], rp).max() <= TOL_PHASE and circ_dist(inc[b] * hop, ri * hop).max() <= TOL_PHASE
    assert ph[0, 1] == 0.0 and ph[0, 0] == np.pi
    nph, sp = V.synthesis_step(hop, prev, (mag, inc), lib=lib)
    for b in range(batch):
        rp, rs = orc.synthesis_step(hop, prev[b], mag[b], inc[b])
        assert np.array_equal(nph[b], rp) and rel_l2(sp[b], rs) <= TOL_L2
    m, p = V.frame_from_complex(spec, lib=lib)
    assert np.array_equal(m, mag) and np.array_equal(p, ph)       # the same device function as analysisStep's
    back = V.frame_to_complex((m, p), lib=lib)
    assert rel_l2(back, spec) <= TOL_L2
    for b in range(batch):
        rm, rp = orc.frame_from_complex(spec[b])
        assert np.abs(m[b] - rm).max() <= 4e-16 * rm.max() and circ_dist(p[b], rp).max() <= TOL_PHASE
        assert rel_l2(back[b], orc.frame_to_complex(rm, rp)) <= TOL_L2
    m2, i2 = V.add_frames((mag, inc), (mag[::-1], inc[::-1]), lib=lib)
    for b in range(batch):
        rm, ri = orc.add_frames(mag[b], inc[b], mag[batch - 1 - b], inc[batch - 1 - b])
        assert rel_l2(m2[b], rm) <= TOL_L2
        assert circ_dist(i2[b], ri)[rm > 1e-9 * rm.max()].max() <= TOL_PHASE
