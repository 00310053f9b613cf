"""Checks at the metric configuration's full shape (1024 streams x 60 s, N = 4096, H = 1024) and
the CPU-reference configuration's full shape (mono 60 s, N = 2048, H = 512), on a real MI355X.

At full size the oracle is run on a few sampled streams only; all streams are covered by
size-independent properties: identity of the round trip under a COLA window, and the telescoping
of the phase chain (with zero initial phases the accumulated synthesis phase equals the raw
analysis phase of the last frame modulo 2 pi)."""
import numpy as np
import pytest

import helpers as Hh

pytestmark = pytest.mark.gpu


def _streams(torch, c, length, dev, seed=0x5EED):
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    n = torch.arange(length, dtype=torch.float64, device=dev)
    s = torch.arange(c, device=dev)
    f = 110.0 * torch.pow(2.0, (s % 72).double() / 12.0)
    theta = torch.rand(c, generator=g, dtype=torch.float64, device=dev) * (2 * np.pi)
    x = torch.empty(c, length, dtype=torch.float64, device=dev)
    for lo in range(0, c, 64):          # build in slabs: keeps temporaries small
        hi = min(c, lo + 64)
        x[lo:hi] = 0.5 * torch.sin((2 * np.pi / 48000.0) * f[lo:hi, None] * n[None, :] + theta[lo:hi, None])
        x[lo:hi] += 1e-2 * (2 * torch.rand(hi - lo, length, generator=g, dtype=torch.float64, device=dev) - 1)
    return x


def test_metric_config_full_size(gpu_lib, orc):
    import torch
    from vocoder_amd import vocoder as V, conduit as Cd
    dev = torch.device("cuda:0")
    n, h, c, length = 4096, 1024, 1024, 2_880_000
    k = np.arange(n)
    win = 0.5 * (1 - np.cos(2 * np.pi * k / n))          # periodic Hann: sum_j w^2(i - jH) = 1.5 exactly
    par = V.VocoderParams(n, h, win, lib=gpu_lib)
    f = Cd.num_frames(par, length)
    assert f == 2809 and c * f == 2_876_416
    x = _streams(torch, c, length, dev)
    y = torch.empty(c, Cd.ola_length(par, f, h), dtype=torch.float64, device=dev)
    aph = torch.zeros(c, par.K, dtype=torch.float64, device=dev)
    sph = torch.zeros_like(aph)
    ws = torch.empty(Cd.roundtrip_workspace_bytes(par, length, c), dtype=torch.uint8, device=dev)
    Cd.process_dev(par, x, y, aph, sph, workspace=ws)
    torch.cuda.synchronize()
    # (1) identity away from the edges, every stream.  The bar is loose on purpose: the reference
    # accumulates the synthesis phase unwrapped (Vocoder.hs:190), so after 2809 frames the top bins
    # sit near 1e7 rad where one ulp is 2e-9 rad, and ANY implementation of the reference's
    # arithmetic (the oracle included) drifts from the exact identity by a few 1e-9.
    lo, hi = n, (f - 4) * h
    err = 0.0
    for s0 in range(0, c, 128):
        err = max(err, (y[s0:s0 + 128, lo:hi] - x[s0:s0 + 128, lo:hi]).abs().max().item())
    assert err <= 1e-7, err
    # (2) telescoping of the phase chain, every stream: with zero initial phases the accumulated
    # synthesis phase is the raw analysis phase of the last frame modulo 2 pi.  Checked on the bins that
    # carry signal in the last frame, told from the library's own STFT (the first half of the workspace
    # holds the moduli, [stream][frame][ldk]).
    d = torch.remainder(sph - aph + np.pi, 2 * np.pi) - np.pi
    ldk = (par.K + 1) & ~1
    mag_last = ws[: c * f * ldk * 8].view(torch.float64).view(c, f, ldk)[:, f - 1, :par.K]
    gate = mag_last >= Hh.GATE * mag_last.max(dim=1, keepdim=True).values
    assert gate.sum(dim=1).min().item() >= 3                    # every stream has its sine and neighbours
    assert d.abs()[gate].max().item() <= 1e-6
    mag_gate = None
    # (3) sampled streams against the oracle, full length
    for s in (0, 371, 1023):
        r = orc.roundtrip(n, h, win, x[s].cpu().numpy(), want_stft=True)
        assert Hh.rel_l2(y[s].cpu().numpy(), r["y"]) <= Hh.TOL_L2
        big = r["mag"][-1] >= Hh.GATE * r["mag"][-1].max()
        assert Hh.circ_dist(aph[s].cpu().numpy(), r["ana_phase"])[big].max() <= Hh.TOL_PHASE
        tol = max(Hh.TOL_PHASE, 8 * f ** 0.5 * np.spacing(np.abs(r["syn_phase"]).max()))
        assert Hh.circ_dist(sph[s].cpu().numpy(), r["syn_phase"])[big].max() <= tol
        assert np.abs(d[s].cpu().numpy())[big].max() <= 1e-6
        mag_gate = big
    assert mag_gate is not None
    par.close()


def test_cpu_reference_config_full_size(gpu_lib, orc):
    """vocoder-file identity on a 48 kHz mono 440 Hz sine, N = 2048, hop = 512, Blackman, 60 s."""
    from vocoder_amd import vocoder as V, conduit as Cd
    n, h, length = 2048, 512, 2_880_000
    win = orc.make_window("blackman", n)
    x = 0.5 * np.sin(2 * np.pi * 440.0 * np.arange(length) / 48000.0)
    par = V.VocoderParams(n, h, win, lib=gpu_lib)
    y, left, (aph, sph) = Cd.process_with(par, x)
    r = orc.roundtrip(n, h, win, x)
    assert r["nframes"] == 5622 and y.shape == r["y"].shape
    assert Hh.rel_l2(y, r["y"]) <= Hh.TOL_L2
    assert np.array_equal(left, r["leftover"])
    par.close()


def test_filtered_configs_full_size(gpu_lib, orc):
    """The filtered benchmark configurations at their full shapes, sampled streams against the oracle
    pipeline: `--pitchShiftInterpolate 2 --lowpassBrickwall 1000` on 1024 channels (N = 4096), and
    `--playSpeed 2` on 512 channels (N = 8192), without and with `--randomPhase` (the increments then
    come from the library's documented generator and the oracle is fed the same numbers)."""
    import torch
    from fractions import Fraction
    from vocoder_amd import vocoder as V, conduit as Cd, filter as Fl
    dev = torch.device("cuda:0")
    length = 2_880_000
    cases = [
        (4096, 1024, 1024, Fl.pitch_shift_interpolate(2.0) >> Fl.lowpass_brickwall(1000.0),
         [("pitch_shift", 2.0), ("lowpass_brickwall", 1000.0)], None),
        (8192, 2048, 512, Fl.play_speed(Fraction(2)), [], Fraction(2)),
        (8192, 2048, 512, Fl.play_speed(Fraction(2)) >> Fl.random_phase_filter(7), "random_phase", Fraction(2)),
    ]
    for n, h, c, filt, chain, speed in cases:
        win = 0.5 * (1 - np.cos(2 * np.pi * np.arange(n) / n))
        par = V.VocoderParams(n, h, win, lib=gpu_lib)
        x = _streams(torch, c, length, dev)
        y, aph, sph = Cd.process_filter_dev(par, filt.bind(par, 48000.0 / n), x)
        torch.cuda.synchronize()
        f_in = Cd.num_frames(par, length)
        f_out = f_in if speed is None else int(f_in * speed)
        assert y.shape == (c, Cd.ola_length(par, f_out, h))
        assert torch.isfinite(y).all()
        for s in (0, c - 1):
            if chain == "random_phase":
                from test_gpu_filters import _oracle_speed_then_random_phase
                yo, spo = _oracle_speed_then_random_phase(orc, n, h, win, x[s].cpu().numpy(), speed, 7, s)
                assert Hh.rel_l2(y[s].cpu().numpy(), yo) <= Hh.TOL_L2
                assert np.array_equal(sph[s].cpu().numpy(), spo)
                continue
            ref = Hh.oracle_filtered(orc, n, h, win, x[s].cpu().numpy(), 48000.0 / n, chain, speed)
            assert Hh.rel_l2(y[s].cpu().numpy(), ref["y"]) <= Hh.TOL_L2
        del x, y
        par.close()
        torch.cuda.empty_cache()
