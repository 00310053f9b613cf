"""Kernel index logic on the CPU: the kernel SOURCE (vocoder_amd/csrc/voc_kernels.h and the C-ABI
layer) compiled for the host through tools/emu/hip_emu.h -- one std::thread per HIP thread, real
barriers -- and checked against the oracle.  This is a development harness for machines without
a GPU; it is not a code path of the product (which fails when libvocoder_hip.so is missing) and
claims nothing about the GPU build's numerics.  The parity tests proper are in test_gpu_*.py."""
import numpy as np
import pytest

import helpers as Hh


@pytest.mark.parametrize("n,w,h", [(8, 8, 2), (16, 16, 4), (64, 48, 16), (128, 100, 30), (512, 512, 128)])
def test_fft_layer(emu_lib, orc, n, w, h):
    Hh.check_fft_layer(emu_lib, orc, n, w, h, np.random.default_rng(n))


@pytest.mark.parametrize("n,w,h,f,c,at,st", [
    (64, 64, 16, 9, 2, 0, 0),      # one tile per channel
    (64, 48, 16, 23, 2, 4, 5),     # zero-padded frames, several tiles with carries
    (128, 100, 30, 17, 1, 3, 7),   # hop that does not divide the window
    (64, 64, 64, 7, 1, 2, 3),      # no overlap
])
def test_stages(emu_lib, orc, n, w, h, f, c, at, st):
    Hh.check_stages(emu_lib, orc, n, w, h, f, c, np.random.default_rng(1), ana_tile=at, syn_tile=st)


@pytest.mark.parametrize("n,w,h,length,c,at,st,chunk", [
    (64, 64, 16, 64 + 16 * 12 + 5, 2, 0, 0, None),
    (64, 48, 16, 48 + 16 * 22 + 3, 2, 4, 5, 48),   # tiles with halo recompute, chunk != hop
    (64, 32, 40, 32 + 40 * 8, 1, 2, 4, None),      # hop larger than the window: gaps of zeros
    (128, 128, 32, 100, 2, 0, 0, None),            # shorter than one frame: no output, all leftover
    (128, 128, 32, 128, 1, 0, 0, 50),              # exactly one frame
])
def test_roundtrip(emu_lib, orc, n, w, h, length, c, at, st, chunk):
    Hh.check_roundtrip(emu_lib, orc, n, w, h, length, c, np.random.default_rng(2), ana_tile=at, syn_tile=st, chunk=chunk)


@pytest.mark.parametrize("ana16,syn16", [(1, 0), (1, 1)])
@pytest.mark.parametrize("n,h,nf,c,at,st", [(4096, 1024, 9, 2, 0, 0), (4096, 1024, 14, 1, 4, 5), (4096, 256, 20, 1, 6, 8),
                                            (4096, 2048, 7, 1, 3, 3), (8192, 2048, 6, 1, 0, 0), (8192, 1024, 9, 1, 4, 4)])
def test_sixteen_points_per_thread(emu_lib, orc, monkeypatch, n, h, nf, c, at, st, ana16, syn16):
    """Index logic of the kernels with sixteen points per thread (N = 4096 / 8192: radix 16 * 16 * 8 or 16)."""
    monkeypatch.setenv("VOC_FAST16_ANA", str(ana16))
    monkeypatch.setenv("VOC_FAST16_SYN", str(syn16))
    Hh.check_roundtrip(emu_lib, orc, n, n, h, n + h * (nf - 1) + 6, c, np.random.default_rng(n + h), ana_tile=at, syn_tile=st)


def test_steps_and_framing(emu_lib, orc):
    from vocoder_amd import vocoder as V, conduit as Cd
    rng = np.random.default_rng(3)
    n, w, h = 64, 48, 12
    win = orc.make_window("hamming", w)
    par = V.VocoderParams(n, h, win, lib=emu_lib)
    k = par.K
    spec = rng.standard_normal((4, k)) + 1j * rng.standard_normal((4, k))
    prev = rng.uniform(-3, 3, (4, k))
    ph, (mag, inc) = V.analysis_step(par, prev, spec)
    for b in range(4):
        rp, rm, ri = orc.analysis_step(h, n, prev[b], spec[b])
        assert np.abs(mag[b] - rm).max() <= 1e-15 * rm.max()
        assert Hh.circ_dist(ph[b], rp).max() <= 1e-14 and Hh.circ_dist(inc[b] * h, ri * h).max() <= 1e-12
    nph, sp = V.synthesis_step(par, prev, (mag, inc))
    for b in range(4):
        rp, rs = orc.synthesis_step(h, prev[b], mag[b], inc[b])
        assert np.array_equal(nph[b], rp) and np.abs(sp[b] - rs).max() <= 1e-14 * np.abs(rs).max()
    x = rng.standard_normal((2, 333))
    frames, left = Cd.frames_of_e(par, x)
    for ch in range(2):
        rf, rl = orc.gen_frames_of_e(w, h, [x[ch]])
        assert np.array_equal(frames[ch], rf) and np.array_equal(left[ch], rl)
    for chunk in (h, 7, 100):
        y = Cd.sum_frames_e(par, frames, chunk)
        for ch in range(2):
            assert np.array_equal(y[ch], orc.sum_frames_e(chunk, h, list(frames[ch])).ravel())
    assert par.lib.dll.voc_volume_coeff(par.handle) == orc.volume_coeff(h, win)
    par.close()


def test_functions_with_the_reference_argument_lists(emu_lib, orc):
    Hh.check_context_free_functions(emu_lib, orc, np.random.default_rng(8), k=37, hop=12, n=72)
    Hh.check_context_free_functions(emu_lib, orc, np.random.default_rng(9), k=33, hop=16, n=64)


@pytest.mark.parametrize("slots,c", [(2, 5), (2, 4), (3, 4), (4, 5)])
def test_uncut_leading_channels_equal_fully_tiled_result(emu_lib, orc, slots, c, monkeypatch):
    """With more channels than workgroup slots, whole rounds of channels run as one synthesis tile each
    and only the last partial round is cut (and carried): same bits as the all-cut schedule."""
    from vocoder_amd import vocoder as V, conduit as Cd
    n, h = 1024, 256
    win = 0.5 * (1 - np.cos(2 * np.pi * np.arange(n) / n))
    rng = np.random.default_rng(slots * 10 + c)
    x = rng.standard_normal((c, n + h * 70 + 9))
    ph = (rng.uniform(-3, 3, (c, n // 2 + 1)), rng.uniform(-3, 3, (c, n // 2 + 1)))
    monkeypatch.setenv("VOC_SYN_TILE_LEN", "0")
    monkeypatch.delenv("VOC_SYN_SLOTS", raising=False)
    ref = Cd.process_with(V.VocoderParams(n, h, win, lib=emu_lib), x, phases=ph)
    monkeypatch.setenv("VOC_SYN_SLOTS", str(slots))
    got = Cd.process_with(V.VocoderParams(n, h, win, lib=emu_lib), x, phases=ph)
    assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])
    assert np.array_equal(got[2][0], ref[2][0]) and np.array_equal(got[2][1], ref[2][1])


def test_uncut_leading_channels_in_the_signal_level_entry(emu_lib, monkeypatch):
    import torch
    from vocoder_amd import vocoder as V, conduit as Cd
    n, h, c = 1024, 256, 5
    win = 0.5 * (1 - np.cos(2 * np.pi * np.arange(n) / n))
    x = np.random.default_rng(77).standard_normal((c, n + h * 70 + 9))
    monkeypatch.delenv("VOC_SYN_SLOTS", raising=False)
    ref = Cd.process_with(V.VocoderParams(n, h, win, lib=emu_lib), x)
    monkeypatch.setenv("VOC_SYN_SLOTS", "2")
    y, aph, sph = Cd.process_filter_dev(V.VocoderParams(n, h, win, lib=emu_lib), None, torch.from_numpy(x))
    assert np.array_equal(y.numpy(), ref[0]) and np.array_equal(sph.numpy(), ref[2][1])
