"""GPU: chunk-at-a-time streaming equals the one-call result bit for bit (host and device entry
points), with and without a filter chain."""
import os
import sys

import numpy as np
import pytest

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
