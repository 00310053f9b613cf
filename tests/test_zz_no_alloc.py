"""GPU: the round trip that allocates nothing (voc_roundtrip_static_dev, scratch in a static device
array of the library) against voc_roundtrip_dev with a caller workspace: same bits, on two calls in
a row, the second with a smaller input; sizes beyond the array are refused."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SCRATCH_BYTES = 64 << 20      # VOC_STATIC_SCRATCH_BYTES in include/vocoder_hip.h


def _hann(n):
    return 0.5 * (1 - np.cos(2 * np.pi * np.arange(n) / n))


def _both(par, x, chunk, seed):
    """(static entry point, existing entry point) on the same input: y, phases, leftover of each."""
    import torch
    from vocoder_amd import conduit as Cd
    c, length = x.shape
    f = Cd.num_frames(par, length)
    n_out = Cd.ola_length(par, f, chunk)
    keep = Cd.leftover_length(par, length)
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    a0 = (torch.rand((c, par.K), generator=g, dtype=torch.float64, device="cuda") - 0.5) * 6
    s0 = (torch.rand((c, par.K), generator=g, dtype=torch.float64, device="cuda") - 0.5) * 6
    res = []
    for static in (True, False):
        y = torch.full((c, n_out), float("nan"), dtype=torch.float64, device="cuda")
        left = torch.full((c, max(keep, 1)), float("nan"), dtype=torch.float64, device="cuda")
        aph, sph = a0.clone(), s0.clone()
        if static:
            Cd.process_static_dev(par, x, y, aph, sph, chunk=chunk, leftover=left)
        else:
            ws = torch.empty(Cd.roundtrip_workspace_bytes(par, length, c), dtype=torch.uint8, device="cuda")
            Cd.process_dev(par, x, y, aph, sph, chunk=chunk, leftover=left, workspace=ws)
        torch.cuda.synchronize()
        res.append((y, aph, sph, left[:, :keep]))
    return res


def test_static_scratch_roundtrip_equals_workspace_roundtrip(gpu_lib, orc):
    import torch
    from vocoder_amd import vocoder as V, conduit as Cd
    n, hop = 4096, 1024
    win = _hann(n)
    par = V.VocoderParams(n, hop, win, lib=gpu_lib)
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    # first call: 24 channels x 300 frames -- more than the array holds at once, so the channels run in groups;
    # second call, right behind it: a smaller input that leaves most of the array's old contents in place
    shapes = [(24, n + hop * 299 + 17, hop), (5, n + hop * 37 + 3, 1000)]
    for i, (c, length, chunk) in enumerate(shapes):
        per_channel = Cd.roundtrip_workspace_bytes(par, length, 1)
        assert per_channel <= SCRATCH_BYTES
        if i == 0:
            assert Cd.roundtrip_workspace_bytes(par, length, c) > SCRATCH_BYTES
        x = torch.randn((c, length), generator=g, dtype=torch.float64, device="cuda")
        (ys, as_, ss, ls), (yw, aw, sw, lw) = _both(par, x, chunk, seed=i)
        assert not torch.isnan(ys).any()
        assert torch.equal(ys, yw) and torch.equal(as_, aw) and torch.equal(ss, sw) and torch.equal(ls, lw)
    # the oracle directly: zero phases, one small call
    x = torch.randn((2, n + hop * 20), generator=g, dtype=torch.float64, device="cuda")
    y = torch.empty((2, Cd.ola_length(par, 21, hop)), dtype=torch.float64, device="cuda")
    aph = torch.zeros((2, par.K), dtype=torch.float64, device="cuda")
    sph = torch.zeros_like(aph)
    Cd.process_static_dev(par, x, y, aph, sph)
    torch.cuda.synchronize()
    for ch in range(2):
        r = orc.roundtrip(n, hop, win, x[ch].cpu().numpy())
        assert np.linalg.norm(y[ch].cpu().numpy() - r["y"]) / np.linalg.norm(r["y"]) <= 1e-9
    par.close()


def test_static_scratch_call_allocates_nothing_and_records_no_launch_times(gpu_lib):
    """The entry point's contract: no device memory is taken by the call (free memory of the device is the same
    before and after, with every tensor of the test allocated beforehand), and, launch timing being built on
    events the library would have to create, the call's launches are not counted while it is on."""
    import torch
    from vocoder_amd import vocoder as V, conduit as Cd
    n, hop = 4096, 1024
    par = V.VocoderParams(n, hop, _hann(n), lib=gpu_lib)
    c, length = 6, n + hop * 150
    x = torch.randn((c, length), dtype=torch.float64, device="cuda")
    y = torch.empty((c, Cd.ola_length(par, 151, hop)), dtype=torch.float64, device="cuda")
    aph = torch.zeros((c, par.K), dtype=torch.float64, device="cuda")
    sph = torch.zeros_like(aph)
    Cd.process_static_dev(par, x, y, aph, sph)      # (first call on a context: nothing left to set up afterwards)
    torch.cuda.synchronize()
    par.lib.dll.voc_profile_enable(par.handle, 1)
    ms, cnt = np.zeros(4), np.zeros(4, dtype=np.int64)
    par.lib.dll.voc_profile_read(par.handle, ms.ctypes.data, cnt.ctypes.data)
    free_before = torch.cuda.mem_get_info()[0]
    for _ in range(3):
        Cd.process_static_dev(par, x, y, aph, sph)
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free_before
    par.lib.dll.voc_profile_read(par.handle, ms.ctypes.data, cnt.ctypes.data)
    assert cnt.sum() == 0
    # the same context with a library-owned workspace does take memory and is timed: the check can fail
    ws_bytes = Cd.roundtrip_workspace_bytes(par, length, c)
    Cd.process_dev(par, x, y, aph, sph)
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] <= free_before - ws_bytes // 2
    par.lib.dll.voc_profile_read(par.handle, ms.ctypes.data, cnt.ctypes.data)
    assert cnt.sum() >= 2
    par.lib.dll.voc_profile_enable(par.handle, 0)
    par.close()


def test_static_scratch_refuses_what_does_not_fit(gpu_lib):
    import torch
    from vocoder_amd import vocoder as V, conduit as Cd
    n, hop = 4096, 1024
    par = V.VocoderParams(n, hop, _hann(n), lib=gpu_lib)
    length = n + hop * 2100                         # one channel: 2101 frames x 2 x 2050 doubles > 64 MiB
    assert Cd.roundtrip_workspace_bytes(par, length, 1) > SCRATCH_BYTES
    x = torch.zeros((1, length), dtype=torch.float64, device="cuda")
    y = torch.full((1, Cd.ola_length(par, 2101, hop)), 7.0, dtype=torch.float64, device="cuda")
    aph = torch.zeros((1, par.K), dtype=torch.float64, device="cuda")
    sph = torch.zeros_like(aph)
    p = lambda t: C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = par.lib.dll.voc_roundtrip_static_dev(par.handle, p(x), x.stride(0), length, 1, hop, p(y), y.stride(0), p(aph),
                                              p(sph), None, 0, st)
    assert rc == -7 and b"static" in par.lib.dll.voc_strerror(rc)
    torch.cuda.synchronize()
    assert (y == 7.0).all()                         # nothing was touched
    par.close()


@pytest.mark.parametrize("seed", range(16))
def test_static_scratch_on_random_shapes(gpu_lib, seed):
    """Random frame lengths, windows, hops, channel counts and chunk sizes, one call after the other on the shared
    array: the bits of voc_roundtrip_dev every time (lengths the entry point does not serve are refused, not run)."""
    import torch
    from vocoder_amd import vocoder as V, conduit as Cd
    rng = np.random.default_rng(17000 + seed)
    n = int(rng.choice([64, 256, 1024, 2048, 4096, 8192, 16384, 1000, 3000]))
    if seed % 3:
        w, h = n, max(1, n // int(rng.choice([2, 4, 8, 16])))
    else:
        w = int(rng.integers(max(3, n // 2), n + 1))
        h = int(rng.integers(max(1, w // 6), w + 1))
    c = int(rng.integers(1, 7))
    length = w + h * int(rng.integers(0, 40)) + int(rng.integers(0, h + 1))
    chunk = int(rng.choice([h, 1, 7, 1000]))
    par = V.VocoderParams(n, h, np.hanning(w + 2)[1:-1], lib=gpu_lib)
    x = torch.from_numpy(rng.standard_normal((c, length))).cuda()
    (ys, as_, ss, ls), (yw, aw, sw, lw) = _both(par, x, chunk, seed=seed)
    assert not torch.isnan(ys).any()
    assert torch.equal(ys, yw) and torch.equal(as_, aw) and torch.equal(ss, sw) and torch.equal(ls, lw)
    par.close()
