"""GPU: single-precision variant of the fused path against the double-precision oracle (bar of
the variant: relative L2 <= 2e-4 on output samples) and against the double-precision GPU path."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL_F32_L2 = 2e-4


def _hann(n):
    return 0.5 * (1 - np.cos(2 * np.pi * np.arange(n) / n))


@pytest.mark.parametrize("n,w,hop", [(4096, 4096, 1024), (2048, 1500, 512), (16384, 16384, 4096), (64, 64, 16)])
def test_f32_roundtrip_on_device(gpu_lib, orc, n, w, hop):
    import torch
    from vocoder_amd import vocoder as V, conduit as Cd
    win = _hann(w)
    p32 = V.VocoderParams(n, hop, win, lib=gpu_lib, precision="f32")
    rng = np.random.default_rng(n + 1)
    c, length = 3, w + hop * 60 + 17
    t = np.arange(length)
    x = np.stack([0.5 * np.sin(2 * np.pi * 0.0071 * (i + 1) * t + i) for i in range(c)]) + 0.01 * rng.uniform(-1, 1, (c, length))
    xf = x.astype(np.float32)
    y_host, left, (aph, sph) = Cd.process_with(p32, xf)

    xt = torch.from_numpy(xf).cuda()
    f = Cd.num_frames(p32, length)
    y = torch.empty((c, Cd.ola_length(p32, f, hop)), dtype=torch.float32, device="cuda")
    a = torch.zeros((c, p32.K), dtype=torch.float32, device="cuda")
    s = torch.zeros_like(a)
    ws = torch.empty(Cd.roundtrip_workspace_bytes(p32, length, c), dtype=torch.uint8, device="cuda")
    Cd.process_dev(p32, xt, y, a, s, workspace=ws)
    torch.cuda.synchronize()
    assert np.array_equal(y.cpu().numpy(), y_host)            # host and device entry points: same kernels
    assert np.array_equal(s.cpu().numpy(), sph)

    ref = orc.roundtrip(n, hop, win, xf[0].astype(np.float64), hop)
    err = np.linalg.norm(y_host[0] - ref["y"]) / np.linalg.norm(ref["y"])
    assert err <= TOL_F32_L2, err
    assert np.abs(sph).max() <= np.pi + 1e-6
