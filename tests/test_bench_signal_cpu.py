"""The metric signal of bench.py is a pure function of (seed, stream, sample index): the generator that fills
HBM (torch, run here on the CPU) and the host generator (numpy) agree with each other and with the Philox-4x32-10
restatement the filter tests use."""
import os
import sys

import numpy as np
import torch

import helpers as Hh

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_noise_term_is_philox_keyed_by_seed_with_counter_sample_stream():
    n = np.arange(123456, 123456 + 257, dtype=np.uint64)
    s = 1000
    c = Hh.philox4x32_10(n, np.full_like(n, s), 0, 0, bench.SEED & 0xFFFFFFFF, bench.SEED >> 32)
    u = (((c[0] << np.uint64(32)) | c[1]) >> np.uint64(11)).astype(np.float64) * 2.0 ** -53
    f, th = bench._stream_params([s])
    assert f[0] == 110.0 * 2.0 ** ((s % 72) / 12.0) and 0.0 <= th[0] < 2 * np.pi
    tone = 0.5 * np.sin((2 * np.pi / bench.RATE) * f[0] * n.astype(np.float64) + th[0])
    assert np.array_equal(bench.stream_samples_host(s, 123456, 123456 + 257), tone + 1e-2 * (2.0 * u - 1.0))


def test_device_generator_equals_host_generator():
    c, length, rank = 5, 4000, 3            # streams 15..19
    x = bench.make_streams(c, length, torch.device("cpu"), rank).numpy()
    for i in range(c):
        h = bench.stream_samples_host(rank * c + i, 0, length)
        # the noise term is bit-identical; sin may differ in its last bit between two libraries
        assert np.abs(x[i] - h).max() <= 2.0 ** -52
    assert np.abs(x).max() <= 0.51 and np.abs(x - 0.5 * np.sin(np.arcsin(np.clip(x / 0.5, -1, 1)))).max() <= 0.011
