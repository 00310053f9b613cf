"""The N > 1 path on CPU: two gloo ranks shard the streams, run the round trip on their blocks and
agree on the slowest rank's time.  The arithmetic runs through the host-emulated build of the
kernel source (development harness); what is under test here is the sharding / reduction logic
that bench.py uses on GPUs."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions_exactly():
    from vocoder_amd.distributed import shard_range
    for total in (0, 1, 7, 8, 1024, 1025):
        for ws in (1, 2, 3, 8):
            blocks = [shard_range(total, ws, r) for r in range(ws)]
            assert blocks[0][0] == 0 and blocks[-1][1] == total
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(ws - 1))
            sizes = [hi - lo for lo, hi in blocks]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def _worker(rank, world_size, port, emu_path, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world_size),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import torch  # noqa: F401
    from vocoder_amd import distributed as D, vocoder as V, conduit as Cd
    from vocoder_amd._ffi import Library
    assert D.init("gloo") == world_size
    lib = Library(emu_path)
    rng = np.random.default_rng(123)                      # same streams on every rank
    x = rng.standard_normal((5, 64 + 16 * 9))
    win = 0.5 * (1 - np.cos(2 * np.pi * np.arange(64) / 64))
    par = V.VocoderParams(64, 16, win, lib=lib)
    lo, hi = D.shard_range(x.shape[0], world_size, rank)
    D.barrier()
    y, left, (aph, sph) = Cd.process_with(par, x[lo:hi])
    D.barrier()
    slowest = D.max_over_ranks(float(rank + 1))
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), lo=lo, hi=hi, y=y, aph=aph, sph=sph, slowest=slowest)
    D.finish()


def test_two_gloo_ranks_shard_streams(emu_lib, tmp_path):
    import torch.multiprocessing as mp
    from vocoder_amd import vocoder as V, conduit as Cd
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, emu_lib.path, str(tmp_path)), nprocs=2, join=True)
    rng = np.random.default_rng(123)
    x = rng.standard_normal((5, 64 + 16 * 9))
    win = 0.5 * (1 - np.cos(2 * np.pi * np.arange(64) / 64))
    par = V.VocoderParams(64, 16, win, lib=emu_lib)
    y_all, _, (a_all, s_all) = Cd.process_with(par, x)
    parts = [np.load(tmp_path / f"rank{r}.npz") for r in range(2)]
    assert (int(parts[0]["lo"]), int(parts[0]["hi"]), int(parts[1]["lo"]), int(parts[1]["hi"])) == (0, 3, 3, 5)
    assert np.array_equal(np.concatenate([p["y"] for p in parts]), y_all)      # sharding changes nothing
    assert np.array_equal(np.concatenate([p["aph"] for p in parts]), a_all)
    assert np.array_equal(np.concatenate([p["sph"] for p in parts]), s_all)
    assert all(float(p["slowest"]) == 2.0 for p in parts)                      # max over ranks reached everyone
