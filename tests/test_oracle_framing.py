"""The oracle's framing / overlap-add state machines (restated from Frames.hs and Dunai.hs)
against the reference's list models, on random integer inputs -- the properties of
vocoder-conduit/test/main.hs:32-33 and vocoder-dunai/test/main.hs:54-63."""
import numpy as np
import pytest

from listmodels import list_frames_of, list_frames_of_e, list_sum_frames, list_sum_frames_e


def _chunks(rng, nmax=8, lmax=40, nonempty=False):
    n = int(rng.integers(0, nmax + 1))
    return [list(map(int, rng.integers(-50, 50, int(rng.integers(1 if nonempty else 0, lmax + 1))))) for _ in range(n)]


@pytest.mark.parametrize("seed", range(60))
def test_frames_of_e_matches_list_model(orc, seed):
    # prop "framesOfE": chunkSize = hopSize + NonNegative, hopSize > 0, arbitrary [[Int]]
    rng = np.random.default_rng(seed)
    hop = int(rng.integers(1, 12))
    chunk_size = hop + int(rng.integers(0, 20))
    inp = _chunks(rng)
    frames, _ = orc.gen_frames_of_e(chunk_size, hop, inp, dtype=np.int64)
    assert frames.tolist() == list_frames_of_e(chunk_size, hop, inp)


@pytest.mark.parametrize("seed", range(60))
def test_sum_frames_e_matches_list_model(orc, seed):
    # prop "sumFramesE": any positive chunkSize, hopSize; frames are non-empty lists of any length
    rng = np.random.default_rng(1000 + seed)
    hop = int(rng.integers(1, 12))
    chunk_size = int(rng.integers(1, 25))
    inp = _chunks(rng, nonempty=True)
    out = orc.sum_frames_e(chunk_size, hop, inp, dtype=np.int64)
    assert out.tolist() == list_sum_frames_e(chunk_size, hop, inp)


def test_frames_of_e_leftover_and_floor_division(orc):
    # buffer shorter than the frame: drop count ((len-W) div H)*H+H is <= 0 under flooring div, so
    # the whole buffer stays queued (Frames.hs:36-37)
    frames, left = orc.gen_frames_of_e(10, 3, [[1, 2, 3, 4]], dtype=np.int64)
    assert frames.shape[0] == 0 and left.tolist() == [1, 2, 3, 4]
    frames, left = orc.gen_frames_of_e(10, 3, [[1, 2, 3, 4], [5, 6, 7], list(range(8, 20))], dtype=np.int64)
    full = list(range(1, 20))
    assert frames.tolist() == list_frames_of_e(10, 3, [full])
    nf = frames.shape[0]
    assert left.tolist() == full[nf * 3:]
    # initial queue (genFramesOfE's third argument) is prepended
    frames2, _ = orc.gen_frames_of_e(10, 3, [list(range(5, 20))], q0=[1, 2, 3, 4], dtype=np.int64)
    assert frames2.tolist() == frames.tolist()


def test_sum_frames_e_last_chunk_zero_padded(orc):
    out = orc.sum_frames_e(4, 2, [[1, 1, 1, 1, 1], [1, 1, 1, 1, 1]], dtype=np.int64)
    assert out.tolist() == [[1, 1, 2, 2], [2, 1, 1, 0]]
    assert orc.sum_frames_e(4, 2, [], dtype=np.int64).shape[0] == 0
    assert orc.ola_length(2, 5, 2, 4) == 8 and orc.ola_length(0, 5, 2, 4) == 0


@pytest.mark.parametrize("seed", range(40))
def test_frames_of_s_matches_list_model(orc, seed):
    # prop "framesOfS": sizes are multiples of the hop; the model prepends chunkSize - hopSize zeros
    rng = np.random.default_rng(2000 + seed)
    hop = int(rng.integers(1, 8))
    chunk_size = int(rng.integers(1, 8)) * hop
    block = int(rng.integers(1, 8)) * hop
    ticks = [list(map(int, rng.integers(-50, 50, block))) for _ in range(int(rng.integers(0, 8)))]
    got = orc.frames_of_s(chunk_size, hop, ticks, dtype=np.int64)
    flat = [f.tolist() for t in got for f in t]
    assert flat == list_frames_of(chunk_size, hop, [[0] * (chunk_size - hop)] + ticks)


@pytest.mark.parametrize("seed", range(40))
def test_sum_frames_s_matches_list_model(orc, seed):
    # prop "sumFramesS": take (length l) . listSumFrames blockSize hopSize on the concatenated frames
    rng = np.random.default_rng(3000 + seed)
    hop = int(rng.integers(1, 8))
    chunk_size = int(rng.integers(1, 8)) * hop   # frame length
    block_m = int(rng.integers(1, 8))
    block = block_m * hop                         # output chunk length
    nticks = int(rng.integers(0, 8))
    ticks = [rng.integers(-50, 50, (block_m, chunk_size)).astype(np.int64) for _ in range(nticks)]
    got = orc.sum_frames_s(hop, ticks, [block] * nticks, dtype=np.int64)
    frames = [f.tolist() for t in ticks for f in t]
    want = list_sum_frames(block, hop, frames)[:nticks]
    assert [g.tolist() for g in got] == want


def test_float_variants_agree_with_int(orc):
    rng = np.random.default_rng(5)
    inp = [rng.integers(-9, 9, 17).tolist() for _ in range(5)]
    a, la = orc.gen_frames_of_e(8, 3, inp, dtype=np.int64)
    b, lb = orc.gen_frames_of_e(8, 3, inp, dtype=np.float64)
    assert np.array_equal(a.astype(float), b) and np.array_equal(la.astype(float), lb)
    assert np.array_equal(orc.sum_frames_e(5, 3, inp, dtype=np.int64).astype(float), orc.sum_frames_e(5, 3, inp))
