"""Pure-Python restatement of the LIST MODELS the reference's own tests check framing and
overlap-add against.  These are the only machine-checkable contract the reference ships, so
they pin the oracle's (and through it the library's) edge behaviour.

  list_frames_of_e / list_sum_frames_e : vocoder-conduit/test/main.hs:18-28
  list_frames_of   / list_sum_frames   : vocoder-dunai/test/main.hs:28-38
"""
from itertools import chain


def list_frames_of_e(chunk_size, hop, inp):
    """map (\\i -> take chunkSize $ drop i cInput) [0, hopSize .. length cInput - chunkSize]"""
    c = list(chain.from_iterable(inp))
    return [c[i:i + chunk_size] for i in range(0, len(c) - chunk_size + 1, hop)]


def list_sum_frames_e(chunk_size, hop, inp):
    """cOutput = foldl1 (zipWith (+)) [replicate k 0 ++ l ++ repeat 0 | k <- [0, hop ..], l <- input]
    chunks at [0, chunkSize .. lastLength], lastLength = maximum (-1 : [k + length l - 1])"""
    last = max([-1] + [k * hop + len(l) - 1 for k, l in enumerate(inp)])
    if last < 0:
        return []
    n = (last // chunk_size + 1) * chunk_size
    out = [0] * n
    for k, l in enumerate(inp):
        for i, v in enumerate(l):
            out[k * hop + i] = out[k * hop + i] + v
    return [out[i:i + chunk_size] for i in range(0, last + 1, chunk_size)]


def list_frames_of(chunk_size, hop, inp):  # dunai: identical definition (test/main.hs:28-32)
    return list_frames_of_e(chunk_size, hop, inp)


def list_sum_frames(chunk_size, hop, inp):  # dunai: identical definition (test/main.hs:34-38)
    return list_sum_frames_e(chunk_size, hop, inp)
