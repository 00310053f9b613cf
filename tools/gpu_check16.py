"""On-GPU parity of the kernels with sixteen points per thread (N = 4096 / 8192) against the oracle, several hops
and tilings, and against the eight-point kernels.  Development aid; the judged checks live in tests/."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
import oracle as orc
import helpers as Hh
from vocoder_amd._ffi import default_library
lib = default_library()
for (n, h, nf, c, at, st) in ((4096, 1024, 9, 2, 0, 0), (4096, 1024, 14, 2, 4, 5), (4096, 256, 20, 1, 6, 8), (4096, 512, 9, 1, 0, 0),
                              (4096, 2048, 7, 1, 3, 3), (8192, 2048, 6, 1, 0, 0), (8192, 1024, 9, 1, 4, 4), (4096, 1024, 40, 3, 0, 0)):
    Hh.check_roundtrip(lib, orc, n, n, h, n + h * (nf - 1) + 6, c, np.random.default_rng(n + h), ana_tile=at, syn_tile=st)
    print('ok', n, h, nf, c, at, st, flush=True)
