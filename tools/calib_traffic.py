"""Known-byte-count launch for calibrating rocprofv3's WRITE_SIZE / FETCH_SIZE on 8-byte-per-lane
accesses: the framing kernel writes exactly C*F*W*8 bytes and reads every input sample (its W/H-fold
re-reads hit L2).  Development aid for profiles/."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vocoder_amd import vocoder as V, conduit as Cd, window as Wn
par = V.VocoderParams(4096, 1024, Wn.hann_window(4096))
x = np.random.default_rng(0).standard_normal((16, 2_880_000))
frames, left = Cd.frames_of_e(par, x)
print('frames bytes written', frames.nbytes, 'input bytes', x.nbytes)
