"""vocoder_amd -- MI355X (gfx950) engine for the STFT round-trip path of tilk/vocoder.

`vocoder` mirrors the reference's core module (vocoder/src/Vocoder.hs), `window` its window
tables, `conduit` the batch forms of its streaming wrappers.  All arithmetic runs in
``libvocoder_hip.so`` (hand-written HIP, C ABI in include/vocoder_hip.h); importing this
package without that library raises.
"""
from ._ffi import Library, VocoderError, default_library, DEFAULT_LIB  # noqa: F401
from . import vocoder, window, conduit  # noqa: F401
from .vocoder import *  # noqa: F401,F403
