"""noble-bls12-381_amd -- MI355X-native batched BLS12-381 pairing engine.

Host-side Python binding (ctypes) of the C ABI in include/nbls.h.  The arithmetic runs only in the HIP library
(libnbls.so, built from csrc/); there is no CPU fallback: importing works anywhere, but creating an Engine
raises if the library or a GPU is missing.

The package directory name contains '-', so load it with importlib.import_module('noble-bls12-381_amd').
"""
from .engine import Engine, MultiEngine, NblsError, lib_path, load_library, PROGRAMS  # noqa: F401
from .pipeline import PairingPipeline  # noqa: F401
