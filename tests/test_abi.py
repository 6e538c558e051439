"""The drop-in boundary on a machine without a GPU: libnbls.so builds, loads and exports every function include/nbls.h
declares; the Python binding names every one of them; and the product path fails loudly (no CPU fallback) when no GPU is
present."""
import ctypes as C
import importlib
import os
import re
import subprocess
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, 'include', 'nbls.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    names = re.findall(r'^\s*(?:int|void|const char\*)\s+(nbls_\w+)\s*\(', src, flags=re.M)
    assert len(names) >= 25
    return names


def test_library_exports_every_declared_symbol():
    subprocess.check_call(['make', '-s', '-C', os.path.join(ROOT, 'noble-bls12-381_amd', 'csrc'), '../libnbls.so'])
    lib = C.CDLL(os.path.join(ROOT, 'noble-bls12-381_amd', 'libnbls.so'))
    missing = [n for n in _declared() if not hasattr(lib, n)]
    assert not missing, missing
    # and nothing of the test-only oracle / simulator leaks into the product library
    out = subprocess.check_output(['nm', '-D', '--defined-only', os.path.join(ROOT, 'noble-bls12-381_amd', 'libnbls.so')]).decode()
    assert 'oracle_' not in out and 'nbls_sim_' not in out


def test_no_cpu_fallback():
    torch = pytest.importorskip('torch')
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    pkg = importlib.import_module('noble-bls12-381_amd')
    with pytest.raises(pkg.NblsError):
        pkg.Engine(0)
    lib = pkg.load_library()
    h = C.c_void_p()
    assert lib.nbls_init(0, C.byref(h)) != 0 and not h.value
    # entry points reject a null context instead of computing anything
    out = C.create_string_buffer(576)
    assert lib.nbls_pairing_batch(None, 1, bytes(96), bytes(192), 1, 0, out, None) != 0
