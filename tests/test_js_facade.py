"""The noble-compatible JS facade: the addon builds and loads on CPU; on a GPU box the facade is checked against the golden vectors."""
import os
import shutil
import subprocess
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JS = os.path.join(ROOT, 'noble-bls12-381_amd', 'js')
needs_node = pytest.mark.skipif(shutil.which('node') is None or not os.path.exists('/usr/include/node/node_api.h'), reason='node / N-API headers not available')


def _build():
    subprocess.check_call(['gcc', '-O2', '-shared', '-fPIC', '-D_GNU_SOURCE', '-I/usr/include/node', '-I' + os.path.join(ROOT, 'include'),
                           os.path.join(JS, 'nbls_napi.c'), '-o', os.path.join(JS, 'nbls_napi.node'), '-ldl'])


@needs_node
def test_addon_builds_and_exports():
    subprocess.check_call(['make', '-s', '-C', os.path.join(ROOT, 'noble-bls12-381_amd', 'csrc'), '../libnbls.so'])
    _build()
    out = subprocess.check_output(['node', '-e', "const b=require('%s'); console.log(Object.keys(b).sort().join(','))" % os.path.join(JS, 'index.js')]).decode()
    for name in ('pairing', 'verify', 'verifyBatch', 'aggregatePublicKeys', 'aggregateSignatures', 'sign', 'getPublicKey', 'PointG1', 'PointG2', 'utils', 'Fp12', 'CURVE'):
        assert name in out.split(',') or name in out


@needs_node
def test_host_side_of_the_facade():
    """field classes, utils and coordinate constructors: pure host code, no GPU call (values pinned on the reference)"""
    subprocess.check_call(['make', '-s', '-C', os.path.join(ROOT, 'noble-bls12-381_amd', 'csrc'), '../libnbls.so'])
    _build()
    out = subprocess.run(['node', os.path.join(ROOT, 'tests', 'js', 'test_host.js')], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and 'JS host ok' in out.stdout, out.stdout + out.stderr


@needs_node
@pytest.mark.gpu
def test_facade_on_gpu():
    _build()
    out = subprocess.run(['node', os.path.join(ROOT, 'tests', 'js', 'test_facade.js')], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and 'JS facade ok' in out.stdout, out.stdout + out.stderr


@needs_node
@pytest.mark.gpu
def test_facade_on_gpu_with_a_context_pool():
    """the same file with NBLS_CONTEXTS=3: asynchronous verifyBatch calls take engine contexts round-robin (js/nbls_napi.c), so concurrent promises overlap on the GPU"""
    _build()
    out = subprocess.run(['node', os.path.join(ROOT, 'tests', 'js', 'test_facade.js')], capture_output=True, text=True, timeout=300, env=dict(os.environ, NBLS_CONTEXTS='3'))
    assert out.returncode == 0 and 'JS facade ok' in out.stdout, out.stdout + out.stderr


@needs_node
@pytest.mark.gpu
def test_facade_on_gpu_with_a_multi_handle():
    """the same file with NBLS_DEVICES=0,0: pairingBatch / millerProduct / verifyBatch(Async) go through nbls_multi_* on ONE handle (two contexts on device 0);
    the Promise.all at the end races six verifyBatch calls -- valid and forged -- on that handle from libuv worker threads (round-2 finding: shared partial /
    gather buffers; every call now owns its buffers, csrc/nbls_multi.cpp)"""
    _build()
    out = subprocess.run(['node', os.path.join(ROOT, 'tests', 'js', 'test_facade.js')], capture_output=True, text=True, timeout=600, env=dict(os.environ, NBLS_DEVICES='0,0'))
    assert out.returncode == 0 and 'JS facade ok' in out.stdout, out.stdout + out.stderr
