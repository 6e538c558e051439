import os
import sys
import pytest
# torch bundles its own HIP runtime: when a test uses both torch.cuda and libnbls.so in one process, torch has to be loaded first
# (libnbls.so then binds to the runtime that is already there); loaded second, torch reports "No HIP GPUs are available"
import torch  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden():
    import goldenio
    return goldenio.load('ref_vectors.json.gz')


@pytest.fixture(scope='session')
def golden2():
    import goldenio
    return goldenio.load('ref_vectors2.json.gz')


@pytest.fixture(scope='session')
def golden3():
    import goldenio
    return goldenio.load('ref_vectors3.json.gz')


@pytest.fixture(scope='session')
def testdata():
    import goldenio
    return goldenio.load('ref_testdata.json.gz')


@pytest.fixture(scope='session')
def oracle():
    import oracle_py
    return oracle_py.load()
