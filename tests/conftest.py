import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:  # noqa: BLE001
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason='no HIP device in this container')
    for it in items:
        if 'gpu' in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope='session')
def orc():
    from oracle import oracle
    oracle.lib()
    return oracle


@pytest.fixture(scope='session')
def g12():
    return dict(np.load(os.path.join(GOLDEN, 'g1_g2_jacobian_hessian.npz')))


@pytest.fixture(scope='session')
def g3():
    return dict(np.load(os.path.join(GOLDEN, 'g3_decode_chain.npz')))


@pytest.fixture(scope='session')
def g4():
    return dict(np.load(os.path.join(GOLDEN, 'g4_pose_head_prep.npz'), allow_pickle=True))


@pytest.fixture(scope='session')
def batch64():
    from monorun_amd import synthetic as syn
    return syn.make_batch(B=64, seed=1234)
