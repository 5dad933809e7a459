import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


@pytest.fixture(scope='session')
def orc():
    """The CPU oracle (test infrastructure): oracle/oracle.py over oracle/liborc.so."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import oracle
    oracle.lib()
    return oracle


@pytest.fixture(scope='session')
def spa():
    """The product package (directory name has a hyphen, so import it by string)."""
    return importlib.import_module('superpixel-align_amd')


@pytest.fixture(scope='session')
def synth(spa):
    return spa.synth


def golden(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)
