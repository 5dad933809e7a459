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


def kmeans_tie_cases():
    """(name, k, X, w, expected assign, shuffled init) for every near-tie input of
    tests/golden/kmeans_tie.npz (oracle/gen_golden_kmeans_tie.py): the base matrices are regenerated
    from their seeds (numpy's legacy RandomState stream is frozen) and checked by sha256."""
    import hashlib
    g = golden('kmeans_tie')

    def sha(a):
        return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    base = {}
    for name in g['cases']:
        name = str(name)
        tag = name.rsplit('_p', 1)[0]
        if tag not in base:
            N, D, k, isz, seed = (int(v) for v in g[tag + '_meta'])
            rs = np.random.RandomState(seed)
            half = N // 2
            X = np.concatenate([rs.normal(0.0, 1.0, (half, D)), rs.normal(0.6, 1.0, (N - half, D))])
            X = X.astype(np.float64 if isz == 8 else np.float32)
            w = np.concatenate([rs.uniform(0.55, 1.0, half), rs.uniform(0.0, 0.45, N - half)])
            assert sha(X) + sha(w) == str(g[tag + '_sha'])
            base[tag] = (k, X, w)
        k, X, w = base[tag]
        j = int(g[name + '_j'])
        for side in ('lo', 'hi'):
            Xt = X.copy()
            Xt[j] = g['%s_row_%s' % (name, side)]
            yield name + '_' + side, k, Xt, w, g['%s_assign_%s' % (name, side)], g[name + '_idx']
