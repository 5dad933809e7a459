"""CPU suite, part 2: the C-ABI library loads, exports what include/spalign.h declares, its
host-side logic (SLIC plan, RNG emulation) matches the golden vectors, and it fails loudly
without a GPU.  No device compute is called here."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT, golden


@pytest.fixture(scope='module')
def L(spa):
    return spa._lib.lib()


def test_header_and_library_agree(spa, L):
    hdr = open(os.path.join(ROOT, 'include', 'spalign.h')).read()
    declared = set(re.findall(r'\b(spa_[a-z0-9_]+)\s*\(', hdr))
    declared -= {'spa_ctx', 'spa_pyrandom', 'spa_nprandom', 'spa_slic_plan', 'spa_fmap_desc'}
    assert declared == set(spa._lib.PROTOTYPES), declared ^ set(spa._lib.PROTOTYPES)
    for name in declared:
        assert hasattr(L, name), name
    assert L.spa_version() >= 100


def test_no_gpu_is_a_loud_error(spa, L):
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    h = ctypes.c_void_p()
    rc = L.spa_ctx_create(0, ctypes.byref(h))
    assert rc == -3 and b'no CPU fallback' in L.spa_last_error()
    engine = __import__('importlib').import_module('superpixel-align_amd.engine')
    with pytest.raises(spa.SpalignError):
        engine.Engine()


def test_slic_plan_matches_skimage(spa, orc):
    for H, W, n, sz, sy, sx, tz, ty, tx in golden('regular_grid')['cases']:
        p = spa._lib.make_plan(int(H), int(W), int(n))
        assert (p.start_y, p.start_x) == (sy, sx)
        assert (p.step_y, p.step_x) == (max(ty, 1), max(tx, 1))
        ny = len(range(int(sy), int(H), max(int(ty), 1))); nx = len(range(int(sx), int(W), max(int(tx), 1)))
        assert p.n_centroids == ny * nx
        assert (p.min_size, p.max_size) == orc.connectivity_sizes(int(H), int(W), ny * nx)
        st2, sp2 = orc.regular_grid(int(H), int(W), ny * nx)
        assert (p.win_step_y, p.win_step_x) == (sp2[1], sp2[2])
    for name in ('slic_s0_64x128_n20', 'slic_s0_1024x2048_n200'):
        seed, H, W, n, nC, mn, mx = (int(v) for v in golden(name)['meta'])
        p = spa._lib.make_plan(H, W, n)
        assert (p.n_centroids, p.min_size, p.max_size) == (nC, mn, mx)


def test_host_rng_streams(spa, orc):
    engine = __import__('importlib').import_module('superpixel-align_amd.engine')
    g = golden('rng')
    for n in (5, 1000, 70000):
        r = engine.PyRandom(1111)
        ranks, nv = r.shuffle_select(np.array([n, n // 2 + 1], np.int32), 32)
        m = min(n, 32)
        assert nv[0] == m and np.array_equal(ranks[0, :m], g['py_%d' % n][:m])
        m2 = min(n // 2 + 1, 32)
        assert np.array_equal(ranks[1, :m2], g['py_%d_second' % n][:m2])
        q = engine.NpRandom(1111)
        a = q.shuffle(np.arange(n, dtype=np.int64))
        assert np.array_equal(a[:32], g['np_%d' % n])
    # against the reference's recorded anchors: ranks -> pixels through the raster order
    gp = golden('pipeline_small')
    sps = gp['superpixels'].astype(np.int64)
    r = engine.PyRandom(1111)
    off = 0
    for b in range(sps.shape[0]):
        S = int(gp['n_per'][b])
        counts = np.bincount(sps[b].ravel(), minlength=S).astype(np.int32)
        ranks, nv = r.shuffle_select(counts, 10)
        order = np.argsort(sps[b].ravel(), kind='stable')
        starts = np.concatenate([[0], np.cumsum(counts)])
        W = sps.shape[2]
        for s in range(S):
            pix = order[starts[s] + ranks[s, :nv[s]]]
            exp = gp['anchors'][off + s, :nv[s]]
            assert np.array_equal(np.stack([pix // W, pix % W], 1), exp)
        off += S


def test_host_rng_vector_and_scalar_forms_agree():
    """spa_pyrandom_shuffle_select_host has an AVX-512 form (16 generator outputs per step of the rejection sampling; the
    anchors' places traced backwards through the swaps with vector compares) chosen at run time and a scalar form
    (SPA_RNG_SCALAR=1): the same ranks on sizes around every boundary of the vector code — lists of 0..40 elements, the 16-lane
    block, the bit-length bands (2^k - 1, 2^k, 2^k + 1), a 200 000-element list — with the stream carried across lists."""
    import hashlib
    import subprocess
    import sys
    code = r'''
import importlib, hashlib, numpy as np
engine = importlib.import_module('superpixel-align_amd.engine')
sizes = list(range(0, 41)) + [63, 64, 65, 127, 128, 129, 255, 256, 257, 1023, 1024, 1025, 4095, 4096, 4097, 10007, 32767, 32768, 32769, 200000]
r = engine.PyRandom(1111)
h = hashlib.sha256()
for rep in range(2):
    for A in (10, 16, 3):
        ranks, nv = r.shuffle_select(np.array(sizes, np.int32), A)
        assert all(nv[i] == min(max(s, 0), A) for i, s in enumerate(sizes))
        for i, s in enumerate(sizes):
            assert len(set(ranks[i, :nv[i]].tolist())) == nv[i] and (nv[i] == 0 or ranks[i, :nv[i]].max() < s)
        h.update(ranks.tobytes()); h.update(nv.tobytes())
print(h.hexdigest())
'''
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for scalar in ('', '1'):
        env = dict(os.environ, PYTHONPATH=root)
        if scalar:
            env['SPA_RNG_SCALAR'] = '1'
        else:
            env.pop('SPA_RNG_SCALAR', None)
        outs.append(subprocess.check_output([sys.executable, '-c', code], env=env, cwd=root).decode().strip())
    assert outs[0] == outs[1] and len(outs[0]) == 64


def test_numpy_generator_state_export_matches_numpy():
    """spa_nprandom_state (what the device-side k-means initialisation uploads, csrc/spa_nprng.hip): the 624 state words and the
    position of numpy's own legacy RandomState, right after seeding and after shuffles of several lengths (block boundaries)."""
    import importlib
    import numpy as np
    engine = importlib.import_module('superpixel-align_amd.engine')
    for seed in (1111, 5):
        host, rs = engine.NpRandom(seed), np.random.RandomState(seed)
        for n in (0, 1, 7, 1000, 623, 4097):
            st = host.state()
            _, key, pos = rs.get_state()[:3]
            assert np.array_equal(st[:624], key.astype(np.uint32)) and int(st[624]) == int(pos) and not st[625:].any()
            a, b = np.arange(n, dtype=np.int64), np.arange(n, dtype=np.int64)
            host.shuffle(a)
            rs.shuffle(b)
            assert np.array_equal(a, b)


def test_numpy_generator_state_import_continues_the_stream():
    """spa_nprandom_set_state (round 6: a batch too large for the device-side initialisation draws on the host from the state it
    downloads, LabelPipeline.cluster): a generator set to numpy's state — taken right after seeding, in the middle of a block and
    at a block boundary — continues numpy's stream."""
    import importlib
    import numpy as np
    engine = importlib.import_module('superpixel-align_amd.engine')
    rs = np.random.RandomState(1111)
    other = engine.NpRandom(7)
    for n in (0, 5, 1000, 623, 4097, 1):
        _, key, pos = rs.get_state()[:3]
        st = np.zeros(628, np.uint32)
        st[:624] = key
        st[624] = pos
        other.set_state(st)
        assert np.array_equal(other.state(), st)
        a, b = np.arange(n, dtype=np.int64), np.arange(n, dtype=np.int64)
        other.shuffle(a)
        rs.shuffle(b)
        assert np.array_equal(a, b)
        # ... and the two stay together without another hand-over
        a, b = np.arange(77, dtype=np.int64), np.arange(77, dtype=np.int64)
        other.shuffle(a)
        rs.shuffle(b)
        assert np.array_equal(a, b)
