#!/opt/conda/bin/python3.9
"""ORACLE — TEST INFRASTRUCTURE ONLY.  Golden vectors for the "KMeans failed -> try again" branch.

    PYTHONDONTWRITEBYTECODE=1 /opt/conda/bin/python3.9 oracle/gen_golden_retry.py

weighted_kmeans (batch_spalign_kmeans.py:186-207) re-runs ITSELF, result discarded, whenever an image ends up
without a cluster-0 pixel (:201-205).  The discarded run calls kmeans() again, which shuffles the initial
assignment with numpy's global generator (:147-149) — so for k > 2 the retries move the random stream every later
batch draws from, and a retry can itself fail and recurse.  The fixture records two consecutive batches of the
reference's own weighted_kmeans (numpy path, seed 1111, k = 4) on engineered descriptors where the first batch
needs several (nested) retries: cluster maps of both batches and every shuffled index vector in call order.
For k = 2 nothing is random, the retry repeats the failure and the reference dies with RecursionError: recorded
as a flag.
"""
import os
import sys
import warnings

sys.dont_write_bytecode = True
warnings.filterwarnings('ignore')
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import refconfig  # noqa: E402,F401  (the reference configuration: before numpy)
import numpy as np  # noqa: E402

import gen_golden as gg  # noqa: E402


def make_case(rs, B, segs_per_image, D, blobs_of_image):
    H, W = 16, 24
    sps, rows, wts, n_per = [], [], [], []
    centres = rs.normal(0, 6, (4, D))
    for b in range(B):
        n = segs_per_image[b]
        lab = (np.arange(H * W) * n // (H * W)).reshape(H, W)       # n horizontal bands, dense labels
        sps.append(lab)
        n_per.append(n)
        for s in range(n):
            blob = blobs_of_image[b][s % len(blobs_of_image[b])]
            rows.append(centres[blob] + rs.normal(0, 0.7, D))
            wts.append(rs.uniform(0.55, 1.0) if blob == 0 else rs.uniform(0.0, 0.45))
    return np.stack(sps).astype(np.int64), np.stack(rows), np.array(wts), n_per


def run(ref, batches, k):
    np.random.seed(1111)
    gg.XP.shuffled.clear()
    out = []
    for sps, X, w, n_per in batches:
        cl, road = ref.weighted_kmeans(sps, X, w, k, n_per)
        out.append(np.asarray(cl))
    return out, [s.copy() for s in gg.XP.shuffled]


def main():
    ref = gg.import_reference()
    sys.setrecursionlimit(300)
    found = None
    for seed in range(400):
        rs = np.random.RandomState(seed)
        # image 1 of the first batch holds only blobs that rarely end in cluster 0
        b0 = make_case(rs, 2, [10, 5], 6, [[0, 1, 2, 3], [3, 2]])
        b1 = make_case(rs, 2, [8, 8], 6, [[0, 1, 2, 3], [0, 1, 2, 3]])
        try:
            import io
            import contextlib
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                cls, shuf = run(ref, [b0, b1], 4)
        except RecursionError:
            continue
        n_retry = buf.getvalue().count('Somehow KMeans')
        # wanted: several retries in batch 0 (3 or more kmeans calls), the second batch clean
        if n_retry >= 3 and len(shuf) == n_retry + 2:
            found = (seed, b0, b1, cls, shuf, n_retry)
            break
    assert found, 'no engineered case found'
    seed, b0, b1, cls, shuf, n_retry = found
    # k = 2, second image made of the low-weight blob only: deterministic -> the retry repeats for ever
    k2_dies = False
    c2 = make_case(np.random.RandomState(77), 2, [6, 3], 6, [[0, 1], [1]])
    try:
        import io
        import contextlib
        with contextlib.redirect_stdout(io.StringIO()):
            np.random.seed(1111)
            ref.weighted_kmeans(c2[0], c2[1], c2[2], 2, c2[3])
    except RecursionError:
        k2_dies = True
    assert k2_dies
    m = max(len(s) for s in shuf)
    shuf_pad = np.full((len(shuf), m), -1, np.int64)
    for i, s in enumerate(shuf):
        shuf_pad[i, :len(s)] = s
    gg.save('kmeans_retry', seed=np.array(seed), n_retry=np.array(n_retry),
            sps0=b0[0].astype(np.int16), X0=b0[1], w0=b0[2], n_per0=np.array(b0[3], np.int64),
            sps1=b1[0].astype(np.int16), X1=b1[1], w1=b1[2], n_per1=np.array(b1[3], np.int64),
            cl0=cls[0].astype(np.uint8), cl1=cls[1].astype(np.uint8), shuffled=shuf_pad,
            shuffled_len=np.array([len(s) for s in shuf], np.int64), k2_recursion_error=np.array(k2_dies),
            sps2=c2[0].astype(np.int16), X2=c2[1], w2=c2[2], n_per2=np.array(c2[3], np.int64))
    print('seed', seed, 'retries in batch 0:', n_retry, 'kmeans calls:', len(shuf), 'k=2 RecursionError:', k2_dies)


if __name__ == '__main__':
    main()
