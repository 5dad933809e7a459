#!/opt/conda/bin/python3.9
"""ORACLE — TEST INFRASTRUCTURE ONLY.  Near-tie golden vectors for the weighted k-means.

    PYTHONDONTWRITEBYTECODE=1 /opt/conda/bin/python3.9 oracle/gen_golden_kmeans_tie.py

The reference's own `kmeans` (batch_spalign_kmeans.py:136-183, imported as in gen_golden.py) is
used as a black box.  For a chosen point j the row X[j] = a + t (b - a) is moved along a segment
between a point the reference puts in cluster 0 and one it puts in another cluster; bisection on
the float64 parameter t down to ADJACENT doubles gives two inputs, X(t_lo) and X(t_hi), that
differ in the last bit of t and on which the reference assigns point j differently.  Both inputs
sit on the decision boundary to within the rounding noise of the distance computation, so an
implementation whose sums are rounded in a different order than numpy's (sequential axis-0 sums for
the centres, pairwise add.reduce for the squared distances) gets each of them right only by
chance.  Written: tests/golden/kmeans_tie.npz with, per case, the base matrix, the weights, the
row index, both rows and both assignment vectors (+ the shuffled init vector for k > 2).
"""
import os
import sys
import warnings

sys.dont_write_bytecode = True
warnings.filterwarnings('ignore')

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import refconfig  # noqa: E402  (the reference configuration: before numpy)
import numpy as np  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402


def run(ref, k, X, w, seed):
    np.random.seed(seed)
    gg.XP.shuffled.clear()
    a = np.asarray(ref.kmeans(k, X, w)).astype(np.int32)
    idx = gg.XP.shuffled[-1].astype(np.int64) if gg.XP.shuffled else np.zeros(0, np.int64)
    return a, idx


def bracket(ref, k, X, w, j, a_row, b_row, seed):
    """Bisect t in [0, 1] until adjacent doubles; returns (row_lo, row_hi, assign_lo, assign_hi, idx)."""
    def at(t):
        Xt = X.copy()
        Xt[j] = (a_row + t * (b_row - a_row)).astype(X.dtype)
        a, idx = run(ref, k, Xt, w, seed)
        return Xt[j].copy(), a, idx
    lo, hi = 0.0, 1.0
    r_lo, a_lo, idx = at(lo)
    r_hi, a_hi, _ = at(hi)
    assert a_lo[j] != a_hi[j], 'segment does not cross the boundary'
    while True:
        mid = lo + (hi - lo) / 2
        if mid == lo or mid == hi:
            break
        r_m, a_m, _ = at(mid)
        if a_m[j] == a_lo[j]:
            lo, r_lo, a_lo = mid, r_m, a_m
        else:
            hi, r_hi, a_hi = mid, r_m, a_m
    # rows may coincide after rounding to X's dtype (float32 case): walk hi up until they differ
    assert not np.array_equal(r_lo, r_hi) or X.dtype == np.float32
    return r_lo, r_hi, a_lo, a_hi, idx


def base_problem(N, D, dt, seed):
    """Seeded base matrix and weights (numpy's legacy RandomState stream is frozen across versions,
    so tests regenerate them instead of storing ~1 MB of incompressible normals; sha256 is stored)."""
    rs = np.random.RandomState(seed)
    half = N // 2
    X = np.concatenate([rs.normal(0.0, 1.0, (half, D)), rs.normal(0.6, 1.0, (N - half, D))]).astype(dt)
    w = np.concatenate([rs.uniform(0.55, 1.0, half), rs.uniform(0.0, 0.45, N - half)])
    return X, w


def main():
    ref = gg.import_reference()
    out = {}
    cases = []
    # (tag, N, D, k, dtype, rng seed)
    for tag, N, D, k, dt, seed in [('f64_d514_k2', 60, 514, 2, np.float64, 1),
                                    ('f64_d200_k2', 48, 200, 2, np.float64, 2),
                                    ('f64_d37_k3', 45, 37, 3, np.float64, 3),
                                    ('f32_d512_k2', 60, 512, 2, np.float32, 4),
                                    ('f64_d514_k4', 90, 514, 4, np.float64, 5)]:
        X, w = base_problem(N, D, dt, seed)
        out[tag + '_meta'] = np.array([N, D, k, 8 if dt == np.float64 else 4, seed], np.int64)
        out[tag + '_sha'] = np.array(gg.sha(X) + gg.sha(w))
        base, idx0 = run(ref, k, X, w, 1111)
        made = 0
        for j in range(N):
            if made == 4:
                break
            others = np.nonzero(base != base[j])[0]
            if others.size == 0:
                continue
            a_row = X[j].astype(np.float64)
            b_row = X[others[(3 * j) % others.size]].astype(np.float64)
            try:
                r_lo, r_hi, a_lo, a_hi, idx = bracket(ref, k, X, w, j, a_row, b_row, 1111)
            except AssertionError:
                continue
            if np.array_equal(r_lo, r_hi):
                continue
            name = '%s_p%d' % (tag, made)
            out[name + '_j'] = np.array(j, np.int64)
            out[name + '_row_lo'] = r_lo
            out[name + '_row_hi'] = r_hi
            out[name + '_assign_lo'] = a_lo
            out[name + '_assign_hi'] = a_hi
            out[name + '_idx'] = idx
            cases.append(name)
            nd = int((r_lo != r_hi).sum())
            print('%-22s j=%2d  rows differ in %d element(s), assign[j] %d | %d, other points differing: %d'
                  % (name, j, nd, a_lo[j], a_hi[j], int((a_lo != a_hi).sum()) - 1))
            made += 1
    out['cases'] = np.array(cases)
    path = os.path.join(gg.GOLD, 'kmeans_tie.npz')
    np.savez_compressed(path, **out)
    print('%s %.1f KB, %d cases' % (path, os.path.getsize(path) / 1024.0, len(cases)))


if __name__ == '__main__':
    main()
