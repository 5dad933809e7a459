#!/opt/conda/bin/python3.9
"""ORACLE — TEST INFRASTRUCTURE ONLY.  Golden vectors for SLIC seeds that lose all their pixels.

    PYTHONDONTWRITEBYTECODE=1 /opt/conda/bin/python3.9 oracle/gen_golden_starve.py

On blocky, noisy images some seeds of scikit-image's `_slic_cython` (0.18.3, float32) end a sweep
with no pixel.  The Cython core then divides 0/0: the centre becomes NaN, a NaN distance never wins
the `distance > dist_center` test, so the seed stays dead for the remaining sweeps and its label never
appears in the result (the NaN -> index casts that size its search window give an empty range on
x86-64).  The fixtures record exactly that behaviour: image, labels before / after the connectivity
pass and the final centres (NaN rows for the dead seeds), all from the compiled scikit-image cores
fed with the deterministic Lab image (as in gen_golden.py).
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
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, 'tests', 'golden')
sys.path.insert(0, HERE)

import oracle as orc  # noqa: E402


def blocky(seed, H, W):
    rs = np.random.RandomState(seed)
    img = np.zeros((3, H, W))
    for c in range(3):
        img[c] = rs.choice([0, 255], size=(H // 4 + 1, W // 4 + 1)).repeat(4, 0).repeat(4, 1)[:H, :W]
    img += rs.normal(0, 30, img.shape)
    return np.clip(img, 0, 255).astype(np.float32)


def main():
    from skimage.segmentation._slic import _slic_cython, _enforce_label_connectivity_cython
    from skimage.segmentation.slic_superpixels import _get_grid_centroids
    for seed, H, W, n in [(0, 24, 40, 30), (5, 32, 32, 60), (6, 40, 64, 60), (2, 40, 64, 30)]:
        img = blocky(seed, H, W)
        lab = orc.rgb2lab_scaled(img)
        image = np.ascontiguousarray(lab[None], dtype=np.float32)
        cent, steps = _get_grid_centroids(image, n)
        nC = cent.shape[0]
        segs = np.ascontiguousarray(np.concatenate([cent, np.zeros((nC, 3))], axis=-1), dtype=np.float32)
        pre = _slic_cython(image, None, segs, max(steps), 10, np.ones(3, np.float32), False,
                           ignore_color=False, start_label=0)
        dead = np.nonzero(np.isnan(segs).any(axis=1))[0]
        assert dead.size > 0
        mn, mx = orc.connectivity_sizes(H, W, nC)
        post = _enforce_label_connectivity_cython(pre, mn, mx, start_label=0)
        path = os.path.join(GOLD, 'slic_starve_s%d_%dx%d_n%d.npz' % (seed, H, W, n))
        np.savez_compressed(path, meta=np.array([seed, H, W, n, nC, mn, mx], np.int64), img=img,
                            pre=pre[0].astype(np.int16), post=post[0].astype(np.int16), centres=segs,
                            dead=dead.astype(np.int64))
        print('%s: %d seeds, dead %s, %.1f KB' % (os.path.basename(path), nC, dead.tolist(),
                                                 os.path.getsize(path) / 1024.0))


if __name__ == '__main__':
    main()
