#!/opt/conda/bin/python3.9
"""ORACLE — TEST INFRASTRUCTURE ONLY.  Golden vectors for scikit-image's float64 SLIC on uint8 images.

    PYTHONDONTWRITEBYTECODE=1 /opt/conda/bin/python3.9 oracle/gen_golden_slic64.py

superpixel_overlaps.py:301-304 calls `slic(img.transpose(1, 2, 0), n_segments)` on the ORIGINAL uint8 image:
`img_as_float` makes it float64 and rgb2lab and the compiled `_slic_cython[double]` run in binary64.  Per
fixture (seeded synthetic uint8 image):
  lab_skimage   rgb2lab(img_as_float(img)) * 0.1 from scikit-image itself (float64)   -> tolerance check of
                the restatement's deterministic Lab (numpy's float64 power / cbrt are not reproducible)
  pre, centres  `_slic_cython` (0.18.3, double) fed with the RESTATEMENT's Lab image  -> bit-exact check of
                the float64 core
  post          `_enforce_label_connectivity_cython` of `pre`
  e2e           the untouched `slic(uint8 image, n)` call                              -> mismatch count
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


def synth_u8(seed, H, W):
    """smooth coloured regions + texture + noise, uint8 CHW"""
    rs = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:H, 0:W]
    img = np.zeros((3, H, W))
    for c in range(3):
        low = rs.uniform(0, 255, (H // 24 + 2, W // 24 + 2))
        img[c] = low.repeat(24, 0).repeat(24, 1)[:H, :W]
        img[c] += 25 * np.sin(xx / rs.uniform(5, 17) + yy / rs.uniform(6, 19))
    img += rs.normal(0, 7, img.shape)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def main():
    from skimage.segmentation._slic import _slic_cython, _enforce_label_connectivity_cython
    from skimage.segmentation.slic_superpixels import _get_grid_centroids
    from skimage.segmentation import slic
    from skimage.color import rgb2lab
    from skimage import img_as_float
    import skimage

    for (seed, H, W, n) in [(0, 64, 96, 20), (1, 120, 200, 60), (2, 96, 96, 30), (3, 200, 320, 100), (4, 37, 100, 7)]:
        img = synth_u8(seed, H, W)
        hwc = np.ascontiguousarray(img.transpose(1, 2, 0))
        lab_sk = np.ascontiguousarray(rgb2lab(img_as_float(hwc)) * 0.1, dtype=np.float64)
        lab = orc.rgb2lab_u8_f64(img)                                   # the restatement's deterministic Lab
        image = np.ascontiguousarray(lab[None], dtype=np.float64)       # (1,H,W,3)
        cent, steps = _get_grid_centroids(image, n)
        nC = cent.shape[0]
        segs = np.ascontiguousarray(np.concatenate([cent, np.zeros((nC, 3))], axis=-1), dtype=np.float64)
        pre = _slic_cython(image, None, segs, max(steps), 10, np.ones(3, np.float64), False,
                           ignore_color=False, start_label=0)
        mn, mx = orc.connectivity_sizes(H, W, nC)
        post = _enforce_label_connectivity_cython(pre, mn, mx, start_label=0)
        e2e = slic(hwc, n)
        np.savez_compressed(os.path.join(GOLD, 'slic64_s%d_%dx%d_n%d.npz' % (seed, H, W, n)),
                            meta=np.array([seed, H, W, n, nC, mn, mx], np.int64), img=img,
                            lab_skimage=lab_sk.astype(np.float64), pre=pre[0].astype(np.int16),
                            post=post[0].astype(np.int16), centres=segs, e2e_skimage=e2e.astype(np.int16))
        print('slic64 s%d %dx%d n=%d: %d seeds, %d labels; max |lab - skimage| = %.3g; e2e mismatch vs restatement %d px'
              % (seed, H, W, n, nC, int(post.max()) + 1, np.abs(lab - lab_sk).max(),
                 int((orc.slic_u8(img, n) != e2e).sum())))
    with open(os.path.join(GOLD, 'PROVENANCE.txt'), 'a') as f:
        f.write('slic64_*.npz: oracle/gen_golden_slic64.py, scikit-image %s, numpy %s\n' % (skimage.__version__, np.__version__))


if __name__ == '__main__':
    main()
