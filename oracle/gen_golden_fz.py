#!/opt/conda/bin/python3.9
"""ORACLE — TEST INFRASTRUCTURE ONLY.  Golden vectors for the felzenszwalb branch.

    PYTHONDONTWRITEBYTECODE=1 /opt/conda/bin/python3.9 oracle/gen_golden_fz.py

Runs scikit-image 0.18.3's compiled core (_felzenszwalb_cython, which the reference reaches through
batch_superpixel, batch_spalign_kmeans.py:301-307) twice per case:
  (a) untouched, as the reference calls it — recorded for the agreement statistic;
  (b) with the module globals `ndi` and `np` proxied so that the smoothing step returns the
      oracle's own (deterministic) smoothed image and np.argsort is stable — this pins everything
      after the smoothing bit for bit (numpy's default sort leaves the order of equal costs
      unspecified and uses an AVX-512 sort on this host).
Also records scipy's Gaussian weights and smoothed image for the smoothing checks.
"""
import os
import sys
import types
import warnings

sys.dont_write_bytecode = True
warnings.filterwarnings('ignore')
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import refconfig  # noqa: E402  (the reference configuration: before numpy)
import numpy as np  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, 'superpixel-align_amd'))
import oracle as orc  # noqa: E402
import synth  # noqa: E402
import skimage.segmentation._felzenszwalb_cy as fzmod  # noqa: E402
from scipy import ndimage as real_ndi  # noqa: E402
from scipy.ndimage.filters import _gaussian_kernel1d  # noqa: E402


class NpStable(object):
    def argsort(self, a, *args, **kw):
        return np.argsort(a, kind='stable')

    def __getattr__(self, name):
        return getattr(np, name)


def run_pinned(img_hwc64, scale, sigma, min_size):
    class Ndi(object):
        def gaussian_filter(self, image, sigma):
            return orc.fz_blur(image, sigma=sigma[0])
    fzmod.ndi, fzmod.np = Ndi(), NpStable()
    try:
        return fzmod._felzenszwalb_cython(img_hwc64, scale=scale, sigma=sigma, min_size=min_size)
    finally:
        fzmod.ndi, fzmod.np = real_ndi, np


if __name__ == '__main__':
    gold = os.path.join(ROOT, 'tests', 'golden')
    for (seed, H, W, scale, sigma, min_size, integer) in [
            (0, 48, 64, 30.0, 0.8, 20, True), (1, 96, 128, 300.0, 0.8, 20, True),
            (2, 224, 224, 300.0, 0.8, 20, True), (3, 64, 96, 10.0, 0.5, 5, False),
            (4, 80, 120, 100.0, 1.2, 50, False), (5, 256, 512, 300.0, 0.8, 20, True),
            (6, 224, 224, 1.0, 0.8, 2, True)]:
        img = synth.synth_scene(seed, H, W, integer_valued=integer)          # CHW f32 0..255
        hwc = img.transpose(1, 2, 0) / 255.                                  # as the reference passes it
        from skimage.segmentation import felzenszwalb
        plain = felzenszwalb(hwc, scale=scale, sigma=sigma, min_size=min_size)
        pinned = run_pinned(np.atleast_3d(hwc), scale, sigma, min_size)
        extra = {}
        if H * W <= 64 * 96:
            extra['scipy_weights'] = _gaussian_kernel1d(sigma, 0, int(4.0 * sigma + 0.5))[::-1].copy()
            extra['scipy_blur'] = real_ndi.gaussian_filter(hwc.astype(np.float64), sigma=[sigma, sigma, 0])
        name = 'fz_s%d_%dx%d' % (seed, H, W)
        np.savez_compressed(os.path.join(gold, name + '.npz'),
                            meta=np.array([seed, H, W, min_size, int(integer)], np.int64),
                            params=np.array([scale, sigma]), pinned=pinned.astype(np.int32),
                            plain=plain.astype(np.int32), **extra)
        print(name, 'segments pinned/plain', pinned.max() + 1, plain.max() + 1,
              'agreement of partitions (same-label pairs along rows): %.4f'
              % np.mean((pinned[:, 1:] == pinned[:, :-1]) == (plain[:, 1:] == plain[:, :-1])),
              '%.0f KB' % (os.path.getsize(os.path.join(gold, name + '.npz')) / 1024.))
