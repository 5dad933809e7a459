#!/opt/conda/bin/python3.9
"""ORACLE — TEST INFRASTRUCTURE ONLY.  Golden vectors for mean-mode pooling.

    PYTHONDONTWRITEBYTECODE=1 /opt/conda/bin/python3.9 oracle/gen_golden_meanpool.py

The definition of mean mode is notebooks/Superpixel_Align.ipynb cell 4 of the reference:

    resized_features = F.resize_images(feature_map, (h, w)).array        # chainer, bilinear
    resized_features = cuda.to_cpu(resized_features)[0].transpose(1, 2, 0)
    for idx in np.sort(np.unique(superpixels)):
        superpixel_feature = np.mean(resized_features[superpixels == idx], axis=0)

This script restates that cell with NumPy only (chainer is not installed in this image):
`resize_images` below follows chainer v4's ResizeImages.forward [3p, from its published
source: float64 linspace(0, n-1, out) sampling grid — corners aligned —, floor/clip of the
upper-left tap, four float64 weights cast to the input dtype, the four products summed left to
right in float32], and the per-superpixel loop is the notebook's own `np.mean` over boolean
masks, over ALL superpixels.  `nearest` is the cheapest variant SURVEY.md 8a-5 names: the value
of feature pixel (y*fh//H, x*fw//W) under every image pixel, averaged by the same np.mean.
Written: tests/golden/meanpool_*.npz — inputs (labels, feature map), expected (S, C) float32
means of both variants and the pixel counts.  Only data is written.
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
sys.path.insert(0, os.path.join(ROOT, 'superpixel-align_amd'))

import oracle as orc  # noqa: E402
import synth  # noqa: E402


def resize_images(x, out_hw):
    """chainer.functions.resize_images forward (v4) on a (B,C,H,W) array."""
    out_H, out_W = out_hw
    B, C, H, W = x.shape
    u_1d = np.linspace(0, W - 1, num=out_W)
    v_1d = np.linspace(0, H - 1, num=out_H)
    grid = np.meshgrid(u_1d, v_1d)
    u = grid[0].ravel()
    v = grid[1].ravel()
    u0 = np.floor(u).astype(np.int32).clip(0, W - 2)
    u1 = u0 + 1
    v0 = np.floor(v).astype(np.int32).clip(0, H - 2)
    v1 = v0 + 1
    w1 = ((u1 - u) * (v1 - v)).astype(x.dtype)
    w2 = ((u - u0) * (v1 - v)).astype(x.dtype)
    w3 = ((u1 - u) * (v - v0)).astype(x.dtype)
    w4 = ((u - u0) * (v - v0)).astype(x.dtype)
    y = (w1[None, None, :] * x[:, :, v0, u0] + w2[None, None, :] * x[:, :, v0, u1] +
         w3[None, None, :] * x[:, :, v1, u0] + w4[None, None, :] * x[:, :, v1, u1])
    return y.reshape(B, C, out_H, out_W)


def resize_nearest(x, out_hw):
    out_H, out_W = out_hw
    B, C, H, W = x.shape
    yy = (np.arange(out_H) * H) // out_H
    xx = (np.arange(out_W) * W) // out_W
    return x[:, :, yy[:, None], xx[None, :]]


def cell4(resized_chw, superpixels):
    """The notebook's loop, verbatim semantics."""
    resized_features = resized_chw.transpose(1, 2, 0)
    out = []
    for idx in np.sort(np.unique(superpixels)):
        out.append(np.mean(resized_features[superpixels == idx], axis=0))
    return np.asarray(out)


def main():
    os.makedirs(GOLD, exist_ok=True)
    # (tag, image seed, H, W, n_segments, C, fh, fw)
    cases = [('a_64x128_c24', 0, 64, 128, 20, 24, 8, 16),
             ('b_128x256_c40', 1, 128, 256, 100, 40, 16, 32),
             ('c_100x37_c8', 5, 100, 37, 12, 8, 13, 5),        # ragged: H, W not multiples of fh, fw
             ('d_256x512_c16', 2, 256, 512, 100, 16, 32, 64)]
    for tag, seed, H, W, n, C, fh, fw in cases:
        img = synth.synth_image(seed, H, W)
        sp = orc.slic(img, n)                                  # any label map does; this one is realistic
        fmap = synth.synth_feature_map(seed + 11, C, fh, fw, batch=1).astype(np.float32)
        assert fmap.shape == (1, C, fh, fw)
        bil = cell4(resize_images(fmap, (H, W))[0], sp)
        nea = cell4(resize_nearest(fmap, (H, W))[0], sp)
        counts = np.bincount(sp.ravel()).astype(np.int64)
        assert bil.dtype == np.float32 and nea.dtype == np.float32 and bil.shape == (counts.size, C)
        path = os.path.join(GOLD, 'meanpool_%s.npz' % tag)
        np.savez_compressed(path, meta=np.array([seed, H, W, n, C, fh, fw], np.int64),
                            labels=sp.astype(np.int16), fmap=fmap[0], mean_bilinear=bil,
                            mean_nearest=nea, counts=counts)
        print('%-28s %8.1f KB  S=%d' % (os.path.basename(path), os.path.getsize(path) / 1024.0, counts.size))


if __name__ == '__main__':
    main()
