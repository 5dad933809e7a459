#!/opt/conda/bin/python3.9
"""ORACLE — TEST INFRASTRUCTURE ONLY.  Golden vectors for the input stage's bicubic resize.

    PYTHONDONTWRITEBYTECODE=1 /opt/conda/bin/python3.9 oracle/gen_golden_resize.py

Outputs of Pillow itself (8.4.0 in the conda environment): PIL.Image.fromarray(channel).resize((w, h),
PIL.Image.BICUBIC) per channel of seeded uint8 images — the computation behind
chainercv.transforms.resize(image, shape, 3) (datasets/resize_image_dataset.py:31-34) when Pillow does the
work on an 8-bit image.  Covers down-scaling by non-integer factors (the 1024x2048 -> 224x224 operating point
scaled down), up-scaling, one unchanged axis and odd sizes.  Only data is written."""
import os
import sys
import numpy as np
import PIL
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(os.path.dirname(HERE), 'tests', 'golden')
rs = np.random.RandomState(21)
out = {}
cases = []
for tag, (H, W, h, w) in {'down_128x256_to_28x28': (128, 256, 28, 28), 'down_100x37_to_33x20': (100, 37, 33, 20),
                          'up_24x40_to_50x90': (24, 40, 50, 90), 'same_h_64x96_to_64x30': (64, 96, 64, 30),
                          'mixed_61x83_to_97x41': (61, 83, 97, 41)}.items():
    # smooth structure + noise + saturated patches (the clip8 path)
    yy, xx = np.mgrid[0:H, 0:W]
    img = np.stack([127 + 100 * np.sin(0.11 * xx + 0.07 * yy * (c + 1)) + 40 * rs.standard_normal((H, W)) for c in range(3)])
    img[:, : H // 5, : W // 4] = 255
    img[:, -H // 6:, -W // 3:] = 0
    img = np.clip(img, 0, 255).astype(np.uint8)
    res = np.stack([np.asarray(Image.fromarray(c).resize((w, h), Image.BICUBIC)) for c in img])
    out[tag + '_img'] = img
    out[tag + '_out'] = res
    cases.append(tag)
out['cases'] = np.array(cases)
out['pillow'] = np.array(PIL.__version__)
path = os.path.join(GOLD, 'resize_bicubic.npz')
np.savez_compressed(path, **out)
print(path, '%.1f KB' % (os.path.getsize(path) / 1024.0), PIL.__version__)
