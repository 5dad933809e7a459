#!/usr/bin/env python3
"""Input-stage measurement: PNG decode (+ optional host bicubic resize) of Cityscapes-sized
frames on worker threads, images/s per core and for the pool.   python tools/decode_bench.py [--threads N]

The frames are synthetic 1024x2048 RGB PNGs (low-frequency structure + sensor-like noise; ~2.3 MB each, close to
a leftImg8bit frame).  cli.ImageList decodes with PIL on `--loader_threads` threads and, by default, resizes on
the GPU (spa_resize_bicubic_u8); `--host_resize` is what the second row times."""
import argparse
import io
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
from PIL import Image

ap = argparse.ArgumentParser()
ap.add_argument('--threads', type=int, default=8)
ap.add_argument('--frames', type=int, default=16)
ap.add_argument('--seconds', type=float, default=6.0)
a = ap.parse_args()

rng = np.random.RandomState(0)
blobs = []
for i in range(a.frames):
    low = rng.rand(17, 33, 3)
    img = np.asarray(Image.fromarray((low * 255).astype(np.uint8)).resize((2048, 1024), Image.BICUBIC), dtype=np.float32)
    img += rng.randn(1024, 2048, 3).astype(np.float32) * 6.0
    buf = io.BytesIO()
    Image.fromarray(np.clip(img, 0, 255).astype(np.uint8)).save(buf, format='PNG')
    blobs.append(buf.getvalue())
print('frames: %d synthetic PNGs, %.2f MB each' % (len(blobs), np.mean([len(b) for b in blobs]) / 1e6))


def decode(b, resize):
    im = Image.open(io.BytesIO(b)).convert('RGB')
    if resize:
        im = im.resize((1024, 512), Image.BICUBIC)
    return np.asarray(im).shape


for resize in (False, True):
    for nthr in (1, a.threads):
        done = 0
        t0 = time.perf_counter()
        with ThreadPoolExecutor(nthr) as ex:
            while time.perf_counter() - t0 < a.seconds:
                done += len(list(ex.map(lambda b: decode(b, resize), blobs)))
        dt = time.perf_counter() - t0
        print('%-28s threads %2d: %6.1f images/s  (%.1f per thread)' % (
            'decode + PIL bicubic /2' if resize else 'decode only', nthr, done / dt, done / dt / nthr))
