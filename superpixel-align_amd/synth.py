"""Synthetic inputs for benchmarks, parity tests and golden-vector generation.

There is no Cityscapes data and no pretrained DRN in the build/test environment,
so every measurement and parity case runs on seeded synthetic data
(SURVEY.md section 8d).  Pure noise collapses to a single SLIC segment after the
connectivity pass, so the generator mixes smooth sinusoids (which give SLIC
non-degenerate, image-like segments) with a little Gaussian noise.

Only numpy's legacy ``RandomState`` is used: its stream is frozen across numpy
versions, so the same seed gives the same image under numpy 1.26 (golden
generation) and numpy 2.x (tests / bench).
"""
import numpy as np


def synth_image(seed, height, width, dtype=np.float32, integer_valued=False):
    """One RGB image, CHW, values in [0, 255].

    per channel: 127 + 60 sin(a0 x + a1 y) + 50 cos(a2 x - a3 y) + 4 N(0, 1),
    a_i ~ U(0.002, 0.022), clipped to [0, 255].
    """
    rs = np.random.RandomState(seed)
    ys, xs = np.mgrid[0:height, 0:width].astype(np.float64)
    out = np.empty((3, height, width), dtype=np.float64)
    for c in range(3):
        a = rs.uniform(0.002, 0.022, size=4)
        out[c] = (127.0 + 60.0 * np.sin(a[0] * xs + a[1] * ys)
                  + 50.0 * np.cos(a[2] * xs - a[3] * ys)
                  + 4.0 * rs.standard_normal((height, width)))
    np.clip(out, 0.0, 255.0, out=out)
    if integer_valued:
        out = np.floor(out)
    return out.astype(dtype)


def synth_batch(seeds, height, width, dtype=np.float32, integer_valued=False):
    """Stack of images, (B, 3, H, W)."""
    return np.stack([synth_image(s, height, width, dtype, integer_valued)
                     for s in seeds])


def synth_feature_map(seed, channels, fh, fw, batch=1):
    """Random-normal feature maps (B, C, fh, fw) float32 (BASELINE config 1)."""
    rs = np.random.RandomState(seed)
    return rs.standard_normal((batch, channels, fh, fw)).astype(np.float32)


def synth_gt_labels(seed, height, width):
    """Cityscapes-style labelIds image (uint8) with a road-ish trapezoid.

    ids 0..6 are void, 7 is road, everything else is non-road
    (reference batch_spalign_kmeans.py:279-296).
    """
    rs = np.random.RandomState(seed + 7919)
    lab = np.full((height, width), 11, dtype=np.uint8)  # building
    horizon = int(height * rs.uniform(0.45, 0.6))
    ys, xs = np.mgrid[0:height, 0:width]
    cx = width * rs.uniform(0.4, 0.6)
    half = (ys - horizon) * (width * 0.5 / max(1, height - horizon))
    road = (ys > horizon) & (np.abs(xs - cx) < half)
    lab[road] = 7
    lab[: int(height * 0.05)] = 0          # void strip (ignored in scoring)
    lab[int(height * 0.95):] = 1           # ego vehicle = void
    return lab


def synth_scene(seed, height, width, n_rect=40, noise=5.0, dtype=np.float32, integer_valued=True):
    """Piecewise-constant 'street scene': random coloured rectangles over a vertical gradient,
    plus Gaussian noise; CHW, 0..255.  Gives graph-based segmentation (felzenszwalb) many regions
    with sharp borders, and — with integer_valued and zero noise inside some rectangles — large
    groups of exactly equal edge costs."""
    rs = np.random.RandomState(seed + 4242)
    ys = np.linspace(0.0, 1.0, height)[:, None]
    img = np.stack([60 + 120 * ys + 0 * np.zeros((1, width)),
                    90 + 60 * ys + 0 * np.zeros((1, width)),
                    140 - 70 * ys + 0 * np.zeros((1, width))]).astype(np.float64)
    for i in range(n_rect):
        h = rs.randint(max(2, height // 12), max(3, height // 2))
        w = rs.randint(max(2, width // 12), max(3, width // 2))
        y0 = rs.randint(0, height - h + 1)
        x0 = rs.randint(0, width - w + 1)
        col = rs.uniform(0, 255, size=3)
        img[:, y0:y0 + h, x0:x0 + w] = col[:, None, None]
        if i % 4 != 0:                       # three quarters of the rectangles are noisy
            img[:, y0:y0 + h, x0:x0 + w] += noise * rs.standard_normal((3, h, w))
    np.clip(img, 0.0, 255.0, out=img)
    if integer_valued:
        img = np.floor(img)
    return img.astype(dtype)
