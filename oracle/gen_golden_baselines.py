#!/opt/conda/bin/python3.9
"""ORACLE — TEST INFRASTRUCTURE ONLY.  Golden vectors for the paper's two baselines
(SURVEY.md section 8f-4): direct_clustering.py and superpixel_overlaps.py.

Run in the BUILD container only:

    PYTHONDONTWRITEBYTECODE=1 /opt/conda/bin/python3.9 oracle/gen_golden_baselines.py

The reference's own `estimate_road_mask` of each script is executed on the CPU path (--gpu -1,
numpy) with
  * stub modules for chainer / cupy / chainercv / cv2 / the dataset helpers (not installed),
  * a stub model whose batch_predict returns seeded synthetic feature maps,
  * PIL.Image.open replaced by a lookup of in-memory arrays (images and label maps of the same
    size as the feature maps, so that the cv2 nearest-neighbour resizes of :327-331 / :355-357 /
    :365-367 are never reached — cv2 is not installed and a stand-in would pin nothing),
  * save_image / save_info replaced by recorders of their (road_mask, clustering_result).
numpy's global RNG is seeded like the scripts seed it (1111) and the shuffled initial cluster
indices are recorded, as in gen_golden.py.  Only data is written to tests/golden/.
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
GOLD = os.path.join(ROOT, 'tests', 'golden')
REF = '/root/reference'
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, 'superpixel-align_amd'))

import synth  # noqa: E402

FILES = {}          # fake path -> array returned by Image.open


class _Img(object):
    @staticmethod
    def open(fn):
        return FILES[fn]


class Variable(object):
    def __init__(self, a):
        self.array = a


def import_reference(name):
    def mod(modname, **attrs):
        m = types.ModuleType(modname)
        m.__dict__.update(attrs)
        sys.modules[modname] = m
        return m

    shuffled = []

    class _R(object):
        def shuffle(self, a):
            np.random.shuffle(a)
            shuffled.append(np.array(a))

        def seed(self, s):
            np.random.seed(s)

    class _NpProxy(object):
        random = _R()

        def __getattr__(self, k):
            return getattr(np, k)

    xp = _NpProxy()
    cuda = mod('chainer.cuda', get_array_module=lambda *a: xp, to_cpu=lambda a: a,
               to_gpu=lambda a, *r: a, cupy=types.SimpleNamespace(ndarray=()))
    ch = mod('chainer', cuda=cuda, Variable=Variable, config=types.SimpleNamespace(train=False))
    ch.datasets = mod('chainer.datasets')
    ch.serializers = mod('chainer.serializers')
    ch.dataset = mod('chainer.dataset', concat_examples=None)
    ch.functions = mod('chainer.functions',
                       concat=lambda xs, axis=1: Variable(np.concatenate([getattr(x, 'array', x) for x in xs], axis)))
    cv = mod('chainercv')
    cv.evaluations = mod('chainercv.evaluations')
    mod('cupy', random=types.SimpleNamespace(seed=lambda s: None))
    mod('cv2', INTER_NEAREST=0)
    mod('drn')
    mod('resize_image_dataset', ResizeImageDataset=None)
    mod('zipped_cityscapes_road_dataset', ZippedCityscapesRoadDataset=None)
    sys.path.insert(0, REF)
    cwd = os.getcwd()
    ref = __import__(name)
    os.chdir(cwd)
    ref.Image = _Img
    return ref, xp, shuffled


def save(name, **arrays):
    path = os.path.join(GOLD, name + '.npz')
    np.savez_compressed(path, **arrays)
    print('%-28s %8.1f KB' % (name, os.path.getsize(path) / 1024.0))


def make_case(seed, n, C, h, w):
    rs = np.random.RandomState(seed)
    # smooth structure + noise, so that k-means takes several sweeps and the clusters are regions
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([np.sin(xx * rs.uniform(0.05, 0.4) + yy * rs.uniform(0.05, 0.4) + rs.uniform(0, 6))
                     for _ in range(n * C)]).reshape(n, C, h, w)
    fmap = (base + 0.3 * rs.standard_normal((n, C, h, w))).astype(np.float32)
    imgs = np.stack([np.clip(synth.synth_image(seed * 10 + i, h, w), 0, 255).astype(np.uint8) for i in range(n)])
    labels = rs.randint(0, 12, (n, h, w)).astype(np.uint8)
    return fmap, imgs, labels


def run(ref, xp, args, fmap, imgs, labels, tag):
    n = len(imgs)
    img_fns = ['%s_img_%d.png' % (tag, i) for i in range(n)]
    label_fns = ['%s_lab_%d.png' % (tag, i) for i in range(n)]
    for i in range(n):
        FILES[img_fns[i]] = imgs[i].transpose(1, 2, 0)          # HWC uint8, like a decoded PNG
        FILES[label_fns[i]] = labels[i]
    got = []
    ref.save_image = lambda *a, **k: None
    ref.save_info = lambda img_fn, label_fn, road_mask, clustering_result, label, et, st: (
        got.append((np.array(road_mask), np.array(clustering_result))) or {'road_iou': 0.0})
    ref.args = args

    class Model(object):
        pass
    model = Model()
    model.xp = xp
    maps = [None] * 7 + [Variable(fmap)]
    model.batch_predict = lambda x: (None, maps)
    np.random.seed(1111)
    ref.estimate_road_mask(np.zeros((n, 3, 8, 8), np.float32), img_fns, labels, label_fns, model, args)
    road = np.stack([g[0] for g in got]).astype(np.uint8)
    cluster = np.stack([g[1] for g in got]).astype(np.int64)
    return road, cluster


def main():
    # ---- direct_clustering.py
    ref, xp, shuffled = import_reference('direct_clustering')
    for (tag, seed, n, C, h, w, k) in [('dc_k2', 3, 3, 16, 12, 20, 2), ('dc_k4', 4, 4, 24, 14, 28, 4),
                                        ('dc_k4_512', 5, 2, 512, 28, 28, 4)]:
        del shuffled[:]
        fmap, imgs, labels = make_case(seed, n, C, h, w)
        args = types.SimpleNamespace(gpu=-1, n_clusters=k, y_rel_pos=0.75, x_rel_pos=0.5, y_rel_sigma=0.1,
                                     x_rel_sigma=0.1, use_feature_maps=[7], out_dir='.')
        prior = ref.create_prior(h, w, args.y_rel_pos, args.x_rel_pos, args.y_rel_sigma, args.x_rel_sigma)
        road, cluster = run(ref, xp, args, fmap, imgs, labels, tag)
        save('baseline_' + tag, fmap=fmap, prior=prior, road=road, cluster=cluster, k=np.int64(k),
             shuffled=(shuffled[0] if shuffled else np.zeros(0, np.int64)))
    for m in ('direct_clustering',):
        sys.modules.pop(m, None)

    # ---- superpixel_overlaps.py (superpixels by the script's own batch_superpixel on uint8 images)
    ref, xp, shuffled = import_reference('superpixel_overlaps')
    for (tag, seed, n, C, h, w, k, method, thr) in [('so_fz_k4', 6, 3, 16, 48, 96, 4, 'felzenszwalb', 0.01),
                                                     ('so_slic_k2', 7, 2, 16, 40, 64, 2, 'slic', 0.05)]:
        del shuffled[:]
        fmap, imgs, labels = make_case(seed, n, C, h, w)
        args = types.SimpleNamespace(gpu=-1, n_clusters=k, y_rel_pos=0.75, x_rel_pos=0.5, y_rel_sigma=0.1,
                                     x_rel_sigma=0.1, use_feature_maps=[7], out_dir='.',
                                     superpixel_method=method, felzenszwalb_scale=500.0,
                                     felzenszwalb_sigma=0.9, felzenszwalb_min_size=20, n_slic_segments=20,
                                     overlap_threshold=thr)
        sp = ref.batch_superpixel(args, imgs)
        refined, cluster = run(ref, xp, args, fmap, imgs, labels, tag)
        save('baseline_' + tag, fmap=fmap, imgs=imgs, superpixels=np.asarray(sp, np.int64), refined=refined,
             cluster=cluster, k=np.int64(k), thr=np.float64(thr),
             shuffled=(shuffled[0] if shuffled else np.zeros(0, np.int64)))


if __name__ == '__main__':
    main()
