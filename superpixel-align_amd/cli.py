"""The two command-line drivers of the label-generation path, on the MI355X pipeline.

  labelled   : batch_spalign_kmeans.py        (reference :38-108 flags, :427-548 loop/outputs)
  label-free : utils/apply_spalign_kmeans.py  (reference :75-145)

Flags, defaults (including the two dead flags), the "keep the batch size" loop, the NPY /
result.json / PNG naming and contents are the reference's; additions are optional flags
(--arch, --dtype, --drn_weights, --pool_mode, --mean_sampling, --no_figure, --balanced).
Under `python -m torch.distributed.run --nproc-per-node N` the labelled driver shards the image
range like utils/create_*_labels.sh does and rank 0 writes result.json after one all_gather.
"""
import argparse
import glob
import json
import os
import sys
import time
import zipfile

# skip MIOpen's naive reference convolution during solver search (slow start-up, never selected)
os.environ.setdefault('MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD', '0')

import numpy as np  # noqa: E402
import torch  # noqa: E402

from . import dist as spdist
from . import ops
from .pipeline import LabelPipeline

# (flag, kwargs) shared by both drivers, in the reference's order
_COMMON_FLAGS = [
    ('--superpixel_method', dict(type=str, default='felzenszwalb', choices=['felzenszwalb', 'slic'])),
    ('--n_clusters', dict(type=int, default=4)),
    ('--y_rel_pos', dict(type=float, default=0.75)),
    ('--x_rel_pos', dict(type=float, default=0.5)),
    ('--y_rel_sigma', dict(type=float, default=0.1)),
    ('--x_rel_sigma', dict(type=float, default=0.1)),
    ('--n_anchors', dict(type=int, default=10)),
    ('--n_neighbors', dict(type=int, default=4)),
    ('--without_pos', dict(action='store_true', default=False)),
    ('--horizontal_line_filtering', dict(action='store_true', default=False)),   # dead in the reference too
    ('--resize_shape', dict(type=int, nargs=2, default=[224, 224])),
    ('--batchsize', dict(type=int, default=30)),
    ('--felzenszwalb_scale', dict(type=float, default=300.0)),
    ('--felzenszwalb_sigma', dict(type=float, default=0.8)),
    ('--felzenszwalb_min_size', dict(type=int, default=20)),
    ('--n_slic_segments', dict(type=int, default=100)),
    ('--use_feature_maps', dict(type=int, nargs='*', default=[7])),
    ('--start_index', dict(type=int)),
    ('--end_index', dict(type=int)),
    # --- additions of this implementation
    ('--arch', dict(type=str, default='drn_c_26', choices=['drn_c_26', 'drn_d_22'])),
    ('--dtype', dict(type=str, default='fp32', choices=['fp32', 'bf16'])),
    ('--drn_weights', dict(type=str, default=None)),
    ('--pool_mode', dict(type=str, default='anchor', choices=['anchor', 'mean'])),
    ('--mean_sampling', dict(type=str, default='nearest', choices=['nearest', 'bilinear'])),
    ('--no_figure', dict(action='store_true', default=False)),
    ('--balanced', dict(action='store_true', default=False)),
    ('--io_threads', dict(type=int, default=8)),
    ('--decode_procs', dict(type=int, default=0)),       # > 0: decode PNGs in this many worker processes (shared-memory slabs)
    ('--strict_retry', dict(action='store_true', default=False)),     # k = 2: die with RecursionError where the reference does
    # k > 2: draw the initial assignment (np.random.shuffle) on the host, synchronously, as round 4 did (default: on the device)
    ('--host_kmeans_init', dict(action='store_true', default=False)),
    ('--host_resize', dict(action='store_true', default=False)),      # resize with Pillow on the host threads instead
    ('--resize_backend', dict(type=str, default='pil', choices=['pil', 'cv2'])),   # cv2: OpenCV's INTER_CUBIC algorithm (not pinned)
]


def _parser(extra):
    p = argparse.ArgumentParser()
    for flag, kw in extra + _COMMON_FLAGS:
        p.add_argument(flag, **kw)
    return p


def get_args(argv=None):
    """batch_spalign_kmeans.py:38-108"""
    extra = [('--gpu', dict(type=int, default=0)),
             ('--out_dir', dict(type=str, default='data/test_images')),
             ('--img_file_list', dict(type=str, default=None)),
             ('--label_file_list', dict(type=str, default=None)),
             ('--cityscapes_img_dir', dict(type=str, default=None)),
             ('--cityscapes_label_dir', dict(type=str, default=None)),
             ('--label_zip', dict(type=str, default=None)),      # extra: write the README's label archive when done
             ('--cityscapes_img_zip', dict(type=str, default=None)),
             ('--cityscapes_label_zip', dict(type=str, default=None)),
             ('--camera_param_dir', dict(type=str, default='data/camera'))]
    args = _parser(extra).parse_args(argv)
    args.resize_shape = tuple(args.resize_shape)
    os.makedirs(args.out_dir, exist_ok=True)
    return args


def get_args_direct(argv=None):
    """direct_clustering.py:38-108 — the labelled driver's flags with --n_clusters 4."""
    args = get_args(argv)
    return args


def get_args_overlaps(argv=None):
    """superpixel_overlaps.py:43-115 — plus --overlap_threshold; felzenszwalb 500 / 0.9 and
    --n_clusters 4 are that script's defaults."""
    argv = list(sys.argv[1:] if argv is None else argv)
    pre = argparse.ArgumentParser(add_help=False)
    pre.add_argument('--overlap_threshold', type=float, default=0.01)
    known, rest = pre.parse_known_args(argv)
    defaults = []
    for flag, val in (('--felzenszwalb_scale', '500.0'), ('--felzenszwalb_sigma', '0.9')):
        if flag not in rest:
            defaults += [flag, val]
    args = get_args(defaults + rest)
    args.overlap_threshold = known.overlap_threshold
    return args


def get_args_labelfree(argv=None):
    """utils/apply_spalign_kmeans.py:75-122"""
    extra = [('--img_list_fn', dict(type=str, default='data/demoVideo_fns.txt')),
             ('--label_shape', dict(type=int, nargs=2, default=[1024, 2048])),
             # (the reference's default here is -1 = its NumPy path, utils/apply_spalign_kmeans.py:84; this build has no CPU path —
             # ops.create_model refuses a negative id — so the script run without --gpu takes device 0 instead of aborting)
             ('--gpu', dict(type=int, default=0)),
             ('--out_dir', dict(type=str))]
    return _parser(extra).parse_args(argv)


# ------------------------------------------------------------------------------- data
def _decode(fp_or_path):
    from PIL import Image
    with Image.open(fp_or_path) as f:
        return np.asarray(f, dtype=np.uint8)


RESIZE_BACKEND = ['pil']         # --resize_backend: 'pil' (pinned against Pillow) or 'cv2' (OpenCV's algorithm, not pinned)


def _cv_cubic_taps(n_src, n_dst):
    """OpenCV INTER_CUBIC along one axis, 8-bit path: (clamped source indices (n_dst, 4), int32 taps (n_dst, 4)) — the
    float32 coefficients of interpolateCubic (A = -0.75) as saturate_cast<short>(c * 2048), cvRound = round half to even."""
    scale = float(n_src) / float(n_dst)
    d = np.arange(n_dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    x = (f - s.astype(np.float32)).astype(np.float32)
    A = np.float32(-0.75)
    one = np.float32(1)
    c0 = ((A * (x + one) - np.float32(5) * A) * (x + one) + np.float32(8) * A) * (x + one) - np.float32(4) * A
    c1 = ((A + np.float32(2)) * x - (A + np.float32(3))) * x * x + one
    c2 = ((A + np.float32(2)) * (one - x) - (A + np.float32(3))) * (one - x) * (one - x) + one
    c3 = one - c0 - c1 - c2
    idx = np.clip(s[:, None] - 1 + np.arange(4)[None, :], 0, n_src - 1)
    taps = np.rint(np.stack([c0, c1, c2, c3], 1).astype(np.float32) * np.float32(2048)).astype(np.int64)
    return idx, np.clip(taps, -32768, 32767).astype(np.int32)


def resize_cvcubic_chw(img_chw, shape):
    """cv2.resize(img.transpose(1, 2, 0), (w, h), interpolation=cv2.INTER_CUBIC).transpose(2, 0, 1) on a UINT8 image, as
    OpenCV's scalar 8-bit path computes it (int32 horizontal sums of byte x tap, int32 vertical sum, (v + 2^21) >> 22,
    saturated to 0..255; border replication): what the reference's datasets do before .astype(float32)
    (datasets/resize_image_dataset.py:20-36).  The host form of spa_resize_cvcubic_u8, same bytes.  Not pinned against cv2."""
    a = np.asarray(img_chw)
    if a.dtype != np.uint8:
        raise ValueError('resize_cvcubic_chw: the reference resizes the decoded uint8 image (got %s)' % a.dtype)
    C, H, W = a.shape
    h, w = int(shape[0]), int(shape[1])
    xi, xc = _cv_cubic_taps(W, w)
    yi, yc = _cv_cubic_taps(H, h)
    ai = a.astype(np.int32)
    r = ai[:, :, xi[:, 0]] * xc[:, 0]
    for k in (1, 2, 3):
        r = r + ai[:, :, xi[:, k]] * xc[:, k]                  # (C, H, w) int32
    o = r[:, yi[:, 0], :] * yc[:, 0][None, :, None]
    for k in (1, 2, 3):
        o = o + r[:, yi[:, k], :] * yc[:, k][None, :, None]
    return np.clip((o + (1 << 21)) >> 22, 0, 255).astype(np.uint8)


def resize_bicubic_chw(img_chw, shape):
    """datasets/resize_image_dataset.py:31-34 (chainercv.transforms.resize(img, size, 3)): Pillow's 8-bit bicubic
    (pinned), or with --resize_backend cv2 OpenCV's 8-bit INTER_CUBIC (the branch the reference environment ran; its
    algorithm restated, not pinned: no cv2 here)."""
    if RESIZE_BACKEND[0] == 'cv2':
        return resize_cvcubic_chw(img_chw, shape)
    from PIL import Image
    h, w = shape
    out = [np.asarray(Image.fromarray(c).resize((w, h), Image.BICUBIC)) for c in img_chw]
    return np.stack(out)


class PinnedRing(object):
    """A few persistent pinned host buffers of one shape, handed out in turn.  The decode workers copy their
    frames straight into a buffer (parallel memcpy, no np.stack into freshly mapped pages: that alone was 100+ ms
    per batch of 30 full-size images) and the upload from it is a true asynchronous DMA; a buffer is reused only
    after the event recorded behind its upload has completed."""

    def __init__(self, depth=3):
        self._bufs, self._depth = {}, depth

    def take(self, shape, dtype=torch.uint8):
        key = (tuple(shape), dtype)
        ring = self._bufs.setdefault(key, {'i': 0, 'slots': []})
        if len(ring['slots']) < self._depth:
            ring['slots'].append([torch.empty(shape, dtype=dtype).pin_memory(), None])
            slot = ring['slots'][-1]
        else:
            slot = ring['slots'][ring['i'] % self._depth]
            ring['i'] += 1
            if slot[1] is not None:
                slot[1].synchronize()
        return slot

    def upload(self, arrays, device, pool=None):
        """list of equally shaped arrays -> (len, ...) device tensor, enqueued on the current stream."""
        slot = self.take((len(arrays),) + tuple(arrays[0].shape), getattr(torch, str(arrays[0].dtype)))
        dst = slot[0].numpy()

        def put(j):
            dst[j] = arrays[j]
        if pool is not None:
            list(pool.map(put, range(len(arrays))))
        else:
            for j in range(len(arrays)):
                put(j)
        t = slot[0].to(device, non_blocking=True)
        slot[1] = torch.cuda.Event()
        slot[1].record(torch.cuda.current_stream(device))
        return t


class ShmTooSmall(RuntimeError):
    """/dev/shm cannot hold the decode slabs: the caller falls back to the decode threads."""


class ProcessDecoder(object):
    """Batches of PNGs decoded by worker processes into shared-memory slabs that are registered as pinned host memory
    (decode_worker.py).  take(idx) -> (images uint8 (B,H,W,3) device tensor, labels uint8 (B,H,W) device tensor, list of
    label arrays on the host) enqueued on the current stream, or None when a frame of the batch has another size
    than the first image of the run (the caller then decodes that batch on its threads)."""

    def __init__(self, imgs_ds, labels_ds, batch, n_procs, device, depth=3, host_labels=False):
        import multiprocessing as mp
        from concurrent.futures import ProcessPoolExecutor
        from multiprocessing import shared_memory
        from . import decode_worker
        self._w = decode_worker
        self.imgs, self.labels, self.device = imgs_ds, labels_ds, device
        first = imgs_ds.get_raw(0)
        gt0 = labels_ds.get_gray(0)
        self.ishape, self.gshape = tuple(first.shape), tuple(gt0.shape)
        self.B, self.host_labels = batch, host_labels
        self.ibytes = int(np.prod(self.ishape))
        self.gbytes = int(np.prod(self.gshape))
        slab = batch * (self.ibytes + self.gbytes)
        # the slabs live in /dev/shm: on a tmpfs smaller than they are (a container's default is 64 MB; 30 full-size frames
        # need 3 x 250 MB) SharedMemory(create=True) still succeeds and the workers die with SIGBUS on their first write
        try:
            st = os.statvfs('/dev/shm')
            free = st.f_bavail * st.f_frsize
        except OSError:
            free = None
        if free is not None and free < depth * slab + (16 << 20):
            raise ShmTooSmall('/dev/shm has %d MB free, the decode slabs need %d MB' % (free >> 20, (depth * slab) >> 20))
        self.slots = []
        self.pool = None
        try:
            for _ in range(depth):
                shm = shared_memory.SharedMemory(create=True, size=slab)
                slot = dict(shm=shm, t=None, pinned=False, ev=None)
                self.slots.append(slot)
                slot['t'] = t = torch.frombuffer(shm.buf, dtype=torch.uint8)
                try:
                    rc = torch.cuda.cudart().cudaHostRegister(t.data_ptr(), slab, 0)
                    slot['pinned'] = rc is None or int(rc) == 0
                except Exception:
                    slot['pinned'] = False                  # pageable slab: the upload is staged, still correct
            self.k = 0
            # spawn, not fork: this process has initialised the GPU
            self.pool = ProcessPoolExecutor(max_workers=n_procs, mp_context=mp.get_context('spawn'))
            list(self.pool.map(decode_worker.warm, range(n_procs)))
        except BaseException:
            self.close()                                # unregister and unlink what exists so far
            raise

    def _src(self, ds, i):
        p = ds._paths[i]
        if ds._open is not None:                       # zipped dataset: (archive, member)
            zf = getattr(ds._open, '__self__', None)
            return (zf.filename, p)
        return p

    def take(self, idx):
        slot = self.slots[self.k % len(self.slots)]
        self.k += 1
        if slot['ev'] is not None:
            slot['ev'].synchronize()                   # the slab's last upload has finished
        n = len(idx)
        name = slot['shm'].name
        tasks = [(name, j * self.ibytes, self.ishape, self._src(self.imgs, i)) for j, i in enumerate(idx)]
        tasks += [(name, self.B * self.ibytes + j * self.gbytes, self.gshape, self._src(self.labels, i)) for j, i in enumerate(idx)]
        shapes = list(self.pool.map(self._w.decode_into, tasks, chunksize=1))
        if any(tuple(sh) != (self.ishape if k < n else self.gshape) for k, sh in enumerate(shapes)):
            return None
        t = slot['t']
        img_h = t[:n * self.ibytes].view((n,) + self.ishape)
        gt_h = t[self.B * self.ibytes:self.B * self.ibytes + n * self.gbytes].view((n,) + self.gshape)
        img_d = img_h.to(self.device, non_blocking=True)
        gt_d = gt_h.to(self.device, non_blocking=True)
        slot['ev'] = torch.cuda.Event()
        slot['ev'].record(torch.cuda.current_stream(self.device))
        # the figure writer wants the labels on the host: copies, the slab comes round again before it runs
        return img_d, gt_d, [np.array(gt_h[j].numpy()) if self.host_labels else None for j in range(n)]

    def close(self):
        if self.pool is not None:
            self.pool.shutdown(wait=True, cancel_futures=True)
            self.pool = None
        for s in self.slots:
            if s.get('t') is not None:
                if s.get('pinned'):
                    try:
                        torch.cuda.cudart().cudaHostUnregister(s['t'].data_ptr())
                    except Exception:
                        pass
                s['t'] = None
            try:
                s['shm'].close()
                s['shm'].unlink()
            except Exception:
                pass
        self.slots = []


class ImageList(object):
    """ResizeImageDataset (datasets/resize_image_dataset.py:8-36): path -> CHW array."""

    def __init__(self, paths, resize_shape=None, dtype=np.float32, opener=None):
        self._paths, self._shape, self._dtype, self._open = list(paths), resize_shape, dtype, opener

    def __len__(self):
        return len(self._paths)

    def get(self, i):
        src = self._open(self._paths[i]) if self._open else self._paths[i]
        img = _decode(src)
        if img.ndim == 2:
            img = img[:, :, None]
        img = img[:, :, :3].transpose(2, 0, 1)
        if self._shape is not None and tuple(img.shape[1:]) != tuple(self._shape):
            img = resize_bicubic_chw(img, self._shape)
        return img.astype(self._dtype)

    def get_raw(self, i):
        """The decoded image as it comes out of the PNG decoder: (H, W, 3) uint8, not resized."""
        src = self._open(self._paths[i]) if self._open else self._paths[i]
        img = _decode(src)
        if img.ndim == 2:
            img = img[:, :, None]
        return np.ascontiguousarray(img[:, :, :3])

    def get_gray(self, i):
        """A single-channel PNG (gtFine labelIds) as it is decoded: (H, W) uint8."""
        src = self._open(self._paths[i]) if self._open else self._paths[i]
        img = _decode(src)
        return np.ascontiguousarray(img if img.ndim == 2 else img[:, :, 0])

    def batch_device(self, lo, hi, pool, engine, ring=None):
        """dataset[lo:hi] as a (B,3,h,w) float32 CUDA tensor: decode on the worker threads, upload the
        8-bit images (3 bytes per pixel) and resize on the GPU (spa_resize_bicubic_u8: bit exact with the
        Pillow resize of `get`).  Falls back to `batch` when the images of the batch differ in size."""
        idx = list(range(len(self))[lo:hi])
        raw = list(pool.map(self.get_raw, idx)) if pool is not None else [self.get_raw(i) for i in idx]
        if len({r.shape for r in raw}) != 1 or raw[0].shape[2] != 3:
            # images of different sizes (or greyscale, which stays one channel like the reference's dataset):
            # host path on the frames already decoded — a NumPy batch, the caller uploads it
            def host(r):
                img = r.transpose(2, 0, 1)
                if self._shape is not None and tuple(img.shape[1:]) != tuple(self._shape):
                    img = resize_bicubic_chw(img, self._shape)
                return img.astype(self._dtype)
            return np.stack(list(pool.map(host, raw)) if pool is not None else [host(r) for r in raw])
        if ring is None:
            if not hasattr(self, '_ring'):
                self._ring = PinnedRing()
            ring = self._ring
        u8 = ring.upload(raw, engine.device, pool)
        shape = tuple(self._shape) if self._shape is not None else raw[0].shape[:2]
        return engine.resize_u8(u8, shape, RESIZE_BACKEND[0])

    def batch(self, lo, hi, pool=None):
        """dataset[lo:hi] stacked (python slice semantics, like concat_examples(dataset[i:end_i]));
        decoded by `pool` threads when given (PIL and numpy release the GIL)."""
        idx = list(range(len(self))[lo:hi])
        imgs = list(pool.map(self.get, idx)) if pool is not None else [self.get(i) for i in idx]
        return np.stack(imgs)


def create_dataset(args):
    """batch_spalign_kmeans.py:486-521 -> (images, labels) ImageLists with ._paths."""
    if args.cityscapes_img_zip and args.cityscapes_label_zip:
        zi, zl = zipfile.ZipFile(args.cityscapes_img_zip), zipfile.ZipFile(args.cityscapes_label_zip)
        key = lambda fn: '_'.join(os.path.basename(fn).split('_')[:3])
        li = {key(f): f for f in zi.namelist() if f.endswith('.png')}
        ll = {key(f): f for f in zl.namelist() if f.endswith('labelIds.png')}
        keys = sorted(ll)
        return (ImageList([li[k] for k in keys], args.resize_shape, np.float32, zi.open),
                ImageList([ll[k] for k in keys], None, np.uint8, zl.open))
    if args.img_file_list and args.label_file_list:
        il = [l.strip() for l in open(args.img_file_list) if l.strip()]
        ll = [l.strip() for l in open(args.label_file_list) if l.strip()]
    else:
        key = lambda fn: '_'.join(os.path.basename(fn).split('_')[:3])
        vi = {key(f): f for f in glob.glob(os.path.join(args.cityscapes_img_dir, '*', '*.png'))}
        vl = {key(f): f for f in glob.glob(os.path.join(args.cityscapes_label_dir, '*', '*labelIds.png'))}
        il, ll = [vi[k] for k in vl], [vl[k] for k in vl]
    return ImageList(il, args.resize_shape, np.float32), ImageList(ll, None, np.uint8)


def create_label_mask(label):
    """:279-296 — void ids 0..6 -> -1, road id 7 -> 1, rest 0."""
    out = np.zeros(label.shape, np.int32)
    out[label <= 6] = -1
    out[label == 7] = 1
    return out


def resize_nearest(a, shape):
    """cv.resize(..., interpolation=cv.INTER_NEAREST): src = min(floor(dst * src/dst), src-1)."""
    h, w = shape
    ys = np.minimum((np.arange(h) * (a.shape[0] / h)).astype(np.int64), a.shape[0] - 1)
    xs = np.minimum((np.arange(w) * (a.shape[1] / w)).astype(np.int64), a.shape[1] - 1)
    return a[ys][:, xs]


# ------------------------------------------------------------------------------- outputs
def score(road_mask, gt):
    """:398-405 — chainercv confusion/IoU on the host copy (integer counts: exact)."""
    m = gt >= 0
    conf = np.bincount(2 * gt[m].astype(np.int64) + road_mask[m].astype(np.int64), minlength=4).reshape(2, 2)
    TP, FP, FN = int(conf[1, 1]), int(conf[0, 1]), int(conf[1, 0])
    with np.errstate(divide='ignore', invalid='ignore'):
        iou = np.diag(conf) / (conf.sum(1) + conf.sum(0) - np.diag(conf))
    return dict(road_iou=float(iou[1]), non_road_iou=float(iou[0]),
                precision=float(TP / (TP + FP)) if TP + FP > 0 else None,
                recall=float(TP / (TP + FN)) if TP + FN > 0 else None, TP=TP, FP=FP, FN=FN)


def score_from_counts(TN, FP, FN, TP):
    """The same scores from the four integer counts (spa_confusion on the device)."""
    conf = np.array([[TN, FP], [FN, TP]], np.int64)
    with np.errstate(divide='ignore', invalid='ignore'):
        iou = np.diag(conf) / (conf.sum(1) + conf.sum(0) - np.diag(conf))
    return dict(road_iou=float(iou[1]), non_road_iou=float(iou[0]),
                precision=float(TP / (TP + FP)) if TP + FP > 0 else None,
                recall=float(TP / (TP + FN)) if TP + FN > 0 else None, TP=TP, FP=FP, FN=FN)


def nearest_index(n_dst, n_src):
    """Source index of every destination index under cv.INTER_NEAREST (see resize_nearest)."""
    return np.minimum((np.arange(n_dst) * (n_src / n_dst)).astype(np.int64), n_src - 1)


def save_npy(args, img_fn, road_mask, clustering_result):
    """:392-396 — <basename>.npy (uint8, 1 = road) and <basename>_all_cluster.npy."""
    out_fn = os.path.splitext(os.path.basename(img_fn))[0]
    np.save(os.path.join(args.out_dir, out_fn), road_mask.astype(np.uint8, copy=False))
    np.save(os.path.join(args.out_dir, out_fn + '_all_cluster'), clustering_result.astype(np.uint8, copy=False))


def write_label_zip(out_dir, zip_path):
    """The README's packaging step (README.md:135)
        find <out_dir> -name "*leftImg8bit.npy" | zip -0r <zip_path> -@
    as a library call: the road masks (not the *_all_cluster.npy files) stored uncompressed under
    the path names `find` prints, which is what the training stage reads the labels from."""
    import fnmatch
    names = []
    for root, _dirs, files in os.walk(out_dir):
        for fn in files:
            if fnmatch.fnmatch(fn, '*leftImg8bit.npy'):
                names.append(os.path.join(root, fn))
    names.sort()
    with zipfile.ZipFile(zip_path, 'w', zipfile.ZIP_STORED, allowZip64=True) as zf:
        for fn in names:
            zf.write(fn, arcname=fn)
    return len(names)


def save_figure(args, img, road_mask, label, clustering_result, img_fn):
    """:361-386 — 2x2 overview figure at 300 dpi (about a second per image: --no_figure).
    Object-oriented matplotlib API (no pyplot state), so figures can be rendered from worker threads."""
    from matplotlib import cm
    from matplotlib.backends.backend_agg import FigureCanvasAgg
    from matplotlib.figure import Figure
    fig = Figure(dpi=300)
    FigureCanvasAgg(fig)
    axes = fig.subplots(2, 2)
    for ax in axes.ravel():
        ax.axis('off')
    axes[0, 0].imshow(img / 255.)
    axes[0, 0].imshow(road_mask, alpha=0.4, cmap=cm.Set1_r)
    axes[0, 0].set_title('Estimated road mask (input image overlayed)', fontsize=8)
    axes[0, 1].imshow(label == 1)
    axes[0, 1].set_title('Ground truth road mask', fontsize=8)
    axes[1, 0].imshow(clustering_result)
    axes[1, 0].set_title('All clusters', fontsize=8)
    axes[1, 1].imshow(clustering_result == 0)
    axes[1, 1].set_title('Estimated road mask', fontsize=8)
    fig.savefig(os.path.join(args.out_dir, os.path.basename(img_fn)), bbox_inches='tight')


def result_line(args, img_fn, label_fn, sc, elapsed_times, st_all):
    """:408-421 — one JSON object per image: scores + every CLI arg + timers."""
    info = dict(img_fn=img_fn, label_fn=label_fn, road_iou=sc['road_iou'],
                non_road_iou=sc['non_road_iou'], precision=sc['precision'], recall=sc['recall'],
                TP=sc['TP'], FP=sc['FP'], FN=sc['FN'])
    info.update(vars(args))
    info.update(elapsed_times)
    info['elapsed_time'] = time.time() - st_all
    return info


# ------------------------------------------------------------------------------- drivers
def _effective_range(args, n_data):
    start = 0 if args.start_index is None else args.start_index
    end = n_data if args.end_index is None else args.end_index
    return start, end


def append_result_line(path, line):
    """:407-422 — one JSON object per image, appended with ONE write on an O_APPEND descriptor: lines of
    the N processes the bash launchers start (utils/create_*_labels.sh) cannot interleave mid-line, and
    every finished image is on disk before the next batch runs."""
    data = (json.dumps(line) + '\n').encode()
    fd = os.open(path, os.O_WRONLY | os.O_APPEND | os.O_CREAT, 0o644)
    try:
        os.write(fd, data)
    finally:
        os.close(fd)


def main_labelled(argv=None, get=None, make_pipe=None, originals=False, make_model=None):
    """The labelled driver loop.  `get` / `make_pipe` select one of the three scripts that share it
    (batch_spalign_kmeans.py, direct_clustering.py, superpixel_overlaps.py); `originals`: the
    pipeline also wants the undecimated uint8 images of the batch (superpixel_overlaps.py:322);
    `make_model`: factory replacing create_model (tests of the host logic inject a stub)."""
    args = (get or get_args)(argv)
    RESIZE_BACKEND[0] = getattr(args, 'resize_backend', 'pil')
    rank, ws, local = spdist.init()
    if ws > 1:
        args.gpu = local
    imgs_ds, labels_ds = create_dataset(args)
    model = (make_model or ops.create_model)(args)
    pipe = make_pipe(args, model, None if make_model else ops.engine()) if make_pipe \
        else LabelPipeline(args, model, ops.engine())
    orig_ds = ImageList(imgs_ds._paths, None, np.uint8, imgs_ds._open) if originals else None
    start, end = _effective_range(args, len(imgs_ds))
    if ws > 1:
        s, e = spdist.shard_range(end - start, ws, rank, args.balanced)
        start, end = start + s, start + e
    # Host side of the loop: decode/resize of the NEXT batch and the per-image
    # outputs of the PREVIOUS one (GT decode, nearest resize, two .npy files, figure, scores) run on
    # worker threads while the GPU labels the current batch.
    from concurrent.futures import ThreadPoolExecutor
    workers = ThreadPoolExecutor(max_workers=max(1, args.io_threads))
    loader = ThreadPoolExecutor(max_workers=2)     # two batches ahead in the asynchronous loop, one otherwise
    # a range shorter than one batch that starts at image 0 makes the reference slice dataset[-k:end],
    # which is empty, and crash in concat_examples; here that range is labelled as one smaller batch
    ranges = [(max(lo, 0), hi) for lo, hi in spdist.batch_ranges(start, end, args.batchsize)] if end > start else []
    path = os.path.join(args.out_dir, 'result.json')

    def finish(i, rm, cl, n_sp, info, times, st_all, conf=None, gt_raw=None):
        """Everything the reference does per image after the masks exist (:461-483, :388-424).  conf: the
        image's {TN, FP, FN, TP} already counted on the device against its ground truth, with rm / cl already at
        the ground truth's size (the asynchronous loop); otherwise decode, resize and count here."""
        img_fn, label_fn = imgs_ds._paths[i], labels_ds._paths[i]
        if conf is not None:
            save_npy(args, img_fn, rm, cl)
            gt = None
            if not args.no_figure:
                gt = create_label_mask(gt_raw)
                full = _decode(imgs_ds._open(img_fn) if imgs_ds._open else img_fn)   # :464 reloads the PNG
                save_figure(args, full, rm, gt, cl, img_fn)
            sc = score_from_counts(*(int(v) for v in conf))
            tn = int(conf[0])
        else:
            gt = create_label_mask(labels_ds.get(i)[0] if gt_raw is None else gt_raw)
            if rm.shape != gt.shape:                                       # :470-477
                rm = resize_nearest(rm, gt.shape)
            if cl.shape != gt.shape:
                cl = resize_nearest(cl, gt.shape)
            save_npy(args, img_fn, rm, cl)
            if not args.no_figure:
                full = _decode(imgs_ds._open(img_fn) if imgs_ds._open else img_fn)   # :464 reloads the PNG
                save_figure(args, full, rm, gt, cl, img_fn)
            sc = score(rm, gt)
            tn = int(((gt == 0) & (rm == 0)).sum())
        line = result_line(args, img_fn, label_fn, sc, times, st_all)
        timers = dict(times, elapsed_time=line['elapsed_time'], gpu=args.gpu)
        return i, line, [i, tn, sc['FP'], sc['FN'], sc['TP'], n_sp, int(info[0]), int(info[1])] \
            + spdist.pack_timers(timers)

    records = []
    own = {}
    drained = set()

    def drain(futs):
        """Collect finished images; a single process appends their result.json lines NOW (the
        reference appends one line per image as it goes, :407-422), so a failure in a later batch
        loses nothing already computed.  Under torchrun rank 0 writes after the gather instead."""
        for f in futs:
            if f in drained:
                continue
            i, line, rec = f.result()
            drained.add(f)
            records.append(rec)
            own[i] = line
            if ws == 1:
                append_result_line(path, line)
            print('Road IoU:', line['road_iou'], os.path.basename(line['img_fn']))

    pending = []
    keep = []               # pinned buffers the writers of `pending` read from
    decoder = None
    # the asynchronous loop: LabelPipeline on the GPU (the baselines and the stub-model tests keep the simple one)
    use_async = make_pipe is None and make_model is None
    try:
        # GPU input stage: decode on the worker threads, resize on the device (bit exact with the host resize)
        # (its own spa_ctx: the loader thread must not share a context with the pipeline's thread)
        from .engine import Engine
        import threading
        have_gpu = make_model is None
        tls = threading.local()         # one spa_ctx + one stream per loader thread (a context's workspaces serve
        #                                 one caller at a time), pinned rings likewise

        def load(lo, hi):
            idx = list(range(len(imgs_ds)))[lo:hi]
            if have_gpu and not hasattr(tls, 'eng'):
                tls.eng = Engine(ops.engine().device.index)
                tls.stream = torch.cuda.Stream(device=tls.eng.device)
                tls.ring = PinnedRing()
                tls.gt_ring = PinnedRing()
            if decoder is not None:
                # worker processes decode images and labels into a pinned shared-memory slab; one DMA each
                with decoder_lock:
                    with torch.cuda.stream(tls.stream):
                        got = decoder.take(idx)
                        if got is not None:
                            u8, gt_dev, gts = got
                            shape = tuple(imgs_ds._shape) if imgs_ds._shape is not None else tuple(u8.shape[1:3])
                            t = tls.eng.resize_u8(u8, shape, RESIZE_BACKEND[0])
                            ev = torch.cuda.Event()
                            ev.record(tls.stream)
                            return t, ev, gts, gt_dev
            # ground truth PNGs of the batch decode alongside its images
            gt_f = [workers.submit(labels_ds.get_gray, i) for i in idx] if use_async else None
            gts = gt_dev = ev = None
            if have_gpu and not args.host_resize:
                with torch.cuda.stream(tls.stream):
                    t = imgs_ds.batch_device(lo, hi, workers, tls.eng, tls.ring)
            else:
                t = imgs_ds.batch(lo, hi, workers)
            if gt_f is not None:
                gts = [f.result() for f in gt_f]
                if len({g.shape for g in gts}) == 1:
                    with torch.cuda.stream(tls.stream):
                        gt_dev = tls.gt_ring.upload(gts, tls.eng.device, workers)
            if have_gpu and (isinstance(t, torch.Tensor) or gt_dev is not None):    # else: host arrays, nothing in flight
                ev = torch.cuda.Event()
                ev.record(tls.stream)
            return t, ev, gts, gt_dev

        def drain_all():
            drain(pending)
            del pending[:]
            del keep[:]

        decoder_lock = threading.Lock()
        if use_async and have_gpu and getattr(args, 'decode_procs', 0) > 0 and not args.host_resize and ranges:
            try:
                decoder = ProcessDecoder(imgs_ds, labels_ds, args.batchsize, args.decode_procs, ops.engine().device,
                                         host_labels=not args.no_figure)
            except ShmTooSmall as exc:
                print('--decode_procs: %s; decoding on the %d io threads instead' % (exc, args.io_threads), file=sys.stderr)
        depth = 2 if use_async else 1
        queue = [loader.submit(load, lo, hi) for lo, hi in ranges[:depth]]
        if use_async:
            # One batch of deferral: while the GPU labels batch k the host finishes batch k-1 (status, timers from
            # its device events, writers) and decodes batch k+1 — the main thread never waits for the batch it
            # has just enqueued.  Scoring runs on the device (create_label_mask, nearest resize to the ground
            # truth's size, spa_confusion), so a worker only writes the two .npy files and the result line.
            eng = ops.engine()
            dev = eng.device
            d2h = torch.cuda.Stream(device=dev)
            index_cache = {}

            def to_gt_size(m, shape):
                if tuple(m.shape[1:]) == tuple(shape):
                    return m
                key = (tuple(m.shape[1:]), tuple(shape))
                if key not in index_cache:
                    index_cache[key] = (torch.from_numpy(nearest_index(shape[0], m.shape[1])).to(dev),
                                        torch.from_numpy(nearest_index(shape[1], m.shape[2])).to(dev))
                ys, xs = index_cache[key]
                return m.index_select(1, ys).index_select(2, xs)

            def finalize(st):
                res = st['res']
                road_d, cl_d, conf_d = st.pop('dev')
                st['computed'].synchronize()
                with torch.cuda.stream(d2h):
                    st.update(road_h=torch.empty(road_d.shape, dtype=torch.uint8, pin_memory=True),
                              cl_h=torch.empty(cl_d.shape, dtype=torch.uint8, pin_memory=True),
                              info_h=torch.empty(res.info.shape, dtype=res.info.dtype, pin_memory=True),
                              nsp_h=torch.empty(res.n_labels.shape, dtype=res.n_labels.dtype, pin_memory=True),
                              conf_h=None if conf_d is None else torch.empty(conf_d.shape, dtype=conf_d.dtype, pin_memory=True),
                              fail_h=None if res.retry_fail is None else torch.empty(res.retry_fail.shape, dtype=torch.bool, pin_memory=True))
                    for h, d in ((st['road_h'], road_d), (st['cl_h'], cl_d), (st['info_h'], res.info),
                                 (st['nsp_h'], res.n_labels), (st['conf_h'], conf_d), (st['fail_h'], res.retry_fail)):
                        if h is not None:
                            h.copy_(d, non_blocking=True)
                            d.record_stream(d2h)
                    st['done'] = torch.cuda.Event()
                    st['done'].record(d2h)
                st['done'].synchronize()
                eng.raise_on_word(st['status_h'])
                if st['fail_h'] is not None:
                    st['res'].check_retry(st['fail_h'])
                elif getattr(st['res'], 'retry_info', None) is not None:
                    st['res'].check_retry()                 # k > 2 on the device: retry runs made / still pending
                times = pipe.elapsed_times(st['events'])
                # `elapsed_time` stays what the reference reports (:420: wall clock since the batch was started, which in this
                # loop includes the batches in flight ahead of it); the batch's own device time goes under its own key
                times['time_device'] = st['events']['start'].elapsed_time(st['computed']) / 1000.0
                info, n_sp = st['info_h'].numpy(), st['nsp_h'].numpy()
                conf = st['conf_h'].numpy() if st['conf_h'] is not None else None
                # a re-labelled image (last batch shifted back, :539-542) must overwrite its earlier files:
                # the previous batch's writers finish before this batch's are queued
                drain_all()
                keep.append(st)
                for j, i in enumerate(st['idx']):
                    pending.append(workers.submit(finish, i, st['road_h'][j].numpy(), st['cl_h'][j].numpy(), int(n_sp[j]),
                                                  info, times, st['st_all'],
                                                  None if conf is None else conf[j],
                                                  None if st['gts'] is None else st['gts'][j]))

            prev = None
            trace = os.environ.get('SPA_DRIVER_TRACE') == '1'      # host-side timeline of the loop, one line per batch
            for bi, (lo, hi) in enumerate(ranges):
                st_all = time.time()
                imgs, ready, gts, gt_dev = queue.pop(0).result()
                if bi + depth < len(ranges):
                    queue.append(loader.submit(load, ranges[bi + depth][0], ranges[bi + depth][1]))
                t_loaded = time.time()
                main = torch.cuda.current_stream()
                if ready is not None:
                    main.wait_event(ready)
                    for t in (imgs, gt_dev):
                        if isinstance(t, torch.Tensor):
                            t.record_stream(main)
                # join=False: the batch's tail (pooling, k-means, paint) runs on the pipeline's second stream and this stream goes
                # straight on to the next batch's DRN forward; everything below is enqueued where the result lives (res.stream)
                res = pipe.run(imgs, check_status=False, join=False)
                events = dict(pipe._ev)
                with torch.cuda.stream(res.stream):
                    road_d, cl_d, conf_d = res.road, res.cluster, None
                    if gt_dev is not None:
                        gt_dev.record_stream(res.stream)
                        road_d, cl_d = to_gt_size(res.road, gt_dev.shape[1:]), to_gt_size(res.cluster, gt_dev.shape[1:])
                        gtm = torch.where(gt_dev <= 6, -1, torch.where(gt_dev == 7, 1, 0)).to(torch.int32)   # :279-296
                        conf_d = eng.confusion(road_d.contiguous(), gtm)
                    status_h = eng.status_take_async()          # the bits raised so far (one atomic exchange, in stream order)
                    computed = torch.cuda.Event(enable_timing=True)
                    computed.record(res.stream)
                # the downloads are enqueued by finalize(), once the batch HAS been computed (the host waits, the copy stream does
                # not): a copy queue whose head is a barrier waiting a batch's time for the compute stream slows every dispatch of
                # that batch (measured: 82 -> 77.6 ms per batch of 30 in pipeline.HostStream, tools/h2h_probe2.py)
                st = dict(idx=list(range(len(imgs_ds)))[lo:hi], res=res, events=events, gts=gts, st_all=st_all,
                          status_h=status_h, computed=computed, dev=(road_d, cl_d, conf_d))
                t_enq = time.time()
                if prev is not None:
                    finalize(prev)
                    if trace:
                        print('trace batch %d: wait_load %.1f ms, enqueue %.1f ms, finalize(prev) %.1f ms, device gap '
                              'done(prev)->start %.1f ms, device batch(prev) %.1f ms'
                              % (bi, (t_loaded - st_all) * 1e3, (t_enq - t_loaded) * 1e3, (time.time() - t_enq) * 1e3,
                                 prev['done'].elapsed_time(events['start']) if False else -1.0,
                                 prev['events']['start'].elapsed_time(prev['computed'])), file=sys.stderr)
                prev = st
            if prev is not None:
                finalize(prev)
        else:
            for bi, (lo, hi) in enumerate(ranges):
                st_all = time.time()
                imgs, ready, _gts, _gt_dev = queue.pop(0).result()
                if ready is not None:
                    torch.cuda.current_stream().wait_event(ready)
                    if isinstance(imgs, torch.Tensor):
                        imgs.record_stream(torch.cuda.current_stream())
                if bi + depth < len(ranges):
                    queue.append(loader.submit(load, ranges[bi + depth][0], ranges[bi + depth][1]))
                res = pipe.run(imgs, orig_ds.batch(lo, hi, workers)) if originals else pipe.run(imgs)
                times = pipe.elapsed_times()
                cluster, road = res.masks_to_host()
                info = res.info.cpu().numpy()
                n_sp = res.n_labels.cpu().numpy()
                # a re-labelled image (last batch shifted back, :539-542) must overwrite its earlier files:
                # wait for the previous batch's writers before queueing this batch's
                # (`pending` keeps them until they are drained, so that the finally clause still sees the
                # successful ones if one of them failed)
                drain_all()
                for j, i in enumerate(list(range(len(imgs_ds)))[lo:hi]):
                    pending.append(workers.submit(finish, i, road[j], cluster[j], int(n_sp[j]), info, times, st_all))
    finally:
        # whatever was computed before an error (corrupt PNG, device status, OOM) still reaches disk
        ok = [f for f in pending if f.exception() is None]
        bad = [f for f in pending if f.exception() is not None]
        drain(ok)
        workers.shutdown()
        loader.shutdown()
        if decoder is not None:
            decoder.close()
    if bad:
        raise bad[0].exception()
    if ws > 1:
        # one collective: fixed-size records (scores + the stage timers of the image's batch); rank 0
        # rebuilds the lines it did not produce from them (file names come from the shared lists)
        allrec = spdist.gather_records(np.array(records, np.int64).reshape(-1, spdist.RECORD_WIDTH))
        if rank == 0:
            order = np.argsort(allrec[:, 0], kind='stable') if len(allrec) else []
            for r in allrec[order]:                                     # index order
                i = int(r[0])
                if i in own:
                    line = own[i]
                else:
                    TP, FP, FN, TN = int(r[4]), int(r[2]), int(r[3]), int(r[1])
                    conf = np.array([[TN, FP], [FN, TP]], np.float64)
                    with np.errstate(divide='ignore', invalid='ignore'):
                        iou = np.diag(conf) / (conf.sum(1) + conf.sum(0) - np.diag(conf))
                    sc = dict(road_iou=float(iou[1]), non_road_iou=float(iou[0]),
                              precision=float(TP / (TP + FP)) if TP + FP else None,
                              recall=float(TP / (TP + FN)) if TP + FN else None, TP=TP, FP=FP, FN=FN)
                    timers = spdist.unpack_timers(r)
                    line = result_line(args, imgs_ds._paths[i], labels_ds._paths[i], sc, {}, time.time())
                    line['gpu'] = int(timers.pop('gpu', args.gpu))
                    line.update(timers)                                  # time_* and elapsed_time of its rank
                append_result_line(path, line)
        spdist.barrier()
    if getattr(args, 'label_zip', None) and rank == 0:
        print('label archive: %d masks -> %s' % (write_label_zip(args.out_dir, args.label_zip), args.label_zip))
    return 0


def main_direct(argv=None):
    from .baselines import DirectClustering
    if argv is None and '--n_clusters' not in sys.argv:
        argv = sys.argv[1:] + ['--n_clusters', '4']                    # direct_clustering.py:44
    return main_labelled(argv, get_args_direct, lambda a, m, e: DirectClustering(a, m, e, ops._rng()[1]))


def main_overlaps(argv=None):
    from .baselines import SuperpixelOverlaps
    if argv is None and '--n_clusters' not in sys.argv:
        argv = sys.argv[1:] + ['--n_clusters', '4']                    # superpixel_overlaps.py:49
    return main_labelled(argv, get_args_overlaps, lambda a, m, e: SuperpixelOverlaps(a, m, e, ops._rng()[1]),
                         originals=True)


def main_labelfree(argv=None):
    args = get_args_labelfree(argv)
    args.resize_shape = tuple(args.resize_shape)
    RESIZE_BACKEND[0] = getattr(args, 'resize_backend', 'pil')
    model = ops.create_model(args)
    pipe = LabelPipeline(args, model, ops.engine())
    img_fns = sorted(fn.strip() for fn in open(args.img_list_fn) if fn.strip())
    print('img_fns:', len(img_fns))
    ds = ImageList(img_fns, args.resize_shape, np.float32)
    os.makedirs(args.out_dir, exist_ok=True)
    start, end = _effective_range(args, len(ds))
    from PIL import Image
    from concurrent.futures import ThreadPoolExecutor
    io_pool = ThreadPoolExecutor(max_workers=max(1, args.io_threads))
    for lo, hi in spdist.batch_ranges(start, end, args.batchsize):
        lo = max(lo, 0)                     # see main_labelled: a range shorter than one batch
        batch = ds.batch(lo, hi, io_pool) if args.host_resize else ds.batch_device(lo, hi, io_pool, ops.engine())   # same thread
        res = pipe.run(batch)
        _, road = res.masks_to_host()
        for j, i in enumerate(list(range(len(ds)))[lo:hi]):
            rm = road[j]
            if rm.shape != tuple(args.label_shape):                    # :62-67
                rm = resize_nearest(rm, tuple(args.label_shape))
            save_fn = os.path.join(args.out_dir, os.path.basename(img_fns[i]))
            Image.fromarray(rm.astype(np.uint8)).save(save_fn)          # cv.imwrite(save_fn, mask)
            print(save_fn)
    return 0
