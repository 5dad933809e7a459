"""ORACLE — TEST INFRASTRUCTURE ONLY.

ctypes front end of oracle/liborc.so, the plain-C CPU restatement of the reference's
label-generation hot path (batch_spalign_kmeans.py / utils/apply_spalign_kmeans.py).
Only tests/, bench.py's ``cpu_baseline`` leg and ``__graft_entry__.smoke()`` may import
this module, and only as the checker; the product (superpixel-align_amd/) never does.

Every function cites the reference lines it restates; the C sources carry the details.
The batch_* functions at the bottom mirror the reference's five boundary ops
(utils/apply_spalign_kmeans.py:17-21) so that parity tests read like the reference driver.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_i64 = ctypes.c_int64
_P = ctypes.c_void_p
_dbl = ctypes.c_double


def build(force=False):
    """Compile liborc.so with gcc (a few seconds). Building the checker is not using it."""
    so = os.path.join(_HERE, 'liborc.so')
    srcs = [os.path.join(_HERE, f) for f in
            ('slic_oracle.c', 'pool_oracle.c', 'kmeans_oracle.c', 'fz_oracle.c', 'resize_oracle.c', 'detmath.h',
             'glibc_flt32.h', 'Makefile')]
    if force or not os.path.exists(so) or \
            any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(['make', '-s', '-C', _HERE, 'liborc.so'])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build())
        L.orc_regular_grid.restype = ctypes.c_int
        L.orc_regular_grid.argtypes = [_i64, _i64, _i64, _P, _P, _P]
        L.orc_slic_core.restype = _i64
        L.orc_slic_core.argtypes = [_P, _i64, _i64, _i64, _i64, _P, _P, _i64]
        L.orc_enforce_connectivity.restype = _i64
        L.orc_enforce_connectivity.argtypes = [_P, _i64, _i64, _i64, _i64, _P]
        L.orc_rgb2lab_scaled.restype = None
        L.orc_rgb2lab_scaled.argtypes = [_P, _i64, _i64, ctypes.c_float, _P]
        L.orc_slic.restype = _i64
        L.orc_slic.argtypes = [_P, _i64, _i64, _i64, _dbl, _i64, _P]
        L.orc_glibc_powf_vec.restype = None
        L.orc_glibc_powf_vec.argtypes = [_P, ctypes.c_float, _i64, _P]
        L.orc_glibc_cbrtf_vec.restype = None
        L.orc_glibc_cbrtf_vec.argtypes = [_P, _i64, _P]
        L.orc_slic_core_f64.restype = _i64
        L.orc_slic_core_f64.argtypes = [_P, _i64, _i64, _i64, _i64, _P, _P, _i64]
        L.orc_rgb2lab_u8_f64.restype = None
        L.orc_rgb2lab_u8_f64.argtypes = [_P, _i64, _i64, _dbl, _P]
        L.orc_slic_u8.restype = _i64
        L.orc_slic_u8.argtypes = [_P, _i64, _i64, _i64, _dbl, _i64, _P]
        L.orc_segment_stats.restype = None
        L.orc_segment_stats.argtypes = [_P, _i64, _i64, _i64, _P, _P, _P]
        L.orc_create_prior.restype = None
        L.orc_create_prior.argtypes = [_P, _i64, _i64, _i64, _dbl, _dbl, _dbl, _dbl, _P]
        L.orc_anchor_pool.restype = None
        L.orc_anchor_pool.argtypes = [_P, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64,
                                      _i64, _P, _P, _P, _P, ctypes.c_int, _P]
        L.orc_mean_pool.restype = ctypes.c_int
        L.orc_mean_pool.argtypes = [_P, _i64, _i64, _i64, _i64, _i64, _i64, _P, _i64, _i64,
                                    _i64, ctypes.c_int, _P]
        L.orc_mt_new.restype = _P
        L.orc_mt_free.argtypes = [_P]
        L.orc_mt_seed_python.argtypes = [_P, ctypes.c_uint64]
        L.orc_mt_seed_numpy.argtypes = [_P, ctypes.c_uint32]
        L.orc_py_shuffle_select.argtypes = [_P, _i64, _i64, _P]
        L.orc_np_shuffle_i64.argtypes = [_P, _P, _i64]
        L.orc_kmeans.restype = _i64
        L.orc_kmeans.argtypes = [_i64, _P, _i64, _i64, _P, _P, _i64, _P, _P]
        L.orc_kmeans_f32.restype = _i64
        L.orc_kmeans_f32.argtypes = [_i64, _P, _i64, _i64, _P, _P, _i64, _P, _P]
        L.orc_paint.restype = None
        L.orc_paint.argtypes = [_P, _i64, _P, _P, _P]
        L.orc_confusion.restype = None
        L.orc_confusion.argtypes = [_P, _P, _i64, _P]
        _LIB = L
    return _LIB


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


# --------------------------------------------------------------------------- SLIC
def regular_grid(H, W, n_points):
    """skimage.util.regular_grid((1, H, W), n) -> (start[3], step[3]) in z, y, x order."""
    st = np.zeros(3, np.int64); sp = np.zeros(3, np.int64); has = np.zeros(3, np.int32)
    rc = lib().orc_regular_grid(H, W, n_points, st.ctypes.data, sp.ctypes.data, has.ctypes.data)
    if rc != 0:
        raise ZeroDivisionError('regular_grid: n_points too large for the image')
    return st, sp


def rgb2lab_scaled(img_chw, compactness=10.0):
    """rgb2lab(img) * (1/compactness) as slic() computes it; (3,H,W) f32 -> (H,W,3) f32."""
    img = _c(img_chw, np.float32)
    _, H, W = img.shape
    out = np.empty((H, W, 3), np.float32)
    lib().orc_rgb2lab_scaled(img.ctypes.data, H, W, np.float32(1.0 / compactness), out.ctypes.data)
    return out


def glibc_powf(x, y):
    """glibc 2.35 powf(x, y) restated (glibc_flt32.h), elementwise on a float32 array."""
    x = _c(x, np.float32)
    out = np.empty_like(x)
    lib().orc_glibc_powf_vec(x.ctypes.data, np.float32(y), x.size, out.ctypes.data)
    return out


def glibc_cbrtf(x):
    """glibc 2.35 cbrtf(x) restated (glibc_flt32.h), elementwise on a float32 array."""
    x = _c(x, np.float32)
    out = np.empty_like(x)
    lib().orc_glibc_cbrtf_vec(x.ctypes.data, x.size, out.ctypes.data)
    return out


def slic_core(lab_hwc, n_segments, max_iter=10):
    """_slic_cython on a given (already scaled) Lab image -> (labels int64 (H,W), centres (n,6))."""
    lab = _c(lab_hwc, np.float32)
    H, W, _ = lab.shape
    labels = np.empty((H, W), np.int64)
    cen = np.zeros((max(8, 4 * n_segments + 64), 6), np.float32)
    n = lib().orc_slic_core(lab.ctypes.data, H, W, n_segments, max_iter,
                            labels.ctypes.data, cen.ctypes.data, cen.shape[0])
    if n <= 0:
        raise RuntimeError('orc_slic_core failed: %d' % n)
    return labels, cen[:n].copy()


def connectivity_sizes(H, W, n_centroids, min_size_factor=0.5, max_size_factor=3):
    """slic_superpixels.py:322-327: int(factor * (prod(shape) / n_centroids))."""
    seg = (H * W) / n_centroids
    return int(min_size_factor * seg), int(max_size_factor * seg)


def enforce_connectivity(labels, min_size, max_size):
    lab = _c(labels, np.int64)
    H, W = lab.shape
    out = np.empty((H, W), np.int64)
    n = lib().orc_enforce_connectivity(lab.ctypes.data, H, W, min_size, max_size, out.ctypes.data)
    return out, int(n)


def slic(img_chw, n_segments, compactness=10.0, max_iter=10):
    """slic(img.transpose(1,2,0), n_segments) as batch_spalign_kmeans.py:311 calls it."""
    img = _c(img_chw, np.float32)
    _, H, W = img.shape
    out = np.empty((H, W), np.int64)
    n = lib().orc_slic(img.ctypes.data, H, W, n_segments, compactness, max_iter, out.ctypes.data)
    if n < 0:
        raise RuntimeError('orc_slic failed: %d' % n)
    return out


def rgb2lab_u8_f64(img_chw_u8, compactness=10.0):
    """rgb2lab(img_as_float(uint8 image)) * (1/compactness) in float64, as slic() computes it for the uint8
    input of superpixel_overlaps.py:303; (3,H,W) u8 -> (H,W,3) f64."""
    img = _c(img_chw_u8, np.uint8)
    _, H, W = img.shape
    out = np.empty((H, W, 3), np.float64)
    lib().orc_rgb2lab_u8_f64(img.ctypes.data, H, W, 1.0 / compactness, out.ctypes.data)
    return out


def slic_core_f64(lab_hwc, n_segments, max_iter=10):
    """_slic_cython[double] on a given (already scaled) float64 Lab image -> (labels (H,W), centres (n,6))."""
    lab = _c(lab_hwc, np.float64)
    H, W, _ = lab.shape
    labels = np.empty((H, W), np.int64)
    cen = np.zeros((max(8, 4 * n_segments + 64), 6), np.float64)
    n = lib().orc_slic_core_f64(lab.ctypes.data, H, W, n_segments, max_iter,
                                labels.ctypes.data, cen.ctypes.data, cen.shape[0])
    if n <= 0:
        raise RuntimeError('orc_slic_core_f64 failed: %d' % n)
    return labels, cen[:n].copy()


def slic_u8(img_chw_u8, n_segments, compactness=10.0, max_iter=10):
    """slic(img.transpose(1,2,0), n_segments) on a uint8 image as superpixel_overlaps.py:303 calls it."""
    img = _c(img_chw_u8, np.uint8)
    _, H, W = img.shape
    out = np.empty((H, W), np.int64)
    n = lib().orc_slic_u8(img.ctypes.data, H, W, n_segments, compactness, max_iter, out.ctypes.data)
    if n < 0:
        raise RuntimeError('orc_slic_u8 failed: %d' % n)
    return out


# --------------------------------------------------------------------------- descriptors
def segment_stats(labels, S=None):
    lab = _c(labels, np.int32)
    H, W = lab.shape
    S = int(lab.max()) + 1 if S is None else S
    cnt = np.zeros(S, np.int64); cy = np.zeros(S); cx = np.zeros(S)
    lib().orc_segment_stats(lab.ctypes.data, H, W, S, cnt.ctypes.data, cy.ctypes.data, cx.ctypes.data)
    return cnt, cy, cx


def create_prior(labels, y_rel_pos=0.75, x_rel_pos=0.5, y_rel_sigma=0.1, x_rel_sigma=0.2, S=None):
    """batch_spalign_kmeans.py:111-129."""
    lab = _c(labels, np.int32)
    H, W = lab.shape
    S = int(lab.max()) + 1 if S is None else S
    out = np.zeros(S)
    lib().orc_create_prior(lab.ctypes.data, H, W, S, y_rel_pos, x_rel_pos, y_rel_sigma,
                           x_rel_sigma, out.ctypes.data)
    return out


class PyRandom(object):
    """CPython ``random`` module state (random.seed(1111), batch_spalign_kmeans.py:33)."""

    def __init__(self, seed=1111):
        self._s = lib().orc_mt_new()
        lib().orc_mt_seed_python(self._s, seed)

    def __del__(self):
        try:
            lib().orc_mt_free(self._s)
        except Exception:
            pass

    def shuffle_select(self, n, n_select):
        """random.shuffle(list of n) then [:n_select] -> original indices."""
        m = min(n, n_select)
        out = np.zeros(max(m, 1), np.int64)
        lib().orc_py_shuffle_select(self._s, n, n_select, out.ctypes.data)
        return out[:m]


class NpRandom(object):
    """numpy legacy global RandomState (np.random.seed(1111), :34)."""

    def __init__(self, seed=1111):
        self._s = lib().orc_mt_new()
        lib().orc_mt_seed_numpy(self._s, seed)

    def __del__(self):
        try:
            lib().orc_mt_free(self._s)
        except Exception:
            pass

    def shuffle(self, a):
        assert a.dtype == np.int64 and a.flags.c_contiguous
        lib().orc_np_shuffle_i64(self._s, a.ctypes.data, a.size)


def select_anchors(labels, n_anchors, pyrandom, S=None):
    """:224-234 — per superpixel (ascending id) shuffle its raster-ordered pixel list with
    the process-global CPython RNG and keep the first n_anchors.
    -> anchors (S, n_anchors, 2) int32 (y, x), n_valid (S,) int32"""
    lab = np.asarray(labels)
    S = int(lab.max()) + 1 if S is None else S
    H, W = lab.shape
    order = np.argsort(lab.ravel(), kind='stable')      # raster order inside each label
    counts = np.bincount(lab.ravel(), minlength=S)
    starts = np.concatenate([[0], np.cumsum(counts)])
    anchors = np.zeros((S, n_anchors, 2), np.int32)
    n_valid = np.zeros(S, np.int32)
    for s in range(S):
        n = int(counts[s])
        pick = pyrandom.shuffle_select(n, n_anchors)
        pix = order[starts[s] + pick]
        anchors[s, :len(pick), 0] = pix // W
        anchors[s, :len(pick), 1] = pix % W
        n_valid[s] = len(pick)
    return anchors, n_valid


def _strides_chw(fmap):
    es = fmap.itemsize
    return [s // es for s in fmap.strides]


def anchor_pool(fmap_chw, labels, img_h, anchors, n_valid, n_neighbor=4, append_pos=True):
    """superpixel_align (:210-276) for one image given the selected anchors."""
    f = np.asarray(fmap_chw)
    assert f.dtype == np.float32
    C, fh, fw = f.shape
    sc, sy, sx = _strides_chw(f)
    S, A, _ = anchors.shape
    cnt, cy, cx = segment_stats(labels, S)
    D = C + (2 if append_pos else 0)
    out = np.zeros((S, D))
    a = _c(anchors, np.int32); nv = _c(n_valid, np.int32)
    lib().orc_anchor_pool(f.ctypes.data, C, fh, fw, sc, sy, sx, img_h, S, A, n_neighbor,
                          a.ctypes.data, nv.ctypes.data, cy.ctypes.data, cx.ctypes.data,
                          1 if append_pos else 0, out.ctypes.data)
    return out if append_pos else out.astype(np.float32)


def mean_pool(fmap_chw, labels, mode='nearest', S=None):
    """Dense per-segment mean (notebooks/Superpixel_Align.ipynb cell 4)."""
    f = np.asarray(fmap_chw)
    assert f.dtype == np.float32
    C, fh, fw = f.shape
    sc, sy, sx = _strides_chw(f)
    lab = _c(labels, np.int32)
    H, W = lab.shape
    S = int(lab.max()) + 1 if S is None else S
    out = np.zeros((S, C), np.float32)
    rc = lib().orc_mean_pool(f.ctypes.data, C, fh, fw, sc, sy, sx, lab.ctypes.data, H, W, S,
                             {'nearest': 0, 'bilinear': 1}[mode], out.ctypes.data)
    if rc != 0:
        raise RuntimeError('orc_mean_pool: slot overflow')
    return out


# --------------------------------------------------------------------------- k-means
def kmeans(k, X, weights, n_iter=1000, nprandom=None):
    """kmeans() :136-183. Returns (assign int32, iterations, status)."""
    f32 = np.asarray(X).dtype == np.float32        # --without_pos descriptors: numpy computes in float32
    X = _c(X, np.float32 if f32 else np.float64); w = _c(weights, np.float64)
    N, D = X.shape
    init_other = None
    ptr = None
    if k > 2:
        thr = np.sort(w)[N // 2]
        m = int((w <= thr).sum())
        init_other = (np.arange(m) % (k - 1) + 1).astype(np.int64)
        (nprandom or NpRandom()).shuffle(init_other)
        ptr = init_other.ctypes.data
    assign = np.zeros(N, np.int32); status = np.zeros(1, np.int32)
    fn = lib().orc_kmeans_f32 if f32 else lib().orc_kmeans
    it = fn(k, X.ctypes.data, N, D, w.ctypes.data, ptr, n_iter, assign.ctypes.data, status.ctypes.data)
    return assign, int(it), int(status[0])


def paint(labels, assign_img):
    lab = _c(labels, np.int32)
    a = _c(assign_img, np.int32)
    cl = np.empty(lab.shape, np.uint8); road = np.empty(lab.shape, np.uint8)
    lib().orc_paint(lab.ctypes.data, lab.size, a.ctypes.data, cl.ctypes.data, road.ctypes.data)
    return cl, road


def create_label_mask(label):
    """:279-296 — ids 0..6 -> -1 (void), 7 -> 1 (road), else 0."""
    out = np.zeros(label.shape, np.int32)
    out[label <= 6] = -1
    out[label == 7] = 1
    return out


def confusion(road_mask, gt_mask):
    """:398-405 -> dict(TP, FP, FN, TN, road_iou, non_road_iou, precision, recall)."""
    p = _c(road_mask, np.uint8); g = _c(gt_mask, np.int32)
    out = np.zeros(4, np.int64)
    lib().orc_confusion(p.ctypes.data, g.ctypes.data, p.size, out.ctypes.data)
    TN, FP, FN, TP = (int(v) for v in out)
    conf = np.array([[TN, FP], [FN, TP]], np.float64)
    with np.errstate(divide='ignore', invalid='ignore'):
        iou = np.diag(conf) / (conf.sum(1) + conf.sum(0) - np.diag(conf))
    return dict(TP=TP, FP=FP, FN=FN, TN=TN, road_iou=float(iou[1]), non_road_iou=float(iou[0]),
                precision=(TP / (TP + FP)) if TP + FP > 0 else None,
                recall=(TP / (TP + FN)) if TP + FN > 0 else None)


# --------------------------------------------------------------------------- boundary ops
def batch_superpixel(args, imgs):
    """:299-313 (SLIC branch)."""
    if args.superpixel_method == 'felzenszwalb':
        return np.asarray([felzenszwalb(img, args.felzenszwalb_scale, args.felzenszwalb_sigma,
                                        args.felzenszwalb_min_size) for img in imgs])
    return np.asarray([slic(img, args.n_slic_segments) for img in imgs])


def batch_superpixel_align(args, imgs, superpixels, feature_maps, pyrandom, pool_mode='anchor',
                           mean_sampling='nearest'):
    """:316-330. feature_maps (B, C, fh, fw) float32."""
    feats, n_per = [], []
    append_pos = not args.without_pos
    for img, sp, fm in zip(imgs, superpixels, feature_maps):
        S = len(np.unique(sp))
        n_per.append(S)
        if pool_mode == 'anchor':
            anchors, n_valid = select_anchors(sp, args.n_anchors, pyrandom, S)
            feats.append(anchor_pool(fm, sp, img.shape[1], anchors, n_valid, args.n_neighbors,
                                     append_pos))
        else:
            f = mean_pool(fm, sp, mean_sampling, S)
            if append_pos:
                _, cy, cx = segment_stats(sp, S)
                f = np.hstack([f.astype(np.float64), cy[:, None], cx[:, None]])
            feats.append(f)
    return np.concatenate(feats, axis=0), n_per


def batch_create_prior(args, superpixels):
    """:333-344."""
    return np.concatenate([create_prior(sp, args.y_rel_pos, args.x_rel_pos, args.y_rel_sigma,
                                        args.x_rel_sigma) for sp in superpixels])


RETRY_DEPTH_LIMIT = 990        # CPython's default recursion limit (1000) minus the frames below weighted_kmeans


def weighted_kmeans(superpixels, superpixel_features, superpixel_weights, k, n_superpixels_per_image,
                    nprandom=None, _depth=0):
    """weighted_kmeans() :186-207 including its retry: after painting image b, if the image has no cluster-0
    pixel the function prints and calls ITSELF with the same arguments, result discarded (:201-205), then goes on
    with image b + 1.  Every call runs kmeans(), i.e. one np.random.shuffle of the initial assignment (:147-149):
    for k > 2 the retries move the stream later batches draw from; for k = 2 the assignment does not depend on the
    shuffle, the retry fails the same way and the reference ends in RecursionError — reproduced as such."""
    if _depth > RETRY_DEPTH_LIMIT:
        raise RecursionError('maximum recursion depth exceeded (weighted_kmeans retry, batch_spalign_kmeans.py:201-205)')
    if nprandom is None:
        nprandom = NpRandom()
    if k == 2:
        # kmeans() shuffles an all-ones vector here: no effect on the result, but the stream advances
        N = np.asarray(superpixel_features).shape[0]
        w = np.asarray(superpixel_weights, np.float64)
        ones = np.ones(int((w <= np.sort(w)[N // 2]).sum()), np.int64)
        nprandom.shuffle(ones)
    assign, it, status = kmeans(k, superpixel_features, superpixel_weights, nprandom=nprandom)
    cl = np.zeros(np.asarray(superpixels).shape, np.uint8)
    off = 0
    for b, n in enumerate(n_superpixels_per_image):
        cl[b], _ = paint(superpixels[b], assign[off:off + n])
        off += n
        if (cl[b] == 0).sum() == 0:
            print('\nSomehow KMeans seems failed. Try again\n')
            if k == 2:          # deterministic: the retry fails identically, ~990 frames deep the interpreter gives up
                raise RecursionError('maximum recursion depth exceeded (weighted_kmeans retry, '
                                     'batch_spalign_kmeans.py:201-205, k = 2 repeats the same failure)')
            weighted_kmeans(superpixels, superpixel_features, superpixel_weights, k, n_superpixels_per_image,
                            nprandom, _depth + 1)
    return cl, cl == 0, dict(assign=assign, n_iter=it, status=status)


def batch_weighted_kmeans(args, superpixels, superpixel_features, superpixel_weights,
                          n_superpixels_per_image, nprandom=None):
    """:347-358 + :186-207 -> (clustering (B,H,W) uint8, road (B,H,W) bool, info)."""
    return weighted_kmeans(superpixels, superpixel_features, superpixel_weights, args.n_clusters,
                           n_superpixels_per_image, nprandom)


# --------------------------------------------------------------------------- felzenszwalb
def _fz_proto():
    L = lib()
    if not hasattr(L, '_fz_ready'):
        L.orc_fz_gauss_weights.restype = ctypes.c_int
        L.orc_fz_gauss_weights.argtypes = [_dbl, _P, ctypes.c_int]
        L.orc_fz_blur.restype = None
        L.orc_fz_blur.argtypes = [_P, _i64, _i64, _i64, _P, ctypes.c_int, _P]
        L.orc_fz_edges.restype = _i64
        L.orc_fz_edges.argtypes = [_P, _i64, _i64, _i64, _P, _P]
        L.orc_fz_segment.restype = _i64
        L.orc_fz_segment.argtypes = [_P, _P, _P, _i64, _i64, _dbl, _i64, _P]
        L.orc_felzenszwalb.restype = _i64
        L.orc_felzenszwalb.argtypes = [_P, _i64, _i64, _dbl, _dbl, _i64, _P]
        L._fz_ready = True
    return L


def fz_gauss_weights(sigma):
    w = np.zeros(64)
    r = _fz_proto().orc_fz_gauss_weights(sigma, w.ctypes.data, 64)
    return w[:2 * r + 1].copy(), r


def fz_blur(img_hwc, weights=None, sigma=0.8):
    """scipy.ndimage.gaussian_filter(img, sigma=[s, s, 0]) on an (H, W, C) float64 image."""
    img = _c(img_hwc, np.float64)
    H, W, C = img.shape
    if weights is None:
        weights, r = fz_gauss_weights(sigma)
    else:
        r = (len(weights) - 1) // 2
    w = _c(weights, np.float64)
    out = np.empty_like(img)
    _fz_proto().orc_fz_blur(img.ctypes.data, H, W, C, w.ctypes.data, r, out.ctypes.data)
    return out


def fz_edges(img_hwc):
    img = _c(img_hwc, np.float64)
    H, W, C = img.shape
    costs = np.empty(4 * H * W); edges = np.empty((4 * H * W, 2), np.int64)
    n = _fz_proto().orc_fz_edges(img.ctypes.data, H, W, C, costs.ctypes.data, edges.ctypes.data)
    return costs[:n].copy(), edges[:n].copy()


def fz_segment(costs, edges, order, npix, scale, min_size):
    c = _c(costs, np.float64); e = _c(edges, np.int64); o = _c(order, np.int64)
    labels = np.empty(npix, np.int64)
    nl = _fz_proto().orc_fz_segment(c.ctypes.data, e.ctypes.data, o.ctypes.data, len(o), npix, scale,
                                    min_size, labels.ctypes.data)
    return labels, int(nl)


def felzenszwalb(img_chw, scale=300.0, sigma=0.8, min_size=20):
    """felzenszwalb(img.transpose(1,2,0) / 255., scale, sigma, min_size) — batch_spalign_kmeans.py:303-307."""
    img = _c(img_chw, np.float32)
    _, H, W = img.shape
    out = np.empty((H, W), np.int64)
    nl = _fz_proto().orc_felzenszwalb(img.ctypes.data, H, W, scale, sigma, min_size, out.ctypes.data)
    if nl < 0:
        raise RuntimeError('orc_felzenszwalb failed')
    return out


# --------------------------------------------------------------------------- the paper's baselines
# (direct_clustering.py, superpixel_overlaps.py — SURVEY.md section 8f-4).  Pure numpy: the
# problems are small and every step is elementwise or a bincount.
def pixel_prior(h, w, y_rel_pos=0.75, x_rel_pos=0.5, y_rel_sigma=0.1, x_rel_sigma=0.2):
    """create_prior(h, w, ...) of direct_clustering.py:188-201 / superpixel_overlaps.py:194-207:
    the Gaussian location prior per feature pixel (integer centre, sigma relative to the map)."""
    y0, x0 = int(h * y_rel_pos), int(w * x_rel_pos)
    sy, sx = h * y_rel_sigma, w * x_rel_sigma
    ty = (np.arange(h)[:, None] - y0) ** 2 / (2 * sy) ** 2
    tx = (np.arange(w)[None, :] - x0) ** 2 / (2 * sx) ** 2
    return np.exp(-(ty + tx))


def pixel_matrix(fmap_nchw):
    """direct_clustering.py:299-306: one row per feature pixel of the batch, the C channels
    followed by the pixel's (x, y); float32 joined with int32 gives float64."""
    n, c, h, w = fmap_nchw.shape
    feats = np.asarray(fmap_nchw).transpose(0, 2, 3, 1).reshape(n * h * w, c)
    yy, xx = np.divmod(np.arange(h * w), w)
    xy = np.tile(np.stack([xx, yy], 1), (n, 1)).astype(np.int32)
    return np.concatenate([feats, xy], axis=1)


def direct_clustering(fmap_nchw, k, y_rel_pos=0.75, x_rel_pos=0.5, y_rel_sigma=0.1, x_rel_sigma=0.1,
                      nprandom=None):
    """estimate_road_mask of direct_clustering.py:286-322 up to the masks: weighted k-means over
    all feature pixels of the batch.  -> (cluster ids (n,h,w) int32, road (n,h,w) bool)."""
    n, c, h, w = fmap_nchw.shape
    prior = np.tile(pixel_prior(h, w, y_rel_pos, x_rel_pos, y_rel_sigma, x_rel_sigma).reshape(-1), n)
    assign, _, _ = kmeans(k, pixel_matrix(fmap_nchw), prior, nprandom=nprandom)
    cl = assign.reshape(n, h, w)
    return cl, cl == 0


def overlap_refine(road_mask, superpixel, threshold):
    """superpixel_overlaps.py:353-361: a superpixel becomes road when it holds more than
    `threshold` of all predicted road pixels."""
    road = np.asarray(road_mask).astype(bool)
    sp = np.asarray(superpixel)
    n_road = float(road.sum())
    out = np.zeros(sp.shape, np.uint8)
    if n_road > 0:
        counts = np.bincount(sp[road].ravel(), minlength=int(sp.max()) + 1)
        out[(counts / n_road > threshold)[sp]] = 1
    return out


def felzenszwalb_u8(img_chw_u8, scale=500.0, sigma=0.9, min_size=20):
    """felzenszwalb(img.transpose(1,2,0) / 255., ...) on a uint8 image (superpixel_overlaps.py:294-300):
    the division happens in float64; equal costs in edge-index order (stable sort), as everywhere."""
    img = np.asarray(img_chw_u8).transpose(1, 2, 0).astype(np.float64) / 255.
    H, W, _ = img.shape
    costs, edges = fz_edges(fz_blur(img, sigma=sigma))
    order = np.argsort(costs, kind='stable')
    labels, _ = fz_segment(costs, edges, order, H * W, float(scale) / 255., min_size)
    return labels.reshape(H, W)


# --------------------------------------------------------------------------- input stage (8f-2)
def resize_bicubic_u8(img_chw_u8, shape):
    """PIL.Image.fromarray(channel).resize((w, h), PIL.Image.BICUBIC) per channel of a uint8 CHW image:
    what datasets/resize_image_dataset.py:31-34 computes when Pillow does the resize (8-bit path)."""
    a = np.ascontiguousarray(img_chw_u8, dtype=np.uint8)
    C, H, W = a.shape
    h, w = int(shape[0]), int(shape[1])
    out = np.empty((C, h, w), np.uint8)
    L = lib()
    L.orc_resize_bicubic_u8.restype = None
    L.orc_resize_bicubic_u8.argtypes = [_P, ctypes.c_int, ctypes.c_int, _P, ctypes.c_int, ctypes.c_int]
    for c in range(C):
        L.orc_resize_bicubic_u8(a[c].ctypes.data, H, W, out[c].ctypes.data, h, w)
    return out


def resize_cvcubic_u8(img_chw, shape):
    """PARITY UNPINNED: cv2.resize(img.transpose(1, 2, 0), (w, h), interpolation=cv2.INTER_CUBIC).transpose(2, 0, 1) on a
    UINT8 CHW image — the OpenCV branch of chainercv.transforms.resize(img, shape, 3) as the reference's datasets call it
    (datasets/resize_image_dataset.py:20-36: resize first, astype(float32) after), OpenCV's 8-bit fixed-point path restated
    from its published algorithm (resize_oracle.c).  Returns uint8 CHW."""
    a = np.ascontiguousarray(np.asarray(img_chw, dtype=np.uint8).transpose(1, 2, 0))
    H, W, C = a.shape
    h, w = int(shape[0]), int(shape[1])
    out = np.empty((h, w, C), np.uint8)
    L = lib()
    L.orc_resize_cvcubic_u8.restype = None
    L.orc_resize_cvcubic_u8.argtypes = [_P, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P, ctypes.c_int, ctypes.c_int]
    L.orc_resize_cvcubic_u8(a.ctypes.data, H, W, C, out.ctypes.data, h, w)
    return np.ascontiguousarray(out.transpose(2, 0, 1))
