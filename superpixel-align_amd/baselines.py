"""The paper's two baselines on the same kernels.

    direct_clustering.py    — weighted k-means over ALL feature pixels of the batch
                              (estimate_road_mask, direct_clustering.py:286-322)
    superpixel_overlaps.py  — the same clustering, then every superpixel of the ORIGINAL image
                              that holds more than --overlap_threshold of the predicted road
                              pixels becomes road (superpixel_overlaps.py:309-361)

Both reuse libspalign: `spa_kmeans_weighted` (N = B*fh*fw points, D = C+2), `spa_felzenszwalb_u8`
(the baselines segment the uint8 image, so /255. is a float64 division) and `spa_overlap_refine`.
PyTorch only moves data: the (N, C+2) float64 matrix is a layout change of the channels-last DRN
map plus two coordinate columns.  There is no CPU fallback.
"""
import time

import numpy as np
import torch

from .engine import NpRandom, default_engine


def pixel_prior(h, w, y_rel_pos=0.75, x_rel_pos=0.5, y_rel_sigma=0.1, x_rel_sigma=0.2):
    """create_prior(h, w, ...) (direct_clustering.py:188-201): Gaussian location prior per feature
    pixel.  fh*fw values: evaluated on the host in float64 exactly as numpy does."""
    cy, cx = int(h * y_rel_pos), int(w * x_rel_pos)
    sy, sx = h * y_rel_sigma, w * x_rel_sigma
    row = (np.arange(h) - cy) ** 2 / (2 * sy) ** 2
    col = (np.arange(w) - cx) ** 2 / (2 * sx) ** 2
    return np.exp(-(row[:, None] + col[None, :]))


def pixel_matrix(fmap):
    """(n,C,h,w) map on the GPU -> (n*h*w, C+2) float64: channels, then x, then y
    (direct_clustering.py:299-306; float32 joined with int32 coordinates is float64)."""
    n, c, h, w = fmap.shape
    X = torch.empty((n * h * w, c + 2), dtype=torch.float64, device=fmap.device)
    X[:, :c] = fmap.permute(0, 2, 3, 1).reshape(n * h * w, c)
    ys, xs = torch.meshgrid(torch.arange(h, device=fmap.device), torch.arange(w, device=fmap.device),
                            indexing='ij')
    X[:, c] = xs.reshape(-1).repeat(n)
    X[:, c + 1] = ys.reshape(-1).repeat(n)
    return X


class DirectClustering(object):
    """estimate_road_mask of direct_clustering.py as a batch pipeline with LabelPipeline's
    interface (run / elapsed_times), so the same command-line driver serves all three scripts."""

    MAX_POINTS = 200000      # the kernel ranks the prior weights in O(N^2): fine for the 224x224 operating point

    def __init__(self, args, model, eng=None, nprandom=None):
        self.args, self.model = args, model
        self.eng = eng or default_engine()
        self.nprandom = nprandom or NpRandom(1111)
        self._t = {}

    def features(self, imgs):
        _, maps = self.model.batch_predict(imgs)
        use = [maps[i] for i in self.args.use_feature_maps]
        return use[0] if len(use) == 1 else torch.cat(use, 1)

    def cluster(self, fmap):
        a, eng = self.args, self.eng
        n, c, h, w = fmap.shape
        N = n * h * w
        if N > self.MAX_POINTS:
            raise ValueError('direct clustering of %d feature pixels: above the supported %d '
                             '(use the reference operating point --resize_shape 224 224)' % (N, self.MAX_POINTS))
        X = pixel_matrix(fmap)
        prior_h = np.tile(pixel_prior(h, w, a.y_rel_pos, a.x_rel_pos, a.y_rel_sigma, a.x_rel_sigma).reshape(-1), n)
        prior = torch.from_numpy(prior_h).to(fmap.device)
        k = a.n_clusters
        init_other = None
        if k > 2:                                            # kmeans() :131-136, numpy's global RNG
            thr = np.sort(prior_h)[N // 2]
            idx = (np.arange(int((prior_h <= thr).sum())) % (k - 1) + 1).astype(np.int64)
            self.nprandom.shuffle(idx)
            init_other = torch.from_numpy(idx).to(fmap.device)
        n_ptr = torch.tensor([N], dtype=torch.int32, device=fmap.device)
        assign, info = eng.kmeans(X, prior, n_ptr, k, 1000, init_other)
        eng.raise_on_status()
        return assign.view(n, h, w), info

    def run(self, imgs):
        dev = self.eng.device
        t0 = time.time()
        imgs = torch.as_tensor(imgs).to(dev)
        fmap = self.features(imgs)
        torch.cuda.synchronize()
        t1 = time.time()
        cl, info = self.cluster(fmap.float() if fmap.dtype != torch.float32 else fmap)
        torch.cuda.synchronize()
        t2 = time.time()
        self._t = {'time_feature_maps': t1 - t0, 'time_prior': 0.0, 'time_kmeans': t2 - t1}
        return BaselineResult(cl.to(torch.uint8), (cl == 0).to(torch.uint8), info,
                              torch.zeros((imgs.shape[0],), dtype=torch.int32))

    def elapsed_times(self):
        return dict(self._t)


class SuperpixelOverlaps(DirectClustering):
    """superpixel_overlaps.py: direct clustering, then refinement by the superpixels of the
    original (uint8, full size) image."""

    def superpixels(self, orig_u8):
        a, eng = self.args, self.eng
        rgb = torch.as_tensor(orig_u8).to(eng.device).float().contiguous()
        if a.superpixel_method == 'felzenszwalb':
            return eng.felzenszwalb(rgb, a.felzenszwalb_scale, a.felzenszwalb_sigma, a.felzenszwalb_min_size,
                                    uint8_image=True)
        # scikit-image runs its float64 SLIC core on a uint8 image (superpixel_overlaps.py:301-304)
        return eng.slic_u8(rgb, a.n_slic_segments)

    def refine(self, road, labels, n_labels):
        """road (n,h,w) u8 at map size, labels (n,H,W): nearest-neighbour resize of the mask to the
        superpixel shape (cv.INTER_NEAREST, :355-357), then the overlap rule."""
        n, H, W = labels.shape
        if road.shape[1:] != (H, W):
            h, w = road.shape[1:]
            yi = torch.clamp((torch.arange(H, device=road.device) * (h / float(H))).floor().long(), max=h - 1)
            xi = torch.clamp((torch.arange(W, device=road.device) * (w / float(W))).floor().long(), max=w - 1)
            road = road[:, yi][:, :, xi].contiguous()
        out = self.eng.overlap_refine(labels, road, int(n_labels.max()), self.args.overlap_threshold)
        self.eng.raise_on_status()
        return out

    def run(self, imgs, orig_u8=None):
        res = DirectClustering.run(self, imgs)
        t0 = time.time()
        labels, n_labels = self.superpixels(orig_u8 if orig_u8 is not None else imgs)
        torch.cuda.synchronize()
        self._t['time_superpixel'] = time.time() - t0
        res.road = self.refine(res.road, labels, n_labels)
        res.n_labels = n_labels
        return res


class BaselineResult(object):
    def __init__(self, cluster, road, info, n_labels):
        self.cluster, self.road, self.info, self.n_labels = cluster, road, info, n_labels

    def masks_to_host(self):
        return self.cluster.cpu().numpy(), self.road.cpu().numpy()
