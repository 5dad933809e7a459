"""The per-batch label-generation pipeline, device resident from the image batch to the masks.

Mirrors estimate_road_mask() of the reference (batch_spalign_kmeans.py:427-458 and
utils/apply_spalign_kmeans.py:26-57): DRN features -> superpixels -> superpixel align ->
location prior -> weighted k-means -> painted masks, in that order, with the same timer keys.
Where the reference crosses the host/device boundary five times per batch, this
path uploads the images once and downloads the two uint8 masks once; in anchor mode the
superpixel sizes additionally visit the host, because the anchors are drawn from the CPython
`random` stream exactly as the reference draws them.
"""
import numpy as np
import torch

from . import _lib
from .engine import Engine, NpRandom, PyRandom


class BatchResult(object):
    """Device tensors of one batch (+ lazily fetched host copies)."""

    def __init__(self, **kw):
        self.__dict__.update(kw)

    def masks_to_host(self):
        """(clustering (B,H,W) uint8, road (B,H,W) uint8) as numpy — one D2H each."""
        return self.cluster.cpu().numpy(), self.road.cpu().numpy()


class LabelPipeline(object):
    def __init__(self, args, model=None, engine=None, pool_mode=None, mean_sampling=None):
        self.args = args
        self.model = model
        self.eng = engine or Engine()
        self.pool_mode = pool_mode or getattr(args, 'pool_mode', 'anchor')
        self.mean_sampling = mean_sampling or getattr(args, 'mean_sampling', 'nearest')
        # the reference seeds both generators once per process (batch_spalign_kmeans.py:33-34);
        # their state carries over from batch to batch
        self.pyrandom = PyRandom(getattr(args, 'seed', 1111))
        self.nprandom = NpRandom(getattr(args, 'seed', 1111))
        self._ev = {}

    # ---------------------------------------------------------------- stages
    def features(self, imgs_dev):
        """model.batch_predict + F.concat(use_maps) (:431-435) -> (B, C, fh, fw), channels-last."""
        _, maps = self.model.batch_predict(imgs_dev, getattr(self.args, 'drn_sub_batch', None))
        use = [maps[i] for i in self.args.use_feature_maps]
        if len(use) == 1:
            return use[0]
        return torch.cat(use, dim=1).contiguous(memory_format=torch.channels_last)

    def superpixels(self, imgs_dev):
        """batch_superpixel (:299-313) -> labels (B,H,W) i32, n_labels (B) i32 on the device."""
        a = self.args
        if a.superpixel_method != 'slic':
            raise NotImplementedError(
                "superpixel_method=%r: only 'slic' runs on the MI355X path so far "
                '(felzenszwalb is the next row of the scope table, SURVEY.md 8f)' % a.superpixel_method)
        return self.eng.slic(imgs_dev, a.n_slic_segments)

    def capacity(self, B, H, W):
        if self.args.superpixel_method == 'slic':
            return B * _lib.make_plan(H, W, self.args.n_slic_segments).max_labels
        return B * H * W

    def describe(self, imgs_shape, labels, n_labels, fmap):
        """batch_superpixel_align (:316-330) + batch_create_prior (:333-344) on the device.
        -> offsets, count, X (Ncap, D), prior (Ncap)"""
        a, eng = self.args, self.eng
        B, _, H, W = imgs_shape
        ncap = self.capacity(B, H, W)
        off = eng.segment_offsets(n_labels)
        append_pos = not a.without_pos
        count, centroid, prior = eng.segment_stats(
            labels, off, ncap, (a.y_rel_pos, a.x_rel_pos, a.y_rel_sigma, a.x_rel_sigma),
            want_centroid=True)
        if self.pool_mode == 'mean':
            X = eng.pool_mean(fmap, labels, off, ncap, count, self.mean_sampling,
                              centroid if append_pos else None, append_pos)
        elif self.pool_mode == 'anchor':
            n = int(off[-1].item())                      # sizes visit the host for the RNG
            cnt_h = count[:n].cpu().numpy()
            ranks_h, nvalid_h = self.pyrandom.shuffle_select(cnt_h, a.n_anchors)
            ranks = torch.zeros((ncap, a.n_anchors), dtype=torch.int32, device=labels.device)
            nvalid = torch.zeros((ncap,), dtype=torch.int32, device=labels.device)
            ranks[:n] = torch.from_numpy(ranks_h).to(labels.device, non_blocking=True)
            nvalid[:n] = torch.from_numpy(nvalid_h).to(labels.device, non_blocking=True)
            anchors = eng.select_anchor_pixels(labels, off, ncap, ranks, nvalid)
            X = eng.pool_anchor(fmap, H, off, ncap, anchors, nvalid, a.n_neighbors,
                                centroid if append_pos else None, append_pos)
        else:
            raise ValueError('pool_mode must be anchor or mean')
        return off, count, X, prior

    def cluster(self, labels, off, X, prior):
        """batch_weighted_kmeans (:347-358) -> assign, info, cluster map, road mask (device)."""
        a, eng = self.args, self.eng
        B = labels.shape[0]
        init_other = None
        if a.n_clusters > 2:
            # idx = arange(M) % (k-1) + 1 shuffled by numpy's global generator (:147-149)
            n = int(off[-1].item())
            w = prior[:n].cpu().numpy()
            thr = np.sort(w)[n // 2]
            m = int((w <= thr).sum())
            idx = (np.arange(m) % (a.n_clusters - 1) + 1).astype(np.int64)
            self.nprandom.shuffle(idx)
            init_other = torch.from_numpy(idx).to(labels.device)
        assign, info = eng.kmeans(X, prior, off[B:], a.n_clusters, 1000, init_other)
        cluster, road = eng.paint(labels, assign, off)
        return assign, info, cluster, road

    # ---------------------------------------------------------------- whole batch
    def _tick(self, name):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        self._ev[name] = ev

    def run(self, imgs, check_status=True):
        """imgs: (B,3,H,W) float32 RGB 0..255, numpy (pinned or not) or CUDA tensor."""
        imgs_dev = torch.as_tensor(imgs)
        if not imgs_dev.is_cuda:
            imgs_dev = imgs_dev.to(self.eng.device, non_blocking=True)
        imgs_dev = imgs_dev.float().contiguous()
        self._tick('start')
        fmap = self.features(imgs_dev)
        self._tick('features')
        labels, n_labels = self.superpixels(imgs_dev)
        self._tick('superpixel')
        off, count, X, prior = self.describe(imgs_dev.shape, labels, n_labels, fmap)
        self._tick('describe')
        assign, info, cluster, road = self.cluster(labels, off, X, prior)
        self._tick('kmeans')
        if check_status:
            self.eng.raise_on_status()
        return BatchResult(labels=labels, n_labels=n_labels, offsets=off, count=count, X=X,
                           prior=prior, assign=assign, info=info, cluster=cluster, road=road,
                           fmap=fmap)

    def elapsed_times(self):
        """Stage times of the last run() in seconds, under the reference's result.json keys
        (:428-458) plus time_feature_maps (the baselines' key, direct_clustering.py:292-294).
        Measured with device events; prior is computed inside the descriptor pass, so its share
        is reported under time_roialign and time_prior is 0."""
        torch.cuda.synchronize()
        e = self._ev
        ms = lambda a, b: e[a].elapsed_time(e[b]) / 1000.0
        return {'time_feature_maps': ms('start', 'features'),
                'time_superpixel': ms('features', 'superpixel'),
                'time_roialign': ms('superpixel', 'describe'),
                'time_prior': 0.0,
                'time_kmeans': ms('describe', 'kmeans')}
