"""The per-batch label-generation pipeline, device resident from the image batch to the masks.

Mirrors estimate_road_mask() of the reference (batch_spalign_kmeans.py:427-458 and
utils/apply_spalign_kmeans.py:26-57): DRN features -> superpixels -> superpixel align ->
location prior -> weighted k-means -> painted masks, with the same timer keys.
Where the reference crosses the host/device boundary five times per batch, this
path uploads the images once and downloads the two uint8 masks once; in anchor mode the
superpixel sizes additionally visit the host, because the anchors are drawn from the CPython
`random` stream exactly as the reference draws them.

Two HIP streams: the superpixel branch (SLIC, per-segment statistics, and in anchor mode the
host-side random draws) does not depend on the DRN features, so it runs on an auxiliary stream
while the DRN forward (libspalign's MFMA kernels) occupies the main stream; they join before pooling.
"""
import os

import numpy as np
import torch

from . import _lib
from .engine import Engine, default_engine, NpRandom, PyRandom


class BatchResult(object):
    """Device tensors of one batch (+ lazily fetched host copies)."""

    def __init__(self, **kw):
        self.__dict__.update(kw)

    def masks_to_host(self):
        """(clustering (B,H,W) uint8, road (B,H,W) uint8) as numpy — one D2H each."""
        self.check_retry()
        return self.cluster.cpu().numpy(), self.road.cpu().numpy()

    def check_retry(self, flags=None, strict=None):
        """k = 2 only (flags: a host copy of `retry_fail` the caller downloaded itself).  An image without a
        cluster-0 pixel sends the reference into its retry (:201-205), which for k = 2 repeats the same
        deterministic failure until the interpreter raises RecursionError.  strict (args.strict_retry /
        SPA_STRICT_RETRY=1): raise RecursionError here as well; default: print the reference's message once per
        such image and keep the batch — a run of 20 k images is not lost to one image without road.  Checked when
        the results are fetched: the flags are device-side and the batch loop stays asynchronous."""
        info = getattr(self, 'retry_info', None)
        if info is not None and flags is None:
            # k > 2 with the generator on the device: [runs still pending, retry runs made] (LabelPipeline.cluster).  Runs still
            # pending after the speculative rounds are made up for by the pipeline — before the next batch draws (its cluster()
            # settles the previous batch first) or, for the last batch of a run, right here.
            self.retry_info = None
            settle = getattr(self, 'retry_settle', None)
            if settle is not None:
                pending, made = settle(info)
            else:
                pending, made = (int(v) for v in info.cpu().tolist())
            for _ in range(made):
                print('\nSomehow KMeans seems failed. Try again\n')
            if pending > 0:
                raise _lib.SpalignError('weighted_kmeans retries (batch_spalign_kmeans.py:201-205): %d run(s) still due and a later batch '
                                        'has already drawn from numpy\'s stream' % pending)
            return
        fail = getattr(self, 'retry_fail', None) if flags is None else flags
        if fail is None or not bool(fail.any().item()):
            return
        bad = fail.nonzero().flatten().tolist()
        if strict is None:
            strict = getattr(self, 'strict_retry', False) or os.environ.get('SPA_STRICT_RETRY') == '1'
        for _ in bad:
            print('\nSomehow KMeans seems failed. Try again\n')
        if strict:
            raise RecursionError(RETRY_MESSAGE + ' (image(s) %s of the batch have no cluster-0 pixel; k = 2 repeats '
                                 'the same failure)' % bad)


RETRY_MESSAGE = 'maximum recursion depth exceeded: weighted_kmeans retry, batch_spalign_kmeans.py:201-205'
RETRY_DEPTH_LIMIT = 990        # CPython's default recursion limit minus the frames below weighted_kmeans


def images_without_cluster0(assign, off, B):
    """(B,) bool on the device: image b has no superpixel in cluster 0 (labels are dense, so no pixel either)."""
    z = torch.cumsum((assign == 0).to(torch.int32), 0)
    c = torch.cat([z.new_zeros(1), z])
    o = off[:B + 1].long()
    return (c[o[1:]] - c[o[:-1]]) == 0


class LabelPipeline(object):
    def __init__(self, args, model=None, engine=None, pool_mode=None, mean_sampling=None,
                 overlap=True):
        self.args = args
        self.model = model
        self.eng = engine or default_engine()     # one spa_ctx per process: the DRN glue kernels use it too
        self.pool_mode = pool_mode or getattr(args, 'pool_mode', 'anchor')
        self.mean_sampling = mean_sampling or getattr(args, 'mean_sampling', 'nearest')
        # the reference seeds both generators once per process (batch_spalign_kmeans.py:33-34);
        # their state carries over from batch to batch
        self.pyrandom = PyRandom(getattr(args, 'seed', 1111))
        self.nprandom = NpRandom(getattr(args, 'seed', 1111))
        # k > 2: the numpy stream moves to the device at the first batch (cluster()); the host form of round 4 on request
        self.host_kmeans_init = bool(getattr(args, 'host_kmeans_init', False)) or os.environ.get('SPA_KM_HOST_INIT') == '1'
        self.retry_rounds = int(os.environ.get('SPA_RETRY_ROUNDS', '2'))
        self._np_state = None
        self._retry_info = None
        if os.environ.get('SPA_PIPE_OVERLAP') in ('0', '1'):
            overlap = os.environ['SPA_PIPE_OVERLAP'] == '1'
        # (the second stream's queue priority, SPA_AUX_PRIORITY: see HISTORY.md section 5 for the A/B)
        prio = int(os.environ.get('SPA_AUX_PRIORITY', '0'))
        self.aux = torch.cuda.Stream(device=self.eng.device, priority=prio) if overlap else None
        # SPA_PIPE_TAIL_AUX=1 (mean pooling on two streams): everything but the DRN forward — superpixels, segment statistics
        # AND the batch's tail (pooling, k-means, paint) — runs on the auxiliary stream, so the next batch's forward starts on the
        # main stream while this batch's tail finishes (run(join=False)).  Built, bit-identical, and measured in same-box A/B runs
        # at 400 against 403 images/s with the tail on the main stream: the tail's latency-bound kernels and the forward's front
        # stretch each other by what the overlap gains, so the default keeps the tail on the main stream.
        self.tail_on_aux = self.aux is not None and os.environ.get('SPA_PIPE_TAIL_AUX', '0') == '1'
        self._ev = {}
        # anchor mode, device_rng: the CPython `random` stream lives on the device (seeded like the reference's
        # module scope), its outputs produced a batch ahead on a side stream, and no superpixel size visits the
        # host (spa_anchor_ranks_dev).  Off by default: the rejection sampling is one sequential pass over
        # ~1.4 stream outputs per pixel whatever runs it — 296 ms per 30 full-size images as one workgroup
        # against 117 ms on a host core hidden under the DRN forward (HISTORY.md section 5)
        self.device_rng = bool(getattr(args, 'device_rng', False))
        self._rng_ready = False
        self._gen_stream = None
        self._gen_done = None

    def reseed(self, seed=None):
        """Both host generators back to their state at process start (batch_spalign_kmeans.py:33-34): a run that follows is
        comparable bit for bit with any other run from the same state (anchor draws, the k > 2 initial assignment)."""
        if self.device_rng and self._rng_ready:
            raise ValueError('reseed: the generator state lives on the device (device_rng)')
        seed = getattr(self.args, 'seed', 1111) if seed is None else seed
        self.pyrandom = PyRandom(seed)
        self.nprandom = NpRandom(seed)
        self._np_state = None                            # (the device copy is re-created from the fresh host state)
        self._np_on_device = False
        self._owed = None

    # ---------------------------------------------------------------- stages
    def features(self, imgs_dev):
        """model.batch_predict + F.concat(use_maps) (:431-435) -> (B, C, fh, fw), channels-last."""
        _, maps = self.model.batch_predict(imgs_dev, getattr(self.args, 'drn_sub_batch', None),
                                           need=self.args.use_feature_maps,
                                           streams=getattr(self.args, 'drn_streams', 1))
        use = [maps[i] for i in self.args.use_feature_maps]
        if len(use) == 1:
            return use[0]
        return torch.cat(use, dim=1).contiguous(memory_format=torch.channels_last)

    def superpixels(self, imgs_dev):
        """batch_superpixel (:299-313) -> labels (B,H,W) i32, n_labels (B) i32 on the device."""
        a = self.args
        if a.superpixel_method == 'slic':
            return self.eng.slic(imgs_dev, a.n_slic_segments)
        if a.superpixel_method == 'felzenszwalb':
            return self.eng.felzenszwalb(imgs_dev, a.felzenszwalb_scale, a.felzenszwalb_sigma,
                                         a.felzenszwalb_min_size)
        raise ValueError('unknown superpixel_method %r' % a.superpixel_method)

    def capacity(self, B, H, W, n_labels):
        """Rows to allocate for the descriptor matrix.  SLIC has a static bound; felzenszwalb's
        segment count is data dependent (1 .. H*W/min_size), so the exact total is read back."""
        if self.args.superpixel_method == 'slic':
            return B * _lib.make_plan(H, W, self.args.n_slic_segments).max_labels
        return max(1, int(n_labels.sum().item()))

    def segments(self, imgs_shape, labels, n_labels):
        """Everything of batch_superpixel_align (:316-330) + batch_create_prior (:333-344) that
        needs only the label maps: offsets, sizes, centres of mass, prior, and in anchor mode the
        anchor pixels (random.shuffle stream of the reference, :231-234)."""
        return self.segments_anchors(imgs_shape, labels, self.segments_stats(imgs_shape, labels, n_labels))

    def segments_stats(self, imgs_shape, labels, n_labels):
        """the part of `segments` that only enqueues: offsets, sizes, centres of mass, prior"""
        a, eng = self.args, self.eng
        B, _, H, W = imgs_shape
        ncap = self.capacity(B, H, W, n_labels)
        off = eng.segment_offsets(n_labels)
        count, centroid, prior = eng.segment_stats(
            labels, off, ncap, (a.y_rel_pos, a.x_rel_pos, a.y_rel_sigma, a.x_rel_sigma),
            want_centroid=True)
        return dict(ncap=ncap, off=off, count=count, centroid=centroid, prior=prior, anchors=None, nvalid=None)

    def segments_anchors(self, imgs_shape, labels, seg):
        """the anchor pixels of `segments` (anchor mode; with the host generator the calling thread waits for the sizes and
        draws: ~47 ms per 30 full-size images)"""
        a, eng = self.args, self.eng
        B, _, H, W = imgs_shape
        ncap, off, count = seg['ncap'], seg['off'], seg['count']
        anchors = nvalid = None
        if self.pool_mode == 'anchor' and self.device_rng and int(1.7 * B * H * W) + (1 << 21) > (1 << 27) - 4096:
            # the shuffles of one batch consume ~1.4-1.5 generator outputs per pixel; the device ring holds 2^27.
            # A batch beyond ~75 M pixels therefore draws on the host (same stream, bit for bit) — possible only
            # while the generator state has not moved to the device yet
            if self._rng_ready:
                raise ValueError('device_rng: a batch of %d pixels needs more generator outputs than the device ring '
                                 'holds and the generator state already lives on the device; use smaller batches or '
                                 'the host generator from the start' % (B * H * W))
            import warnings
            warnings.warn('device_rng: batch of %d pixels exceeds the device ring; drawing the anchors on the host'
                          % (B * H * W))
            self.device_rng = False
        if self.pool_mode == 'anchor' and self.device_rng:
            # no superpixel size visits the host: rejection sampling of every shuffle swap and the first
            # n_anchors places of every shuffled list on the device (spa_anchor_ranks_dev)
            cur = torch.cuda.current_stream(eng.device)
            want = min(int(1.7 * B * H * W) + (1 << 21), (1 << 27) - 4096)
            if not self._rng_ready:
                eng.pyrandom_seed(getattr(a, 'seed', 1111))
                eng.pyrandom_generate(want)
                self._gen_stream = torch.cuda.Stream(device=eng.device, priority=0)
                self._rng_ready = True
            elif self._gen_done is not None:
                cur.wait_event(self._gen_done)
            ranks, nvalid = eng.anchor_ranks(count, off[B:], ncap, a.n_anchors, B * H * W)
            anchors = eng.select_anchor_pixels(labels, off, ncap, ranks, nvalid)
            # top the ring up for the next batch in the background
            used = torch.cuda.Event()
            used.record(cur)
            self._gen_stream.wait_event(used)
            with torch.cuda.stream(self._gen_stream):
                eng.pyrandom_generate(want)
                self._gen_done = torch.cuda.Event()
                self._gen_done.record(self._gen_stream)
        elif self.pool_mode == 'anchor':
            n = int(off[-1].item())                      # sizes visit the host for the RNG
            cnt_h = count[:n].cpu().numpy()
            ranks_h, nvalid_h = self.pyrandom.shuffle_select(cnt_h, a.n_anchors)
            ranks = torch.zeros((ncap, a.n_anchors), dtype=torch.int32, device=labels.device)
            nvalid = torch.zeros((ncap,), dtype=torch.int32, device=labels.device)
            ranks[:n] = torch.from_numpy(ranks_h).to(labels.device, non_blocking=True)
            nvalid[:n] = torch.from_numpy(nvalid_h).to(labels.device, non_blocking=True)
            anchors = eng.select_anchor_pixels(labels, off, ncap, ranks, nvalid)
        elif self.pool_mode != 'mean':
            raise ValueError('pool_mode must be anchor or mean')
        seg['anchors'], seg['nvalid'] = anchors, nvalid
        return seg

    def pool(self, imgs_shape, labels, seg, fmap):
        """The part of batch_superpixel_align that reads the feature map -> X (Ncap, D)."""
        a, eng = self.args, self.eng
        append_pos = not a.without_pos
        cen = seg['centroid'] if append_pos else None
        if self.pool_mode == 'mean':
            return eng.pool_mean(fmap, labels, seg['off'], seg['ncap'], seg['count'],
                                 self.mean_sampling, cen, append_pos)
        return eng.pool_anchor(fmap, imgs_shape[2], seg['off'], seg['ncap'], seg['anchors'],
                               seg['nvalid'], a.n_neighbors, cen, append_pos)

    def cluster(self, labels, off, X, prior, _depth=0):
        """batch_weighted_kmeans (:347-358) -> assign, info, cluster map, road mask, retry flags (device).

        The reference's weighted_kmeans re-runs itself, result discarded, for every image that ends without a
        cluster-0 pixel (:201-205).  k > 2: each run shuffles the initial assignment with numpy's global
        generator, so the retries are executed here for their effect on the stream later batches draw from.
        k = 2: see BatchResult.check_retry.

        k > 2 WITHOUT a host round trip (round 5; `--host_kmeans_init` / SPA_KM_HOST_INIT=1 keeps round 4's synchronous form):
        numpy's generator state lives in device memory and the initial assignment — threshold, the number m of points at or
        below it (known only on the device), arange(m) % (k - 1) + 1, np.random.shuffle — is drawn by one workgroup
        (Engine.np_kmeans_init, csrc/spa_nprng.hip).  Retries: every run of the reference consumes ONE shuffle when it starts
        and runs are sequential (the recursion is depth first and a run draws before it recurses), so the stream only depends
        on HOW MANY runs there are: pending = failures of the first run; while pending: run again, pending += its failures - 1.
        SPA_RETRY_ROUNDS (default 2) such runs are enqueued speculatively behind a device-side gate — a gated run that is not
        wanted draws nothing and does nothing — and the counters go to `retry_info` = [runs still pending, runs made].  Runs still
        pending after those rounds (three or more failing images, or an image that keeps failing) are made up for by
        _settle_retries() BEFORE the next batch draws — one host wait for the previous batch's counters per batch, the reference's
        recursion limit as the bound — or when the last batch is fetched (BatchResult.check_retry, which prints the reference's
        message per run).  A batch with more points than the device initialisation holds (felzenszwalb: data dependent) draws on
        the host from the downloaded state and hands the stream back to the device."""
        a, eng = self.args, self.eng
        B = labels.shape[0]
        init_other = None
        if _depth == 0:
            self._retry_info = None
        on_device = (a.n_clusters > 2 and not self.host_kmeans_init and X.shape[0] <= getattr(self, 'np_init_max', eng.NP_INIT_MAX))
        if a.n_clusters > 2 and not self.host_kmeans_init and _depth == 0:
            # runs the previous batch still owes (more failures than the speculative rounds covered) come BEFORE this batch's draw
            self._settle_retries()
        if on_device:
            if self._np_state is None:
                self._np_state = torch.from_numpy(self.nprandom.state().view(np.int32)).to(labels.device)
            self._np_on_device = True                    # the stream lives on the device from here on
            init_other = eng.np_kmeans_init(self._np_state, prior, off[B:], a.n_clusters)
        elif a.n_clusters > 2:
            if getattr(self, '_np_on_device', False) and _depth == 0:
                # more points than the device initialisation holds (felzenszwalb: data dependent): the stream comes back to the
                # host for this batch — synchronously — and returns to the device afterwards
                self.nprandom.set_state(self._np_state.cpu().numpy().view(np.uint32))
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
        if on_device:
            counters = torch.zeros((2,), dtype=torch.int32, device=labels.device)       # [runs pending, retry runs made]
            eng.retry_update(assign, off, B, counters)
            for _ in range(self.retry_rounds):
                gate = counters[0:1].clone()              # the round runs iff a retry is pending now
                init_r = eng.np_kmeans_init(self._np_state, prior, off[B:], a.n_clusters, gate=gate)
                assign_r, _ = eng.kmeans(X, prior, off[B:], a.n_clusters, 1000, init_r, gate=gate)
                eng.retry_update(assign_r, off, B, counters, gate=gate)
            self._retry_info = counters
            # what a later settlement needs: the batch's operands and an event behind its last speculative round
            ev = torch.cuda.Event()
            ev.record()
            self._owed = dict(counters=counters, X=X, prior=prior, off=off, B=B, k=a.n_clusters, event=ev, settled=None)
            return assign, info, cluster, road, None
        fail = images_without_cluster0(assign, off, B)
        if a.n_clusters > 2:
            for b in fail.cpu().numpy().nonzero()[0]:
                print('\nSomehow KMeans seems failed. Try again\n')
                if _depth >= RETRY_DEPTH_LIMIT:
                    raise RecursionError(RETRY_MESSAGE)
                self.cluster(labels, off, X, prior, _depth + 1)        # discarded, as in the reference
            fail = None
            if _depth == 0 and getattr(self, '_np_on_device', False) and not self.host_kmeans_init:
                # (an oversize batch drawn on the host: the stream goes back to the device for the batches that follow)
                self._np_state.copy_(torch.from_numpy(self.nprandom.state().view(np.int32)))
        return assign, info, cluster, road, fail

    def _settle_retries(self, info=None):
        """k > 2, generator on the device: make the retry runs a batch still owes (batch_spalign_kmeans.py:201-205: the reference
        recurses until no image fails, depth first — only the NUMBER of runs matters to the stream, see cluster()).  The speculative
        rounds cover SPA_RETRY_ROUNDS of them without a host round trip; here the host reads the counters (one wait for the
        batch's event — it precedes everything enqueued since) and enqueues run after run until none is pending, up to the
        reference's recursion limit.  Called before the next batch draws, so numpy's stream stays the reference's; also by
        BatchResult.check_retry (the last batch of a run).  -> (pending, made) of the settled batch, or None if nothing is owed."""
        owed = getattr(self, '_owed', None)
        if owed is None or (info is not None and info is not owed['counters']):
            if info is not None:
                # a batch that is no longer the latest: whatever it owed was settled when its successor drew
                done = getattr(self, '_settled', {}).pop(id(info), None)
                if done is not None:
                    return done
                return tuple(int(v) for v in info.cpu().tolist())
            return None
        self._owed = None
        owed['event'].synchronize()
        eng, c = self.eng, owed['counters']
        pending, made = (int(v) for v in c.cpu().tolist())
        one = torch.ones((1,), dtype=torch.int32, device=c.device)
        while pending > 0:
            if made >= RETRY_DEPTH_LIMIT:
                raise RecursionError(RETRY_MESSAGE)
            init_r = eng.np_kmeans_init(self._np_state, owed['prior'], owed['off'][owed['B']:], owed['k'], gate=one)
            assign_r, _ = eng.kmeans(owed['X'], owed['prior'], owed['off'][owed['B']:], owed['k'], 1000, init_r, gate=one)
            eng.retry_update(assign_r, owed['off'], owed['B'], c, gate=one)
            pending, made = (int(v) for v in c.cpu().tolist())
        if not hasattr(self, '_settled'):
            self._settled = {}
        if len(self._settled) > 64:
            self._settled.clear()
        self._settled[id(c)] = (pending, made)
        if info is not None:
            return self._settled.pop(id(c))
        return (pending, made)

    # ---------------------------------------------------------------- whole batch
    def _tick(self, name):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        self._ev[name] = ev

    def run(self, imgs, check_status=True, join=True):
        """imgs: (B,3,H,W) float32 RGB 0..255, numpy (pinned or not) or CUDA tensor.
        join=False (batch loops): the result is complete on `res.stream` (the auxiliary stream in the two-stream mean-pooling
        flow, else the current one) at the event `res.ready`; whoever consumes it enqueues there or waits for the event, and the
        calling stream is NOT made to wait — the next batch's DRN forward runs under this batch's pooling / k-means / paint.
        join=True (default): the calling stream waits for the result, as a plain call must."""
        main = torch.cuda.current_stream()
        imgs_dev = torch.as_tensor(imgs)
        if not imgs_dev.is_cuda:
            imgs_dev = imgs_dev.to(self.eng.device, non_blocking=True)
        imgs_dev = imgs_dev.float().contiguous()
        self._tick('start')
        if self.aux is not None and self.pool_mode == 'anchor' and not self.device_rng:
            # anchor mode with the host generator: the superpixels and their statistics run FIRST, on the main stream, the DRN
            # forward behind them; the host then waits for the sizes (14 ms in), draws the anchors (~47 ms) while the forward
            # (~65 ms) runs, and sends them back on the auxiliary stream.  (Superpixels BESIDE the forward on a second stream —
            # the layout below — cost 27.6 + 77.6 ms instead of 14 + 65: the forward's persistent kernels and the SLIC sweeps
            # take the compute units from each other, and the draws started 14 ms later.)
            self._tick('sp_start')
            labels, n_labels = self.superpixels(imgs_dev)
            self._tick('superpixel')
            seg = self.segments_stats(imgs_dev.shape, labels, n_labels)
            sizes = torch.cuda.Event()
            sizes.record(main)
            fmap = self.features(imgs_dev)
            self._tick('features')
            self.aux.wait_event(sizes)
            with torch.cuda.stream(self.aux):
                for t in [labels, n_labels] + [v for v in seg.values() if isinstance(v, torch.Tensor)]:
                    t.record_stream(self.aux)
                seg = self.segments_anchors(imgs_dev.shape, labels, seg)
                self._tick('segments')
            main.wait_stream(self.aux)
            for t in [v for v in seg.values() if isinstance(v, torch.Tensor)]:
                t.record_stream(main)
        elif self.aux is not None and self.tail_on_aux and self.pool_mode == 'mean':
            # main stream: the DRN forward only.  Auxiliary stream: superpixels, statistics, then — once the forward's maps
            # exist — pooling, k-means, paint.  Every kernel that touches the context's shared workspaces is on the auxiliary
            # stream, in order, batch after batch; the two streams meet at two events per batch.
            aux = self.aux
            given = torch.cuda.Event()
            given.record(main)                       # the batch is complete on the calling stream
            aux.wait_event(given)
            imgs_dev.record_stream(aux)
            with torch.cuda.stream(aux):
                self._tick('sp_start')
                labels, n_labels = self.superpixels(imgs_dev)
                self._tick('superpixel')
            fmap = self.features(imgs_dev)
            self._tick('features')
            feat = torch.cuda.Event()
            feat.record(main)
            fmap.record_stream(aux)
            with torch.cuda.stream(aux):
                seg = self.segments(imgs_dev.shape, labels, n_labels)
                self._tick('segments')
                aux.wait_event(feat)
                self._tick('joined')
                X = self.pool(imgs_dev.shape, labels, seg, fmap)
                self._tick('describe')
                assign, info, cluster, road, fail = self.cluster(labels, seg['off'], X, seg['prior'])
                self._tick('kmeans')
                done = torch.cuda.Event(enable_timing=True)
                done.record(aux)
            res = BatchResult(labels=labels, n_labels=n_labels, offsets=seg['off'], count=seg['count'],
                              X=X, prior=seg['prior'], assign=assign, info=info, cluster=cluster,
                              road=road, fmap=fmap, retry_fail=fail, retry_info=self._retry_info, retry_settle=self._settle_retries,
                              strict_retry=bool(getattr(self.args, 'strict_retry', False)), stream=aux, ready=done)
            if join:
                main.wait_event(done)
                for t in res.__dict__.values():
                    if isinstance(t, torch.Tensor) and t.is_cuda:
                        t.record_stream(main)
                res.stream = main
            if check_status:
                self.eng.raise_on_status()
                res.check_retry()
            return res
        elif self.aux is not None:
            # superpixel branch on the auxiliary stream; it also waits for the previous batch's
            # consumers of the shared workspaces, which ran on the main stream
            self.aux.wait_stream(main)
            box = {}

            def start_sp(after_layer=None):
                if box:
                    return
                if after_layer is not None:
                    ev = torch.cuda.Event()
                    ev.record(main)                  # the forward up to and including that layer
                    self.aux.wait_event(ev)
                with torch.cuda.stream(self.aux):
                    self._tick('sp_start')
                    box['sp'] = self.superpixels(imgs_dev)
                    self._tick('superpixel')
            # SPA_SP_AFTER_LAYER = n: the superpixel branch starts when layer n of the forward has run (beside the Winograd layers
            # instead of beside the network's front); unset: at once
            after = int(os.environ.get('SPA_SP_AFTER_LAYER', '0'))
            if after > 0 and self.model is not None:
                self.model._layer_hook = lambda i: start_sp(i) if i == after else None
            else:
                start_sp()
            # enqueue the DRN forward BEFORE anything on the aux branch can block the host
            # (anchor mode synchronises the aux stream to draw the anchors on the host)
            try:
                fmap = self.features(imgs_dev)
            finally:
                if after > 0 and self.model is not None:
                    self.model._layer_hook = None
            start_sp()                               # (a forward that never passed the hook, e.g. a replayed graph)
            labels, n_labels = box['sp']
            self._tick('features')
            with torch.cuda.stream(self.aux):
                seg = self.segments(imgs_dev.shape, labels, n_labels)
                self._tick('segments')
            main.wait_stream(self.aux)
            for t in [labels, n_labels] + [v for v in seg.values() if isinstance(v, torch.Tensor)]:
                t.record_stream(main)
        else:
            fmap = self.features(imgs_dev)
            self._tick('features')
            self._tick('sp_start')
            labels, n_labels = self.superpixels(imgs_dev)
            self._tick('superpixel')
            seg = self.segments(imgs_dev.shape, labels, n_labels)
            self._tick('segments')
        self._tick('joined')
        X = self.pool(imgs_dev.shape, labels, seg, fmap)
        self._tick('describe')
        assign, info, cluster, road, fail = self.cluster(labels, seg['off'], X, seg['prior'])
        self._tick('kmeans')
        done = torch.cuda.Event(enable_timing=True)
        done.record(main)
        res = BatchResult(labels=labels, n_labels=n_labels, offsets=seg['off'], count=seg['count'],
                          X=X, prior=seg['prior'], assign=assign, info=info, cluster=cluster,
                          road=road, fmap=fmap, retry_fail=fail, retry_info=self._retry_info, retry_settle=self._settle_retries,
                          strict_retry=bool(getattr(self.args, 'strict_retry', False)), stream=main, ready=done)
        if check_status:
            self.eng.raise_on_status()
            res.check_retry()
        return res

    def stage_ms(self, events=None):
        """Device-event durations of the last run() in ms (or of the run whose `events` = dict(pipe._ev) the
        caller kept and knows to be complete: then nothing is synchronised).  With two streams the superpixel
        branch overlaps the DRN forward, so the stages do not add up to the step time."""
        if events is None:
            torch.cuda.synchronize()
        e = self._ev if events is None else events
        return {'time_feature_maps': e['start'].elapsed_time(e['features']),
                'time_superpixel': e['sp_start'].elapsed_time(e['superpixel']),
                'time_roialign': e['superpixel'].elapsed_time(e['segments']) + e['joined'].elapsed_time(e['describe']),
                'time_prior': 0.0,
                'time_kmeans': e['describe'].elapsed_time(e['kmeans'])}

    def elapsed_times(self, events=None):
        """Stage times of the last run() in seconds, under the reference's result.json keys
        (:428-458) plus time_feature_maps (the baselines' key, direct_clustering.py:292-294).
        The prior is computed inside the segment-statistics pass: its share is reported under
        time_roialign and time_prior is 0."""
        return {k: v / 1000.0 for k, v in self.stage_ms(events).items()}


class HostStream(object):
    """Host memory to host memory, double buffered (SURVEY.md 8d's timed region: "from batch tensor
    resident in pinned host memory to masks in host memory").

    Two device input buffers and two pinned output buffers; the upload of batch s+1 (h2d stream) and
    the download of batch s's two uint8 masks (d2h stream) run under the kernels of the neighbouring
    batches, so PCIe (25 MB up + 4 MB down per 1024x2048 image) never sits on the critical path as
    long as a step takes longer than its own transfers.

        hs = HostStream(pipe, B, H, W)
        for cluster, road, res in hs.process(batches):   # batches: iterable of pinned (B,3,H,W) f32
            ...            # cluster / road: numpy views of pinned memory, valid until the next-but-one yield

    `after(res, s)` may enqueue extra device work on the compute stream for the s-th batch of the
    call (bench: confusion counts)."""

    def __init__(self, pipe, B, H, W, after=None, u8_hwc=False):
        """u8_hwc: the batches are decoded 8-bit images, (B,H,W,3) uint8 as a PNG decoder leaves them (what the drivers
        of cli.py hand over): 3 bytes per pixel cross PCIe instead of 12, and the planar float32 batch the kernels
        take is made on the device (spa_resize_bicubic_u8 at unchanged size = layout and type change only)."""
        self.pipe, self.after, self.u8_hwc = pipe, after, bool(u8_hwc)
        self.late_download = os.environ.get('SPA_LATE_DOWNLOAD', '1') != '0'
        self.upload_after = os.environ.get('SPA_UPLOAD_AFTER', '') or None      # a pipe._ev key to delay the next upload to; default: at once
        dev = pipe.eng.device
        self.dev = dev
        if self.u8_hwc:
            self.inp = [torch.empty((B, H, W, 3), dtype=torch.uint8, device=dev) for _ in range(2)]
        else:
            self.inp = [torch.empty((B, 3, H, W), dtype=torch.float32, device=dev) for _ in range(2)]
        self.out = [(torch.empty((B, H, W), dtype=torch.uint8).pin_memory(),
                     torch.empty((B, H, W), dtype=torch.uint8).pin_memory()) for _ in range(2)]
        self.h2d = torch.cuda.Stream(device=dev)
        self.d2h = torch.cuda.Stream(device=dev)
        self.up_done = [torch.cuda.Event() for _ in range(2)]       # inp[i] holds its batch
        self.in_free = [torch.cuda.Event() for _ in range(2)]       # kernels reading inp[i] finished
        self.down_done = [torch.cuda.Event() for _ in range(2)]     # out[i] holds its masks
        self.staging = None

    def pinned_batch(self, B=None):
        """A pinned (B,3,H,W) float32 array for the decode workers to fill in place."""
        shape = tuple(self.inp[0].shape) if B is None else (B,) + tuple(self.inp[0].shape[1:])
        return torch.empty(shape, dtype=self.inp[0].dtype).pin_memory()

    def _upload(self, slot, batch, after=None):
        """after: an event of the compute stream the copy waits for (SPA_UPLOAD_AFTER=features|superpixel|joined delays the
        next batch's upload to that point of the current batch).  Measured (tools/h2h_probe2.py, 30 x 1024x2048, 8-bit
        uploads of 189 MB = 3.85 ms of DMA): the loop costs 80-82 ms per batch wherever the upload is placed and with 4, 8
        or 16 hardware queues, 77.8 without uploads, 76 device resident — every kernel of the batch runs a few per cent
        slower while host traffic is in flight, so the default stays 'at once'."""
        t = torch.as_tensor(batch)
        if not t.is_pinned():
            # pageable memory: stage through a pinned buffer (a host memcpy; the drivers decode straight
            # into pinned_batch() arrays instead)
            if self.staging is None or self.staging.shape != t.shape:
                self.staging = torch.empty(t.shape, dtype=self.inp[0].dtype).pin_memory()
            self.h2d.synchronize()
            self.staging.copy_(t)
            t = self.staging
        with torch.cuda.stream(self.h2d):
            self.h2d.wait_event(self.in_free[slot])
            if after is not None:
                self.h2d.wait_event(after)
            self.inp[slot][:t.shape[0]].copy_(t, non_blocking=True)
            self.up_done[slot].record(self.h2d)
        return t.shape[0]

    def _download(self, slot, res, n, after):
        with torch.cuda.stream(self.d2h):
            if after is not None:
                self.d2h.wait_event(after)
            oc, orr = self.out[slot]
            oc[:n].copy_(res.cluster, non_blocking=True)
            orr[:n].copy_(res.road, non_blocking=True)
            res.cluster.record_stream(self.d2h)
            res.road.record_stream(self.d2h)
            self.down_done[slot].record(self.d2h)

    def process(self, batches):
        main = torch.cuda.current_stream(self.dev)
        it = iter(batches)
        for ev in self.in_free:
            ev.record(main)
        nxt = next(it, None)
        if nxt is None:
            return
        nb = self._upload(0, nxt)
        s = 0
        prev = None                                   # (slot, result, n) whose download is in flight
        while nxt is not None:
            slot = s & 1
            cur_n = nb
            nxt = next(it, None)
            main.wait_event(self.up_done[slot])
            src = self.inp[slot][:cur_n]
            if self.u8_hwc:
                src = self.pipe.eng.resize_u8(src, (src.shape[1], src.shape[2]))       # same size: (B,3,H,W) float32 planar
                self.in_free[slot].record(main)        # the 8-bit buffer is free as soon as it has been widened
            res = self.pipe.run(src, check_status=False, join=False)
            with torch.cuda.stream(res.stream):      # the stream the result lives on (the pipeline's auxiliary one, or main)
                if self.after is not None:
                    self.after(res, s)
                done = torch.cuda.Event()
                done.record(res.stream)              # the forward (main) has finished before the tail (res.stream) could start
                if not self.u8_hwc:
                    self.in_free[slot].record(res.stream)
            if nxt is not None:
                # under this batch's kernels (optionally behind one of its stage events, see _upload)
                nb = self._upload(slot ^ 1, nxt, after=self.pipe._ev.get(self.upload_after) if self.upload_after else None)
            if not self.late_download:
                self._download(slot, res, cur_n, done)
            if prev is not None:
                pslot, pres, pn, pdone = prev
                if self.late_download:
                    # the download is enqueued only once the batch HAS finished (the host waits, the copy stream does not):
                    # a copy queue whose head is a barrier waiting ~a batch's time for the compute stream was measured to
                    # slow every dispatch of that batch (tools/h2h_probe2.py)
                    pdone.synchronize()
                    self._download(pslot, pres, pn, None)
                self.down_done[pslot].synchronize()
                yield self.out[pslot][0][:pn].numpy(), self.out[pslot][1][:pn].numpy(), pres
            prev = (slot, res, cur_n, done)
            s += 1
        pslot, pres, pn, pdone = prev
        if self.late_download:
            pdone.synchronize()
            self._download(pslot, pres, pn, None)
        self.down_done[pslot].synchronize()
        yield self.out[pslot][0][:pn].numpy(), self.out[pslot][1][:pn].numpy(), pres
