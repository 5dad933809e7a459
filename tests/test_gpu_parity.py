"""GPU parity tests proper: every op of the hot path, called through the C ABI
(libspalign.so via superpixel-align_amd/engine.py), against the CPU oracle on the same seeded
inputs and against the golden vectors generated from the reference.

Bar: bit exact for every integer output (labels, counts, assignments, masks, confusion) and
for the float32 stages whose arithmetic order is pinned (Lab image, SLIC centroids, anchor
and mean pooling); 1e-12 relative for the float64 prior (different summation order).
"""
import glob
import importlib
import os

import numpy as np
import pytest

from conftest import GOLDEN, golden, kmeans_tie_cases

pytestmark = pytest.mark.gpu

torch = pytest.importorskip('torch')


@pytest.fixture(scope='module')
def eng():
    engine = importlib.import_module('superpixel-align_amd.engine')
    e = engine.Engine()
    yield e
    e.close()


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def test_loaded_native_library(eng, spa):
    import os
    assert os.path.exists(spa._lib.LIB_PATH)
    maps = open('/proc/self/maps').read()
    assert 'libspalign.so' in maps          # the HIP path is the one that runs


@pytest.mark.parametrize('seed,H,W', [(0, 64, 128), (3, 96, 96), (5, 100, 37), (2, 256, 512)])
def test_rgb2lab_bit_exact(eng, orc, synth, seed, H, W):
    img = synth.synth_image(seed, H, W)
    img[:, 0, :5] = 0.0           # linear branch of the sRGB curve and the Lab toe
    img[:, 1, :5] = 0.03
    ref = orc.rgb2lab_scaled(img)                                   # (H,W,3)
    got = eng.rgb2lab(dev(img[None]), 0.1)[0].permute(1, 2, 0).cpu().numpy()
    assert np.array_equal(got.view(np.int32), ref.view(np.int32))


SLIC_CASES = [(0, 64, 128, 20), (1, 128, 256, 100), (3, 96, 96, 30), (5, 100, 37, 12),
              (4, 224, 224, 100), (2, 256, 512, 100), (7, 61, 83, 9)]


@pytest.mark.parametrize('seed,H,W,n', SLIC_CASES)
def test_slic_core_bit_exact(eng, orc, synth, seed, H, W, n):
    lab = orc.rgb2lab_scaled(synth.synth_image(seed, H, W))
    pre, centres = orc.slic_core(lab, n)
    lab_planar = dev(lab.transpose(2, 0, 1)[None])
    labels, cen = eng.slic_core(lab_planar, n, 10, want_centres=True)
    eng.raise_on_status()
    assert np.array_equal(labels[0].cpu().numpy().astype(np.int64), pre)
    assert np.array_equal(cen[0].cpu().numpy().view(np.int32), centres.view(np.int32))


def test_slic_core_batch_and_golden(eng, orc, synth):
    """A batch of different images in one call; image 0 is the skimage-pinned fixture."""
    g = golden('slic_s1_128x256_n100')
    labs = [orc.rgb2lab_scaled(synth.synth_image(s, 128, 256)) for s in (1, 8, 9)]
    lab_planar = dev(np.stack([l.transpose(2, 0, 1) for l in labs]))
    labels = eng.slic_core(lab_planar, 100, 10).cpu().numpy()
    eng.raise_on_status()
    assert np.array_equal(labels[0], g['pre'].astype(np.int32))
    for b in (1, 2):
        assert np.array_equal(labels[b].astype(np.int64), orc.slic_core(labs[b], 100)[0])


@pytest.mark.parametrize('seed,H,W,n', SLIC_CASES)
def test_connectivity_bit_exact(eng, orc, synth, seed, H, W, n):
    lab = orc.rgb2lab_scaled(synth.synth_image(seed, H, W))
    pre, centres = orc.slic_core(lab, n)
    mn, mx = orc.connectivity_sizes(H, W, centres.shape[0])
    post, nl = orc.enforce_connectivity(pre, mn, mx)
    out, n_labels = eng.enforce_connectivity(dev(pre[None], torch.int32), mn, mx)
    eng.raise_on_status()
    assert np.array_equal(out[0].cpu().numpy().astype(np.int64), post)
    assert int(n_labels[0]) == max(nl, 1)


@pytest.mark.parametrize('name', ['connectivity_stress_30_200', 'connectivity_stress_8_5000',
                                  'connectivity_stress_100_400', 'connectivity_stress_1_50'])
def test_connectivity_stress_golden(eng, orc, name):
    """Noisy label maps with hundreds of tiny fragments, several of them with max_size small
    enough that components are cut in BFS order (skimage-pinned fixtures)."""
    g = golden(name)
    mn, mx = (int(v) for v in g['meta'])
    seg = g['seg'].astype(np.int32)
    out, n_labels = eng.enforce_connectivity(dev(seg[None]), mn, mx)
    eng.raise_on_status()
    assert np.array_equal(out[0].cpu().numpy(), g['post'])
    assert int(n_labels[0]) == int(g['post'].max()) + 1


@pytest.mark.parametrize('seed,H,W,n', [(1, 40, 56, 6), (2, 33, 65, 5), (3, 17, 300, 9), (4, 300, 17, 9),
                                        (5, 64, 64, 64), (6, 128, 128, 400), (7, 250, 250, 3),
                                        (8, 480, 640, 100), (11, 512, 1024, 800), (10, 1024, 2048, 400),
                                        (3, 256, 512, 1000), (4, 256, 512, 8), (6, 64, 2048, 40), (7, 300, 70, 25),
                                        (3, 96, 4000, 60), (5, 200, 4096, 120), (6, 128, 2049, 40)])
def test_slic_edge_shapes(eng, orc, synth, seed, H, W, n):
    """Odd widths (no float4 path), one-seed grids, more seeds than fit, n up to 800, full size."""
    img = synth.synth_image(seed, H, W)
    ref = orc.slic(img, n)
    labels, n_labels = eng.slic(dev(img[None]), n)
    eng.raise_on_status()
    assert np.array_equal(labels[0].cpu().numpy().astype(np.int64), ref)
    assert int(n_labels[0]) == ref.max() + 1


def test_connectivity_random_cuts(eng, orc):
    """Random label maps with max_size small enough that many components are cut in BFS order."""
    rs = np.random.RandomState(11)
    for trial in range(6):
        H, W = 64 + 16 * trial, 96 + 8 * trial
        base = (np.arange(H)[:, None] // 16) * 7 + (np.arange(W)[None, :] // 24)
        seg = np.where(rs.uniform(size=(H, W)) < 0.15, rs.randint(0, 30, size=(H, W)), base).astype(np.int64)
        mn, mx = int(rs.randint(1, 60)), int(rs.randint(60, 400))
        ref, nl = orc.enforce_connectivity(seg, mn, mx)
        out, n_labels = eng.enforce_connectivity(dev(seg[None], torch.int32), mn, mx)
        eng.raise_on_status()
        assert np.array_equal(out[0].cpu().numpy().astype(np.int64), ref), (trial, mn, mx)


@pytest.mark.parametrize('case', [
    # H, W, labels, noise, min_size, max_size
    (24, 2048, 4, 1.0, 6, 100000),        # every pixel a random label: > 4 096 runs per strip of 8 rows -> the
    (40, 1536, 3, 0.8, 12, 100000),       #   strips link their rows through global memory
    (96, 640, 6, 0.35, 40, 100000),       # strips in LDS, many multi-row specks (lane tier) and mid-size boxes
    (300, 700, 5, 0.05, 3000, 100000),    # large small-components: the 26.5 KB / 80 KB LDS tiers
    (64, 4000, 3, 0.5, 25, 2000),         # wide rows + oversize cuts
])
def test_connectivity_run_tables_and_tiers(eng, orc, case):
    """Label maps built to reach every branch of the run-level pass: strips whose run count exceeds the LDS
    table, the lane tier, each LDS replay tier and the oversize (pixel-level) path — all against the restatement
    of scikit-image's scan-order rule."""
    H, W, nl, noise, mn, mx = case
    rs = np.random.RandomState(H * 7 + W)
    base = (np.arange(H)[:, None] // 29) * 5 + (np.arange(W)[None, :] // 131)
    seg = np.where(rs.uniform(size=(H, W)) < noise, rs.randint(0, nl, size=(H, W)), base).astype(np.int64)
    ref, n_ref = orc.enforce_connectivity(seg, mn, mx)
    out, n_labels = eng.enforce_connectivity(dev(np.stack([seg, seg[::-1].copy()]), torch.int32), mn, mx)
    eng.raise_on_status(ignore=0x04)
    assert np.array_equal(out[0].cpu().numpy().astype(np.int64), ref)
    assert int(n_labels[0]) == max(n_ref, 1)
    ref2, n_ref2 = orc.enforce_connectivity(seg[::-1].copy(), mn, mx)
    assert np.array_equal(out[1].cpu().numpy().astype(np.int64), ref2)
    assert int(n_labels[1]) == max(n_ref2, 1)


E2E_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, 'slic_s[0-9]_*.npz')))


@pytest.mark.parametrize('name', E2E_CASES)
def test_slic_from_rgb_is_the_untouched_skimage_call(eng, name):
    """Whole spa_slic call from RGB (BASELINE size 1024x2048 included) against `e2e_skimage`: the label map the
    reference's own batch_superpixel (batch_spalign_kmeans.py:308-311 -> skimage.segmentation.slic, untouched)
    returned for the same image under the reference configuration of tests/golden/PROVENANCE.txt.  0 pixels
    differ; the Lab image is scikit-image's bit for bit (sha256 of the float32 words)."""
    import hashlib
    import importlib
    synth = importlib.import_module('superpixel-align_amd.synth')
    g = golden(name)
    seed, H, W, n = (int(v) for v in g['meta'][:4])
    img = synth.synth_image(seed, H, W)
    lab = eng.rgb2lab(dev(img[None]), 0.1)[0].permute(1, 2, 0).contiguous().cpu().numpy()
    assert hashlib.sha256(lab.tobytes()).hexdigest() == str(g['skimage_lab_sha256'])
    labels, n_labels = eng.slic(dev(img[None]), n)
    eng.raise_on_status()
    assert np.array_equal(labels[0].cpu().numpy(), g['e2e_skimage'].astype(np.int32))
    assert int(n_labels[0]) == int(g['e2e_skimage'].max()) + 1


def test_slic_full_batch_vs_oracle(eng, orc, synth):
    imgs = synth.synth_batch([11, 12, 13, 14], 192, 320)
    labels, n_labels = eng.slic(dev(imgs), 60)
    eng.raise_on_status()
    for b in range(4):
        ref = orc.slic(imgs[b], 60)
        assert np.array_equal(labels[b].cpu().numpy().astype(np.int64), ref)
        assert int(n_labels[b]) == ref.max() + 1


def test_slic_rows_wider_than_4096_pixels(orc, synth):
    """Images the tuned SLIC kernels do not take (their occupancy masks hold 64 pieces of 64 pixels per row) run on
    the general float32 kernels (csrc/spa_slic64.hip, the Cython core's other instantiation): a 40 x 4500 and a
    96 x 5000 image against the oracle, whole call from RGB; and the general kernels forced onto a golden fixture
    (SPA_SLIC_GENERAL=1) give the tuned kernels' / scikit-image's labels and float32 centres bit for bit."""
    engine = importlib.import_module('superpixel-align_amd.engine')
    e = engine.Engine()
    try:
        for seed, H, W, n in [(31, 40, 4500, 30), (32, 96, 5000, 120)]:
            img = synth.synth_image(seed, H, W)
            labels, n_labels = e.slic(dev(img[None]), n)
            e.raise_on_status()
            ref = orc.slic(img, n)
            assert np.array_equal(labels[0].cpu().numpy().astype(np.int64), ref), (H, W)
            assert int(n_labels[0]) == ref.max() + 1
    finally:
        e.close()
    os.environ['SPA_SLIC_GENERAL'] = '1'
    try:
        eg = engine.Engine()
    finally:
        del os.environ['SPA_SLIC_GENERAL']
    try:
        g = golden('slic_s2_256x512_n100')
        img = synth.synth_image(2, 256, 512)
        lab = orc.rgb2lab_scaled(img)
        pre, cen = eg.slic_core(dev(lab.transpose(2, 0, 1)[None]), 100, want_centres=True)
        assert np.array_equal(pre[0].cpu().numpy(), g['pre'].astype(np.int32))
        assert np.array_equal(cen[0].cpu().numpy().view(np.int32), g['centres'].view(np.int32))
        labels, _ = eg.slic(dev(img[None]), 100)
        assert np.array_equal(labels[0].cpu().numpy(), g['e2e_skimage'].astype(np.int32))
        eg.raise_on_status()
    finally:
        eg.close()


def _batch_labels(orc, synth, seeds, H, W, n):
    imgs = synth.synth_batch(seeds, H, W)
    sps = np.stack([orc.slic(im, n) for im in imgs])
    return imgs, sps


def test_segment_stats_and_prior(eng, orc, synth):
    imgs, sps = _batch_labels(orc, synth, [0, 1, 2], 128, 256, 40)
    n_per = [int(s.max()) + 1 for s in sps]
    labels = dev(sps, torch.int32)
    off = eng.segment_offsets(dev(np.array(n_per, np.int32)))
    assert off.cpu().tolist() == [0] + list(np.cumsum(n_per))
    N = sum(n_per)
    count, centroid, prior = eng.segment_stats(labels, off, N + 3, (0.75, 0.5, 0.1, 0.1))
    eng.raise_on_status()
    o = 0
    for b, S in enumerate(n_per):
        cnt, cy, cx = orc.segment_stats(sps[b], S)
        assert np.array_equal(count[o:o + S].cpu().numpy(), cnt)
        assert np.array_equal(centroid[o:o + S, 0].cpu().numpy(), cy)     # exact integer sums
        assert np.array_equal(centroid[o:o + S, 1].cpu().numpy(), cx)
        p = orc.create_prior(sps[b], 0.75, 0.5, 0.1, 0.1, S)
        np.testing.assert_allclose(prior[o:o + S].cpu().numpy(), p, rtol=1e-12, atol=0)
        o += S


@pytest.mark.parametrize('tag', ['small', 'config1'])
def test_anchor_pipeline_against_reference_golden(eng, orc, synth, tag):
    """anchors -> pooled descriptors -> k-means -> paint, all against vectors recorded from the
    reference's own functions (superpixel_align, kmeans, weighted_kmeans)."""
    engine = importlib.import_module('superpixel-align_amd.engine')
    g = golden('pipeline_' + tag)
    seed, H, W, n, C, B = (int(v) for v in g['meta'])
    sps = g['superpixels'].astype(np.int32)
    n_per = [int(v) for v in g['n_per']]
    N = sum(n_per)
    fm = synth.synth_feature_map(seed + 1, C, H // 8, W // 8, batch=B)
    fmap = dev(fm).contiguous(memory_format=torch.channels_last)
    # the reference's own batch_superpixel output (skimage slic from RGB) is what the GPU computes from RGB
    labels, n_labels = eng.slic(dev(synth.synth_batch([seed + b for b in range(B)], H, W)), n)
    eng.raise_on_status()
    assert np.array_equal(labels.cpu().numpy(), sps) and [int(v) for v in n_labels] == n_per
    off = eng.segment_offsets(dev(np.array(n_per, np.int32)))
    count, centroid, prior = eng.segment_stats(labels, off, N, (0.75, 0.5, 0.1, 0.1))
    np.testing.assert_allclose(prior.cpu().numpy(), g['prior'], rtol=1e-12, atol=0)
    # host RNG needs only the counts
    ranks, n_valid = engine.PyRandom(1111).shuffle_select(count.cpu().numpy(), 10)
    assert np.array_equal(n_valid, g['n_valid'])
    anchors = eng.select_anchor_pixels(labels, off, N, dev(ranks), dev(n_valid))
    assert np.array_equal(anchors.cpu().numpy(), g['anchors'])
    X = eng.pool_anchor(fmap, H, off, N, anchors, dev(n_valid), 4, centroid, True)
    assert X.dtype == torch.float64
    assert np.array_equal(X.cpu().numpy(), g['feats'])                  # bit exact
    Xn = eng.pool_anchor(fmap, H, off, N, anchors, dev(n_valid), 4, None, False)
    assert Xn.dtype == torch.float32
    assert np.array_equal(Xn.cpu().numpy(), g['feats_nopos'])
    # k-means on the reference's own descriptors and prior
    Xr, wr = dev(g['feats']), dev(g['prior'])
    assign, info = eng.kmeans(Xr, wr, off[B:], 2)
    assert np.array_equal(assign.cpu().numpy(), g['k2_assign'])
    assert info.cpu().tolist()[1] == 0
    assign4, _ = eng.kmeans(Xr, wr, off[B:], 4, init_other=dev(g['k4_shuffled_idx']))
    assert np.array_equal(assign4.cpu().numpy(), g['k4_assign'])
    # same through the host RNG emulation of np.random.shuffle
    thr = np.sort(g['prior'])[N // 2]
    m = int((g['prior'] <= thr).sum())
    idx = engine.NpRandom(1111).shuffle((np.arange(m) % 3 + 1).astype(np.int64))
    assert np.array_equal(idx, g['k4_shuffled_idx'])
    # paint
    cluster, road = eng.paint(labels, assign, off)
    eng.raise_on_status()
    assert np.array_equal(cluster.cpu().numpy(), g['clustering'])
    assert np.array_equal(road.cpu().numpy(), g['road'])


def test_kmeans_engineered(eng, orc):
    g = golden('kmeans_engineered')
    X, w = dev(g['X']), dev(g['w'])
    n = dev(np.array([X.shape[0]], np.int32))
    a, info = eng.kmeans(X, w, n, 2)
    assert np.array_equal(a.cpu().numpy(), g['assign'])
    Xe, we = dev(g['Xe']), dev(g['we'])
    ne = dev(np.array([Xe.shape[0]], np.int32))
    ae, info = eng.kmeans(Xe, we, ne, 5, init_other=dev(g['idx_e']))
    assert np.array_equal(ae.cpu().numpy(), g['assign_e'])
    ao, it, st = orc.kmeans(5, g['Xe'], g['we'], nprandom=orc.NpRandom(5))
    assert info.cpu().tolist()[:2] == [it, st]
    eng.raise_on_status()


def test_numpy_shuffle_stream_on_the_device(eng):
    """Engine.np_kmeans_init (csrc/spa_nprng.hip): threshold, m, arange(m) % (k - 1) + 1 and np.random.shuffle drawn by one
    workgroup from numpy's MT19937 state in device memory — against the host restatement of the same stream (spa_rng.cpp's
    NpRandom, itself pinned by tests/golden/rng.npz and kmeans_retry.npz): call after call on ONE state (the position inside
    a 624-output block carries over; several blocks per call), sizes from 3 to 40 000 points, ties at the threshold, and a
    closed gate must neither draw nor write."""
    engine = importlib.import_module('superpixel-align_amd.engine')
    host = engine.NpRandom(1111)
    state = torch.from_numpy(host.state().view(np.int32)).cuda()
    rs = np.random.RandomState(7)
    zero, one = dev(np.zeros(1, np.int32)), dev(np.ones(1, np.int32))
    for N, k in [(3, 3), (17, 4), (600, 4), (5617, 4), (5617, 3), (1249, 8), (40000, 5), (2, 4), (1, 3)]:
        w = rs.uniform(0.0, 1.0, N)
        if N > 10:
            w[rs.randint(0, N, N // 7)] = np.sort(w)[N // 2]          # ties at the threshold count as "other"
        thr = np.sort(w)[N // 2]
        m = int((w <= thr).sum())
        want = (np.arange(m) % (k - 1) + 1).astype(np.int64)
        host.shuffle(want)
        ncap = N + 5
        wd = dev(np.concatenate([w, np.full(5, 2.0)]))
        before = state.clone()
        closed = eng.np_kmeans_init(state, wd, dev(np.array([N], np.int32)), k, gate=zero)
        assert torch.equal(state, before) and int(closed.abs().sum()) == 0             # nothing drawn, nothing written
        got = eng.np_kmeans_init(state, wd, dev(np.array([N], np.int32)), k, gate=None if N % 2 else one)
        assert got.shape == (ncap,) and np.array_equal(got[:m].cpu().numpy(), want), (N, k)
    assert np.array_equal(state.cpu().numpy().view(np.uint32), host.state())               # the streams are at the same place
    eng.raise_on_status()


def test_kmeans_retry_branch_follows_the_reference(eng, capsys):
    """weighted_kmeans :201-205 on the GPU paths.  The drop-in op (ops.batch_weighted_kmeans) and the fused
    pipeline (LabelPipeline.cluster) execute the reference's discarded retries, so two consecutive k = 4 batches
    reproduce the cluster maps recorded from the reference's own weighted_kmeans (oracle/gen_golden_retry.py);
    k = 2 ends in RecursionError like the reference."""
    import types
    ops = importlib.import_module('superpixel-align_amd.ops')
    pipeline = importlib.import_module('superpixel-align_amd.pipeline')
    g = golden('kmeans_retry')
    args = types.SimpleNamespace(n_clusters=4, seed=1111)
    ops.seed(1111)
    cl0, road0 = ops.batch_weighted_kmeans(args, g['sps0'].astype(np.int64), g['X0'], g['w0'], [int(v) for v in g['n_per0']])
    cl1, road1 = ops.batch_weighted_kmeans(args, g['sps1'].astype(np.int64), g['X1'], g['w1'], [int(v) for v in g['n_per1']])
    assert capsys.readouterr().out.count('Somehow KMeans seems failed') == int(g['n_retry'])
    assert np.array_equal(cl0, g['cl0']) and np.array_equal(cl1, g['cl1']) and np.array_equal(road0, g['cl0'] == 0)
    # fused pipeline: same stages on device tensors.  Round 5: numpy's generator lives on the device, the initial assignment is
    # drawn there and the retry runs are enqueued speculatively behind a device-side gate (LabelPipeline.cluster); batch 0 of
    # the fixture needs three of them.  Round 4's synchronous host form (--host_kmeans_init) must give the same maps.
    def run_pipe(rounds, host_init, seq=('0', '1'), defer=False, oversize=()):
        monkey = pytest.MonkeyPatch()
        monkey.setenv('SPA_RETRY_ROUNDS', str(rounds))
        try:
            a = types.SimpleNamespace(n_clusters=4, seed=1111, host_kmeans_init=host_init)
            pipe = pipeline.LabelPipeline(a, model=None, engine=eng, pool_mode='mean', overlap=False)
        finally:
            monkey.undo()
        outs, results = [], []
        for i, t in enumerate(seq):
            n_per = g['n_per' + t].astype(np.int32)
            off = dev(np.concatenate([[0], np.cumsum(n_per)]).astype(np.int32))
            pipe.np_init_max = 8 if i in oversize else eng.NP_INIT_MAX      # (8: this batch is "too large" for the device initialisation)
            assign, info, cluster, road, fail = pipe.cluster(dev(g['sps' + t].astype(np.int32)), off, dev(g['X' + t]), dev(g['w' + t]))
            assert fail is None
            res = pipeline.BatchResult(retry_info=pipe._retry_info, retry_settle=pipe._settle_retries)
            if not host_init and not defer:
                res.check_retry()                 # prints the reference's message per retry run
            results.append(res)
            outs.append(cluster.cpu().numpy())
        if defer:
            for res in results:                   # fetched late, as an asynchronous batch loop does
                res.check_retry()
        return outs
    for host_init in (False, True):
        capsys.readouterr()
        outs = run_pipe(4, host_init)
        assert capsys.readouterr().out.count('Somehow KMeans seems failed') == int(g['n_retry'])
        assert np.array_equal(outs[0], g['cl0']) and np.array_equal(outs[1], g['cl1'])
    # round 6: fewer speculative rounds than the data needs (batch 0 needs three) — the runs still owed are made before the next
    # batch draws (or when the last batch is fetched): same maps, same messages, whenever the results are fetched
    for rounds, defer in ((2, False), (0, False), (1, True), (0, True)):
        capsys.readouterr()
        outs = run_pipe(rounds, False, defer=defer)
        assert capsys.readouterr().out.count('Somehow KMeans seems failed') == int(g['n_retry']), (rounds, defer)
        assert np.array_equal(outs[0], g['cl0']) and np.array_equal(outs[1], g['cl1']), (rounds, defer)
    # a batch too large for the device initialisation in the middle of a run (felzenszwalb: the row count is data dependent):
    # it draws on the host from the downloaded state and hands the stream back — the sequence equals the all-host one
    seq = ('0', '1', '0', '1')
    want = run_pipe(4, True, seq=seq)
    for oversize in ((1,), (0, 2), (3,)):
        got = run_pipe(1, False, seq=seq, defer=True, oversize=oversize)
        assert all(np.array_equal(x, y) for x, y in zip(got, want)), oversize
    capsys.readouterr()
    # k = 2
    args2 = types.SimpleNamespace(n_clusters=2, seed=1111, strict_retry=True)
    with pytest.raises(RecursionError):
        ops.batch_weighted_kmeans(args2, g['sps2'].astype(np.int64), g['X2'], g['w2'], [int(v) for v in g['n_per2']])
    pipe2 = pipeline.LabelPipeline(args2, model=None, engine=eng, pool_mode='mean', overlap=False)
    n_per = g['n_per2'].astype(np.int32)
    off = dev(np.concatenate([[0], np.cumsum(n_per)]).astype(np.int32))
    assign, info, cluster, road, fail = pipe2.cluster(dev(g['sps2'].astype(np.int32)), off, dev(g['X2']), dev(g['w2']))
    assert fail.cpu().tolist() == [False, True]
    with pytest.raises(RecursionError):
        pipeline.BatchResult(cluster=cluster, road=road, retry_fail=fail, strict_retry=True).masks_to_host()
    # default: the reference's message, the batch survives
    pipeline.BatchResult(cluster=cluster, road=road, retry_fail=fail).masks_to_host()
    args2.strict_retry = False
    ops.batch_weighted_kmeans(args2, g['sps2'].astype(np.int64), g['X2'], g['w2'], [int(v) for v in g['n_per2']])
    assert capsys.readouterr().out.count('Somehow KMeans seems failed') >= 3
    eng.raise_on_status()


@pytest.mark.parametrize('name', ['slic_starve_s0_24x40_n30', 'slic_starve_s2_40x64_n30',
                                  'slic_starve_s5_32x32_n60', 'slic_starve_s6_40x64_n60'])
def test_slic_starved_seeds(eng, orc, name):
    """A seed that loses all its pixels gets a NaN centre and stays dead, as in scikit-image
    (fixtures from the compiled core); the status bit is informational, nothing raises."""
    g = golden(name)
    seed, H, W, n, nC, mn, mx = (int(v) for v in g['meta'])
    lab = orc.rgb2lab_scaled(g['img'])
    labels, centres = eng.slic_core(dev(lab.transpose(2, 0, 1)[None]), n, want_centres=True)
    assert np.array_equal(labels[0].cpu().numpy(), g['pre'].astype(np.int32))
    c = centres[0].cpu().numpy()
    assert np.array_equal(np.nonzero(np.isnan(c).any(axis=1))[0], g['dead'])
    assert np.array_equal(np.isnan(c), np.isnan(g['centres']))
    alive = ~np.isnan(c).any(axis=1)
    assert np.array_equal(c[alive].view(np.int32), g['centres'][alive].view(np.int32))
    # the whole slic() call from RGB, connectivity included, and the drivers' status check
    full, n_labels = eng.slic(dev(g['img'][None]), n)
    assert np.array_equal(full[0].cpu().numpy(), g['post'].astype(np.int32))
    assert int(n_labels[0]) == int(g['post'].max()) + 1
    eng.raise_on_status()
    assert eng.last_info & 0x01


SLIC64 = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, 'slic64_*.npz')))


@pytest.mark.parametrize('name', SLIC64)
def test_slic_float64_on_uint8_images(eng, orc, name):
    """superpixel_overlaps.py:303: slic(uint8 image, n) = scikit-image's float64 core.  Lab bit-equal to the
    restatement, core labels and float64 centres bit-equal to the compiled core's (fixtures), and the whole call
    identical to the untouched scikit-image call."""
    g = golden(name)
    seed, H, W, n, nC, mn, mx = (int(v) for v in g['meta'])
    rgb = dev(g['img'].astype(np.float32)[None])
    lab = eng.rgb2lab_u8_f64(rgb)
    assert np.array_equal(lab[0].cpu().numpy().transpose(1, 2, 0), orc.rgb2lab_u8_f64(g['img']))
    labels, cen = eng.slic_core_f64(lab, n, want_centres=True)
    assert np.array_equal(labels[0].cpu().numpy(), g['pre'].astype(np.int32))
    assert np.array_equal(cen[0].cpu().numpy(), g['centres'], equal_nan=True)
    full, n_labels = eng.slic_u8(rgb, n)
    assert np.array_equal(full[0].cpu().numpy(), g['e2e_skimage'].astype(np.int32))
    assert int(n_labels[0]) == int(g['e2e_skimage'].max()) + 1
    eng.raise_on_status()


def test_slic_float64_batch_vs_oracle(eng, orc):
    """a batch of larger uint8 images against the restatement (which is identical to scikit-image on the fixtures)"""
    rs = np.random.RandomState(7)
    B, H, W, n = 3, 192, 416, 150
    imgs = np.zeros((B, 3, H, W), np.uint8)
    for b in range(B):
        low = rs.randint(0, 256, (3, H // 16 + 1, W // 16 + 1)).repeat(16, 1).repeat(16, 2)[:, :H, :W]
        imgs[b] = np.clip(low + rs.normal(0, 9, (3, H, W)), 0, 255).astype(np.uint8)
    full, n_labels = eng.slic_u8(dev(imgs.astype(np.float32)), n)
    for b in range(B):
        ref = orc.slic_u8(imgs[b], n)
        assert np.array_equal(full[b].cpu().numpy(), ref.astype(np.int32))
        assert int(n_labels[b]) == int(ref.max()) + 1
    eng.raise_on_status()


def test_device_anchor_selection_is_cpython_shuffle(spa, orc):
    """spa_anchor_ranks_dev (device MT19937 + _randbelow rejection + backward trace of the first places)
    against CPython's own random.shuffle (tests/golden/rng.npz) and the host emulation, with the generator
    state carried across calls, a large superpixel (crosses many 1 024-output steps), empty and 1-pixel
    lists, and more anchors than pixels."""
    engine = importlib.import_module('superpixel-align_amd.engine')
    g = golden('rng')
    for n in (5, 1000, 70000):
        e = engine.Engine()
        e.pyrandom_seed(1111)
        e.pyrandom_generate(400000)
        for cnt, key in ((n, 'py_%d' % n), (n // 2 + 1, 'py_%d_second' % n)):
            c = dev(np.array([cnt], np.int32))
            ranks, nv = e.anchor_ranks(c, dev(np.array([1], np.int32)), 1, 32, cnt)
            k = min(cnt, 32)
            assert int(nv[0]) == k and np.array_equal(ranks[0, :k].cpu().numpy(), g[key][:k]), (n, key)
        e.raise_on_status()
        e.close()
    rs = np.random.RandomState(8)
    counts = np.concatenate([rs.randint(0, 4000, 300), [0, 1, 2, 250000, 7, 1, 0, 33]]).astype(np.int32)
    e = engine.Engine()
    e.pyrandom_seed(1111)
    host = engine.PyRandom(1111)
    for rep in range(2):                                   # second call continues the stream
        e.pyrandom_generate(int(1.7 * counts.sum()) + (1 << 20))
        ranks, nv = e.anchor_ranks(dev(counts), dev(np.array([counts.size], np.int32)), counts.size, 10, int(counts.sum()))
        hr, hn = host.shuffle_select(counts, 10)
        assert np.array_equal(nv.cpu().numpy(), hn)
        got = ranks.cpu().numpy()
        for s in range(counts.size):
            assert np.array_equal(got[s, :hn[s]], hr[s, :hn[s]]), (rep, s, counts[s])
    e.raise_on_status()
    e.close()


def test_resize_bicubic_bit_exact(eng, orc):
    """spa_resize_bicubic_u8 (8f-2) against Pillow's own outputs (fixture), the oracle and the live
    Pillow on this box, including the 1024x2048 -> 224x224 operating point of every reference launcher."""
    g = golden('resize_bicubic')
    for tag in g['cases']:
        img, ref = g[str(tag) + '_img'], g[str(tag) + '_out']
        src = dev(np.ascontiguousarray(img.transpose(1, 2, 0))[None])          # (1, H, W, 3) uint8
        out = eng.resize_bicubic_u8(src, ref.shape[1:])[0].cpu().numpy()
        assert out.dtype == np.float32 and np.array_equal(out, ref.astype(np.float32)), tag
    rs = np.random.RandomState(4)
    big = rs.randint(0, 256, (2, 1024, 2048, 3)).astype(np.uint8)
    out = eng.resize_bicubic_u8(dev(big), (224, 224)).cpu().numpy()
    for b in range(2):
        assert np.array_equal(out[b], orc.resize_bicubic_u8(big[b].transpose(2, 0, 1), (224, 224)).astype(np.float32))
    from PIL import Image
    live = np.stack([np.asarray(Image.fromarray(big[0, :, :, c]).resize((224, 224), Image.BICUBIC)) for c in range(3)])
    assert np.array_equal(out[0], live.astype(np.float32))
    same = eng.resize_bicubic_u8(dev(big[:1, :64, :96]), (64, 96))[0].cpu().numpy()   # layout / dtype change only
    assert np.array_equal(same, big[0, :64, :96].transpose(2, 0, 1).astype(np.float32))


def test_resize_opencv_cubic_follows_the_restatement(eng, orc):
    """spa_resize_cvcubic_u8 — the OpenCV branch of the reference's resize (cv2.resize(uint8 HWC, INTER_CUBIC), then
    .astype(float32): datasets/resize_image_dataset.py:20-36), which cannot be pinned here (no cv2, no fixture): the kernel is
    bit-identical to the restatement of OpenCV's published 8-bit scalar algorithm (oracle/resize_oracle.c:
    orc_resize_cvcubic_u8) on down- and up-scaling, odd sizes and the 1024x2048 -> 224x224 operating point, and to the host
    form the driver falls back to; its values are bytes."""
    import importlib
    cli = importlib.import_module('superpixel-align_amd.cli')
    rs = np.random.RandomState(8)
    for (H, W, h, w) in ((37, 53, 16, 24), (64, 100, 224, 224), (30, 30, 30, 61), (9, 7, 20, 3), (1024, 2048, 224, 224)):
        img = rs.randint(0, 256, (2, H, W, 3)).astype(np.uint8)
        out = eng.resize_cvcubic_u8(dev(img), (h, w)).cpu().numpy()
        assert out.dtype == np.float32 and out.shape == (2, 3, h, w)
        for b in range(2):
            ref = orc.resize_cvcubic_u8(img[b].transpose(2, 0, 1), (h, w))
            assert np.array_equal(out[b], ref.astype(np.float32)), (H, W, h, w)
        assert np.array_equal(out[0], cli.resize_cvcubic_chw(img[0].transpose(2, 0, 1), (h, w)).astype(np.float32))
        assert np.array_equal(out, np.rint(out)) and out.min() >= 0 and out.max() <= 255
    same = eng.resize_cvcubic_u8(dev(img[:1, :64, :96]), (64, 96))[0].cpu().numpy()      # layout / dtype change only
    assert np.array_equal(same, img[0, :64, :96].transpose(2, 0, 1).astype(np.float32))


def test_kmeans_near_ties(eng):
    """Inputs bisected onto the reference's decision boundary (tests/golden/kmeans_tie.npz): the
    kernel's sums must round exactly like numpy's (sequential axis-0 centre sums, pairwise
    add.reduce distances, float32 arithmetic for float32 descriptors) to get all 40 right."""
    wrong = []
    for name, k, X, w, expect, idx in kmeans_tie_cases():
        n = dev(np.array([X.shape[0]], np.int32))
        init = dev(idx) if k > 2 else None
        a, info = eng.kmeans(dev(X), dev(w), n, k, init_other=init)
        if not np.array_equal(a.cpu().numpy(), expect):
            wrong.append(name)
    eng.raise_on_status()
    assert not wrong, wrong


@pytest.mark.parametrize('sampling', ['nearest', 'bilinear'])
def test_mean_pool_bit_exact(eng, orc, synth, sampling):
    imgs, sps = _batch_labels(orc, synth, [4, 5], 128, 256, 40)
    n_per = [int(s.max()) + 1 for s in sps]
    N = sum(n_per)
    C = 96
    fm = synth.synth_feature_map(9, C, 16, 32, batch=2)
    fmap = dev(fm).contiguous(memory_format=torch.channels_last)
    labels = dev(sps, torch.int32)
    off = eng.segment_offsets(dev(np.array(n_per, np.int32)))
    count, centroid, _ = eng.segment_stats(labels, off, N)
    X = eng.pool_mean(fmap, labels, off, N, count, sampling, centroid, True).cpu().numpy()
    Xn = eng.pool_mean(fmap, labels, off, N, count, sampling, None, False).cpu().numpy()
    eng.raise_on_status()
    o = 0
    for b, S in enumerate(n_per):
        ref = orc.mean_pool(fm[b], sps[b], sampling, S)
        assert np.array_equal(Xn[o:o + S].view(np.int32), ref.view(np.int32))
        assert np.array_equal(X[o:o + S, :C], ref.astype(np.float64))
        _, cy, cx = orc.segment_stats(sps[b], S)
        assert np.array_equal(X[o:o + S, C], cy) and np.array_equal(X[o:o + S, C + 1], cx)
        # and the notebook definition (mean of the upsampled map over the mask), to tolerance
        if sampling == 'nearest':
            up = np.repeat(np.repeat(fm[b], 8, axis=1), 8, axis=2)
            s = S // 2
            np.testing.assert_allclose(ref[s], up[:, sps[b] == s].mean(axis=1), rtol=1e-4, atol=1e-5)
        o += S


@pytest.mark.parametrize('sampling', ['nearest', 'bilinear'])
def test_mean_pool_more_than_sixteen_superpixels_per_feature_pixel(eng, orc, synth, sampling):
    """Mean pooling when a feature pixel is touched by more superpixels than its 16 slots hold (used to be a
    hard SpalignError): (a) a label map with one segment per 2x2 block — 16 segments under every 8x8 cell, up to
    ~80 within reach of a bilinear cell; (b) felzenszwalb (300, 0.8, 20) at 224x224, the reference launchers'
    setting (batch_spalign_kmeans.py:301-307), whose 20-pixel segments crowd the 28x28 map.  Bit exact with the
    oracle, no status bit."""
    H, W, C = 64, 96, 24
    fine = (np.arange(H)[:, None] // 2) * (W // 2) + (np.arange(W)[None, :] // 2)
    coarse = (np.arange(H)[:, None] // 16) * (W // 16) + (np.arange(W)[None, :] // 16)
    sps = np.stack([fine, coarse, np.where(np.arange(W)[None, :] < W // 2, fine, fine.max() + 1 + coarse)])
    sps = np.stack([np.unique(m, return_inverse=True)[1].reshape(H, W) for m in sps]).astype(np.int32)
    n_per = [int(m.max()) + 1 for m in sps]
    N = sum(n_per)
    fm = synth.synth_feature_map(21, C, H // 8, W // 8, batch=3)
    fmap = dev(fm).contiguous(memory_format=torch.channels_last)
    labels = dev(sps)
    off = eng.segment_offsets(dev(np.array(n_per, np.int32)))
    count, centroid, _ = eng.segment_stats(labels, off, N)
    X = eng.pool_mean(fmap, labels, off, N, count, sampling, None, False).cpu().numpy()
    eng.raise_on_status(ignore=0)
    o = 0
    for b, S in enumerate(n_per):
        ref = orc.mean_pool(fm[b], sps[b], sampling, S)
        assert np.array_equal(X[o:o + S].view(np.int32), ref.view(np.int32)), b
        o += S
    # (b) the reference operating point
    imgs = np.stack([synth.synth_scene(60 + i, 224, 224) for i in range(2)])
    lab, n_lab = eng.felzenszwalb(dev(imgs), 300.0, 0.8, 20)
    sp = lab.cpu().numpy()
    n_per = [int(v) for v in n_lab]
    cells = [len(np.unique(sp[0][8 * u:8 * u + 8, 8 * v:8 * v + 8])) for u in range(28) for v in range(28)]
    fm = synth.synth_feature_map(22, 32, 28, 28, batch=2)
    off = eng.segment_offsets(n_lab)
    N = sum(n_per)
    count, centroid, _ = eng.segment_stats(lab, off, N)
    X = eng.pool_mean(dev(fm).contiguous(memory_format=torch.channels_last), lab, off, N, count, sampling, None, False).cpu().numpy()
    eng.raise_on_status(ignore=0)
    o = 0
    for b, S in enumerate(n_per):
        ref = orc.mean_pool(fm[b], sp[b], sampling, S)
        assert np.array_equal(X[o:o + S].view(np.int32), ref.view(np.int32)), b
        o += S
    print('most superpixels under one 8x8 cell of the felzenszwalb map:', max(cells))


def test_pool_bf16_features(eng, orc, synth):
    imgs, sps = _batch_labels(orc, synth, [4], 128, 256, 40)
    S = int(sps[0].max()) + 1
    fm = synth.synth_feature_map(9, 64, 16, 32, batch=1)
    fb = dev(fm).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    labels = dev(sps, torch.int32)
    off = eng.segment_offsets(dev(np.array([S], np.int32)))
    count, centroid, _ = eng.segment_stats(labels, off, S)
    X = eng.pool_mean(fb, labels, off, S, count, 'nearest', None, False).cpu().numpy()
    ref = orc.mean_pool(fb.float().cpu().numpy()[0], sps[0], 'nearest', S)
    assert np.array_equal(X, ref)


def test_layout_error_is_loud(eng, spa):
    import ctypes
    L = spa._lib.lib()
    d = spa._lib.FmapDesc(8, 4, 4, 128, 16, 4, 1, 0)        # NCHW strides
    rc = L.spa_pool_mean(eng._ctx, 1, ctypes.byref(d), 1, 1, 32, 32, 1, 1, 1, 0, None, 0, 1, 0, 8, None)
    assert rc in (-1, -4)


def test_confusion_exact(eng, orc):
    rs = np.random.RandomState(0)
    pred = (rs.uniform(size=(2, 64, 96)) < 0.4).astype(np.uint8)
    gt = rs.randint(-1, 2, size=(2, 64, 96)).astype(np.int32)
    out = eng.confusion(dev(pred), dev(gt)).cpu().numpy()
    for b in range(2):
        r = orc.confusion(pred[b], gt[b])
        assert out[b].tolist() == [r['TN'], r['FP'], r['FN'], r['TP']]


def test_status_reports_bad_labels(eng):
    labels = torch.full((1, 16, 16), 5, dtype=torch.int32, device='cuda')
    off = torch.tensor([0, 2], dtype=torch.int32, device='cuda')
    eng.segment_stats(labels, off, 2)
    assert eng.status() & 0x20
