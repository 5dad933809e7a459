"""BASELINE configs[1] as a WHOLE, at its real size, against the oracle (-m gpu):

    1 x MI355X: DRN-D-22 fp32 features + HIP SLIC(200 sp) / pool / k-means on one 1024x2048 image,
    bit-match label map vs reference

i.e. DRN-D-22 -> SLIC(200) -> mean pooling of the real C=512, 128x256 map -> prior -> weighted
k-means over D=514 -> painted uint8 masks, one image, a batch of 30 (the bench's step) and the
config-5 shape (bf16 features, 400 superpixels).  The DRN is checked against a float32 PyTorch-CPU
forward of the same module at full size (floating-point stage, 1e-4 of scale); every later stage is
compared bit for bit with the oracle fed with the GPU's own feature map.  Mean pooling is also
checked against the committed restatement of the notebook (tests/golden/meanpool_*.npz).
"""
import glob
import importlib
import os
import types
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from conftest import GOLDEN, golden

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')

H, W = 1024, 2048


def _args(**kw):
    d = dict(superpixel_method='slic', n_slic_segments=200, n_anchors=10, n_neighbors=4,
             without_pos=False, y_rel_pos=0.75, x_rel_pos=0.5, y_rel_sigma=0.1, x_rel_sigma=0.1,
             gpu=0, n_clusters=2, use_feature_maps=[7], pool_mode='mean', mean_sampling='nearest',
             arch='drn_d_22', dtype='fp32', drn_weights=None)
    d.update(kw)
    return types.SimpleNamespace(**d)


@pytest.fixture(scope='module')
def mods():
    names = ('ops', 'pipeline', 'drn', 'engine')
    return types.SimpleNamespace(**{n: importlib.import_module('superpixel-align_amd.' + n) for n in names})


def _oracle_rest(orc, args, imgs, fmap, pool_mode='mean', sampling='nearest', threads=16):
    """Everything behind the DRN on the host oracle (images in parallel: ctypes releases the GIL)."""
    with ThreadPoolExecutor(threads) as ex:
        sps = np.stack(list(ex.map(lambda im: orc.slic(im, args.n_slic_segments), imgs)))
        n_per = [int(s.max()) + 1 for s in sps]

        def pool(b):
            f = orc.mean_pool(fmap[b], sps[b], sampling, n_per[b])
            _, cy, cx = orc.segment_stats(sps[b], n_per[b])
            return np.hstack([f.astype(np.float64), cy[:, None], cx[:, None]])
        feats = np.concatenate(list(ex.map(pool, range(len(imgs)))), axis=0)
    prior = orc.batch_create_prior(args, sps)
    cl, road, info = orc.batch_weighted_kmeans(args, sps, feats, prior, n_per)
    return sps, n_per, feats, prior, cl, road, info


def _check(res, orc, args, imgs, sampling='nearest'):
    fmap = res.fmap.float().cpu().numpy()
    assert fmap.shape[1:] == (512, H // 8, W // 8)
    sps, n_per, feats, prior, cl, road, info = _oracle_rest(orc, args, imgs, fmap, sampling=sampling)
    N = sum(n_per)
    assert np.array_equal(res.labels.cpu().numpy().astype(np.int64), sps), 'label maps differ'
    assert res.n_labels.cpu().tolist() == n_per
    assert res.X.shape[1] == 514
    assert np.array_equal(res.X[:N].cpu().numpy(), feats), 'descriptors differ'
    np.testing.assert_allclose(res.prior[:N].cpu().numpy(), prior, rtol=1e-12)
    assert res.info.cpu().tolist()[:3] == [info['n_iter'], info['status'], N]
    assert np.array_equal(res.assign[:N].cpu().numpy(), info['assign'])
    cluster = res.cluster.cpu().numpy()
    assert cluster.dtype == np.uint8 and cluster.shape == (len(imgs), H, W)
    assert np.array_equal(cluster, cl), 'cluster maps differ'
    assert np.array_equal(res.road.cpu().numpy().astype(bool), road), 'road masks differ'
    return n_per, info


def test_config1_one_image_end_to_end(mods, orc, synth):
    """configs[1] literally: one 1024x2048 image, DRN-D-22 fp32, SLIC 200, k = 2 -> uint8 mask."""
    args = _args()
    imgs = synth.synth_batch([0], H, W)
    model = mods.drn.create_drn('drn_d_22', device='cuda')
    pipe = mods.pipeline.LabelPipeline(args, model, mods.ops.engine())
    res = pipe.run(imgs)
    n_per, info = _check(res, orc, args, imgs)
    assert 150 <= n_per[0] <= 200
    # the DRN at full size against a float32 PyTorch-CPU forward of the same module (unfolded BN,
    # no library kernel): 1e-4 of the map's scale (north_star's floating-point tolerance)
    ref_model = mods.drn.create_drn('drn_d_22', device='cpu', fold_bn=False)      # same seed, same weights
    with torch.no_grad():
        x = mods.drn.DRN.normalise(torch.from_numpy(imgs))
        ref = ref_model.forward_maps(x)[7].numpy()
    got = res.fmap.float().cpu().numpy()
    scale = float(np.abs(ref).max())
    assert np.abs(got - ref).max() <= 1e-4 * scale, float(np.abs(got - ref).max() / scale)


def test_config1_batch_of_30(mods, orc, synth):
    """The bench's step: 30 distinct 1024x2048 images, k-means joint over the batch (N ~ 5 600)."""
    args = _args()
    imgs = synth.synth_batch(list(range(100, 130)), H, W)
    model = mods.drn.create_drn('drn_d_22', device='cuda')
    pipe = mods.pipeline.LabelPipeline(args, model, mods.ops.engine())
    res = pipe.run(imgs)
    n_per, info = _check(res, orc, args, imgs)
    assert len(n_per) == 30 and sum(n_per) > 4500
    # the next batch reuses every workspace: same answer for the same images
    res2 = pipe.run(imgs)
    assert torch.equal(res2.cluster, res.cluster) and torch.equal(res2.labels, res.labels)


@pytest.mark.parametrize('sampling', ['nearest', 'bilinear'])
def test_config5_shape_bf16_400_superpixels(mods, orc, synth, sampling):
    """configs[4]'s per-GPU shape: bf16 DRN features, 400 superpixels, 1024x2048 (2 images)."""
    args = _args(n_slic_segments=400, dtype='bf16', mean_sampling=sampling)
    imgs = synth.synth_batch([7, 8], H, W)
    model = mods.drn.create_drn('drn_d_22', device='cuda', dtype=torch.bfloat16)
    pipe = mods.pipeline.LabelPipeline(args, model, mods.ops.engine())
    res = pipe.run(imgs)
    assert res.fmap.dtype == torch.bfloat16
    n_per, info = _check(res, orc, args, imgs, sampling)
    assert all(300 <= n <= 400 for n in n_per)


def test_config1_anchor_mode_full_size(mods, orc, synth):
    """The reference's own descriptor (anchor mode, random.shuffle stream) at 1024x2048, C = 512."""
    args = _args(pool_mode='anchor')
    imgs = synth.synth_batch([3], H, W)
    model = mods.drn.create_drn('drn_d_22', device='cuda')
    pipe = mods.pipeline.LabelPipeline(args, model, mods.ops.engine())
    res = pipe.run(imgs)
    fmap = res.fmap.float().cpu().numpy()
    sps = np.stack([orc.slic(im, 200) for im in imgs])
    assert np.array_equal(res.labels.cpu().numpy().astype(np.int64), sps)
    feats, n_per = orc.batch_superpixel_align(args, imgs, sps, fmap, orc.PyRandom(1111), 'anchor')
    N = sum(n_per)
    assert np.array_equal(res.X[:N].cpu().numpy(), feats)
    prior = orc.batch_create_prior(args, sps)
    cl, road, info = orc.batch_weighted_kmeans(args, sps, feats, prior, n_per)
    assert np.array_equal(res.cluster.cpu().numpy(), cl)


MEANPOOL_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, 'meanpool_*.npz')))


@pytest.mark.parametrize('name', MEANPOOL_CASES)
@pytest.mark.parametrize('mode', ['nearest', 'bilinear'])
def test_mean_pool_against_notebook_fixture(mods, orc, name, mode):
    """k_pool_mean vs notebooks/Superpixel_Align.ipynb cell 4 (restated by
    oracle/gen_golden_meanpool.py), ALL superpixels, both samplings: 1e-4 relative per descriptor;
    and bit for bit vs the oracle."""
    from test_oracle_golden import assert_pooled_close
    eng = mods.ops.engine()
    g = golden(name)
    labels = torch.from_numpy(g['labels'].astype(np.int32))[None].cuda()
    S = int(g['counts'].size)
    fmap = torch.from_numpy(g['fmap'])[None].cuda().contiguous(memory_format=torch.channels_last)
    off = eng.segment_offsets(torch.tensor([S], dtype=torch.int32, device='cuda'))
    count, _, _ = eng.segment_stats(labels, off, S)
    assert np.array_equal(count.cpu().numpy(), g['counts'])
    X = eng.pool_mean(fmap, labels, off, S, count, mode, None, False).cpu().numpy()
    eng.raise_on_status()
    assert_pooled_close(X, g['mean_' + mode])
    ref = orc.mean_pool(g['fmap'], g['labels'].astype(np.int32), mode, S)
    assert np.array_equal(X.view(np.int32), ref.view(np.int32))
