"""north_star's floating-point contract in ITS terms, -m gpu, full size:

    "Outputs match the reference ... identical per-pixel integer label maps for fixed SLIC/k-means seeds,
     pooled feature vectors within 1e-4 relative"

The reference's arithmetic is float32 convolutions (models/drn.py:304-325 -> cuDNN) pooled into per-superpixel
descriptors (batch_spalign_kmeans.py:431-435, :444).  The default network here multiplies on the 16-bit matrix
cores (two half-precision planes per operand, three products, float32 accumulation).  These tests run the WHOLE
label pipeline twice on the same 1024 x 2048 images and the same weights — once with the split planes, once with
float32 matrix instructions (`SPA_SPLIT_GEMM=0`, exact fmaf chains) — and compare

  * the pooled descriptors, superpixel by superpixel: max_c |X_split - X_f32| <= 1e-4 * max_c |X_f32|  (asserted),
  * the final cluster / road maps pixel by pixel: the number of differing pixels is printed and asserted 0,

for DRN-D-22 and DRN-C-26, with random-init weights and with hostile ones (BatchNorm-folded channel scales over
1e-1.5 .. 1e1.5, tests/test_gpu_hostile.py's construction made architecture independent).  One case also pools a
float32 PyTorch-CPU forward of the unfolded module (no library kernel anywhere) through the oracle's pooling and
holds the default path's descriptors against that."""
import importlib
import os
import types

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')

H, W = 1024, 2048
TOL = 1e-4                      # north_star: pooled feature vectors within 1e-4 relative


def _args(**kw):
    d = dict(superpixel_method='slic', n_slic_segments=200, n_anchors=10, n_neighbors=4,
             without_pos=False, y_rel_pos=0.75, x_rel_pos=0.5, y_rel_sigma=0.1, x_rel_sigma=0.1,
             gpu=0, n_clusters=2, use_feature_maps=[7], pool_mode='mean', mean_sampling='nearest',
             arch='drn_d_22', dtype='fp32', drn_weights=None)
    d.update(kw)
    return types.SimpleNamespace(**d)


@pytest.fixture(scope='module')
def mods():
    names = ('ops', 'pipeline', 'drn', 'engine', 'synth')
    return types.SimpleNamespace(**{n: importlib.import_module('superpixel-align_amd.' + n) for n in names})


def _hostile_pth(drn, arch, path, seed=0, spread=1.5):
    """A checkpoint of `arch` with calibrated BatchNorm statistics whose BasicBlocks are rescaled per channel between
    conv1/bn1 and conv2 by g = 10^U(-spread, spread) (the same function: ReLU is positively homogeneous), so the folded
    weights and the activations inside every block span three decades."""
    g = torch.Generator().manual_seed(seed)
    m = drn.DRN(arch, bn_eps=drn.CHAINER_BN_EPS, with_fc=True).double()
    synth = importlib.import_module('superpixel-align_amd.synth')
    x = drn.DRN.normalise(torch.from_numpy(synth.synth_batch([5, 6], 128, 256))).double()
    with torch.no_grad():
        for bn in (mod for mod in m.modules() if isinstance(mod, torch.nn.BatchNorm2d)):
            bn.weight.copy_(torch.rand(bn.weight.shape, generator=g, dtype=torch.float64) + 0.5)
            bn.bias.copy_(torch.randn(bn.bias.shape, generator=g, dtype=torch.float64) * 0.2)
            bn.momentum = 1.0
        m.train()
        m.forward_maps(x)
        m.eval()
        n = 0
        for blk in m.modules():
            if all(hasattr(blk, a) for a in ('conv1', 'bn1', 'conv2', 'bn2')):
                s = torch.pow(10.0, (torch.rand(blk.bn1.weight.shape, generator=g, dtype=torch.float64) * 2 - 1) * spread)
                blk.bn1.weight.mul_(s)
                blk.bn1.bias.mul_(s)
                blk.conv2.weight.div_(s.view(1, -1, 1, 1))
                n += 1
        assert n >= 6
    sd = {k: v.float() for k, v in m.state_dict().items() if not k.endswith('num_batches_tracked')}
    torch.save(sd, path)
    return path


def _run(mods, arch, weights, imgs, split):
    """The whole pipeline with the split planes (default) or float32 matrix instructions; returns host copies."""
    E = mods.drn._EPILOGUE
    saved = E['split_gemm']
    try:
        E['split_gemm'] = split
        for k in ('gemm16_launches', 'gemm16n_launches', 'conv16_launches'):
            E[k] = 0
        model = mods.drn.create_drn(arch, weights=weights, device='cuda', dtype=torch.float32)
        pipe = mods.pipeline.LabelPipeline(_args(arch=arch), model, mods.ops.engine())
        res = pipe.run(imgs)
        torch.cuda.synchronize()
        ran16 = E['gemm16_launches'] + E['gemm16n_launches'] + E['conv16_launches']
        assert (ran16 > 0) == split, 'the arithmetic asked for is not the one that ran'
        N = int(res.info.cpu()[2])
        out = types.SimpleNamespace(X=res.X[:N].cpu().numpy().copy(), labels=res.labels.cpu().numpy().copy(),
                                    cluster=res.cluster.cpu().numpy().copy(), road=res.road.cpu().numpy().copy(),
                                    assign=res.assign[:N].cpu().numpy().copy(), fmap=res.fmap, N=N)
    finally:
        E['split_gemm'] = saved
    return out


def _descriptor_error(Xs, Xr, C=512):
    """per superpixel: max_c |Xs - Xr| / max_c |Xr| over the C feature columns (the two position columns are integers' means,
    identical by construction)."""
    a, r = Xs[:, :C].astype(np.float64), Xr[:, :C].astype(np.float64)
    scale = np.abs(r).max(axis=1)
    assert (scale > 0).all()
    return np.abs(a - r).max(axis=1) / scale


@pytest.mark.parametrize('arch,weights', [('drn_d_22', 'random'), ('drn_d_22', 'hostile'),
                                          ('drn_c_26', 'random'), ('drn_c_26', 'hostile')])
def test_pooled_descriptors_and_label_maps_split_planes_vs_float32(mods, synth, tmp_path, arch, weights):
    path = _hostile_pth(mods.drn, arch, str(tmp_path / (arch + '-hostile.pth'))) if weights == 'hostile' else None
    imgs = synth.synth_batch([11, 12, 13], H, W)
    s = _run(mods, arch, path, imgs, split=True)
    r = _run(mods, arch, path, imgs, split=False)
    assert np.array_equal(s.labels, r.labels) and s.N == r.N            # superpixels do not depend on the network
    assert s.X.shape == r.X.shape and s.X.shape[1] == 514
    assert np.array_equal(s.X[:, 512:], r.X[:, 512:])
    err = _descriptor_error(s.X, r.X)
    # element-wise view (informational): relative error of every element that is at least 1e-3 of its descriptor's largest
    a, b = s.X[:, :512].astype(np.float64), r.X[:, :512].astype(np.float64)
    big = np.abs(b) >= 1e-3 * np.abs(b).max(axis=1, keepdims=True)
    elem = float((np.abs(a - b)[big] / np.abs(b)[big]).max())
    diff_cluster = int((s.cluster != r.cluster).sum())
    diff_road = int((s.road != r.road).sum())
    diff_assign = int((s.assign != r.assign).sum())
    print('%s / %s weights: %d descriptors, worst per-descriptor relative error %.2e (median %.2e), worst element (>= 1e-3 of its '
          'row) %.2e; superpixels assigned differently %d; label-map pixels that differ: cluster %d, road %d of %d'
          % (arch, weights, s.N, float(err.max()), float(np.median(err)), elem, diff_assign, diff_cluster, diff_road, s.cluster.size))
    assert float(err.max()) <= TOL, float(err.max())
    assert diff_assign == 0 and diff_cluster == 0 and diff_road == 0


def test_default_descriptors_against_a_float32_pytorch_cpu_forward(mods, orc, synth):
    """No library kernel on the reference side at all: float32 PyTorch-CPU forward of the UNFOLDED module (BatchNorm with
    Chainer's eps), the oracle's superpixels and pooling on that map — against the default pipeline's descriptors."""
    imgs = synth.synth_batch([21], H, W)
    s = _run(mods, 'drn_d_22', None, imgs, split=True)
    ref_model = mods.drn.create_drn('drn_d_22', device='cpu', fold_bn=False)      # same seed, same weights
    with torch.no_grad():
        ref = ref_model.forward_maps(mods.drn.DRN.normalise(torch.from_numpy(imgs)))[7].numpy()
    sp = orc.slic(imgs[0], 200)
    assert np.array_equal(s.labels[0].astype(np.int64), sp)
    n = int(sp.max()) + 1
    Xr = orc.mean_pool(ref[0], sp, 'nearest', n)
    err = _descriptor_error(s.X, np.asarray(Xr))
    print('default path vs float32 PyTorch-CPU: %d descriptors, worst per-descriptor relative error %.2e (median %.2e)'
          % (n, float(err.max()), float(np.median(err))))
    assert float(err.max()) <= TOL, float(err.max())


def _pool64(fmap64, labels, n):
    """float64 mean pooling with nearest sampling (notebooks/Superpixel_Align.ipynb cell 4 on the map sampled at (y // 8, x // 8)):
    X[s, c] = sum_cells count[s, cell] * F[c, cell] / pixels(s), everything in float64 (the yardstick, not the oracle's float32 sums)."""
    import scipy.sparse as sp
    C, fh, fw = fmap64.shape
    Hh, Ww = labels.shape
    yy, xx = np.meshgrid(np.arange(Hh) * fh // Hh, np.arange(Ww) * fw // Ww, indexing='ij')
    cell = (yy * fw + xx).ravel()
    M = sp.coo_matrix((np.ones(cell.size), (labels.ravel().astype(np.int64), cell)), shape=(n, fh * fw)).tocsr()
    tot = np.asarray(M.sum(axis=1)).ravel()
    return np.asarray(M @ fmap64.reshape(C, fh * fw).T) / tot[:, None]


@pytest.mark.parametrize('arch', ['drn_d_22', 'drn_c_26'])
def test_elementwise_descriptor_error_against_a_float64_network(mods, synth, tmp_path, arch):
    """north_star says "pooled feature vectors within 1e-4 relative" without naming the norm.  Per descriptor
    (max norm) the split planes are 1e-5 from float32 instructions (test above); ELEMENT by element (elements >= 1e-3 of their row) the
    two differ by up to 2e-4 on hostile weights.  Whose error is that?  A float64 forward of the same hostile float32 weights
    (unfolded BatchNorm, PyTorch CPU, no library convolution in float32 anywhere) pooled in float64 is the yardstick: the
    element-wise error of (a) the split planes and (b) float32 matrix instructions against it, full size.  Asserted: (a) <= 2 x (b)
    + 1e-6 — the element-wise deviation between the two arithmetics is float32's own rounding noise on small elements of a
    descriptor, not something the planes add."""
    path = _hostile_pth(mods.drn, arch, str(tmp_path / (arch + '-hostile.pth')))
    imgs = synth.synth_batch([31], H, W)
    s = _run(mods, arch, path, imgs, split=True)
    r = _run(mods, arch, path, imgs, split=False)
    assert np.array_equal(s.labels, r.labels) and s.N == r.N
    ref_model = mods.drn.DRN(arch, bn_eps=mods.drn.CHAINER_BN_EPS, with_fc=True).double()
    ref_model.load_state_dict({k: v.double() for k, v in torch.load(path).items()}, strict=False)
    ref_model.eval()
    with torch.no_grad():
        f64 = ref_model.forward_maps(mods.drn.DRN.normalise(torch.from_numpy(imgs)).double())[7][0].numpy()
    X64 = _pool64(f64, s.labels[0], s.N)
    big = np.abs(X64) >= 1e-3 * np.abs(X64).max(axis=1, keepdims=True)
    out = {}
    for name, X in (('split planes', s.X), ('float32 instructions', r.X)):
        d = np.abs(X[:, :512].astype(np.float64) - X64)
        out[name] = (float((d[big] / np.abs(X64)[big]).max()), float((d.max(axis=1) / np.abs(X64).max(axis=1)).max()))
    print('%s hostile weights, 1024 x 2048, %d descriptors, against the float64 network: worst ELEMENT (>= 1e-3 of its row) relative '
          'error: split planes %.2e, float32 instructions %.2e; worst per-descriptor (max norm): %.2e / %.2e'
          % (arch, s.N, out['split planes'][0], out['float32 instructions'][0], out['split planes'][1], out['float32 instructions'][1]))
    assert out['split planes'][0] <= 2.0 * out['float32 instructions'][0] + 1e-6, out
    assert out['split planes'][1] <= TOL
