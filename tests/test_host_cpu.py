"""CPU suite, part 3: host logic of the path — DRN module structure and numerics against the
reference's PyTorch definition (golden maps), the reference batch loop / sharding rules, the
N > 1 record exchange over gloo (world_size 2), output helpers and the result.json summary."""
import importlib
import json
import os
import subprocess
import sys
import zlib

import numpy as np
import pytest
import torch

from conftest import ROOT, golden

drn = importlib.import_module('superpixel-align_amd.drn')
dist = importlib.import_module('superpixel-align_amd.dist')
cli = importlib.import_module('superpixel-align_amd.cli')


def det_fill(model):
    """Same name-keyed deterministic weights as oracle/gen_golden_drn.py."""
    with torch.no_grad():
        for name, t in list(model.named_parameters()) + list(model.named_buffers()):
            if name.endswith('num_batches_tracked'):
                continue
            g = torch.Generator().manual_seed(zlib.crc32(name.encode()) & 0x7fffffff)
            if name.endswith('running_var'):
                t.copy_(torch.rand(t.shape, generator=g) * 0.5 + 0.75)
            elif name.endswith('running_mean'):
                t.copy_(torch.randn(t.shape, generator=g) * 0.1)
            elif t.ndim == 4:
                fan = t.shape[1] * t.shape[2] * t.shape[3]
                t.copy_(torch.randn(t.shape, generator=g) * (2.0 / fan) ** 0.5)
            elif name.endswith('weight'):
                t.copy_(torch.rand(t.shape, generator=g) * 0.5 + 0.75)
            else:
                t.copy_(torch.randn(t.shape, generator=g) * 0.1)


@pytest.mark.parametrize('name', ['drn_c_26', 'drn_d_22'])
def test_drn_matches_reference_definition(name):
    g = golden('drn_maps')
    m = drn.DRN(name)
    assert sum(p.numel() for p in m.parameters()) == int(g[name + '_nparams'])
    det_fill(m)
    x = torch.from_numpy(g['x'])
    keep = x.clone()
    _, maps = m.batch_predict(x)                 # un-folded BatchNorm, CPU float32
    assert torch.equal(x, keep)                  # the input batch is never modified
    assert len(maps) == 8                        # Chainer map convention for both archs
    assert [list(t.shape) for t in maps] == g[name + '_shapes'].tolist()
    np.testing.assert_allclose([float(t.double().mean()) for t in maps], g[name + '_means'], rtol=1e-4, atol=1e-6)
    scale = float(np.abs(g[name + '_map7']).max())
    # 24 float32 convolutions deep: accumulation order differs between conv back ends
    np.testing.assert_allclose(maps[7].numpy(), g[name + '_map7'], rtol=1e-4, atol=1e-4 * scale)
    # folding BatchNorm into the convolutions is an identity in eval mode (to float32 rounding)
    m.fold_batchnorm()
    assert not any(isinstance(mod, torch.nn.BatchNorm2d) for mod in m.modules())
    _, folded = m.batch_predict(x)
    np.testing.assert_allclose(folded[7].numpy(), g[name + '_map7'], rtol=1e-4, atol=1e-4 * scale)
    # sub-batching does not change results
    _, sub = m.batch_predict(x, sub_batch=1)
    np.testing.assert_allclose(sub[7].numpy(), folded[7].numpy(), rtol=1e-4, atol=1e-4 * scale)


def test_drn_state_dict_names_match_upstream_checkpoint_layout():
    keys = set(drn.DRN('drn_c_26').state_dict())
    for k in ('conv1.weight', 'bn1.running_var', 'layer1.0.conv1.weight', 'layer2.0.downsample.0.weight',
              'layer2.0.downsample.1.bias', 'layer6.1.bn2.weight', 'layer8.0.conv2.weight'):
        assert k in keys
    keys = set(drn.DRN('drn_d_22').state_dict())
    for k in ('layer0.0.weight', 'layer0.1.running_mean', 'layer1.0.weight', 'layer2.1.bias',
              'layer3.0.downsample.0.weight', 'layer8.0.weight'):
        assert k in keys


def chainer_npz_arrays(model):
    """The arrays chainer.serializers.save_npz would write for the reference's model of the same architecture,
    built from Chainer's published naming rules, NOT from the module's own state_dict keys: a Chain names children
    by attribute; the reference's Sequential is a ChainList (models/sequential.py:9, :272-290) whose add_link
    numbers links only (activation functions take no number); Convolution2D holds W (+ b), BatchNormalization
    gamma, beta and the persistents avg_mean, avg_var, N."""
    import torch.nn as nn
    out = {}

    def link(prefix, m):
        if isinstance(m, nn.Conv2d):
            out[prefix + '/W'] = m.weight.detach().numpy()
            if m.bias is not None:
                out[prefix + '/b'] = m.bias.detach().numpy()
        elif isinstance(m, nn.BatchNorm2d):
            out[prefix + '/gamma'] = m.weight.detach().numpy()
            out[prefix + '/beta'] = m.bias.detach().numpy()
            out[prefix + '/avg_mean'] = m.running_mean.numpy()
            out[prefix + '/avg_var'] = m.running_var.numpy()
            out[prefix + '/N'] = np.array(0)
        elif isinstance(m, nn.Sequential):                      # ChainList: links only, numbered as inserted
            k = 0
            for child in m:
                if any(True for _ in child.parameters()):
                    link('%s/%d' % (prefix, k), child)
                    k += 1
        else:                                                   # Chain (BasicBlock): attribute names
            for name, child in m.named_children():
                if any(True for _ in child.parameters()):
                    link(prefix + '/' + name, child)
    for name, child in model.named_children():
        if any(True for _ in child.parameters()):
            link(name, child)
    return out


def test_chainer_npz_loader(tmp_path):
    """load_chainer_npz against keys written the way Chainer names them.  DRN-C-26 / DRN-D-22 plus a DRN-D stem
    with TWO convolutions per plain layer, where ChainList numbering (0,1,2,3) and nn.Sequential numbering
    (0,1,3,4) part ways."""
    import torch.nn as nn
    for arch in ('drn_c_26', 'drn_d_22'):
        src = drn.DRN(arch)
        det_fill(src)
        arrays = chainer_npz_arrays(src)
        if arch == 'drn_d_22':
            assert 'layer0/0/W' in arrays and 'layer0/1/avg_var' in arrays and 'layer3/0/downsample/1/gamma' in arrays
            assert 'layer8/1/beta' in arrays and not any(k.startswith('layer0/2') for k in arrays)
        path = str(tmp_path / (arch + '.npz'))
        np.savez(path, **arrays)
        dst = drn.DRN(arch).load_chainer_npz(path)
        for k, v in src.state_dict().items():
            if not k.endswith('num_batches_tracked'):
                assert torch.equal(v, dst.state_dict()[k]), k
    # two convolutions in a plain layer: conv, bn, relu, conv, bn, relu
    def widen(m):
        m._cin = 16
        m.layer1 = m._plain(16, 2, 1, 1, 2e-5)
        return m
    src = widen(drn.DRN('drn_d_22'))
    det_fill(src)
    arrays = chainer_npz_arrays(src)
    assert 'layer1/2/W' in arrays and 'layer1/3/gamma' in arrays and 'layer1/4/W' not in arrays
    assert 'layer1.3.weight' in src.state_dict() and isinstance(src.layer1[2], nn.ReLU)
    path = str(tmp_path / 'two.npz')
    np.savez(path, **arrays)
    dst = widen(drn.DRN('drn_d_22')).load_chainer_npz(path)
    for k, v in src.state_dict().items():
        if not k.endswith('num_batches_tracked'):
            assert torch.equal(v, dst.state_dict()[k]), k
    np.savez(path, **dict(arrays, **{'layer1/4/W': arrays['layer1/2/W']}))
    with pytest.raises(KeyError):
        widen(drn.DRN('drn_d_22')).load_chainer_npz(path)


def test_batch_loop_and_sharding_rules():
    # batch_spalign_kmeans.py:538-544 — the last batch is shifted back to keep its size
    assert dist.batch_ranges(0, 100, 30) == [(0, 30), (30, 60), (60, 90), (70, 100)]
    assert dist.batch_ranges(0, 90, 30) == [(0, 30), (30, 60), (60, 90)]
    assert dist.batch_ranges(38, 76, 30) == [(38, 68), (46, 76)]
    assert dist.batch_ranges(0, 10, 30) == [(-20, 10)]              # reference quirk kept
    # utils/create_random300_labels.sh:37-51 — step = 300/8+1 = 38
    got = [dist.shard_range(300, 8, r) for r in range(8)]
    assert got == [(0, 38), (38, 76), (76, 114), (114, 152), (152, 190), (190, 228), (228, 266), (266, 300)]
    assert [dist.shard_range(10, 4, r) for r in range(4)] == [(0, 3), (3, 6), (6, 9), (9, 10)]
    bal = [dist.shard_range(10, 4, r, balanced=True) for r in range(4)]
    assert bal == [(0, 3), (3, 6), (6, 8), (8, 10)]


_WORKER = r'''
import importlib, os, sys, numpy as np
sys.path.insert(0, %r)
dist = importlib.import_module('superpixel-align_amd.dist')
rank, ws, local = dist.init('gloo')
lo, hi = dist.shard_range(11, ws, rank)
rec = np.zeros((hi - lo, dist.RECORD_WIDTH), np.int64)
rec[:, 0] = np.arange(lo, hi); rec[:, 4] = 100 * rank + np.arange(hi - lo)
allrec = dist.gather_records(rec)
mx = dist.max_over_ranks(1.5 + rank)
vals = dist.all_values(10.0 * rank + 0.25)          # bench.py's per-rank rates / gather times / device ids
if rank == 0:
    np.save(os.environ['OUT'], allrec)
    open(os.environ['OUT'] + '.max', 'w').write(str(mx))
    open(os.environ['OUT'] + '.vals', 'w').write(repr(vals))
dist.barrier()
'''


def test_two_rank_gloo_gather(tmp_path):
    """N > 1 path on CPU: 2 processes, gloo, uneven shard sizes, one all_gather of records."""
    script = tmp_path / 'worker.py'
    script.write_text(_WORKER % ROOT)
    out = str(tmp_path / 'rec.npy')
    env = dict(os.environ, OUT=out, MASTER_ADDR='127.0.0.1', MASTER_PORT='29577')
    procs = []
    for r in range(2):
        e = dict(env, RANK=str(r), WORLD_SIZE='2', LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=e))
    for p in procs:
        assert p.wait(timeout=120) == 0
    rec = np.load(out)
    assert rec[:, 0].tolist() == list(range(11))                      # rank order == index order
    assert rec[:6, 4].tolist() == list(range(6)) and rec[6:, 4].tolist() == [100 + i for i in range(5)]
    assert float(open(out + '.max').read()) == 2.5
    assert eval(open(out + '.vals').read()) == [0.25, 10.25]           # one value per rank, in rank order, on every rank


def test_output_helpers_and_summary(tmp_path, orc):
    rs = np.random.RandomState(1)
    a = rs.randint(0, 4, size=(7, 9)).astype(np.uint8)
    up = cli.resize_nearest(a, (14, 27))
    assert np.array_equal(up[::2, ::3], a) and np.array_equal(up[1::2, 2::3], a)
    gt = orc.create_label_mask(rs.randint(0, 12, size=(40, 50)).astype(np.uint8))
    assert np.array_equal(cli.create_label_mask(rs.randint(0, 2, size=(2, 2))), [[-1, -1], [-1, -1]])
    pred = (rs.uniform(size=(40, 50)) < 0.5).astype(np.uint8)
    sc, ref = cli.score(pred, gt), orc.confusion(pred, gt)
    for k in ('TP', 'FP', 'FN', 'road_iou', 'non_road_iou', 'precision', 'recall'):
        assert sc[k] == ref[k]
    # result.json round trip through utils/mean_result.py
    import types
    args = types.SimpleNamespace(out_dir=str(tmp_path), n_clusters=2)
    cli.save_npy(args, 'x/aachen_000000_000019_leftImg8bit.png', pred, pred * 3)
    assert np.load(tmp_path / 'aachen_000000_000019_leftImg8bit.npy').dtype == np.uint8
    assert np.load(tmp_path / 'aachen_000000_000019_leftImg8bit_all_cluster.npy').max() == 3
    with open(tmp_path / 'result.json', 'w') as fp:
        for i in range(3):
            line = cli.result_line(args, 'img%d.png' % (i % 2), 'lab.png', sc, {'time_kmeans': 0.1}, 0.0)
            print(json.dumps(line), file=fp)
    sys.path.insert(0, os.path.join(ROOT, 'utils'))
    mean_result = importlib.import_module('mean_result')
    s, text = mean_result.summarise(str(tmp_path / 'result.json'))
    assert abs(s['Road mean IoU'] - sc['road_iou']) < 1e-12 and 'N\t:2' in text      # de-duplicated
    s2, text2 = mean_result.summarise(str(tmp_path / 'result.json'), count_duplicated=True)
    assert 'N\t:3' in text2


def test_cli_flags_match_reference_defaults():
    a = cli.get_args(['--out_dir', '/tmp/spa_cli_test'])
    assert (a.gpu, a.superpixel_method, a.n_clusters, a.n_anchors, a.n_neighbors) == (0, 'felzenszwalb', 4, 10, 4)
    assert (a.resize_shape, a.batchsize, a.n_slic_segments, a.use_feature_maps) == ((224, 224), 30, 100, [7])
    assert (a.y_rel_pos, a.x_rel_pos, a.y_rel_sigma, a.x_rel_sigma) == (0.75, 0.5, 0.1, 0.1)
    assert a.camera_param_dir == 'data/camera' and a.horizontal_line_filtering is False
    b = cli.get_args_labelfree([])
    # --gpu: the reference's default is -1 (its NumPy path, utils/apply_spalign_kmeans.py:84); this build has no CPU path, so
    # the script run with default arguments takes device 0 — create_model still refuses an explicit negative id
    assert (b.gpu, b.label_shape, b.img_list_fn) == (0, [1024, 2048], 'data/demoVideo_fns.txt')
    assert cli.get_args_labelfree(['--gpu', '-1']).gpu == -1


def test_label_archive_matches_find_zip_pipeline(tmp_path):
    """README.md:135 `find <dir> -name "*leftImg8bit.npy" | zip -0r <zip> -@` vs cli.write_label_zip:
    same member names, stored (not deflated), same payload bytes and CRCs."""
    import shutil
    import zipfile
    if shutil.which('zip') is None:
        pytest.skip('zip binary not available')
    out = tmp_path / 'results' / 'estimated_train_labels'
    out.mkdir(parents=True)
    rng = np.random.RandomState(5)
    for city in ('aachen_000000_000019', 'bochum_000000_000313', 'ulm_000094_000019'):
        np.save(str(out / (city + '_leftImg8bit')), (rng.rand(16, 32) > 0.5).astype(np.uint8))
        np.save(str(out / (city + '_leftImg8bit_all_cluster')), rng.randint(0, 4, (16, 32)).astype(np.uint8))
    (out / 'result.json').write_text('{}\n')
    cwd = os.getcwd()
    os.chdir(str(tmp_path))
    try:
        subprocess.check_call('find results/estimated_train_labels -name "*leftImg8bit.npy" | '
                              'zip -0r ref.zip -@ > /dev/null', shell=True)
        n = cli.write_label_zip('results/estimated_train_labels', 'ours.zip')
    finally:
        os.chdir(cwd)
    assert n == 3
    with zipfile.ZipFile(str(tmp_path / 'ref.zip')) as a, zipfile.ZipFile(str(tmp_path / 'ours.zip')) as b:
        assert sorted(a.namelist()) == sorted(b.namelist())
        for name in a.namelist():
            ia, ib = a.getinfo(name), b.getinfo(name)
            assert ia.compress_type == ib.compress_type == zipfile.ZIP_STORED
            assert ia.CRC == ib.CRC and ia.file_size == ib.file_size
            assert a.read(name) == b.read(name)
            assert not name.endswith('_all_cluster.npy')


# --------------------------------------------------------------------------- the labelled driver, N ranks
_CLI_WORKER = r'''
import importlib, os, sys, time, numpy as np, torch
sys.path.insert(0, %r)
cli = importlib.import_module('superpixel-align_amd.cli')


class StubResult(object):
    def __init__(self, imgs):
        # "road" = bright pixels of the red channel; three fake clusters
        self.road = (imgs[:, 0] > 127).astype(np.uint8)
        self.cluster = np.where(self.road == 1, 0, 1 + (imgs[:, 1] > 127)).astype(np.uint8)
        self.info = torch.tensor([3, 0, 7, 0], dtype=torch.int32)
        self.n_labels = torch.full((imgs.shape[0],), 5, dtype=torch.int32)

    def masks_to_host(self):
        return self.cluster, self.road


class StubPipe(object):
    """Stands in for LabelPipeline (no GPU here): the host logic around it is what is under test."""
    def __init__(self, args, model, engine):
        self.fail_at = int(os.environ.get('FAIL_AT_BATCH', '-1'))
        self.n = 0

    def run(self, imgs):
        if self.n == self.fail_at:
            raise RuntimeError('injected failure in batch %%d' %% self.n)
        self.n += 1
        return StubResult(imgs)

    def elapsed_times(self):
        return {'time_superpixel': 0.25, 'time_roialign': 0.5, 'time_prior': 0.0, 'time_kmeans': 0.125,
                'time_feature_maps': 1.0}


sys.exit(cli.main_labelled(sys.argv[1:], make_pipe=StubPipe, make_model=lambda a: None))
'''


def _make_png_dataset(tmp_path, n, H=16, W=24):
    from PIL import Image
    rs = np.random.RandomState(11)
    imgs, labs = [], []
    for i in range(n):
        fn = tmp_path / ('city_%06d_000019_leftImg8bit.png' % i)
        Image.fromarray(rs.randint(0, 256, size=(H, W, 3)).astype(np.uint8)).save(str(fn))
        ln = tmp_path / ('city_%06d_000019_gtFine_labelIds.png' % i)
        Image.fromarray(rs.randint(0, 12, size=(H, W)).astype(np.uint8)).save(str(ln))
        imgs.append(str(fn)); labs.append(str(ln))
    (tmp_path / 'imgs.txt').write_text('\n'.join(imgs) + '\n')
    (tmp_path / 'labels.txt').write_text('\n'.join(labs) + '\n')
    return imgs, labs


def _run_cli_ranks(tmp_path, ws, extra, port, env_extra=None):
    script = tmp_path / 'cli_worker.py'
    script.write_text(_CLI_WORKER % ROOT)
    out = tmp_path / ('out_ws%d_%d' % (ws, port))
    argv = ['--img_file_list', str(tmp_path / 'imgs.txt'), '--label_file_list', str(tmp_path / 'labels.txt'),
            '--out_dir', str(out), '--resize_shape', '16', '24', '--no_figure', '--io_threads', '2'] + extra
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), SPA_DIST_BACKEND='gloo',
               **(env_extra or {}))
    procs = []
    for r in range(ws):
        e = dict(env, RANK=str(r), WORLD_SIZE=str(ws), LOCAL_RANK=str(r)) if ws > 1 else env
        procs.append(subprocess.Popen([sys.executable, str(script)] + argv, env=e,
                                      stdout=subprocess.DEVNULL, stderr=subprocess.PIPE))
    rcs = [p.wait(timeout=180) for p in procs]
    lines = []
    if (out / 'result.json').exists():
        lines = [json.loads(l) for l in (out / 'result.json').read_text().splitlines()]
    return rcs, lines, out, [p.stderr.read().decode() for p in procs]


@pytest.mark.parametrize('ws,extra,n', [(2, [], 11), (4, [], 11), (4, ['--balanced'], 11), (4, [], 3)])
def test_labelled_driver_under_n_ranks_gloo(tmp_path, ws, extra, n):
    """cli.main_labelled end to end on N gloo ranks with a stub pipeline: the reference partition
    (step = n // N + 1, so 3 images on 4 ranks leave rank 3 with an EMPTY shard), --balanced, one
    gather, rank 0 writes result.json in index order with one schema for every row."""
    imgs, labs = _make_png_dataset(tmp_path, n)
    rcs, lines, out, err = _run_cli_ranks(tmp_path, ws, ['--batchsize', '2'] + extra, 29580 + ws + len(extra) + n)
    assert rcs == [0] * ws, err
    # every image once per batch it appears in; shards in rank order -> indices non-decreasing
    idx = [imgs.index(l['img_fn']) for l in lines]
    assert idx == sorted(idx) and set(idx) == set(range(n))
    shards = [dist.shard_range(n, ws, r, balanced=bool(extra)) for r in range(ws)]
    expect = []
    for s, e in shards:
        for lo, hi in (dist.batch_ranges(s, e, 2) if e > s else []):
            expect += list(range(n))[max(lo, 0):hi]
    assert sorted(expect) == idx                                   # incl. the shifted-back duplicates
    keys = set(lines[0])
    assert {'img_fn', 'label_fn', 'road_iou', 'TP', 'FP', 'FN', 'time_superpixel', 'time_roialign',
            'time_prior', 'time_kmeans', 'elapsed_time', 'gpu', 'n_clusters'} <= keys
    assert all(set(l) == keys for l in lines)                      # rows of other ranks: same schema
    assert all(l['time_roialign'] == 0.5 for l in lines)
    owner = {i: r for r, (s, e) in enumerate(shards) for i in range(s, e)}
    if n > ws:
        assert all(l['gpu'] == owner[imgs.index(l['img_fn'])] for l in lines)
    # scores match a host recomputation, and every image has its two .npy files
    from PIL import Image
    for l in lines:
        im = np.asarray(Image.open(l['img_fn']))
        gt = cli.create_label_mask(np.asarray(Image.open(l['label_fn'])))
        sc = cli.score((im[:, :, 0] > 127).astype(np.uint8), gt)
        assert (l['TP'], l['FP'], l['FN']) == (sc['TP'], sc['FP'], sc['FN'])
        base = os.path.splitext(os.path.basename(l['img_fn']))[0]
        assert np.load(out / (base + '.npy')).dtype == np.uint8 and (out / (base + '_all_cluster.npy')).exists()


def test_labelled_driver_keeps_finished_images_on_failure(tmp_path):
    """A failure in batch 2 must not lose the result.json lines of batches 0 and 1 (the reference
    appends one line per image as it goes, batch_spalign_kmeans.py:407-422)."""
    imgs, labs = _make_png_dataset(tmp_path, 8)
    rcs, lines, out, err = _run_cli_ranks(tmp_path, 1, ['--batchsize', '3'], 29571, {'FAIL_AT_BATCH': '2'})
    assert rcs[0] != 0 and 'injected failure' in err[0]
    assert [imgs.index(l['img_fn']) for l in lines] == [0, 1, 2, 3, 4, 5]


def test_result_lines_from_concurrent_processes_do_not_interleave(tmp_path):
    """The bash launchers start N processes that append to ONE result.json: every line must be
    written by a single write() on an O_APPEND descriptor."""
    path = str(tmp_path / 'result.json')
    code = ("import importlib,sys;sys.path.insert(0,%r);cli=importlib.import_module('superpixel-align_amd.cli');"
            "[cli.append_result_line(%r,{'p':int(sys.argv[1]),'i':i,'pad':'x'*9000}) for i in range(200)]"
            % (ROOT, path))
    procs = [subprocess.Popen([sys.executable, '-c', code, str(p)]) for p in range(4)]
    assert [p.wait(timeout=120) for p in procs] == [0] * 4
    rows = [json.loads(l) for l in open(path)]                     # any torn line fails to parse
    assert len(rows) == 800 and sorted((r['p'], r['i']) for r in rows) == [(p, i) for p in range(4) for i in range(200)]


def test_eight_ranks_label_the_full_training_split(tmp_path):
    """BASELINE configs[3] as far as a box without GPUs can run it: cli.main_labelled on EIGHT gloo ranks over
    n_data = 2975 (the Cityscapes train split; the files are links to one small image) with the reference batch size 30.
    Every rank labels the range the reference launcher gives its GPU — step = 2975 // 8 + 1 = 372
    (utils/create_random300_labels.sh:37-51) — the per-image records cross ranks in one all_gather and rank 0 writes
    result.json in index order, the shifted-back last batch of every range included (batch_spalign_kmeans.py:538-544)."""
    from PIL import Image
    n, ws = 2975, 8
    rs = np.random.RandomState(3)
    src_i, src_l = tmp_path / 'src_img.png', tmp_path / 'src_lab.png'
    Image.fromarray(rs.randint(0, 256, size=(16, 24, 3)).astype(np.uint8)).save(str(src_i))
    Image.fromarray(rs.randint(0, 12, size=(16, 24)).astype(np.uint8)).save(str(src_l))
    data = tmp_path / 'data'
    data.mkdir()
    imgs, labs = [], []
    for i in range(n):
        fn, ln = data / ('city_%06d_000019_leftImg8bit.png' % i), data / ('city_%06d_000019_gtFine_labelIds.png' % i)
        os.symlink(str(src_i), str(fn)); os.symlink(str(src_l), str(ln))
        imgs.append(str(fn)); labs.append(str(ln))
    (tmp_path / 'imgs.txt').write_text('\n'.join(imgs) + '\n')
    (tmp_path / 'labels.txt').write_text('\n'.join(labs) + '\n')
    rcs, lines, out, err = _run_cli_ranks(tmp_path, ws, ['--batchsize', '30'], 29641)
    assert rcs == [0] * ws, err
    shards = [dist.shard_range(n, ws, r) for r in range(ws)]
    assert shards == [(372 * r, min(372 * (r + 1), n)) for r in range(ws)] and shards[-1] == (2604, 2975)
    expect = []
    for s, e in shards:
        for lo, hi in dist.batch_ranges(s, e, 30):
            expect += list(range(lo, hi))
    pos = {fn: i for i, fn in enumerate(imgs)}
    idx = [pos[l['img_fn']] for l in lines]
    assert idx == sorted(idx) and idx == sorted(expect) and set(idx) == set(range(n))
    assert len(idx) > n                                            # 372 = 12 x 30 + 12: every range re-labels 18 images
    owner = {i: r for r, (s, e) in enumerate(shards) for i in range(s, e)}
    assert all(l['gpu'] == owner[pos[l['img_fn']]] for l in lines)
    keys = set(lines[0])
    assert all(set(l) == keys for l in lines)
    assert len(list(out.glob('*_leftImg8bit.npy'))) == n


def test_graph_replay_bookkeeping_and_eligibility(monkeypatch):
    """The host side of the captured-graph forward (drn.py): the FLOP / byte counters bench.py prices the DRN with advance by one
    forward's worth per replay (snapshot / delta / add), and the eligibility rule reads SPA_DRN_GRAPH and the pixel bound."""
    E = drn._EPILOGUE
    saved = drn._epi_snapshot()
    try:
        before = drn._epi_snapshot()
        E['gemm16_flops'] += 5.0
        E['library_convs'] += 2
        drn._c16('front', 7.0, 11.0, 3)
        delta = drn._epi_delta(before)
        assert delta['gemm16_flops'] == 5.0 and delta['library_convs'] == 2 and delta['c16']['front'] == [7.0, 11.0, 3]
        assert isinstance(E['split_gemm'], bool) and 'split_gemm' not in delta and 'engine' not in delta      # settings are not counters
        drn._epi_restore(before)
        assert drn._epi_delta(before)['gemm16_flops'] == 0.0 and E['c16'].get('front', [0, 0, 0])[2] == before['c16'].get('front', [0, 0, 0])[2]
        drn._epi_add(delta)
        drn._epi_add(delta)
        assert drn._epi_delta(before)['gemm16_flops'] == 10.0 and drn._epi_delta(before)['c16']['front'] == [14.0, 22.0, 6]
    finally:
        drn._epi_restore(saved)

    class FakeX(object):
        def __init__(self, shape, cuda):
            self.shape, self.is_cuda = shape, cuda
    small, big = FakeX((30, 3, 224, 224), True), FakeX((30, 3, 1024, 2048), True)
    monkeypatch.delenv('SPA_DRN_GRAPH', raising=False)
    monkeypatch.delenv('SPA_DRN_GRAPH_PIXELS', raising=False)
    assert drn._graph_wanted(small) and not drn._graph_wanted(big) and not drn._graph_wanted(FakeX((1, 3, 8, 8), False))
    monkeypatch.setenv('SPA_DRN_GRAPH', '0')
    assert not drn._graph_wanted(small)
    monkeypatch.setenv('SPA_DRN_GRAPH', '1')
    assert drn._graph_wanted(big)
    monkeypatch.delenv('SPA_DRN_GRAPH')
    monkeypatch.setenv('SPA_DRN_GRAPH_PIXELS', '1000')
    assert not drn._graph_wanted(small)
