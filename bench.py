#!/usr/bin/env python
"""bench.py — images/sec of end-to-end superpixel-align labelling on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic 1024x2048 RGB images: DRN features
(libspalign's own MFMA kernels) -> HIP SLIC -> per-superpixel pooling -> location prior -> weighted
k-means -> painted road masks -> per-image confusion counts (BASELINE.json configs[1]: DRN-D-22 fp32,
SLIC 200 superpixels, k = 2).  Three distinct batches rotate through the steps.  With N > 1 every rank labels its own batches (images shard with no
data-path collective; weak scaling) and the per-image score records are exchanged by one RCCL
all_gather at the end of the timed region, the native replacement of the reference's shared
result.json append.

Prints ONE JSON line on rank 0 (see README/DESIGN for the fields).  `value` / `ms_per_step` are SURVEY.md
8d's region — the reference's own step (batch_spalign_kmeans.py:427-458: H->D at :431, D->H at :355-357):
batches of decoded 8-bit images in pinned host memory -> cluster + road masks in pinned host memory
through pipeline.HostStream (uploads and downloads double buffered on copy streams under the kernels).
`device_resident_value` / `device_resident_ms_per_step` time the same K steps with the batches already in
HBM (a loop of its own, run first).  Both loops run the pipeline as the drivers do (LabelPipeline's default): the superpixel
branch on a second stream beside the front of the DRN forward; `--one_stream` puts every kernel on its own.  `kernels` holds one roofline-shaped entry per hand-written kernel
family of libspalign (HIP events on the launch stream, recorded during the headline loop): HBM-bound ones against 8 TB/s
with their algorithmic bytes, the float32-MFMA stem against the fp32 matrix peak; `roofline` is the
family with the most time per step, whichever it is.  `drn` reports the matrix side of the step (libspalign's own
convolutions; none is MIOpen's in the float32 forward).  `cpu_baseline` times, on the host cores, the C
restatement (oracle, single thread and all cores with images sharded over threads), the same DRN
through PyTorch-CPU and BASELINE.json's named comparator (NumPy pooling + scipy.cluster.vq.kmeans2)
on a bounded sample, rank 0 at N = 1 only.
"""
import argparse
import importlib
import json
import os
import sys
import time
import types

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# MIOpen's exhaustive find (cudnn.benchmark) otherwise also times its naive reference convolution
# on every layer shape: ~100 s of start-up per process at 1024x2048 for a solver that never wins
os.environ.setdefault('MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD', '0')

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
FP32_MATRIX_PEAK_TF = 157.3    # fp32 MFMA peak; bf16 dense 2500
BF16_MATRIX_PEAK_TF = 2500.0


def parse():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=6)
    p.add_argument('--warmup', type=int, default=2)
    p.add_argument('--batch', type=int, default=30, help='images per step per GPU (reference batchsize)')
    p.add_argument('--height', type=int, default=1024)
    p.add_argument('--width', type=int, default=2048)
    p.add_argument('--arch', default='drn_d_22', choices=['drn_d_22', 'drn_c_26'])
    p.add_argument('--dtype', default='fp32', choices=['fp32', 'bf16'])
    p.add_argument('--n_slic_segments', type=int, default=200)
    p.add_argument('--superpixel_method', default='slic', choices=['slic', 'felzenszwalb'],
                   help="felzenszwalb (scale 300, sigma 0.8, min_size 20) is the reference launchers' setting")
    p.add_argument('--n_clusters', type=int, default=2)
    p.add_argument('--pool_mode', default='mean', choices=['mean', 'anchor'])
    p.add_argument('--drn_sub_batch', type=int, default=0,
                   help='DRN forward in sub-batches of this many images (0 = the whole batch at once: fewest, largest '
                        'convolution launches; 30 x 1024x2048 fp32 needs ~25 GB of activations)')
    p.add_argument('--drn_streams', type=int, default=1,
                   help='parts of the batch run on side streams inside the DRN forward (epilogue passes of one part '
                        'under the convolutions of the other): 2 shortens the forward by 2 %% but the label kernels '
                        'that follow run slower by as much (measured), so the default is 1')
    p.add_argument('--no_cpu_baseline', action='store_true')
    p.add_argument('--no_exact_fp32', action='store_true', help='skip the extra K steps on float32 matrix instructions (`exact_fp32_value`)')
    p.add_argument('--cpu_sample', type=int, default=1, help='images of the PyTorch-CPU DRN sample')
    p.add_argument('--cpu_threads', type=int, default=64,
                   help='threads (= images in flight) of the all-cores oracle row: min(host cores, this)')
    p.add_argument('--no_host_loop', action='store_true', help='skip the pinned-host to pinned-host loop')
    p.add_argument('--n_batches', type=int, default=3, help='distinct batches rotating through the steps')
    p.add_argument('--no_prof', action='store_true', help='do not record per-kernel events')
    p.add_argument('--integer_images', action='store_true', help='(default since round 4; kept for old command lines)')
    p.add_argument('--float_images', action='store_true',
                   help='non-integer synthetic float32 images (SURVEY 8d\'s generator before quantisation) instead of 8-bit-valued '
                        'ones (what a decoded PNG is): the host loop then uploads float32 planes, 12 bytes per pixel instead of 3')
    p.add_argument('--device_rng', action='store_true', help='anchor mode: draw the anchors on the device (spa_anchor_ranks_dev)')
    p.add_argument('--miopen_conv', action='store_true', help='leave the stride-1 3x3 layers to MIOpen (A/B against spa_conv3x3_bf16 / spa_conv3x3_f32)')
    p.add_argument('--winograd', type=int, default=4, choices=[2, 4], help='float32: F(4x4,3x3) (default) or F(2x2,3x3) tiles')
    p.add_argument('--fp32_mfma_gemm', action='store_true', help='float32: the Winograd GEMMs on the float32 matrix instructions '
                   '(v_mfma_f32_16x16x4_f32) instead of two half-precision planes per operand on the 16-bit ones (same accuracy, 2.7x the matrix time)')
    p.add_argument('--no_winograd', action='store_true', help='float32: direct convolution (spa_conv3x3_f32) on the 256/512-channel layers too')
    p.add_argument('--one_stream', action='store_true',
                   help='run the superpixel branch on the main stream instead of a second one beside the front of the DRN forward '
                        '(the pipeline\'s and the drivers\' default, LabelPipeline(overlap=True): +3.5-4 %% images/s in same-box A/B runs). '
                        'On one stream every label kernel\'s duration is its own; on two, the SLIC / connectivity kernels and the '
                        'stem / narrow layers they run beside stretch each other (the Winograd launches, the headline roofline entry, '
                        'are not touched: 6.08 vs 6.12 ms).')
    p.add_argument('--overlap', action='store_true', help='(the default since round 4; kept for old command lines)')
    return p.parse_args()


# what the PMC counters say holds each kernel below the HBM roofline
LIMITERS = {
    'k_slic_assign': 'VALU issue: ~20 candidate centres per pixel x ~17 separately rounded float32 operations '
                     '(SQ_INSTS_VALU x 4 cycles / SIMD = the whole launch time)',
    'k_slic_update': 'serial float32 chains (6-cycle dependent add x pixels of the largest segment) + VALU issue',
    'connectivity(all)': 'latency and issue, not bytes: run tables + strip-local union-find ~1.1 ms, then the BFS replays '
                         '(exact scan-order semantics): a replay is a chain of ~1 100-cycle steps, one wave each; the tiers '
                         'compete for LDS bytes x time and VALU issue (~1.1 ms), relabel 0.17 ms',
    'k_bias_act(all)': 'HBM bandwidth: one in-place pass over each convolution output (PMC traffic = 1.00x the algorithmic bytes); '
                       '~5 TB/s of the ~6.3 TB/s a streaming kernel reaches on this part, small layers pay the launch ramp',
    'k_kmeans': 'latency: numpy-ordered float64 sums (one serial chain per cluster and column, ~10 cycles per member row) and '
                'two grid barriers per Lloyd iteration',
    'k_rgb2lab': 'DP VALU: binary64 exp/log emulation of the float32 power and cube root (bit-defined transcendental)',
    'k_gemm_f16x3_stag<256, 256>': '16-bit MFMA pipe + its feed: the Winograd GEMMs with three half-precision products per float32 product, the two '
                                   'waves of every SIMD half a K step apart (one stages / reads / splits while the other multiplies), the split as '
                                   'single-issue mixed-precision fmas partly between the matrix instructions.  In-kernel stamps (profiles/'
                                   'r5_gemm16_stagger_stamps.txt): a half period is one wave\'s 96 matrix instructions (1 536 matrix cycles, 1 800-2 000 '
                                   'with the split riding between them) + ~350 cycles of waits and barriers, the two waves\' matrix phases back to back: '
                                   '~68 % of the cycles multiply.  Timing-only ablations: no split 3.30 ms, no global loads 3.20, neither 2.49 '
                                   '(1.4 PFLOP/s) against 3.35-3.42 as shipped (512 -> 512, 30 images); round 3\'s lock-step kernel 3.70',
    'k_conv3x3_p16': 'the staging, not the matrix pipe (round 6, csrc/spa_convp.hip; in-kernel stamps, tools/convp_stamps.py): planes built once per staged '
                     'segment in place in LDS, one barrier per group of three taps, every LDS read between matrix instructions — a group of 72 matrix '
                     'instructions per wave is ~5 000 cycles for 2 304 cycles of matrix work per SIMD: the tap that carries the 7-8 LDS-DMA instructions '
                     'per wave takes 1 450-1 900 cycles (~100 cycles each, both waves of a SIMD staging at once), the tap beside the fragment reads of '
                     'the next 840-1 650, the third runs at the matrix rate (760); matrix pipe 44-52 % busy (was 29-40 % in k_conv3x3_f32<split>, '
                     'same bits: 64 -> 64 1.32 -> 1.11 ms, 128 -> 128 1.19 -> 0.99 ms per 30 images alone); fewer staged bytes per matrix '
                     'instruction needs a 128 x 256 tile, which three segment buffers leave no LDS for',
    'k_conv3x3_f32<split>(all)': 'instruction issue, not the matrix pipe (SQ counters, profiles/r3_sq_counters_drn_split.txt): a K step of these '
                                 'narrow layers is 12-24 matrix instructions per wave next to 4-5 other vector and ~5 scalar instructions per '
                                 'matrix instruction (split of the pixels, staging addresses, scalar state spilled to vector lanes); matrix pipe busy '
                                 '20-37 % of the cycles, waves 40 % waiting; measured around it: 3 / 2 / 1 workgroups per CU 1.77 / 1.68 / 2.25 ms '
                                 '(64 -> 64), 256-pixel tiles 1.53, delayed pixel staging +5 %, K loop unrolled by the three taps -5 % on 64 '
                                 'channels but +25 % on the 256-channel tile (code size)',
    'k_conv3x3_f32<0, 256, 1, 256>': 'float32 MFMA pipe.  K = Cin is 8-16 K steps per 256 x 256 tile against 144 in the 3x3 form, so the '
                                             'tile prologue and the store of the output tile weigh more (0.83 against 0.88 of the peak) although the '
                                             'workgroups are persistent and stage the next tile before their epilogue',
    'k_wino_in': 'HBM: reads X, writes V = 2.25x X (position-major, dense rows); 5.3 TB/s on the 512-channel layers',
    'k_wino_out': 'HBM: reads M = 2.25x Y (+ the residual), writes Y; 4.9 TB/s on the 512-channel layers',
    'k_conv3x3_f32<taps 9>(all)': 'float32 MFMA pipe: 0.88-0.89 of the 157.3 TFLOP/s peak on the 256/512-channel layers (MIOpen\'s hand-written '
                          'assembly reaches 0.88 on the same box, without the epilogue); the 64/128-channel layers (a sixth of the '
                          'launches\' time) run at 0.65-0.78: a K step is short there and its barrier + load wait shows',
    'k_conv3x3_bf16(all)': 'LDS-read + MFMA loop on random operands tops out at 1 220-1 300 TFLOP/s with the global loads '
                           'compiled out (clock give-back under dense bf16 MFMA); pixel-tile loads cost the rest',
}


def pmc_traffic(kernel, B, H, W, algorithmic_bytes=None):
    """HBM bytes per launch measured with the PMC counters (committed summary), or None.  Kernels whose launches
    differ in size (the DRN epilogue) are recorded as a ratio to their algorithmic bytes."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'pmc_traffic.json')
    if (H, W) != (1024, 2048) or not os.path.exists(path):
        return None
    with open(path) as f:
        d = json.load(f)
    per_image = d['bytes_per_image_per_launch'].get(kernel)
    if per_image is not None:
        return int(per_image * B)
    ratio = d.get('ratio_to_algorithmic_bytes', {}).get(kernel)
    return int(ratio * algorithmic_bytes) if ratio is not None and algorithmic_bytes else None


def algorithmic_bytes(kernel, B, H, W, C, fh, fw, n_seg, feat_bytes):
    """Algorithmic HBM bytes per launch (DESIGN.md 'Kernels'), B images/launch."""
    px = H * W
    per_image = {
        'k_rgb2lab': px * (12 + 12),                      # f32 RGB in, f32 Lab out
        # SURVEY.md 8d prices a SLIC iteration at 16 B/pixel (Lab in, label out, centre sums on chip).  This build needs a
        # second pass per iteration for scikit-image's raster-order float32 sums (a serial chain per segment: no tile-partial
        # sums reproduce its bits), so an iteration is two launches: the 16 bytes are split between them, 8 each, and the two
        # fractions ADD UP to the iteration's share of the roofline instead of each claiming the whole budget
        'k_slic_assign': px * 8,
        'k_slic_update': px * 8,
        'connectivity(all)': px * (4 + 4),
        'segment_stats(all)': px * 4,
        'k_cell_weights': px * 4,
        'k_pool_mean': C * fh * fw * feat_bytes + n_seg * C * 4,
        'k_pool_anchor': n_seg * 10 * 4 * C * feat_bytes + n_seg * C * 8,
        'k_paint': px * (4 + 1 + 1),
        'k_drn_stem_d(+normalise)': px * (12 + 16 * feat_bytes),     # raw image in, layer1 map out
        'k_kmeans': 0,
    }
    return per_image.get(kernel, 0) * B


def stem_flops(B, H, W):
    """2 * MACs of the fused DRN-D stem: conv7x7 3->16 (K = 147) + conv3x3 16->16 (K = 144)."""
    return 2.0 * (147 + 144) * 16 * H * W * B


def make_batch(synth, B, H, W, scene=False, seed0=0, out=None, integer=False):
    """B synthetic images from 4 generated ones (rolled copies are new images for SLIC/DRN).
    scene=True: piecewise-constant scenes (what graph-based felzenszwalb needs to find regions).
    seed0: first generator seed (different batches use different seeds); out: array to fill."""
    gen = synth.synth_scene if scene else synth.synth_image
    base = [gen(seed0 + s, H, W, integer_valued=True) if integer else gen(seed0 + s, H, W) for s in range(min(4, B))]
    imgs = np.empty((B, 3, H, W), np.float32) if out is None else out
    gts = np.empty((B, H, W), np.int32)
    gt0 = [synth.synth_gt_labels(seed0 + s, H, W) for s in range(min(4, B))]
    for b in range(B):
        shift = 37 * (b // 4)
        imgs[b] = np.roll(base[b % 4], shift, axis=2)
        g = np.roll(gt0[b % 4], shift, axis=1).astype(np.int32)
        gts[b] = np.where(g <= 6, -1, np.where(g == 7, 1, 0))     # create_label_mask (:279-296)
    return imgs, gts


def cpu_baseline(a, synth, n_img):
    """SURVEY.md 8d's CPU rows on the host cores of this box, on a bounded sample of the same workload:
      drn       : the same DRN, PyTorch-CPU float32 on all cores, n_img image(s);
      port_1t   : the C restatement (oracle: Lab, SLIC, connectivity, pooling, prior, k-means, paint),
                  ONE thread, one image;
      port_all  : the same with images sharded over `cpu_threads` threads (one image per thread);
      comparator: BASELINE.json's named CPU path as far as this box has it — NumPy pooling
                  (label x cell count matrix @ feature matrix) + scipy.cluster.vq.kmeans2(k) on the
                  oracle's SLIC labels (scikit-image does not travel to the GPU box).
    `value` = images/s of the whole path on all cores = 1 / (drn s/image + port_all s/image)."""
    import torch
    from concurrent.futures import ThreadPoolExecutor
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import oracle as orc
    drn = importlib.import_module('superpixel-align_amd.drn')
    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    H, W = a.height, a.width
    nthr = max(1, min(a.cpu_threads, cores))
    args = types.SimpleNamespace(superpixel_method='slic', n_slic_segments=a.n_slic_segments,
                                 n_anchors=10, n_neighbors=4, without_pos=False, y_rel_pos=0.75,
                                 x_rel_pos=0.5, y_rel_sigma=0.1, x_rel_sigma=0.1, n_clusters=a.n_clusters)
    orc.lib()
    with ThreadPoolExecutor(nthr) as ex:
        imgs = np.stack(list(ex.map(lambda i: synth.synth_image(100 + i, H, W), range(max(n_img, nthr)))))
    model = drn.create_drn(a.arch, device='cpu', dtype=torch.float32)
    t0 = time.time()
    with torch.no_grad():
        fm = [model.batch_predict(imgs[i:i + 1])[1][7].float().numpy() for i in range(n_img)]
    t_drn = (time.time() - t0) / n_img
    fmap = fm[0][0]                                   # (C, fh, fw); reused by every image of the port rows

    def rest(i):                                      # everything behind the DRN for ONE image
        one = imgs[i:i + 1]
        sps = orc.batch_superpixel(args, one)
        feats, n_per = orc.batch_superpixel_align(args, one, sps, fmap[None], orc.PyRandom(1111),
                                                  a.pool_mode, 'nearest')
        prior = orc.batch_create_prior(args, sps)
        orc.batch_weighted_kmeans(args, sps, feats, prior, n_per)
        return sps[0]

    t0 = time.time()
    sp0 = rest(0)
    t_1t = time.time() - t0
    t0 = time.time()
    with ThreadPoolExecutor(nthr) as ex:
        list(ex.map(rest, range(nthr)))
    t_all = (time.time() - t0) / nthr                 # seconds per image with nthr images in flight

    # BASELINE.json's comparator: NumPy pooling + scipy k-means on given labels
    comp = None
    try:
        from scipy.cluster.vq import kmeans2
        t0 = time.time()
        C, fh, fw = fmap.shape
        S = int(sp0.max()) + 1
        ys = (np.arange(H) * fh // H)[:, None]
        xs = (np.arange(W) * fw // W)[None, :]
        pair = sp0.astype(np.int64) * (fh * fw) + (ys * fw + xs)
        Wt = np.bincount(pair.ravel(), minlength=S * fh * fw).reshape(S, fh * fw).astype(np.float32)
        pooled = (Wt @ fmap.reshape(C, -1).T) / Wt.sum(axis=1, keepdims=True)
        kmeans2(pooled.astype(np.float64), a.n_clusters, minit='points', seed=1111)
        comp = time.time() - t0
    except Exception as exc:                          # scipy missing or too old: report, do not fail
        comp = repr(exc)
    return {'value': round(1.0 / (t_drn + t_all), 4), 'unit': 'images/sec', 'cores': cores, 'kind': 'port',
            'sample': '%d synthetic %dx%d image(s) through PyTorch-CPU %s fp32 on %d threads; the C restatement '
                      '(SLIC %d / pool(%s) / prior / k-means / paint) on 1 image with 1 thread and on %d images '
                      'with %d threads; value = 1 / (DRN s/image + all-cores restatement s/image)'
                      % (n_img, H, W, a.arch, cores, a.n_slic_segments, a.pool_mode, nthr, nthr),
            'rows': {'drn_pytorch_cpu_s_per_image': round(t_drn, 3),
                     'port_1_thread_s_per_image': round(t_1t, 3),
                     'port_all_cores_s_per_image': round(t_all, 3), 'port_all_cores_threads': nthr,
                     'comparator_numpy_pool_scipy_kmeans2_s_per_image': comp if isinstance(comp, str) else round(comp, 3),
                     'comparator_note': 'NumPy pooling + scipy.cluster.vq.kmeans2 on the restatement\'s SLIC labels '
                                        '(scikit-image SLIC, 1.8-3.8 s/image in the build container, is not on this box)'}}


def self_launch(a):
    """`python bench.py --gpus N` with N > 1 and no launcher environment: start the N ranks ourselves, exactly as
    the driver would (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`), as a
    CHILD process — this process has not touched the GPU and never will — forward rank 0's line and exit with the
    child's code.  (The reference's equivalent is the bash fan-out of utils/create_random300_labels.sh:37-51.)"""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(a.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    return subprocess.call(cmd, env=env)


def main():
    a = parse()
    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(self_launch(a))
    import torch
    spa = importlib.import_module('superpixel-align_amd')
    dist = importlib.import_module('superpixel-align_amd.dist')
    pipeline = importlib.import_module('superpixel-align_amd.pipeline')
    drn = importlib.import_module('superpixel-align_amd.drn')

    rank, ws, local = dist.init()
    if os.environ.get('SPA_BENCH_SAME_DEVICE') == '1':
        local = 0               # test hook: several ranks on one GPU (with SPA_DIST_BACKEND=gloo)
    local = local % max(torch.cuda.device_count(), 1)       # ranks isolated by device visibility see one device each
    if ws != a.gpus:
        raise SystemExit('bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks' % (a.gpus, ws))
    torch.cuda.set_device(local)
    torch.backends.cudnn.benchmark = True
    dtype = {'fp32': torch.float32, 'bf16': torch.bfloat16}[a.dtype]

    args = types.SimpleNamespace(
        superpixel_method=a.superpixel_method, n_slic_segments=a.n_slic_segments, n_anchors=10, n_neighbors=4,
        felzenszwalb_scale=300.0, felzenszwalb_sigma=0.8, felzenszwalb_min_size=20,
        without_pos=False, y_rel_pos=0.75, x_rel_pos=0.5, y_rel_sigma=0.1, x_rel_sigma=0.1,
        gpu=local, n_clusters=a.n_clusters, use_feature_maps=[7], pool_mode=a.pool_mode,
        mean_sampling='nearest', drn_sub_batch=a.drn_sub_batch or None, drn_streams=a.drn_streams,
        device_rng=a.device_rng)
    drn._EPILOGUE['own_conv'] = drn._EPILOGUE['own_conv32'] = not a.miopen_conv
    drn._EPILOGUE['winograd'] = 0 if a.no_winograd else a.winograd
    if a.fp32_mfma_gemm:
        drn._EPILOGUE['split_gemm'] = False
    model = drn.create_drn(a.arch, device='cuda:%d' % local, dtype=dtype)
    overlap = (not a.one_stream) or a.pool_mode == 'anchor'
    pipe = pipeline.LabelPipeline(args, model, overlap=overlap)
    eng = pipe.eng

    B, H, W = a.batch, a.height, a.width
    NB = max(1, a.n_batches)
    # distinct batches in pinned host memory (the host loop's source) and their device copies.  Default: 8-bit-valued
    # images, held on the host as a PNG decoder leaves them — (B,H,W,3) uint8 — like the drivers of cli.py hold them
    integer = not a.float_images
    host, gts, dev = [], [], []
    for nb in range(NB):
        pin = torch.empty((B, 3, H, W), dtype=torch.float32).pin_memory()
        _, g = make_batch(spa.synth, B, H, W, scene=(a.superpixel_method == 'felzenszwalb'),
                          seed0=1000 * rank + 4 * nb, out=pin.numpy(), integer=integer)
        dev.append(pin.cuda())
        if integer:
            u8 = torch.empty((B, H, W, 3), dtype=torch.uint8).pin_memory()
            u8.copy_(pin.permute(0, 2, 3, 1))
            pin = u8
        host.append(pin)
        gts.append(torch.from_numpy(g).cuda())
    conf_total = torch.zeros((B, 4), dtype=torch.int64, device='cuda')
    cur = [0]

    def score(res):
        conf_total.add_(eng.confusion(res.road, gts[cur[0] % NB]))

    def step(s):
        cur[0] = s
        # join=False: the calling stream is not made to wait for the batch's tail (pooling, k-means, paint run on the pipeline's
        # second stream), so the next step's DRN forward starts under it — the drivers' loops do the same
        res = pipe.run(dev[s % NB], check_status=False, join=False)
        with torch.cuda.stream(res.stream):
            score(res)
        return res

    for s in range(a.warmup):
        res = step(s)
    eng.raise_on_status()
    torch.cuda.synchronize()
    COUNTERS = ('bytes', 'launches', 'conv_flops', 'conv16_flops', 'conv16_launches', 'conv16_bytes', 'gemm16_flops', 'gemm16_launches',
                'gemm16_bytes', 'gemm16n_flops', 'gemm16n_launches', 'gemm16n_bytes', 'gemm_flops', 'gemm_launches', 'gemm_bytes',
                'gemmn_flops', 'gemmn_launches', 'gemmn_bytes', 'wino_direct_flops', 'wino_saved_flops', 'wino_in_bytes',
                'wino_out_bytes', 'wino_launches', 'gemm16_layer_bytes', 'light_flops', 'light_bytes', 'light_launches')

    gather_s, local_s = [0.0], [0.0]

    def gather(res):
        """the result.json reduction: one all_gather of per-image records (inside the timed region)"""
        res.stream.synchronize()                   # the scores were accumulated on the stream the results live on
        # k > 2: retry runs a batch still owes are made by the pipeline before the next batch draws; the last batch settles here
        # (the reference's messages go to stderr: stdout carries the one JSON line)
        import contextlib
        with contextlib.redirect_stdout(sys.stderr):
            res.check_retry()
        info = res.info.cpu().numpy()
        conf = conf_total.cpu().numpy()
        n_sp = res.n_labels.cpu().numpy()
        rec = np.zeros((B, dist.RECORD_WIDTH), np.int64)
        rec[:, 0] = rank * B + np.arange(B)
        rec[:, 1:5] = conf
        rec[:, 5] = n_sp
        rec[:, 6], rec[:, 7] = info[0], info[1]
        tg = time.perf_counter()
        allrec = dist.gather_records(rec)          # nccl = RCCL over xGMI when N > 1
        gather_s[0] = time.perf_counter() - tg
        return allrec, info, n_sp

    def loop_device(headline):
        """K steps on batches already resident in HBM"""
        evs = []
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s in range(a.steps):
            res = step(a.warmup + s)
            evs.append(dict(pipe._ev))             # device events, read after the timed region
        g = gather(res) if headline else None
        torch.cuda.synchronize()
        local_s[0] = time.perf_counter() - t0      # this rank's own time, before it waits for the others
        dist.barrier()
        torch.cuda.synchronize()
        dt = dist.max_over_ranks(time.perf_counter() - t0)
        eng.raise_on_status()
        return dt, res, evs, g

    def loop_host(headline):
        """SURVEY.md 8d's region, K steps: batches in pinned host memory -> masks in pinned host memory, double buffered"""
        first = [a.warmup]
        evs = []

        def after(r, s):
            conf_total.add_(eng.confusion(r.road, gts[(first[0] + s) % NB]))
            evs.append(dict(pipe._ev))
        hs = pipeline.HostStream(pipe, B, H, W, after=after, u8_hwc=integer)
        order = [host[(a.warmup + s) % NB] for s in range(a.steps + 1)]
        for _ in hs.process(iter([order[0]] * 3)):   # warm the copy streams, buffers and the allocator's pools (not timed)
            pass
        torch.cuda.synchronize()
        del evs[:]
        conf_total.zero_()
        if headline and not a.no_prof:
            eng.prof_enable(True)
        for key in COUNTERS:
            drn._EPILOGUE[key] = 0
        drn._EPILOGUE['c16'] = {}
        first[0] = a.warmup + 1
        del evs[:]
        dist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        n_out = 0
        for cluster_h, road_h, res in hs.process(iter(order[1:])):
            n_out += int(road_h.shape[0])
        g = gather(res) if headline else None
        torch.cuda.synchronize()
        local_s[0] = time.perf_counter() - t1      # this rank's own time, before it waits for the others
        dist.barrier()
        torch.cuda.synchronize()
        dt = dist.max_over_ranks(time.perf_counter() - t1)
        eng.raise_on_status()
        assert n_out == B * a.steps
        return dt, res, evs, g

    # the device-resident loop first (plain), then the headline loop with the per-kernel events
    host_headline = not a.no_host_loop
    conf_total.zero_()
    for key in COUNTERS:
        drn._EPILOGUE[key] = 0
    drn._EPILOGUE['c16'] = {}
    if not host_headline and not a.no_prof:
        eng.prof_enable(True)
    dt_dev, res, evs, g = loop_device(headline=not host_headline)
    dt_h = None
    if host_headline:
        dt_h, res, evs, g = loop_host(headline=True)
    allrec, info, n_sp = g
    dt = dt_h if host_headline else dt_dev
    # multi-GPU bookkeeping (every rank takes part in the collectives; rank 0 reports): each rank's own rate, the time of the one
    # data-path collective, and where each rank's spa_ctx lives — ranks must sit on distinct devices
    rank_rates = [round(B * a.steps / max(v, 1e-9), 3) for v in dist.all_values(local_s[0])]
    rank_gather_ms = [round(v * 1e3, 3) for v in dist.all_values(gather_s[0])]
    same_dev = os.environ.get('SPA_BENCH_SAME_DEVICE') == '1'
    dev_index = eng.device.index if eng.device.index is not None else torch.cuda.current_device()
    rank_devices = [int(v) for v in dist.all_values(dev_index)]
    uuid = getattr(torch.cuda.get_device_properties(dev_index), 'uuid', None)
    uuid_hash = float(int.from_bytes(__import__('hashlib').sha1(str(uuid).encode()).digest()[:6], 'big')) if uuid is not None else float(dev_index)
    rank_uuid = dist.all_values(uuid_hash)
    # (a launcher that isolates ranks by device visibility gives every rank ordinal 0: then only the uuids can tell the devices
    # apart, and a runtime that reports no usable uuid leaves the placement unverified — reported, not fatal)
    ndev = torch.cuda.device_count()
    assert dev_index == (0 if same_dev else local), 'rank %d: spa_ctx on device %d, LOCAL_RANK %d' % (rank, dev_index, local)
    placement = 'one device (test layout)' if same_dev else 'single rank'
    if ws > 1 and not same_dev:
        if len(set(rank_devices)) == ws:
            placement = 'distinct device ordinals' + (', distinct uuids' if len(set(rank_uuid)) == ws else '')
        elif len(set(rank_uuid)) == ws and uuid is not None:
            placement = 'distinct uuids (ranks isolated by device visibility)'
        elif ndev >= ws:
            raise AssertionError('two ranks share a device: ordinals %r with %d devices visible' % (rank_devices, ndev))
        else:
            placement = 'unverified: %d device(s) visible per rank and no distinct uuids reported' % ndev
            if rank == 0:
                sys.stderr.write('[bench] rank placement unverified: ordinals %r\n' % (rank_devices,))
    assert allrec.shape[0] == ws * B, 'gathered %d records, expected N x B = %d' % (allrec.shape[0], ws * B)
    assert sorted(allrec[:, 0].tolist()) == list(range(ws * B)), 'record indices are not 0 .. N x B - 1'
    backend = (torch.distributed.get_backend() if torch.distributed.is_initialized() else None)
    stage = {'time_feature_maps': 0.0, 'time_superpixel': 0.0, 'time_roialign': 0.0, 'time_kmeans': 0.0}
    bias_bytes, bias_launches = drn._EPILOGUE['bytes'], drn._EPILOGUE['launches']
    conv_flops = drn._EPILOGUE['conv_flops']
    E = dict(drn._EPILOGUE)
    E['c16'] = {k: list(v) for k, v in drn._EPILOGUE['c16'].items()}
    wino_direct = E['wino_direct_flops']
    wino_saved = E['wino_saved_flops']                             # multiplications Winograd does not execute
    for e in evs:
        pipe._ev = e
        for k2, v2 in pipe.stage_ms().items():
            if k2 in stage:
                stage[k2] += v2
    prof = eng.prof_read() if not a.no_prof else {}
    eng.prof_enable(False)
    # the same step with the reference's own arithmetic — float32 matrix instructions (v_mfma_f32_16x16x4_f32, exact fmaf
    # chains) instead of two half-precision planes per operand: `exact_fp32_value` (VERDICT r4, next #2 / missing #3), the
    # same region as `value`, K steps after two warm-up steps, counters and events of the headline loop untouched
    exact_fp32 = None
    if a.dtype == 'fp32' and drn._EPILOGUE['split_gemm'] and not a.no_exact_fp32:
        saved_counters = {k: drn._EPILOGUE[k] for k in COUNTERS}
        drn._EPILOGUE['split_gemm'] = False
        try:
            for s in range(2):
                step(s)
            torch.cuda.synchronize()
            dt_x = (loop_host(headline=False) if host_headline else loop_device(headline=False))[0]
            exact_fp32 = {'value': round(ws * B * a.steps / dt_x, 3), 'ms_per_step': round(dt_x / a.steps * 1e3, 3)}
        finally:
            drn._EPILOGUE['split_gemm'] = True
            drn._EPILOGUE.update(saved_counters)
    px_bytes = 3 if integer else 12
    h2h = None
    if host_headline:
        h2h = {'value': round(ws * B * a.steps / dt_h, 3), 'unit': 'images/sec',
               'ms_per_step': round(dt_h / a.steps * 1e3, 3), 'images_downloaded': B * a.steps * ws,
               'pcie_bytes_per_step': B * (px_bytes * H * W + 2 * H * W),
               'host_batch': '(B,H,W,3) uint8, a decoded PNG\'s layout' if integer else '(B,3,H,W) float32',
               'region': 'batches in pinned host memory -> cluster + road masks in pinned host memory; uploads '
                         '(h2d stream) and downloads (d2h stream) double buffered under the kernels'}

    if rank != 0:
        return
    total_images = ws * B * a.steps
    C = res.fmap.shape[1]
    fh, fw = res.fmap.shape[2], res.fmap.shape[3]
    n_seg = float(n_sp.mean())
    feat_bytes = 4 if a.dtype == 'fp32' else 2
    peak_tf = FP32_MATRIX_PEAK_TF if a.dtype == 'fp32' else BF16_MATRIX_PEAK_TF
    kernels = {}
    for name, (ms, n) in prof.items():
        avg = ms / n
        ent = {'launches_per_step': n / a.steps, 'avg_ms': round(avg, 4), 'ms_per_step': round(ms / a.steps, 3)}
        if name.startswith('k_drn_stem_d') and a.dtype == 'fp32' and not drn._EPILOGUE['split_gemm']:
            # the float32 network's fused stem is float32 MFMA arithmetic (the bf16 network's runs on the bf16
            # matrix cores and is priced against HBM like the other streaming kernels)
            tf = stem_flops(B, H, W) / (avg * 1e-3) / 1e12
            ent.update(bound='mfma', achieved=round(tf, 2), peak=FP32_MATRIX_PEAK_TF, unit='TFLOP/s',
                       frac=round(tf / FP32_MATRIX_PEAK_TF, 4), flops_per_launch=stem_flops(B, H, W))
        elif name == 'k_conv3x3_f32<0, 256, 1, 256>' or name.startswith('k_conv3x3_f32<taps 1'):
            # the GEMM form of the float32 MFMA kernel: the 16 / 36 GEMMs of every Winograd layer (one launch each) and
            # the 1x1 projections.  The 256 x 256-tile instance is one entry (= one rocprofv3 row), the narrow-tile
            # instances (128-channel layers) another.  FLOPs = the products actually executed (a Winograd layer
            # multiplies 1/4 resp. 16/36 of what the direct form would), counted by drn.py
            pre = 'gemm' if name == 'k_conv3x3_f32<0, 256, 1, 256>' else 'gemmn'
            fl, nl, by = E[pre + '_flops'], max(1, E[pre + '_launches']), E[pre + '_bytes']
            tf = fl / a.steps / (ms / a.steps * 1e-3) / 1e12
            ent.update(bound='mfma', achieved=round(tf, 1), peak=FP32_MATRIX_PEAK_TF, unit='TFLOP/s',
                       frac=round(tf / FP32_MATRIX_PEAK_TF, 4), flops_per_step=fl / a.steps, flops_per_launch=fl / nl,
                       hbm_bytes_per_launch_by_construction=int(by / nl),
                       traffic=pmc_traffic('k_conv3x3_f32<taps 1>(GEMM form, all)', B, H, W, by / nl))
        elif name.startswith('k_conv3x3_f32<split') or name.startswith('k_conv3x3_p16') or name.startswith('split-plane front'):
            # the direct split-plane kernels, one entry per instantiation (= one rocprofv3 row): the 64- / 128- / 256-channel tiles of
            # the stride-1 3x3 layers, the 1x1 projections, and the front (stride-2 openers with their projections, layer 2, DRN-C's
            # layers 1-2) — on the 16-bit matrix cores (three half-precision products per float32 product): executed FLOPs against
            # the dense 16-bit peak, and the activations' bytes against HBM (what binds the narrow layers)
            kind = ('64' if '64-channel' in name else '128' if '128-channel' in name else '256' if '256-channel' in name
                    else '1x1' if '1x1' in name else 'front')
            fl, by, nl = E['c16'].get(kind, [0.0, 0.0, 0])
            nl = max(1, nl)
            tf = fl / a.steps / (ms / a.steps * 1e-3) / 1e12
            gbs = by / a.steps / (ms / a.steps * 1e-3) / 1e9
            ent.update(bound='mfma', achieved=round(3 * tf, 1), peak=BF16_MATRIX_PEAK_TF, unit='TFLOP/s',
                       frac=round(3 * tf / BF16_MATRIX_PEAK_TF, 4), float32_equivalent_tflops=round(tf, 1),
                       flops_per_step=3 * fl / a.steps, flops_per_launch=3 * fl / nl,
                       algorithmic_bytes_per_launch=int(by / nl),
                       achieved_hbm_GBs=round(gbs, 1), hbm_frac=round(gbs / HBM_PEAK_GBS, 4))
            ent['limiter'] = LIMITERS.get('k_conv3x3_p16' if name.startswith('k_conv3x3_p16') else 'k_conv3x3_f32<split>(all)')
        elif name.startswith('k_gemm_f16x3'):
            # the Winograd GEMMs on the 16-bit matrix cores at float32 accuracy: every product of the float32 GEMM is three
            # half-precision matrix products (csrc/spa_gemm16.hip).  achieved = EXECUTED half-precision FLOPs (3 x the
            # GEMM's) against the dense 16-bit peak; float32_equivalent_tflops = the GEMM's own FLOPs per second
            pre = 'gemm16' if '<256, 256>' in name else 'gemm16n'
            fl, nl, by = E[pre + '_flops'], max(1, E[pre + '_launches']), E[pre + '_bytes']
            tf = fl / a.steps / (ms / a.steps * 1e-3) / 1e12
            ent.update(bound='mfma', achieved=round(3 * tf, 1), peak=BF16_MATRIX_PEAK_TF, unit='TFLOP/s',
                       frac=round(3 * tf / BF16_MATRIX_PEAK_TF, 4), float32_equivalent_tflops=round(tf, 1),
                       flops_per_step=3 * fl / a.steps, flops_per_launch=3 * fl / nl,
                       hbm_bytes_per_launch_by_construction=int(by / nl),
                       traffic=pmc_traffic(name, B, H, W, by / nl))
            if pre == 'gemm16' and E['gemm16_layer_bytes'] > 0:
                # SURVEY.md 8(d) prices the DRN by FLOPs against the MFMA peak (`frac` above, this launch alone).  The LAYER view
                # beside it: the layer's own bytes — X, Y, the residual, the weight planes once — are the algorithmic
                # bytes; V and M (2.25x the activations, written and read back by the three launches of the Winograd form) are the
                # implementation's traffic, and the layer's executed FLOPs over the time of its three launches is what the matrix
                # cores deliver per layer
                layer_by = E['gemm16_layer_bytes'] / nl
                ent['algorithmic_bytes_per_launch'] = int(layer_by)
                ent['layer_bytes_per_launch'] = int(layer_by)
                ent['v_m_scratch_over_layer_bytes'] = round((by / nl) / layer_by, 2)
        elif name in ('k_wino_in', 'k_wino_out'):
            ab = (E['wino_in_bytes'] if name == 'k_wino_in' else E['wino_out_bytes']) / max(1, E['wino_launches'])
            gbs = ab / (avg * 1e-3) / 1e9
            ent.update(bound='hbm', achieved=round(gbs, 1), peak=HBM_PEAK_GBS, unit='GB/s', frac=round(gbs / HBM_PEAK_GBS, 4),
                       algorithmic_bytes_per_launch=int(ab), traffic=pmc_traffic(name, B, H, W, ab))
        elif name.startswith('k_conv_bf16_light'):
            # the light layers of the bf16 network (stride 2, thin, 1x1): priced by their activations' bytes (in + out [+ residual], bf16)
            ab = E['light_bytes'] / max(1, E['light_launches'])
            gbs = ab / (avg * 1e-3) / 1e9
            ent.update(bound='hbm', achieved=round(gbs, 1), peak=HBM_PEAK_GBS, unit='GB/s', frac=round(gbs / HBM_PEAK_GBS, 4),
                       algorithmic_bytes_per_launch=int(ab), traffic=None, flops_per_step=E['light_flops'] / a.steps,
                       limiter='the pixel operand is loaded per tap straight from the image (9 reads of a pixel through L1 / L2 per 3x3 '
                               'layer, no LDS staging): bound by the CU\'s vector memory path, not by HBM or the matrix pipe')
        elif name.startswith('k_conv3x3_'):
            # libspalign's implicit-GEMM convolutions (the stride-1 3x3 layers, bf16 or float32 matrix cores):
            # family entry over all launches; FLOPs = 2 * MACs of exactly those layers (counted by drn.py)
            pk = BF16_MATRIX_PEAK_TF if name.startswith('k_conv3x3_bf16') else FP32_MATRIX_PEAK_TF
            tf = conv_flops / a.steps / (ms / a.steps * 1e-3) / 1e12
            ent.update(bound='mfma', achieved=round(tf, 1), peak=pk, unit='TFLOP/s',
                       frac=round(tf / pk, 4), flops_per_step=conv_flops / a.steps,
                       flops_per_launch=conv_flops / max(1, n))
        else:
            if name.startswith('k_bias_act'):
                ab = bias_bytes / max(1, bias_launches)          # average over the 23 layers' shapes
            else:
                ab = algorithmic_bytes(name, B, H, W, C, fh, fw, n_seg, feat_bytes)
            gbs = ab / (avg * 1e-3) / 1e9 if ab else None
            ent.update(bound='hbm' if ab else 'latency', achieved=round(gbs, 1) if gbs else None, peak=HBM_PEAK_GBS,
                       unit='GB/s', frac=round(gbs / HBM_PEAK_GBS, 4) if gbs else None,
                       algorithmic_bytes_per_launch=int(ab), traffic=pmc_traffic(name, B, H, W, ab))
        ent['achieved_GBs'] = ent['achieved'] if ent.get('unit') == 'GB/s' else None
        if name in LIMITERS:
            ent['limiter'] = LIMITERS[name]
        kernels[name] = ent
    # headline roofline: the hand-written kernel family with the most time per step — no name filter
    roof = None
    if kernels:
        dom = max(kernels, key=lambda k: kernels[k]['ms_per_step'])
        e = kernels[dom]
        roof = {'kernel': dom, 'bound': e['bound'], 'achieved': e['achieved'], 'peak': e['peak'], 'unit': e['unit'],
                'frac': e['frac'], 'traffic': e.get('traffic'), 'avg_launch_ms': e['avg_ms'],
                'ms_per_step': e['ms_per_step'],
                'algorithmic_bytes_per_launch': e.get('algorithmic_bytes_per_launch'),
                'flops_per_launch': e.get('flops_per_launch'),
                'hbm_frac': e.get('hbm_frac'), 'mfma_frac': e.get('mfma_frac'),
                'layer_bytes_per_launch': e.get('layer_bytes_per_launch'),
                'bytes_by_construction_over_layer_bytes': e.get('bytes_by_construction_over_layer_bytes'),
                'traffic_source': 'profiles/pmc_traffic.json: HBM bytes per launch from separate rocprofv3 --pmc '
                                  'FETCH_SIZE / WRITE_SIZE passes (gfx950 corrections applied), scaled to this batch',
                'limiter': e.get('limiter'),
                'selection': 'largest ms_per_step among all hand-written kernel families of libspalign (see `kernels`)'}
    drn_ms = stage['time_feature_maps'] / a.steps
    split16 = a.dtype == 'fp32' and E['gemm16_launches'] + E['gemm16n_launches'] + E['conv16_launches'] > 0
    flops_direct = drn.flops_per_image(a.arch, H, W) * B
    flops = flops_direct - wino_saved / a.steps                        # executed: the Winograd layers multiply 16/36 as much
    drn_tf = flops / (drn_ms * 1e-3) / 1e12
    tp = allrec[:, 4].sum(); fp = allrec[:, 2].sum(); fn = allrec[:, 3].sum()
    out = {
        'metric': 'images/sec superpixel-align labelling (1024x2048)' if (H, W) == (1024, 2048)
                  else 'images/sec superpixel-align labelling (%dx%d)' % (H, W),
        'value': round(total_images / dt, 3), 'unit': 'images/sec', 'n_gpus': ws, 'steps': a.steps,
        'warmup': a.warmup, 'ms_per_step': round(dt / a.steps * 1e3, 3), 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None,
        'dtype': ('f32 as 2 x f16 planes, f32 accumulate' if split16 else 'f32') if a.dtype == 'fp32' else 'bf16',
        'data': 'synthetic',
        'timed_region': ('host to host (SURVEY.md 8d; reference batch_spalign_kmeans.py:427-458): decoded batches in pinned host '
                         'memory -> masks in pinned host memory' if host_headline else
                         'device resident (--no_host_loop): batches already in HBM -> masks in HBM'),
        'exact_fp32_value': exact_fp32['value'] if exact_fp32 else None,
        'exact_fp32_ms_per_step': exact_fp32['ms_per_step'] if exact_fp32 else None,
        'exact_fp32_note': ('the same region and steps with every matrix product on float32 matrix instructions (--fp32_mfma_gemm): the '
                            'reference\'s arithmetic (cuDNN float32, models/drn.py:304-325); `value` multiplies two half-precision planes per '
                            'operand instead — pooled descriptors within 1e-4 relative of this path, label maps identical '
                            '(tests/test_gpu_descriptor_parity.py)') if exact_fp32 else None,
        'device_resident_value': round(total_images / dt_dev, 3),
        'device_resident_ms_per_step': round(dt_dev / a.steps * 1e3, 3),
        'config': {'workload': 'BASELINE configs[1] x batch: %s %s features + HIP %s/%s-pool/'
                               'prior/k-means(k=%d)/paint, %dx%d, %d images per step per GPU, '
                               '%d distinct batches rotating, %s images, random-init weights'
                               % (a.arch, a.dtype, 'SLIC(%d)' % a.n_slic_segments if a.superpixel_method == 'slic'
                                  else 'felzenszwalb(300,0.8,20)', a.pool_mode, a.n_clusters, H, W, B, NB, '8-bit-valued' if integer else 'float-valued'),
                   'images_per_step_per_gpu': B, 'sharding': 'images (no data-path collective), '
                   'one all_gather of score records',
                   'drn_arithmetic': ('float32 tensors; matrix products as three v_mfma_f32_16x16x32_f16 per float32 product on two '
                                      'half-precision planes per operand (22 significand bits after an exact power-of-two scaling), float32 '
                                      'accumulation: the final map is as close to the float64 network as with float32 matrix instructions '
                                      '(--fp32_mfma_gemm selects those)') if split16 else
                                     ('float32 matrix instructions' if a.dtype == 'fp32' else 'bf16 operands, float32 accumulation')},
        'roofline': roof,
        'host_to_host': h2h,
        'drn': {'bound': 'mfma' if not split16 else 'mixed: 16-bit mfma (GEMMs) + hbm (Winograd transforms)',
                'achieved': round(drn_tf, 2), 'peak': peak_tf if not split16 else None, 'unit': 'TFLOP/s',
                'frac': round(drn_tf / peak_tf, 4) if not split16 else None, 'ms_per_step': round(drn_ms, 3),
                'effective_TFLOPs_direct_equivalent': round(flops_direct / (drn_ms * 1e-3) / 1e12, 2),
                # small batches: the forward is replayed as one captured HIP graph (drn.py); its kernels then carry no per-kernel timers
                'captured_graph': any(e is not False for e in getattr(model, '_graphs', {}).values()),
                'library_convolutions': int(E.get('library_convs', 0)),      # F.conv2d calls of the whole run (0: every convolution was libspalign's)
                'note': ('%s' % (('float32 with the Winograd GEMMs on the 16-bit matrix cores: every float32 operand of a GEMM is two half-precision '
                                  'planes (22 significand bits after an exact power-of-two scaling) and every product three '
                                  'v_mfma_f32_16x16x32_f16 accumulated in float32 — the final map is as close to the float64 network as with '
                                  'float32 operands (DESIGN.md section 4); `achieved` = float32-equivalent FLOPs of the network per second, '
                                  'which the float32 matrix peak (157 TFLOP/s) no longer bounds.  ' if split16 else '') +
                                 'float32: every stride-1 3x3 layer from 128 channels up (from 256 input channels up with the split planes; the layers below take the direct kernel) runs as Winograd F(4x4,3x3) (input transform, 36 '
                                 'GEMMs in one launch of the float32-MFMA kernel, output transform with the epilogue fused), the '
                                 '64-channel layers and the 1x1 projections on the same kernel directly, the stem on its own MFMA kernel; ' +
                                 ('the stride-2 layers (layer 2, the openers of layers 3 / 4 with their projections) have their own split-plane kernels: no MIOpen convolution is left.  ' if split16 else 'the five stride-2 layers are PyTorch-ROCm (MIOpen).  ') + '`achieved` counts the products actually '
                                 'executed, `effective_TFLOPs_direct_equivalent` what a direct convolution would have to sustain'
                                 if a.dtype == 'fp32' else
                                 'bf16: the stride-1 3x3 layers from 64 channels up are libspalign\'s bf16 implicit-GEMM convolution with '
                                 'the epilogue fused (k_conv3x3_bf16), the stem its bf16-MFMA kernel, the light layers (layer 2, the stride-2 openers and the 1x1 '
                                 'projections) libspalign\'s plain bf16 kernel (k_conv_bf16_light): no MIOpen convolution is left')
                         if (conv_flops > 0 or split16 or wino_direct > 0) else
                         'the big convolutions are PyTorch-ROCm (MIOpen); libspalign adds the fused float32-MFMA stem '
                         'of DRN-D (normalise + layer0 + layer1, k_drn_stem_d) and the bias/residual/ReLU epilogues')},
        'stage_ms_per_step': dict({k: round(v / a.steps, 3) for k, v in stage.items()},
                                  streams='two: the superpixel branch runs beside the front of the DRN forward (stage times overlap; --one_stream for '
                                          'every kernel on its own)' if overlap else 'one'),
        'kernels': kernels,
        'quality': {'superpixels_per_image': round(n_seg, 1), 'kmeans_iterations': int(info[0]),
                    'synthetic_road_iou': round(float(tp) / max(1.0, float(tp + fp + fn)), 4),
                    'records_gathered': int(allrec.shape[0])},
        'multi_gpu': {'ranks': ws, 'backend': backend, 'per_rank_images_per_sec': rank_rates,
                      'gather_ms_per_rank': rank_gather_ms, 'records_gathered': int(allrec.shape[0]),
                      'records_expected': ws * B, 'rank_devices': rank_devices, 'rank_placement': placement,
                      'note': 'one all_gather of %d-word records per image at the end of the timed region (the result.json reduction); '
                              'no data-path collective; every rank\'s spa_ctx asserted on its own device' % dist.RECORD_WIDTH},
    }
    if ws == 1 and not a.no_cpu_baseline:
        try:
            out['cpu_baseline'] = cpu_baseline(a, spa.synth, a.cpu_sample)
        except Exception as exc:                      # the baseline must never hide the GPU number
            out['cpu_baseline'] = {'error': repr(exc)}
    print(json.dumps(out))


if __name__ == '__main__':
    main()
