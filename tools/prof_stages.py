#!/usr/bin/env python3
"""Development aid: run the non-DRN stages of the hot path on a synthetic batch and print the
per-kernel HIP-event times recorded by libspalign (spa_prof_*).  Fast to launch (no MIOpen
tuning), so it is the loop used while optimising kernels:

    python tools/prof_stages.py --batch 8 --reps 3
    rocprofv3 --kernel-trace --stats ... -- python3 tools/prof_stages.py
"""
import argparse
import importlib
import os
import sys
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=8)
ap.add_argument('--height', type=int, default=1024)
ap.add_argument('--width', type=int, default=2048)
ap.add_argument('--n', type=int, default=200)
ap.add_argument('--channels', type=int, default=512)
ap.add_argument('--reps', type=int, default=3)
ap.add_argument('--pool_mode', default='mean')
ap.add_argument('--bias_act', action='store_true', help='also run the DRN epilogue kernel (k_bias_act) on the shapes of '
                'the heavy layers, for the PMC traffic passes')
ap.add_argument('--wino', action='store_true', help='also run one Winograd F(4x4,3x3) layer (512 -> 512, dilation 2) and one direct '
                'float32 layer (128 -> 128) at 1/8 resolution, for the PMC traffic passes')
a = ap.parse_args()

spa = importlib.import_module('superpixel-align_amd')
pipeline = importlib.import_module('superpixel-align_amd.pipeline')
bench = importlib.import_module('bench')
args = types.SimpleNamespace(superpixel_method='slic', n_slic_segments=a.n, n_anchors=10, n_neighbors=4,
                             without_pos=False, y_rel_pos=0.75, x_rel_pos=0.5, y_rel_sigma=0.1,
                             x_rel_sigma=0.1, gpu=0, n_clusters=2, use_feature_maps=[7],
                             pool_mode=a.pool_mode, mean_sampling='nearest')
pipe = pipeline.LabelPipeline(args, model=None, overlap=False)
imgs_h, _ = bench.make_batch(spa.synth, a.batch, a.height, a.width)
imgs = torch.from_numpy(imgs_h).cuda()
fmap = torch.randn((a.batch, a.channels, a.height // 8, a.width // 8), device='cuda').contiguous(
    memory_format=torch.channels_last)
pipe.features = lambda x: fmap
pipe.run(imgs)
pipe.eng.prof_enable(True)
for _ in range(a.reps):
    pipe.run(imgs)
torch.cuda.synchronize()
t = pipe.elapsed_times()
print('last run: superpixel %.3f ms  describe %.3f ms  kmeans+paint %.3f ms' %
      (t['time_superpixel'] * 1e3, t['time_roialign'] * 1e3, t['time_kmeans'] * 1e3))
for name, (ms, n) in pipe.eng.prof_read().items():
    print('%-22s launches/run %5.1f  avg %9.1f us  per run %9.3f ms' % (name, n / a.reps, ms / n * 1e3, ms / a.reps))

if a.bias_act:
    # the 512- and 256-channel layers of DRN-D-22 at 1/8 resolution, with and without the residual operand
    for C, res in ((512, True), (512, False), (256, True)):
        y = torch.randn((a.batch, C, a.height // 8, a.width // 8), device='cuda').contiguous(memory_format=torch.channels_last)
        r = torch.randn_like(y) if res else None
        bias = torch.randn((C,), device='cuda')
        for _ in range(a.reps):
            pipe.eng.bias_act_(y, bias, r, True)
    torch.cuda.synchronize()
    print('bias_act: 512 ch + residual, 512 ch, 256 ch + residual at %dx%d, %d launches each' % (a.height // 8, a.width // 8, a.reps))

if a.wino:
    eng = pipe.eng
    h, w = a.height // 8, a.width // 8
    x = torch.relu(torch.randn((a.batch, 512, h, w), device='cuda')).contiguous(memory_format=torch.channels_last)
    wt = torch.randn((512, 512, 3, 3), device='cuda') * (2.0 / (9 * 512)) ** 0.5
    b = torch.randn((512,), device='cuda')
    u2, cs = eng.winograd_weights_split(wt)         # F(4x4,3x3) with the GEMMs on the 16-bit matrix cores: the network's default
    am = eng.amax(x)
    for _ in range(a.reps):
        eng.conv3x3_wino_f16s(x, u2, cs, b, None, True, 2, amax_in=am)
    x2 = torch.relu(torch.randn((a.batch, 128, h, w), device='cuda')).contiguous(memory_format=torch.channels_last)
    w2 = (torch.randn((128, 128, 3, 3), device='cuda') * (2.0 / (9 * 128)) ** 0.5).permute(0, 2, 3, 1).reshape(128, 9, 128).contiguous()
    b2 = torch.randn((128,), device='cuda')
    w22, inv2 = eng.split_planes(w2)
    am2 = eng.amax(x2)
    for _ in range(a.reps):
        eng.conv3x3_f16s(x2, w22, inv2, b2, None, True, 1, amax_in=am2)
    torch.cuda.synchronize()
    print('winograd 512 -> 512 dil 2 and direct 128 -> 128 at %dx%d, %d launches each' % (h, w, a.reps))
