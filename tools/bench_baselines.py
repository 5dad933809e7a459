#!/usr/bin/env python3
"""Throughput of the two baseline pipelines at the reference operating point:
224x224 network input (28x28 maps, N = 30*784 feature pixels, D = 514), DRN-C-26 fp32, k = 4;
superpixel_overlaps additionally segments the ORIGINAL 1024x2048 uint8 image (felzenszwalb 500/0.9/20).
Synthetic images, random-init weights.  One JSON line per pipeline.

    python tools/bench_baselines.py [--steps 5] [--warmup 2] [--batch 30]
"""
import argparse
import importlib
import json
import os
import sys
import time
import types

os.environ.setdefault('MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD', '0')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--steps', type=int, default=5)
ap.add_argument('--warmup', type=int, default=2)
ap.add_argument('--batch', type=int, default=30)
ap.add_argument('--orig', type=int, nargs=2, default=[1024, 2048])
a = ap.parse_args()
spa = importlib.import_module('superpixel-align_amd')
baselines = importlib.import_module('superpixel-align_amd.baselines')
drn = importlib.import_module('superpixel-align_amd.drn')
engine = importlib.import_module('superpixel-align_amd.engine')
torch.backends.cudnn.benchmark = True
args = types.SimpleNamespace(n_clusters=4, y_rel_pos=0.75, x_rel_pos=0.5, y_rel_sigma=0.1, x_rel_sigma=0.1,
                             use_feature_maps=[7], superpixel_method='felzenszwalb', felzenszwalb_scale=500.0,
                             felzenszwalb_sigma=0.9, felzenszwalb_min_size=20, overlap_threshold=0.01)
model = drn.create_drn('drn_c_26', None, device='cuda', dtype=torch.float32)
eng = engine.default_engine()
B = a.batch
small = torch.from_numpy(np.stack([spa.synth.synth_image(i, 224, 224) for i in range(B)])).cuda()
orig = torch.from_numpy(np.stack([np.clip(spa.synth.synth_image(i, a.orig[0], a.orig[1]), 0, 255).astype(np.uint8)
                                  for i in range(B)])).cuda()
for name, cls in (('direct_clustering', baselines.DirectClustering), ('superpixel_overlaps', baselines.SuperpixelOverlaps),
                  ('superpixel_overlaps_slic', baselines.SuperpixelOverlaps)):
    if name == 'superpixel_overlaps_slic':
        args = types.SimpleNamespace(**dict(args.__dict__, superpixel_method='slic', n_slic_segments=100))
    pipe = cls(args, model, eng, engine.NpRandom(1111))
    run = (lambda: pipe.run(small, orig)) if name.startswith('superpixel_overlaps') else (lambda: pipe.run(small))
    stages = {}
    for i in range(a.warmup + a.steps):
        if i == a.warmup:
            torch.cuda.synchronize()
            t0 = time.time()
        res = run()
        if i >= a.warmup:
            for k2, v in pipe.elapsed_times().items():
                stages[k2] = stages.get(k2, 0.0) + v
    torch.cuda.synchronize()
    dt = time.time() - t0
    info = res.info.cpu().numpy()
    print(json.dumps({'pipeline': name, 'value': round(B * a.steps / dt, 2), 'unit': 'images/sec', 'n_gpus': 1,
                      'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': round(dt / a.steps * 1e3, 2),
                      'config': {'workload': '%s: drn_c_26 fp32 on 224x224, weighted k-means (k=4) over %d feature '
                                             'pixels x 514 dims%s' % (name, B * 28 * 28,
                                                                      ', %s of the %dx%d uint8 originals + overlap refinement'
                                                                      % (('float64 slic(n=100)' if name.endswith('slic') else 'felzenszwalb(500,0.9,20)',) + tuple(a.orig))
                                                                      if name.startswith('superpixel_overlaps') else ''),
                                 'images_per_step': B},
                      'stage_ms_per_step': {k2: round(v / a.steps * 1e3, 2) for k2, v in stages.items()},
                      'kmeans_iterations': int(info[0]), 'data': 'synthetic', 'dtype': 'f32'}), flush=True)
