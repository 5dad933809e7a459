"""Micro-benchmark of spa_kmeans_weighted on the bench's shape (N ~ 4 600 superpixels, D = 514, k = 2):
    python tools/km_bench.py [N] [D] [k]
prints ms per call and per Lloyd iteration for a few SPA_KM_DIV settings (points per sweep workgroup)."""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

engine = importlib.import_module('superpixel-align_amd.engine')
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4600
D = int(sys.argv[2]) if len(sys.argv) > 2 else 514
k = int(sys.argv[3]) if len(sys.argv) > 3 else 2
rs = np.random.RandomState(0)
X = np.concatenate([rs.normal(0, 1, (N // 2, D)), rs.normal(0.3, 1, (N - N // 2, D))])
w = np.concatenate([rs.uniform(0.4, 1, N // 2), rs.uniform(0, 0.6, N - N // 2)])
eng = engine.Engine()
Xd, wd = torch.from_numpy(X).cuda(), torch.from_numpy(w).cuda()
nd = torch.tensor([N], dtype=torch.int32, device='cuda')
for div in (16, 32, 64, 128):
    os.environ['SPA_KM_DIV'] = str(div)
    for _ in range(2):
        a, info = eng.kmeans(Xd, wd, nd, k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        a, info = eng.kmeans(Xd, wd, nd, k)
    e1.record()
    torch.cuda.synchronize()
    it = int(info[0])
    ms = e0.elapsed_time(e1) / 10
    print('SPA_KM_DIV %3d: %.3f ms per call, %d iterations, %.1f us per iteration, status %d'
          % (div, ms, it, ms * 1e3 / max(it, 1), int(info[1])))
