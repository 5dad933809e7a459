#!/usr/bin/env python3
"""Development aid: how many small components does the connectivity pass see per image in the
benchmark workload, and how big are they / their bounding boxes?  (reads ConnMisc and the size /
box tables through spa_debug_peek after one spa_slic call)"""
import argparse
import ctypes
import importlib
import os
import re
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=2)
ap.add_argument('--n', type=int, default=200)
a = ap.parse_args()
spa = importlib.import_module('superpixel-align_amd')
eng_mod = importlib.import_module('superpixel-align_amd.engine')
lib_mod = importlib.import_module('superpixel-align_amd._lib')
bench = importlib.import_module('bench')
eng = eng_mod.default_engine()
H, W = 1024, 2048
imgs_h, _ = bench.make_batch(spa.synth, a.batch, H, W)
imgs = torch.from_numpy(imgs_h).cuda()
src = open(os.path.join(os.path.dirname(lib_mod.__file__), 'csrc', 'spa_common.h')).read()
names = re.findall(r'^\s*(WS_[A-Z_0-9]+)\s*(?:=\s*0)?,', src, re.M)


def peek(name, offset, count, dtype):
    buf = np.empty(count, dtype=dtype)
    lib_mod.check(lib_mod.lib().spa_debug_peek(eng._ctx, names.index(name), offset, buf.nbytes,
                                               buf.ctypes.data_as(ctypes.c_void_p)))
    return buf


labels, n_labels = eng.slic(imgs, a.n)
torch.cuda.synchronize()
plan = lib_mod.make_plan(H, W, a.n)
print('min_size', plan.min_size, 'max_size', plan.max_size)
misc = peek('WS_CONNMISC', 0, 12 * a.batch, np.int32).reshape(a.batch, 12)[:, :8]
npix = H * W
for b in range(a.batch):
    n_small, first_kept, qalloc, n_kept, n_todo1, n_todo2, n_big, n_over = misc[b]
    print('image %d: tiny(<=32px) %d  big-small %d  -> past 40KB tier %d  -> past 156KB tier %d  kept %d' %
          (b, n_small, n_big, n_todo1, n_todo2, n_kept))
    big = peek('WS_SMALL', (a.batch * npix + b * npix) * 4, int(n_big), np.int32)
    size = peek('WS_SIZE', b * npix * 4, npix, np.int32)
    sz = size[big]
    sbox = peek('WS_SBOX', b * 65536 * 16, min(int(n_big), 65536) * 4, np.int32).reshape(-1, 4)
    area = (sbox[:, 1] - sbox[:, 0] + 3) * (sbox[:, 3] - sbox[:, 2] + 3)
    qs = [50, 90, 99, 100]
    print('   sizes   pct', qs, np.percentile(sz, qs).astype(int), ' sum', sz.sum())
    print('   box area pct', qs, np.percentile(area, qs).astype(int))
    for cap in (256, 512, 1024, 2048, 4096, 9000):
        print('   box area <= %5d: %5d components (%.1f%%), %.1f%% of their pixels' %
              (cap, (area <= cap).sum(), 100.0 * (area <= cap).mean(), 100.0 * sz[area <= cap].sum() / max(sz.sum(), 1)))
