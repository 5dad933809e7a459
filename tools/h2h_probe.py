#!/usr/bin/env python3
"""Where does the host-to-host loop (pipeline.HostStream) lose time against the device-resident loop?
Runs the bf16 pipeline on 30 synthetic 1024x2048 images: device resident, full host loop, host loop without uploads,
host loop without downloads.   python tools/h2h_probe.py [--dtype bf16]"""
import argparse
import importlib
import os
import sys
import time
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

ap = argparse.ArgumentParser()
ap.add_argument('--dtype', default='bf16')
ap.add_argument('--steps', type=int, default=8)
a = ap.parse_args()
spa = importlib.import_module('superpixel-align_amd')
pipeline = importlib.import_module('superpixel-align_amd.pipeline')
drn = importlib.import_module('superpixel-align_amd.drn')
bench = importlib.import_module('bench')
torch.backends.cudnn.benchmark = True
B, H, W = 30, 1024, 2048
args = types.SimpleNamespace(superpixel_method='slic', n_slic_segments=200, n_anchors=10, n_neighbors=4, without_pos=False,
                             y_rel_pos=0.75, x_rel_pos=0.5, y_rel_sigma=0.1, x_rel_sigma=0.1, gpu=0, n_clusters=2,
                             use_feature_maps=[7], pool_mode='mean', mean_sampling='nearest')
model = drn.create_drn('drn_d_22', None, device='cuda', dtype=torch.bfloat16 if a.dtype == 'bf16' else torch.float32)
pipe = pipeline.LabelPipeline(args, model, overlap=False)
imgs_h, _ = bench.make_batch(spa.synth, B, H, W)
pinned = [torch.from_numpy(imgs_h).pin_memory() for _ in range(2)]
dev = torch.from_numpy(imgs_h).cuda()
for _ in range(2):
    pipe.run(dev)
torch.cuda.synchronize()


def timed(fn, label):
    torch.cuda.synchronize(); t = time.time(); fn(); torch.cuda.synchronize()
    print('%-34s %7.2f ms per batch' % (label, (time.time() - t) * 1e3 / a.steps))


def resident():
    for _ in range(a.steps):
        pipe.run(dev, check_status=False)


def host_loop(hs):
    for _ in hs.process(pinned[i & 1] for i in range(a.steps)):
        pass


timed(resident, 'device resident')
hs = pipeline.HostStream(pipe, B, H, W)
host_loop(hs)
timed(lambda: host_loop(hs), 'host loop')
hs2 = pipeline.HostStream(pipe, B, H, W)
hs2._upload_real = hs2._upload
state = {'n': 0}
def no_upload(slot, batch):
    state['n'] += 1
    if state['n'] <= 2:
        return hs2._upload_real(slot, batch)
    with torch.cuda.stream(hs2.h2d):
        hs2.h2d.wait_event(hs2.in_free[slot]); hs2.up_done[slot].record(hs2.h2d)
    return B
hs2._upload = no_upload
host_loop(hs2)
timed(lambda: host_loop(hs2), 'host loop, uploads skipped')
