#!/usr/bin/env python3
"""Development aid: the split-plane direct kernel on the 64 -> 64 layer of 30 full-size images (256 x 512 maps), time and digest.
    SPA_CONV32_BN512=0|1 python tools/conv64_ab.py"""
import hashlib, importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
eng_mod = importlib.import_module('superpixel-align_amd.engine')
eng = eng_mod.default_engine()
torch.manual_seed(0)
for (B, C, H, W, res) in ((30, 64, 256, 512, True), (30, 64, 256, 512, False), (4, 64, 77, 500, True), (30, 128, 128, 256, True), (30, 128, 128, 256, False)):
    x = torch.randn(B, C, H, W, device='cuda').contiguous(memory_format=torch.channels_last)
    r = torch.randn(B, C, H, W, device='cuda').contiguous(memory_format=torch.channels_last) if res else None
    w = torch.randn(C, C, 3, 3, device='cuda') * 0.05
    b = torch.randn(C, device='cuda')
    wt = w.permute(0, 2, 3, 1).reshape(C, 9, C).contiguous()
    wt2, inv_t = eng_mod.Engine.split_planes(wt)
    am = eng.amax(x)
    y, _ = eng.conv3x3_f16s(x, wt2, inv_t, b, r, True, 1, amax_in=am)
    torch.cuda.synchronize()
    t = time.time()
    for _ in range(10):
        y, a2 = eng.conv3x3_f16s(x, wt2, inv_t, b, r, True, 1, amax_in=am)
    torch.cuda.synchronize()
    dt = (time.time() - t) / 10 * 1e3
    ref = torch.nn.functional.conv2d(x[:1].double(), w.double(), b.double(), 1, 1)
    if res: ref = ref + r[:1].double()
    ref = ref.relu()
    err = float((y[:1].double() - ref).abs().max() / ref.abs().max())
    print('BN512=%s C %d B %d %dx%d res %s: %.3f ms, digest %s, max err vs float64 %.1e' % (os.environ.get('SPA_CONV32_BN512', 'default'), C, B, H, W, res, dt,
          hashlib.sha1(y.cpu().numpy().tobytes()).hexdigest()[:10], err))
