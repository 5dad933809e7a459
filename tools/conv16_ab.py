#!/usr/bin/env python3
"""Development aid (round 5): spa_conv3x3_bf16 on the DRN's 256 / 512-channel shapes — time, TFLOP/s and a digest of the output,
so that the settings of SPA_CONV16_STAGGER (read once per process) can be compared bit for bit from two processes:
    SPA_CONV16_STAGGER=0 python tools/conv16_ab.py ; SPA_CONV16_STAGGER=1 python tools/conv16_ab.py"""
import argparse, hashlib, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=30)
ap.add_argument('--reps', type=int, default=6)
ap.add_argument('--shapes', default='512:512:4:1,512:512:1:0,256:512:2:0,256:256:2:1')
a = ap.parse_args()
eng = importlib.import_module('superpixel-align_amd.engine').Engine()
torch.manual_seed(0)
for spec in a.shapes.split(','):
    Cin, Cout, dil, res = (int(v) for v in spec.split(':'))
    B, H, W = a.batch, 128, 256
    x = torch.relu(torch.randn((B, Cin, H, W), device='cuda')).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = torch.randn((Cout, Cin, 3, 3), device='cuda') * (2.0 / (9 * Cin)) ** 0.5
    wt = w.permute(0, 2, 3, 1).reshape(Cout, 9, Cin).contiguous().to(torch.bfloat16)
    bias = torch.randn((Cout,), device='cuda')
    r = torch.randn((B, Cout, H, W), device='cuda').to(torch.bfloat16).contiguous(memory_format=torch.channels_last) if res else None
    for _ in range(2):
        y = eng.conv3x3_bf16(x, wt, bias, r, True, dil)
    torch.cuda.synchronize()
    dig = hashlib.sha256(y.view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:16]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        y = eng.conv3x3_bf16(x, wt, bias, r, True, dil)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.reps
    fl = 2.0 * B * H * W * Cout * 9 * Cin
    print('stagger=%s B %d %d->%d dil %d res %d | y %s | %.3f ms  %.0f TFLOP/s' % (os.environ.get('SPA_CONV16_STAGGER', 'unset'), B, Cin, Cout, dil, res, dig, ms, fl / ms / 1e9), flush=True)
