"""The fused Winograd layer launch (spa_conv3x3_wino4_fused) against the three-launch form (spa_conv3x3_wino4_f16s) on the
DRN's layer shapes at 30 x 128 x 256 feature pixels: bit equality and time.
    python tools/winof_bench.py [--batch 30] [--reps 5]"""
import argparse
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=30)
ap.add_argument('--reps', type=int, default=5)
ap.add_argument('--shapes', default='256:256:2,256:512:2,512:512:4,512:512:1')
a = ap.parse_args()
engine = importlib.import_module('superpixel-align_amd.engine')
eng = engine.Engine()
torch.manual_seed(0)


def t(fn, reps):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for spec in a.shapes.split(','):
    Cin, Cout, dil = (int(v) for v in spec.split(':'))
    B, H, W = a.batch, 128, 256
    x = torch.relu(torch.randn((B, Cin, H, W), device='cuda')).contiguous(memory_format=torch.channels_last)
    w = torch.randn((Cout, Cin, 3, 3), device='cuda') * (2.0 / (9 * Cin)) ** 0.5
    bias = torch.randn((Cout,), device='cuda')
    r = torch.randn((B, Cout, H, W), device='cuda').contiguous(memory_format=torch.channels_last)
    u2, cs = eng.winograd_weights_split(w)
    am = eng.amax(x)
    y3, a3 = eng.conv3x3_wino_f16s(x, u2, cs, bias, r, True, dil, amax_in=am)
    y1, a1 = eng.conv3x3_wino_f16s(x, u2, cs, bias, r, True, dil, amax_in=am, fused=True)
    torch.cuda.synchronize()
    same = bool(torch.equal(y1, y3)) and int(a1) == int(a3)
    ms3 = t(lambda: eng.conv3x3_wino_f16s(x, u2, cs, bias, r, True, dil, amax_in=am), a.reps)
    ms1 = t(lambda: eng.conv3x3_wino_f16s(x, u2, cs, bias, r, True, dil, amax_in=am, fused=True), a.reps)
    fl = 2.0 * B * H * W * Cout * 9 * Cin / 4 * 3            # executed half-precision FLOPs
    print('B %d %d -> %d dil %d | same bits %s | three launches %.3f ms | fused %.3f ms (%.2fx) | fused = %.0f TFLOP/s executed'
          % (B, Cin, Cout, dil, same, ms3, ms1, ms3 / ms1, fl / ms1 / 1e9), flush=True)
    st = eng.status()
    if st:
        print('  device status 0x%x' % st)
