#!/usr/bin/env python3
"""Development aid (round 5): the three-launch Winograd layer on the DRN's 16-bit-plane GEMM shapes — per-family launch
times (library HIP events) and a digest of the output, so that two builds / two settings of SPA_GEMM16_STAGGER can be
compared bit for bit from two processes:
    SPA_GEMM16_STAGGER=0 python tools/gemm16_ab.py ; SPA_GEMM16_STAGGER=1 python tools/gemm16_ab.py"""
import argparse, hashlib, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=30)
ap.add_argument('--reps', type=int, default=6)
ap.add_argument('--shapes', default='512:512:4,512:512:1,256:512:2,256:256:2')
a = ap.parse_args()
eng = importlib.import_module('superpixel-align_amd.engine').Engine()
torch.manual_seed(0)
for spec in a.shapes.split(','):
    Cin, Cout, dil = (int(v) for v in spec.split(':'))
    B, H, W = a.batch, 128, 256
    x = torch.relu(torch.randn((B, Cin, H, W), device='cuda')).contiguous(memory_format=torch.channels_last)
    w = torch.randn((Cout, Cin, 3, 3), device='cuda') * (2.0 / (9 * Cin)) ** 0.5
    bias = torch.randn((Cout,), device='cuda')
    u2, cs = eng.winograd_weights_split(w)
    am = eng.amax(x)
    for _ in range(2):
        y, ao = eng.conv3x3_wino_f16s(x, u2, cs, bias, None, True, dil, amax_in=am)
    torch.cuda.synchronize()
    dig = hashlib.sha256(y.cpu().numpy().tobytes()).hexdigest()[:16]
    eng.prof_enable(True)
    for _ in range(a.reps):
        y, ao = eng.conv3x3_wino_f16s(x, u2, cs, bias, None, True, dil, amax_in=am)
    torch.cuda.synchronize()
    fl = 3 * 2.0 * 36 * (B * H * W / 16) * Cin * Cout
    out = []
    for name, (ms, n) in eng.prof_read().items():
        if n:
            out.append('%s %.3f ms%s' % (name, ms / n, ' (%.0f TFLOP/s executed)' % (fl / (ms / n) / 1e9) if 'gemm' in name.lower() else ''))
    eng.prof_enable(False)
    print('stagger=%s B %d %d->%d dil %d | y %s amax %08x | %s' % (os.environ.get('SPA_GEMM16_STAGGER', '0'), B, Cin, Cout, dil, dig,
          int(ao) & 0xffffffff, ' | '.join(out)), flush=True)
    st = eng.status()
    if st:
        print('  device status 0x%x' % st)
