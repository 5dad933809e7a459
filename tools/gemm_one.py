#!/usr/bin/env python3
"""Development aid: the GEMM form of k_conv3x3_f32 alone (36 batched problems of a Winograd F(4x4,3x3) layer)."""
import importlib, os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
lib_mod = importlib.import_module('superpixel-align_amd._lib')
eng = importlib.import_module('superpixel-align_amd.engine').Engine()
C = int(sys.argv[1]) if len(sys.argv) > 1 else 512
x = torch.relu(torch.randn((30, C, 128, 256), device='cuda')).contiguous(memory_format=torch.channels_last)
w = torch.randn((C, C, 3, 3), device='cuda') * (2.0 / (9 * C)) ** 0.5
u = eng.winograd_weights(w, 4)
b = torch.randn((C,), device='cuda')
eng.prof_enable(True)
for _ in range(4):
    y = eng.conv3x3_wino_f32(x, u, b, None, True, 2)
torch.cuda.synchronize()
fl = 36 * 61440 * C * C * 2.0
for name, (ms, n) in eng.prof_read().items():
    print('%-40s avg %8.3f ms' % (name, ms / n) + ('  %.1f TFLOP/s' % (fl / (ms / n) / 1e9) if 'GEMM' in name else ''))
