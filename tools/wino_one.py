#!/usr/bin/env python3
"""Development aid: one Winograd layer (512 -> 512, dilation 2, 30 x 128 x 256) a few times, for rocprofv3."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
eng = importlib.import_module('superpixel-align_amd.engine').Engine()
C = int(sys.argv[1]) if len(sys.argv) > 1 else 512
x = torch.relu(torch.randn((30, C, 128, 256), device='cuda')).contiguous(memory_format=torch.channels_last)
w = torch.randn((C, C, 3, 3), device='cuda') * (2.0 / (9 * C)) ** 0.5
u = eng.winograd_weights(w)
b = torch.randn((C,), device='cuda')
for _ in range(4):
    y = eng.conv3x3_wino_f32(x, u, b, None, True, 2)
torch.cuda.synchronize()
