"""One shape of spa_conv3x3_bf16 for profilers:  python tools/conv_one.py [reps]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
engine = importlib.import_module('superpixel-align_amd.engine')
eng = engine.Engine()
B, Cin, Cout, H, W, dil = 30, 512, 512, 128, 256, 4
x = torch.randn((B, Cin, H, W), device='cuda').to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
wt = (torch.randn((Cout, 9, Cin), device='cuda') * 0.02).to(torch.bfloat16)
bias = torch.randn((Cout,), device='cuda')
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    y = eng.conv3x3_bf16(x, wt, bias, None, True, dil)
torch.cuda.synchronize()
