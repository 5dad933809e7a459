#!/usr/bin/env python3
"""Development aid: one F(4x4,3x3) layer three ways — float32 operands (v_mfma_f32_16x16x4_f32), two half-precision
planes per operand (three v_mfma_f32_16x16x32_f16 per product), and a float64 convolution — error and time.
    python tools/wino_split_one.py [channels] [images] [dilation]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
eng = importlib.import_module('superpixel-align_amd.engine').Engine()
C = int(sys.argv[1]) if len(sys.argv) > 1 else 512
B = int(sys.argv[2]) if len(sys.argv) > 2 else 15
d = int(sys.argv[3]) if len(sys.argv) > 3 else 2
torch.manual_seed(1)
x = (torch.relu(torch.randn((B, C, 128, 256), device='cuda')) * 3.7).contiguous(memory_format=torch.channels_last)
w = torch.randn((C, C, 3, 3), device='cuda') * (2.0 / (9 * C)) ** 0.5
b = torch.randn((C,), device='cuda')
res = torch.randn((B, C, 128, 256), device='cuda').contiguous(memory_format=torch.channels_last)
u = eng.winograd_weights(w, 4)
u2, cs = eng.winograd_weights_split(w)
def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    t = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.time() - t) / n * 1e3
y32 = eng.conv3x3_wino_f32(x, u, b, res, True, d)
am = eng.amax(x)
y16, am_out = eng.conv3x3_wino_f16s(x, u2, cs, b, res, True, d, amax_in=am)
torch.cuda.synchronize()
print('amax in %.6g (torch %.6g)  amax out %.6g (torch %.6g)' % (am.view(torch.float32).item(), x.abs().max().item(),
                                                             am_out.view(torch.float32).item(), y16.abs().max().item()))
nb = min(B, 2)
ref = torch.relu(F.conv2d(x[:nb].double(), w.double(), b.double(), 1, d, d) + res[:nb].double())
s = ref.abs().max().item()
print('float32 operands  vs float64: %.3e of scale' % ((y32[:nb].double() - ref).abs().max().item() / s))
print('two f16 planes    vs float64: %.3e of scale' % ((y16[:nb].double() - ref).abs().max().item() / s))
print('two f16 planes vs float32 operands: %.3e of scale' % ((y16 - y32).abs().max().item() / s))
t32 = timed(lambda: eng.conv3x3_wino_f32(x, u, b, res, True, d))
t16 = timed(lambda: eng.conv3x3_wino_f16s(x, u2, cs, b, res, True, d, amax_in=am))
fl = 2.0 * 36 * int(eng._lib.spa_wino4_tiles(B, 128, 256, d)) * C * C
print('layer: float32 operands %.3f ms, two f16 planes %.3f ms' % (t32, t16))
def show(tag, fn):
    eng.prof_enable(True)
    for _ in range(3): fn()
    for name, (ms, n) in eng.prof_read().items():
        print('%-18s %-40s %3d launches  avg %8.1f us%s' % (tag, name, n, ms / n * 1e3, ('  = %.0f TFLOP/s of the GEMM (executed %.0f)' % (fl / (ms / n) / 1e9, (3 if 'f16' in name else 1) * fl / (ms / n) / 1e9)) if 'gemm' in name or '1, 256' in name else ''))
show('float32 operands', lambda: eng.conv3x3_wino_f32(x, u, b, res, True, d))
show('two f16 planes', lambda: eng.conv3x3_wino_f16s(x, u2, cs, b, res, True, d, amax_in=am))
