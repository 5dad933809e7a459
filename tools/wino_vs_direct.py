import importlib, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
eng = importlib.import_module('superpixel-align_amd.engine').Engine()
torch.manual_seed(1)
def timed(fn, n=5):
    fn(); torch.cuda.synchronize(); t = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time() - t) / n * 1e3
for Cin, Cout, d in ((128, 256, 2), (256, 512, 4), (128, 128, 1)):
    B = 30
    x = torch.relu(torch.randn((B, Cin, 128, 256), device='cuda')).contiguous(memory_format=torch.channels_last)
    w = torch.randn((Cout, Cin, 3, 3), device='cuda') * (2.0 / (9 * Cin)) ** 0.5
    b = torch.randn((Cout,), device='cuda')
    u2, cs = eng.winograd_weights_split(w)
    wt2, inv_t = eng.split_planes(w.permute(0, 2, 3, 1).reshape(Cout, 9, Cin).contiguous())
    am = eng.amax(x)
    tw = timed(lambda: eng.conv3x3_wino_f16s(x, u2, cs, b, None, True, d, amax_in=am))
    td = timed(lambda: eng.conv3x3_f16s(x, wt2, inv_t, b, None, True, d, amax_in=am))
    print('%d->%d d%d: winograd split %.3f ms, direct split %.3f ms' % (Cin, Cout, d, tw, td))
