import importlib, sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
eng = importlib.import_module('superpixel-align_amd.engine').Engine()
torch.manual_seed(3)
for (C, K, taps, H, W, d, B) in ((128, 128, 9, 128, 256, 1, 30), (256, 256, 9, 128, 256, 2, 30), (64, 64, 9, 256, 512, 1, 30), (128, 256, 1, 128, 256, 1, 30), (256, 512, 1, 128, 256, 1, 30), (64, 64, 9, 60, 130, 2, 2)):
    x = (torch.relu(torch.randn((B, C, H, W), device='cuda')) * 2.3).contiguous(memory_format=torch.channels_last)
    w = torch.randn((K, C, 3, 3) if taps == 9 else (K, C, 1, 1), device='cuda') * (2.0 / (taps * C)) ** 0.5
    b = torch.randn((K,), device='cuda')
    res = torch.randn((B, K, H, W), device='cuda').contiguous(memory_format=torch.channels_last)
    wt = (w.permute(0, 2, 3, 1).reshape(K, taps, C)).contiguous()
    wt2, inv_t = eng.split_planes(wt)
    y32 = eng.conv3x3_f32(x, wt, b, res, True, d)
    y16, am = eng.conv3x3_f16s(x, wt2, inv_t, b, res, True, d)
    nb = min(B, 2)
    ref = torch.relu(F.conv2d(x[:nb].double(), w.double(), b.double(), 1, d if taps == 9 else 0, d) + res[:nb].double())
    s = ref.abs().max().item()
    def timed(fn, n=5):
        fn(); torch.cuda.synchronize(); t = time.time()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.time() - t) / n * 1e3
    am_in = eng.amax(x)
    t32 = timed(lambda: eng.conv3x3_f32(x, wt, b, res, True, d)); t16 = timed(lambda: eng.conv3x3_f16s(x, wt2, inv_t, b, res, True, d, amax_in=am_in))
    print('%d->%d taps %d %dx%d d%d B%d: f32 %.2e  split %.2e of scale vs float64; amax %.5g (torch %.5g); %.3f ms vs %.3f ms' % (
        C, K, taps, H, W, d, B, (y32[:nb].double() - ref).abs().max().item() / s, (y16[:nb].double() - ref).abs().max().item() / s,
        am.view(torch.float32).item(), y16.abs().max().item(), t32, t16))
