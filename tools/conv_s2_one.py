import importlib, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
eng = importlib.import_module('superpixel-align_amd.engine').Engine()
torch.manual_seed(3)
for (C, K, Hi, Wi, B) in ((32, 64, 512, 1024, 30), (64, 128, 256, 512, 30), (32, 64, 37, 301, 2), (64, 128, 20, 270, 1)):
    x = (torch.relu(torch.randn((B, C, Hi, Wi), device='cuda')) * 2.3).contiguous(memory_format=torch.channels_last)
    w = torch.randn((K, C, 3, 3), device='cuda') * (2.0 / (9 * C)) ** 0.5
    wd = torch.randn((K, C, 1, 1), device='cuda') * (2.0 / C) ** 0.5
    b = torch.randn((2 * K,), device='cuda')
    wc = torch.zeros((2 * K, 9, C), device='cuda')
    wc[:K] = w.permute(0, 2, 3, 1).reshape(K, 9, C); wc[K:, 4] = wd.reshape(K, C)
    wt2, inv_t = eng.split_planes(wc)
    y, y2, am = eng.conv3x3_s2_f16s(x, wt2, inv_t, b, K, True)
    nb = min(B, 2)
    r1 = torch.relu(F.conv2d(x[:nb].double(), w.double(), b[:K].double(), 2, 1))
    r2 = F.conv2d(x[:nb].double(), wd.double(), b[K:].double(), 2, 0)
    am_in = eng.amax(x)
    def timed(fn, n=5):
        fn(); torch.cuda.synchronize(); t = time.time()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.time() - t) / n * 1e3
    t16 = timed(lambda: eng.conv3x3_s2_f16s(x, wt2, inv_t, b, K, True, amax_in=am_in))
    def mi():
        a = F.conv2d(x, w, None, 2, 1); eng.bias_act_(a, b[:K].contiguous(), None, True)
        c = F.conv2d(x, wd, None, 2, 0); eng.bias_act_(c, b[K:].contiguous(), None, False)
    tm = timed(mi)
    print('%d->%d %dx%d B%d: conv %.2e  projection %.2e of scale vs float64; shapes %s %s; amax %.5g (torch %.5g); %.3f ms vs MIOpen+epilogues %.3f ms' % (
        C, K, Hi, Wi, B, (y[:nb].double() - r1).abs().max().item() / r1.abs().max().item(), (y2[:nb].double() - r2).abs().max().item() / r2.abs().max().item(),
        tuple(y.shape), tuple(y2.shape), am.view(torch.float32).item(), y.abs().max().item(), t16, tm))
