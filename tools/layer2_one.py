import importlib, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
eng = importlib.import_module('superpixel-align_amd.engine').Engine()
torch.manual_seed(5)
for (H, W, B) in ((37, 61, 2), (64, 130, 1), (1024, 2048, 30)):
    x = (torch.relu(torch.randn((B, 16, H, W), device='cuda')) * 1.7).contiguous(memory_format=torch.channels_last)
    w = torch.randn((32, 16, 3, 3), device='cuda') * (2.0 / (9 * 16)) ** 0.5
    b = torch.randn((32,), device='cuda')
    wp, inv_t = eng.layer2_planes(w)
    y = eng.drn_layer2_f16s(x, wp, inv_t, b)
    nb = min(B, 2)
    ref = torch.relu(F.conv2d(x[:nb].double(), w.double(), b.double(), 2, 1))
    err = (y[:nb].double() - ref).abs().max().item() / ref.abs().max().item()
    am = eng.amax(x)
    def timed(fn, n=5):
        fn(); torch.cuda.synchronize(); t = time.time()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.time() - t) / n * 1e3
    t16 = timed(lambda: eng.drn_layer2_f16s(x, wp, inv_t, b, amax_in=am))
    def mi():
        a = F.conv2d(x, w, None, 2, 1); eng.bias_act_(a, b, None, True)
    tm = timed(mi)
    print('%dx%d B%d: %.2e of scale vs float64, shape %s, amax %.5g (torch %.5g); %.3f ms vs MIOpen + epilogue %.3f ms' % (
        H, W, B, err, tuple(y.shape), y._spa_amax.view(torch.float32).item(), y.abs().max().item(), t16, tm))
