import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
engine = importlib.import_module('superpixel-align_amd.engine')
eng = engine.Engine()
torch.manual_seed(0)
for (B, Cin, Cout, H, W, dil) in [(2, 128, 128, 32, 64, 1), (2, 128, 128, 32, 64, 2), (2, 128, 256, 32, 64, 1), (2, 256, 256, 32, 64, 2), (30, 128, 128, 128, 256, 1), (2, 128, 384, 32, 64, 1)]:
    x = torch.relu(torch.randn((B, Cin, H, W), device='cuda')).contiguous(memory_format=torch.channels_last)
    w = torch.randn((Cout, Cin, 3, 3), device='cuda') * (2.0 / (9 * Cin)) ** 0.5
    bias = torch.randn((Cout,), device='cuda')
    u2, cs = eng.winograd_weights_split(w)
    am = eng.amax(x)
    ks = []
    ys = []
    for rep in range(4):
        k = {}
        y, a = eng.conv3x3_wino_f16s(x, u2, cs, bias, None, True, dil, amax_in=am, _keep=k)
        torch.cuda.synchronize()
        ks.append({n: t.clone() for n, t in k.items()}); ys.append(y)
    T = ks[0]['v'].shape[1]
    print((B, Cin, Cout, H, W, dil), 'Tpad', T, 'v diff', [int((ks[i]['v'] != ks[0]['v']).sum()) for i in range(1, 4)],
          'm diff', [int((ks[i]['m'] != ks[0]['m']).sum()) for i in range(1, 4)], 'y diff', [int((ys[i] != ys[0]).sum()) for i in range(1, 4)])
    d = (ks[1]['m'] != ks[0]['m'])
    if d.any():
        idx = d.nonzero()
        print('   m first', idx[0].tolist(), 'last', idx[-1].tolist(), 'positions', sorted(set(idx[:, 0].tolist()))[:40], 'rows', int(idx[:, 1].min()), int(idx[:, 1].max()), 'cols', int(idx[:, 2].min()), int(idx[:, 2].max()))
