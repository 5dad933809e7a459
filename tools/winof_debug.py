"""Where does the fused Winograd launch differ from the three launches?  Compares V, M and Y."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
engine = importlib.import_module('superpixel-align_amd.engine')
eng = engine.Engine()
torch.manual_seed(0)
B, Cin, Cout, H, W, dil = (int(v) for v in (sys.argv[1:7] if len(sys.argv) > 6 else (1, 128, 256, 24, 300, 2)))
x = torch.relu(torch.randn((B, Cin, H, W), device='cuda')).contiguous(memory_format=torch.channels_last)
w = torch.randn((Cout, Cin, 3, 3), device='cuda') * (2.0 / (9 * Cin)) ** 0.5
bias = torch.randn((Cout,), device='cuda')
u2, cs = eng.winograd_weights_split(w)
am = eng.amax(x)
k3, k1 = {}, {}
y3, _ = eng.conv3x3_wino_f16s(x, u2, cs, bias, None, True, dil, amax_in=am, _keep=k3)
torch.cuda.synchronize()
y1, _ = eng.conv3x3_wino_f16s(x, u2, cs, bias, None, True, dil, amax_in=am, fused=True, _keep=k1)
torch.cuda.synchronize()
print('status 0x%x' % eng.status())
T = int(eng._lib.spa_wino4_tiles(B, H, W, dil))
import math
d = dil
hs, ws = -(-H // d), -(-W // d)
Treal = B * d * d * (-(-hs // 4)) * (-(-ws // 4))
print('Tpad', T, 'T', Treal)
for name in ('v', 'm'):
    a, b = k3[name][:, :Treal], k1[name][:, :Treal]
    ne = (a != b)
    print(name, 'differing', int(ne.sum()), 'of', a.numel(), '| positions with differences:', ne.flatten(1).any(1).nonzero().flatten().tolist()[:40])
    if ne.any():
        idx = ne.nonzero()[:5]
        for i in idx:
            print('   ', i.tolist(), float(a[tuple(i)]), float(b[tuple(i)]))
        rows = ne.any(2).any(0).nonzero().flatten()
        print('    rows with differences: %d, first %s last %s' % (rows.numel(), rows[:5].tolist(), rows[-5:].tolist()))
ne = y1 != y3
print('y differing', int(ne.sum()), 'of', y1.numel())
if os.environ.get('WF_NET'):
    # layer by layer inside the network: the same input through both forms
    drn = importlib.import_module('superpixel-align_amd.drn')
    synth = importlib.import_module('superpixel-align_amd.synth')
    m = drn.create_drn('drn_d_22', device='cuda', dtype=torch.float32)
    xs = synth.synth_batch([3, 4], 256, 512)
    orig = eng.conv3x3_wino_f16s
    n = [0]
    def both(x, u2, cs, bias, residual=None, relu=True, dilation=1, amax_in=None, track_amax=True, fused=False, _keep=None):
        y3, a3 = orig(x, u2, cs, bias, residual, relu, dilation, amax_in, track_amax, False)
        if u2.shape[1] % 256 == 0 and x.shape[1] >= 160:
            for rep in range(3):
                y1, a1 = orig(x, u2, cs, bias, residual, relu, dilation, amax_in, track_amax, True)
                ne = int((y1 != y3).sum())
                print('layer %d: %s -> %d dil %d res %s | differing %d | amax %d %d' % (n[0], tuple(x.shape), u2.shape[1], dilation, residual is not None, ne, int(a1), int(a3)))
                if ne:
                    idx = (y1 != y3).nonzero()
                    print('   first', idx[0].tolist(), float(y1[tuple(idx[0])]), float(y3[tuple(idx[0])]), 'last', idx[-1].tolist())
        n[0] += 1
        return y3, a3
    drn._EPILOGUE['engine'].conv3x3_wino_f16s = both
    m.batch_predict(xs, need=[7])
    print('status 0x%x' % eng.status())
