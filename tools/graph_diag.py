#!/usr/bin/env python3
"""Development aid: which map first differs between the launch-by-launch and the captured-graph forward (and between two eager runs)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
drn = importlib.import_module('superpixel-align_amd.drn')
synth = importlib.import_module('superpixel-align_amd.synth')
arch = sys.argv[1] if len(sys.argv) > 1 else 'drn_d_22'
model = drn.create_drn(arch, device='cuda', dtype=torch.float32)
Hh, Ww, Bb = (int(v) for v in (sys.argv[2:5] if len(sys.argv) > 4 else (224, 224, 4)))
x = torch.from_numpy(synth.synth_batch(list(range(70, 70 + Bb)), Hh, Ww)).cuda()
def run(mode):
    os.environ['SPA_DRN_GRAPH'] = mode
    _, maps = model.batch_predict(x)
    torch.cuda.synchronize()
    return [m.clone() for m in maps]
e1, e2 = run('0'), run('0')
g1, g2 = run('1'), run('1')
g3 = run('1')
for name, a, b in (('eager vs eager', e1, e2), ('graph vs graph', g1, g2), ('eager vs graph (capture call)', e1, g1), ('eager vs graph (replay)', e1, g2), ('replay vs replay', g2, g3)):
    print(name, [('=' if torch.equal(u, v) else '%.1e' % float((u.float() - v.float()).abs().max() / v.float().abs().max())) for u, v in zip(a, b)])
print('library convs', drn._EPILOGUE['library_convs'], 'graphs', [e is not False for e in model._graphs.values()])
