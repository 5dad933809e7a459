"""A kernel that fills an LDS table and reads it back per lane (spa_debug_lds_probe), alone and beside the split-plane stem on a
second stream: is its output the same?  (development aid for HISTORY.md section 5's finding; DESIGN.md section 7, open items)"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
spa = importlib.import_module('superpixel-align_amd')
engine = importlib.import_module('superpixel-align_amd.engine')
lib_mod = importlib.import_module('superpixel-align_amd._lib')
drn = importlib.import_module('superpixel-align_amd.drn')
bench = importlib.import_module('bench')
eng = engine.default_engine()
torch.manual_seed(0)
model = drn.create_drn('drn_d_22', None, device='cuda', dtype=torch.float32)
B = 30
x = torch.from_numpy(bench.make_batch(spa.synth, B, 1024, 2048, seed0=0, integer=True)[0]).cuda()
aux = torch.cuda.Stream()
NWG = 61440                      # the tiles of 30 images of 1024 x 2048
def probe(mode, steps=20):
    out = torch.empty((NWG, 256, 4), dtype=torch.float32, device='cuda')
    lib_mod.check(lib_mod.lib().spa_debug_lds_probe(eng._ctx, out.data_ptr(), NWG, steps, mode, eng._s()))
    return out
def stem(n=1):
    for _ in range(n): eng.drn_stem_d(x, *model._stem, dtype=torch.float32, split=True)
stem(); torch.cuda.synchronize()
for mode, name in ((0, '16-byte reads'), (1, '4-byte reads'), (2, 'broadcast reads')):
    ref = probe(mode); torch.cuda.synchronize()
    again = probe(mode); torch.cuda.synchronize()
    res = []
    for rep in range(6):
        main = torch.cuda.current_stream()
        aux.wait_stream(main)
        stem(2)
        with torch.cuda.stream(aux):
            outs = [probe(mode) for _ in range(4)]
        torch.cuda.synchronize()
        bad = sum(int((o != ref).any(dim=2).sum()) for o in outs)
        res.append(bad)
        if bad and rep == 0:
            o = next(o for o in outs if bool((o != ref).any()))
            idx = (o != ref).any(dim=2).nonzero()[:400]
            lanes = sorted(set(int(i[1]) % 64 for i in idx))
            comps = (o != ref).sum(dim=(0, 1)).tolist()
            print('   first failing launch: lanes (mod 64) %s, differing per component %s' % (lanes, comps))
    print('%-16s alone twice: %d differing threads | beside the stem (4 launches each): %s' % (name, int((again != ref).any(dim=2).sum()), res), flush=True)
