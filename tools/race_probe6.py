"""slic_core with 1 sweep (assign + update) and 2 sweeps, the split stem enqueued BEFORE it so that every kernel overlaps it.  (development aid)"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
spa = importlib.import_module('superpixel-align_amd')
engine = importlib.import_module('superpixel-align_amd.engine')
drn = importlib.import_module('superpixel-align_amd.drn')
bench = importlib.import_module('bench')
eng = engine.default_engine()
torch.manual_seed(0)
model = drn.create_drn('drn_d_22', None, device='cuda', dtype=torch.float32)
B = 30
x = torch.from_numpy(bench.make_batch(spa.synth, B, 1024, 2048, seed0=0, integer=True)[0]).cuda()
lab = eng.rgb2lab(x, 0.1)
aux = torch.cuda.Stream()
eng.drn_stem_d(x, *model._stem, dtype=torch.float32, split=True); torch.cuda.synchronize()
def stem(n):
    for _ in range(n): eng.drn_stem_d(x, *model._stem, dtype=torch.float32, split=True)
for iters in (1, 2, 3):
    ref = eng.slic_core(lab, 200, iters, want_centres=True); torch.cuda.synchronize()
    for nstem in (1, 4):
        bad = []
        for rep in range(6):
            main = torch.cuda.current_stream()
            aux.wait_stream(main)
            stem(nstem)                                   # main first
            with torch.cuda.stream(aux):
                out = eng.slic_core(lab, 200, iters, want_centres=True)
            torch.cuda.synchronize()
            bad.append((int((out[0] != ref[0]).sum()), int((out[1] != ref[1]).sum())))
        print('sweeps %d, %d stems first: differing (labels, centre words) %s' % (iters, nstem, bad), flush=True)
