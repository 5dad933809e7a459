"""Which property of the neighbour kernel matters?  slic_core (2 sweeps) beside several kernels.  (development aid)"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
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
model.batch_predict(x, None, need=[7])
l1 = eng.drn_stem_d(x, *model._stem, dtype=torch.float32, split=True)
xs = x[:8].contiguous()
wc = torch.randn(16, 3, 7, 7, device='cuda')
torch.cuda.synchronize()
def wl_stem_split(): eng.drn_stem_d(x, *model._stem, dtype=torch.float32, split=True)
def wl_stem_f32(): eng.drn_stem_d(x, *model._stem, dtype=torch.float32, split=False)
def wl_stem_bf16(): eng.drn_stem_d(x, *model._stem, dtype=torch.bfloat16, split=False)
def wl_layer2():
    for _ in range(3): model._layer2_call(l1) if hasattr(model, '_layer2_call') else None
def wl_miopen():
    for _ in range(2): F.conv2d(xs, wc, padding=3)
def wl_sort():
    torch.sort(lab.view(-1)[:1 << 24])
def wl_cumsum():
    for _ in range(4): torch.cumsum(lab.view(-1)[:1 << 26], 0)
ref = eng.slic_core(lab, 200, 2, want_centres=True); torch.cuda.synchronize()
for name, wl in (('stem split', wl_stem_split), ('stem float32 mfma', wl_stem_f32), ('stem bf16', wl_stem_bf16), ('MIOpen conv 7x7', wl_miopen), ('torch sort', wl_sort), ('torch cumsum', wl_cumsum)):
    try:
        wl(); torch.cuda.synchronize()
    except Exception as e:
        print('beside %-18s skipped: %s' % (name, str(e)[:80])); continue
    bad = []
    for rep in range(6):
        main = torch.cuda.current_stream()
        aux.wait_stream(main)
        with torch.cuda.stream(aux):
            out = eng.slic_core(lab, 200, 2, want_centres=True)
        wl()
        torch.cuda.synchronize()
        bad.append(int((out[1] != ref[1]).sum()))
    print('beside %-18s differing centre words: %s  status 0x%x' % (name, bad, eng.status()), flush=True)
