"""Bisect: which DRN kernel beside slic_core changes its output, and after how many sweeps?  (development aid)"""
import hashlib, importlib, os, sys
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
# workloads for the main stream
h, w = 128, 256
x512 = torch.relu(torch.randn((B, 512, h, w), device='cuda')).contiguous(memory_format=torch.channels_last)
wt = torch.randn((512, 512, 3, 3), device='cuda') * 0.01
u2, cs = eng.winograd_weights_split(wt); b512 = torch.zeros(512, device='cuda'); am = eng.amax(x512)
x64 = torch.relu(torch.randn((B, 64, 256, 512), device='cuda')).contiguous(memory_format=torch.channels_last)
w64 = (torch.randn((64, 64, 3, 3), device='cuda') * 0.05).permute(0, 2, 3, 1).reshape(64, 9, 64).contiguous()
w64p, inv64 = eng.split_planes(w64); b64 = torch.zeros(64, device='cuda'); am64 = eng.amax(x64)
def wl_forward(): model.batch_predict(x, None, need=[7])
def wl_fused():
    for _ in range(3): eng.conv3x3_wino_f16s(x512, u2, cs, b512, None, True, 2, amax_in=am, fused=True)
def wl_three():
    for _ in range(3): eng.conv3x3_wino_f16s(x512, u2, cs, b512, None, True, 2, amax_in=am)
def wl_direct():
    for _ in range(8): eng.conv3x3_f16s(x64, w64p, inv64, b64, None, True, 1, amax_in=am64)
def wl_matmul():
    y = torch.randn(8192, 8192, device='cuda'); z = y @ y
def wl_none(): pass
for iters in (1, 2, 10):
    ref = eng.slic_core(lab, 200, iters, want_centres=True); torch.cuda.synchronize()
    for name, wl in (('nothing', wl_none), ('matmul', wl_matmul), ('direct 64->64', wl_direct), ('three-launch wino', wl_three), ('fused wino', wl_fused), ('forward', wl_forward)):
        bad = []
        for rep in range(3):
            main = torch.cuda.current_stream()
            aux.wait_stream(main)
            with torch.cuda.stream(aux):
                out = eng.slic_core(lab, 200, iters, want_centres=True)
            wl()
            torch.cuda.synchronize()
            bad.append((int((out[0] != ref[0]).sum()), int((out[1] != ref[1]).sum())))
        print('sweeps %2d beside %-18s differing (labels, centre words): %s  status 0x%x' % (iters, name, bad, eng.status()))
