"""Bisect further: which kernels beside slic_core (2 sweeps) change its centres?  (development aid)"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
spa = importlib.import_module('superpixel-align_amd')
engine = importlib.import_module('superpixel-align_amd.engine')
bench = importlib.import_module('bench')
eng = engine.default_engine()
torch.manual_seed(0)
B = 30
x = torch.from_numpy(bench.make_batch(spa.synth, B, 1024, 2048, seed0=0, integer=True)[0]).cuda()
lab = eng.rgb2lab(x, 0.1)
aux = torch.cuda.Stream()
def mk(C, K, H, W, taps=9):
    xx = torch.relu(torch.randn((B, C, H, W), device='cuda')).contiguous(memory_format=torch.channels_last)
    w = (torch.randn((K, C, 3, 3) if taps == 9 else (K, C, 1, 1), device='cuda') * 0.05).permute(0, 2, 3, 1).reshape(K, taps, C).contiguous()
    wp, inv = eng.split_planes(w)
    return xx, w, wp, inv, torch.zeros(K, device='cuda'), eng.amax(xx)
x64, w64, w64p, inv64, b64, am64 = mk(64, 64, 256, 512)
x128, w128, w128p, inv128, b128, am128 = mk(128, 128, 128, 256)
x1, w1, w1p, inv1, b1, am1 = mk(128, 256, 128, 256, taps=1)
big = torch.randn(64, 1024, 1024, device='cuda')
def wl_elementwise():
    for _ in range(20): big.mul_(1.0001)
def wl_direct64():
    for _ in range(8): eng.conv3x3_f16s(x64, w64p, inv64, b64, None, True, 1, amax_in=am64)
def wl_direct64_noamax():
    for _ in range(8): eng.conv3x3_f16s(x64, w64p, inv64, b64, None, True, 1, amax_in=am64, track_amax=False)
def wl_direct64_f32():
    for _ in range(4): eng.conv3x3_f32(x64, w64, b64, None, True, 1)
def wl_direct128():
    for _ in range(8): eng.conv3x3_f16s(x128, w128p, inv128, b128, None, True, 1, amax_in=am128)
def wl_1x1():
    for _ in range(16): eng.conv3x3_f16s(x1, w1p, inv1, b1, None, True, 1, amax_in=am1)
def wl_amax():
    for _ in range(20): eng.amax(x64)
def wl_none(): pass
ref = eng.slic_core(lab, 200, 2, want_centres=True); torch.cuda.synchronize()
for name, wl in (('nothing', wl_none), ('elementwise', wl_elementwise), ('k_amax', wl_amax), ('direct 64 split', wl_direct64), ('direct 64 split, no amax', wl_direct64_noamax),
                 ('direct 64 float32 mfma', wl_direct64_f32), ('direct 128 split', wl_direct128), ('1x1 128->256 split', wl_1x1)):
    bad = []
    for rep in range(6):
        main = torch.cuda.current_stream()
        aux.wait_stream(main)
        with torch.cuda.stream(aux):
            out = eng.slic_core(lab, 200, 2, want_centres=True)
        wl()
        torch.cuda.synchronize()
        bad.append(int((out[1] != ref[1]).sum()))
    print('beside %-26s differing centre words: %s  status 0x%x' % (name, bad, eng.status()), flush=True)
