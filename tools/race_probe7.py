"""Stop the SLIC sweep loop after n launches (SPA_SLIC_STOP, read per call) and compare labels / centre table / masks with a run
alone: which launch is the first whose output changes beside the split stem?  (development aid)"""
import ctypes, importlib, os, re, sys
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
lab = eng.rgb2lab(x, 0.1)
aux = torch.cuda.Stream()
src = open(os.path.join(os.path.dirname(lib_mod.__file__), 'csrc', 'spa_common.h')).read()
names = re.findall(r'^\s*(WS_[A-Z_0-9]+)\s*(?:=\s*0)?,', src, re.M)
def peek(name, nbytes):
    host = np.zeros(nbytes // 4, np.uint32)
    lib_mod.check(lib_mod.lib().spa_debug_peek(eng._ctx, names.index(name), 0, nbytes, host.ctypes.data_as(ctypes.c_void_p)))
    return host
eng.drn_stem_d(x, *model._stem, dtype=torch.float32, split=True); torch.cuda.synchronize()
def run(n, beside):
    os.environ['SPA_SLIC_STOP'] = str(n)
    main = torch.cuda.current_stream()
    aux.wait_stream(main)
    if beside:
        for _ in range(2): eng.drn_stem_d(x, *model._stem, dtype=torch.float32, split=True)
    with torch.cuda.stream(aux):
        out = eng.slic_core(lab, 200, 3, want_centres=False)
    torch.cuda.synchronize()
    cen = peek('WS_CENTRES', B * 200 * 16 * 4).reshape(B, 200, 16)
    rm = peek('WS_ROWMASK', 8 << 20)
    fm = peek('WS_FINEMASK', 64 << 20)
    return out.cpu().numpy(), cen.copy(), rm.copy(), fm.copy()
kinds = ['assign1', 'update1', 'assign2', 'update2', 'assign3']
for n in range(1, 6):
    ref = run(n, False)
    again = run(n, False)
    same_alone = [int((a != b).sum()) for a, b in zip(ref, again)]
    diffs = []
    for rep in range(4):
        got = run(n, True)
        diffs.append([int((a != b).sum()) for a, b in zip(ref, got)])
    print('after %-8s alone twice: %s | beside the stem, differing (labels, centre words [cols 0-11], rowmask words, fine words): %s' % (kinds[n - 1], same_alone, diffs), flush=True)
# where are the differing pixels of assign2?
ref = run(3, False)
for rep in range(2):
    got = run(3, True)
    d = np.argwhere(ref[0] != got[0])
    print('assign2: %d differing pixels' % len(d))
    for (b, y, xx) in d[:24]:
        print('   image %2d y %4d x %4d (tile row %d, x mod 32 = %2d, y mod 32 = %2d): alone %3d beside %3d' % (b, y, xx, y // 32, xx % 32, y % 32, ref[0][b, y, xx], got[0][b, y, xx]))
    import collections
    print('   x mod 32 histogram:', sorted(collections.Counter((d[:, 2] % 32).tolist()).items()))
    print('   y mod 8 histogram:', sorted(collections.Counter((d[:, 1] % 8).tolist()).items()))
    print('   tiles:', len(set((int(b), int(y) // 32, int(xx) // 32) for b, y, xx in d)))
