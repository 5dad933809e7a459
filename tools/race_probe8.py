"""Round 6: the REPRODUCER of the co-residency miscompare (HISTORY.md section 5).  k_slic_assign's shelved variant — the x part
of the spatial term shared through an LDS table, selected with spa_debug_set(ctx, 2, 1) in a library built with `make EXTRA=-DSPA_DIAG` — runs
(a) alone after a device synchronise behind another kernel (nothing runs beside it; the CUs hold what that kernel left) and (b) beside it on a
second stream; the centre table after two sweeps is compared with a run alone.  Expected: (a) always identical; (b) 50-700 differing words beside
the stem and beside the direct matrix-instruction kernels of the forward, 0 beside a fill, a plain LDS kernel, a library GEMM or a Winograd layer
(those cannot share a CU with it).  With SPA_LDSX=0 the shipped kernel runs: 0 everywhere.
    make -C superpixel-align_amd/csrc EXTRA=-DSPA_DIAG && python tools/race_probe8.py"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
spa = importlib.import_module('superpixel-align_amd')
engine = importlib.import_module('superpixel-align_amd.engine')
drn = importlib.import_module('superpixel-align_amd.drn')
bench = importlib.import_module('bench')
eng = engine.default_engine()
if os.environ.get('SPA_LDSX', '1') != '0':
    eng.debug_set(2, 1)          # (needs a diagnostic build)
torch.manual_seed(0)
model = drn.create_drn('drn_d_22', None, device='cuda', dtype=torch.float32)
B = 30
x = torch.from_numpy(bench.make_batch(spa.synth, B, 1024, 2048, seed0=0, integer=True)[0]).cuda()
lab = eng.rgb2lab(x, 0.1)
aux = torch.cuda.Stream()
model.batch_predict(x, None, need=[7])
l1 = eng.drn_stem_d(x, *model._stem, dtype=torch.float32, split=True)
torch.cuda.synchronize()
def wl_stem(): eng.drn_stem_d(x, *model._stem, dtype=torch.float32, split=True)
def wl_rest():
    with torch.no_grad(): model.forward_maps(None, layer1_out=l1)
def wl_fill():          # a tensor fill: no LDS
    torch.empty(64 << 20, device='cuda').fill_(1.0)
lib_mod = importlib.import_module('superpixel-align_amd._lib')
probe_out = torch.empty((40000, 256, 4), device='cuda')
def wl_ldsprobe():      # plain LDS traffic (per-lane 16-byte reads of a 4 KB table in a loop), no LDS-DMA, no matrix instructions
    lib_mod.check(lib_mod.lib().spa_debug_lds_probe(eng._ctx, probe_out.data_ptr(), 40000, 32, 0, eng._s()))
ma, mb = torch.randn(8192, 8192, device='cuda'), torch.randn(8192, 8192, device='cuda')
def wl_matmul():        # a library float32 GEMM (matrix instructions, the library's own LDS use)
    torch.matmul(ma, mb)
def wl_wino():          # one Winograd layer of the network: transforms (no LDS) + the GEMM (LDS-DMA, matrix instructions, 128 KB of LDS)
    eng.conv3x3_wino_f16s(wx, wu2, wcs, wbias, None, True, 2, amax_in=wam)
wx = torch.relu(torch.randn((15, 256, 128, 256), device='cuda')).contiguous(memory_format=torch.channels_last)
ww = torch.randn((256, 256, 3, 3), device='cuda') * 0.03
wbias = torch.randn((256,), device='cuda')
wu2, wcs = eng.winograd_weights_split(ww)
wam = eng.amax(wx)
wl_wino(); wl_matmul(); wl_ldsprobe(); torch.cuda.synchronize()
ref = eng.slic_core(lab, 200, 2, want_centres=True); torch.cuda.synchronize()
for name, wl in (('stem', wl_stem), ('behind the stem', wl_rest), ('a fill', wl_fill), ('LDS probe kernel', wl_ldsprobe), ('torch.matmul f32', wl_matmul), ('Winograd layer', wl_wino)):
    after, beside = [], []
    for rep in range(4):
        wl(); torch.cuda.synchronize()
        out = eng.slic_core(lab, 200, 2, want_centres=True); torch.cuda.synchronize()
        after.append(int((out[1] != ref[1]).sum()))
        main = torch.cuda.current_stream()
        aux.wait_stream(main)
        with torch.cuda.stream(aux):
            out = eng.slic_core(lab, 200, 2, want_centres=True)
        wl()
        torch.cuda.synchronize()
        beside.append(int((out[1] != ref[1]).sum()))
    print('%-16s differing centre words: ALONE AFTER it %s | BESIDE it %s   status 0x%x' % (name, after, beside, eng.status()), flush=True)
