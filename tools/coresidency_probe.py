"""Do the Winograd transforms (HBM streaming) and the Winograd GEMMs (matrix cores) overlap when they are separate launches
on two streams?  Two independent 512 -> 512 layers (three launches each) run back to back on one stream, then on two streams
with one of them delayed by about one transform, for the GEMM tile sizes the environment selects:

    python tools/coresidency_probe.py                       (256 x 256 tiles, one workgroup per CU: the default)
    SPA_GEMM16_TILE=128 SPA_GEMM16_PER_CU=2 python tools/coresidency_probe.py
"""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

engine = importlib.import_module('superpixel-align_amd.engine')
eng = engine.Engine()
torch.manual_seed(0)
B, H, W, C, dil, LAYERS = 15, 128, 256, 512, 2, 4
xs = [torch.relu(torch.randn((B, C, H, W), device='cuda')).contiguous(memory_format=torch.channels_last) for _ in range(2)]
w = torch.randn((C, C, 3, 3), device='cuda') * (2.0 / (9 * C)) ** 0.5
bias = torch.randn((C,), device='cuda')
u2, cs = eng.winograd_weights_split(w)
ams = [eng.amax(x) for x in xs]
keep = [{}, {}]


def layer(i):
    return eng.conv3x3_wino_f16s(xs[i], u2, cs, bias, None, True, dil, amax_in=ams[i])


ref = layer(0)[0].clone()
torch.cuda.synchronize()
eng.prof_enable(True)
for _ in range(3):
    layer(0)
torch.cuda.synchronize()
for name, (ms, n) in eng.prof_read().items():
    if n:
        print('%-24s avg %8.1f us (15 images)' % (name, ms / n * 1e3))
eng.prof_enable(False)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def timed(two_streams, delay_cycles):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    cur = torch.cuda.current_stream()
    e0.record(cur)
    if two_streams:
        sa.wait_stream(cur)
        sb.wait_stream(cur)
        with torch.cuda.stream(sa):
            for _ in range(LAYERS):
                layer(0)
        with torch.cuda.stream(sb):
            if delay_cycles:
                torch.cuda._sleep(delay_cycles)
            for _ in range(LAYERS):
                layer(1)
        cur.wait_stream(sa)
        cur.wait_stream(sb)
    else:
        for _ in range(LAYERS):
            layer(0)
            layer(1)
    e1.record(cur)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)


for _ in range(2):
    timed(False, 0)
    timed(True, 0)
print('tile %s per CU %s: %d + %d layers of 15 images' % (os.environ.get('SPA_GEMM16_TILE', '256'), os.environ.get('SPA_GEMM16_PER_CU', 'auto'), LAYERS, LAYERS))
print('  one stream            %.3f ms' % min(timed(False, 0) for _ in range(3)))
for d in (0, 1000000, 2000000, 4000000):
    print('  two streams, delay %7d cycles   %.3f ms' % (d, min(timed(True, d) for _ in range(3))))
print('  same bits as before the runs:', bool(torch.equal(layer(0)[0], ref)), ' status 0x%x' % eng.status())
