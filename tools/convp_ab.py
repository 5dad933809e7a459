#!/usr/bin/env python3
"""Development aid (round 6): the split-plane direct 3x3 layers of 64 / 128 channels on the planes-in-LDS kernel
(spa_convp.hip, SPA_CONVP unset or 1) against the kernel it replaces (spa_conv32.hip, SPA_CONVP=0): time per launch and a
digest of the output and of the tracked maximum, so that two processes can be compared bit for bit:
    SPA_CONVP=0 python tools/convp_ab.py ; SPA_CONVP=1 python tools/convp_ab.py
    python tools/convp_ab.py --both        (runs the two settings as child processes and compares the digests)"""
import hashlib, importlib, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SHAPES = (  # B, C, H, W, dilation, residual, relu
    (30, 64, 256, 512, 1, True, True), (30, 64, 256, 512, 1, False, True), (30, 128, 128, 256, 1, True, True),
    (30, 128, 128, 256, 1, False, True), (4, 64, 77, 500, 1, True, True), (3, 128, 50, 130, 2, False, True),
    (2, 64, 40, 300, 4, True, False), (2, 128, 33, 100, 1, True, True), (1, 64, 9, 37, 2, False, True),
    (2, 128, 64, 257, 3, True, True), (1, 64, 300, 1024, 1, False, True))


def run():
    import torch
    eng_mod = importlib.import_module('superpixel-align_amd.engine')
    eng = eng_mod.default_engine()
    torch.manual_seed(0)
    for (B, C, H, W, dil, res, relu) in (SHAPES[:4] if '--big' in sys.argv else SHAPES):
        x = torch.randn(B, C, H, W, device='cuda').contiguous(memory_format=torch.channels_last)
        r = torch.randn(B, C, H, W, device='cuda').contiguous(memory_format=torch.channels_last) if res else None
        w = torch.randn(C, C, 3, 3, device='cuda') * 0.05
        b = torch.randn(C, device='cuda')
        wt = w.permute(0, 2, 3, 1).reshape(C, 9, C).contiguous()
        wt2, inv_t = eng_mod.Engine.split_planes(wt)
        am = eng.amax(x)
        y, a2 = eng.conv3x3_f16s(x, wt2, inv_t, b, r, relu, dil, amax_in=am)
        torch.cuda.synchronize()
        reps = 10 if B >= 30 else 3
        t = time.time()
        for _ in range(reps):
            y, a2 = eng.conv3x3_f16s(x, wt2, inv_t, b, r, relu, dil, amax_in=am)
        torch.cuda.synchronize()
        dt = (time.time() - t) / reps * 1e3
        ref = torch.nn.functional.conv2d(x[:1].double(), w.double(), b.double(), 1, dil, dil)
        if res:
            ref = ref + r[:1].double()
        if relu:
            ref = ref.relu()
        err = float((y[:1].double() - ref).abs().max() / ref.abs().max())
        print('C %3d B %2d %4dx%4d dil %d res %d relu %d: %7.3f ms  y %s amax %08x  err vs float64 %.1e' % (
            C, B, H, W, dil, res, relu, dt, hashlib.sha1(y.cpu().numpy().tobytes()).hexdigest()[:12],
            int(a2.item()) & 0xffffffff if hasattr(a2, 'item') else int(a2) & 0xffffffff, err), flush=True)
    st = eng.status()
    print('device status 0x%x' % st)


if __name__ == '__main__':
    if '--both' in sys.argv:
        outs = []
        for v in ('0', '1'):
            env = dict(os.environ, SPA_CONVP=v)
            o = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True)
            print('SPA_CONVP=%s\n%s%s' % (v, o.stdout, o.stderr[-2000:] if o.returncode else ''))
            outs.append([l.split('ms')[1].split('err')[0] for l in o.stdout.splitlines() if ' ms ' in l])
        same = outs[0] == outs[1] and len(outs[0]) == len(SHAPES)
        print('bit-identical on %d shapes: %s' % (len(SHAPES), same))
        sys.exit(0 if same else 1)
    run()
