"""Per-convolution timing of the DRN forward on the GPU (HIP events), next to each layer's own
roofline: 2*MACs against the dense MFMA peak of the dtype and (input + output + weight) bytes
against 8 TB/s.  Shows which layers are MIOpen-efficient and which are memory/launch bound.

    python tools/prof_drn.py [--arch drn_d_22] [--dtype fp32|bf16] [--batch 30] [--hw 1024 2048]
"""
import argparse
import importlib
import os
import sys

os.environ.setdefault('MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD', '0')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--arch', default='drn_d_22')
    ap.add_argument('--dtype', default='fp32')
    ap.add_argument('--batch', type=int, default=30)
    ap.add_argument('--hw', type=int, nargs=2, default=[1024, 2048])
    ap.add_argument('--reps', type=int, default=3)
    args = ap.parse_args()
    drn = importlib.import_module('superpixel-align_amd.drn')
    dtype = {'fp32': torch.float32, 'bf16': torch.bfloat16}[args.dtype]
    model = drn.create_drn(args.arch, None, device='cuda', dtype=dtype)
    H, W = args.hw
    x = torch.rand(args.batch, 3, H, W, device='cuda') * 255

    records = []
    real_conv2d = F.conv2d

    def timed_conv2d(inp, weight, bias=None, stride=1, padding=0, dilation=1, groups=1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = real_conv2d(inp, weight, bias, stride, padding, dilation, groups)
        e1.record()
        records.append((tuple(inp.shape), tuple(weight.shape), tuple(out.shape),
                        stride, dilation, e0, e1))
        return out

    model.batch_predict(x)            # MIOpen find + warm-up
    torch.cuda.synchronize()
    F.conv2d = timed_conv2d
    drn.F.conv2d = timed_conv2d
    tot0, tot1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    per = {}
    whole = []
    for _ in range(args.reps):
        del records[:]
        tot0.record()
        model.batch_predict(x)
        tot1.record()
        torch.cuda.synchronize()
        whole.append(tot0.elapsed_time(tot1))
        for i, r in enumerate(records):
            per.setdefault(i, []).append(r[5].elapsed_time(r[6]))
    es = 4 if dtype == torch.float32 else 2
    peak = 157.3e12 if dtype == torch.float32 else 2500e12
    print('| # | in (B,C,H,W) | weight | stride/dil | ms | TFLOP/s | %%MFMA | GB/s(min traffic) | %%HBM |')
    print('|---|---|---|---|---|---|---|---|---|')
    s_ms = s_fl = 0.0
    for i, r in enumerate(records):
        ish, wsh, osh = r[0], r[1], r[2]
        ms = min(per[i])
        macs = osh[0] * osh[1] * osh[2] * osh[3] * wsh[1] * wsh[2] * wsh[3]
        byts = es * (ish[0] * ish[1] * ish[2] * ish[3] + osh[0] * osh[1] * osh[2] * osh[3]
                     + wsh[0] * wsh[1] * wsh[2] * wsh[3])
        tf = 2 * macs / (ms * 1e-3)
        gb = byts / (ms * 1e-3)
        s_ms += ms
        s_fl += 2 * macs
        print('| %d | %s | %s | %s/%s | %.3f | %.1f | %.0f | %.0f | %.0f |' % (
            i, ish, wsh, r[3], r[4], ms, tf / 1e12, 100 * tf / peak, gb / 1e9, 100 * gb / 8e12))
    print('conv sum %.2f ms (%.1f TFLOP/s), whole forward %.2f ms -> glue %.2f ms'
          % (s_ms, s_fl / s_ms / 1e9, min(whole), min(whole) - s_ms))


if __name__ == '__main__':
    main()
