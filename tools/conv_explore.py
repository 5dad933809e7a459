"""Which MIOpen configuration is fastest for the DRN's convolution shapes?  Times each shape in
NCHW and NHWC storage, and the dilated layers also as dilation-1 convolutions over the
space-to-batch rearrangement (a d-dilated 3x3 conv == d*d independent undilated convs on the
d-strided sub-grids).  Exploration tool, not part of the product path.
"""
import argparse
import os
import sys

os.environ.setdefault('MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD', '0')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402


def timeit(fn, reps=5):
    fn()
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--dtype', default='fp32')
    ap.add_argument('--batch', type=int, default=30)
    args = ap.parse_args()
    torch.backends.cudnn.benchmark = True
    dt = {'fp32': torch.float32, 'bf16': torch.bfloat16}[args.dtype]
    B = args.batch
    shapes = [  # cin, cout, H, W, k, stride, dil
        (3, 16, 1024, 2048, 7, 1, 1), (16, 16, 1024, 2048, 3, 1, 1), (16, 32, 1024, 2048, 3, 2, 1),
        (64, 64, 256, 512, 3, 1, 1), (128, 128, 128, 256, 3, 1, 1),
        (256, 256, 128, 256, 3, 1, 2), (512, 512, 128, 256, 3, 1, 4), (512, 512, 128, 256, 3, 1, 2),
        (512, 512, 128, 256, 3, 1, 1)]
    for cin, cout, H, W, k, st, d in shapes:
        x = torch.randn(B, cin, H, W, device='cuda', dtype=dt)
        w = torch.randn(cout, cin, k, k, device='cuda', dtype=dt) * 0.05
        pad = d * (k // 2)
        flops = 2.0 * B * cout * (H // st) * (W // st) * cin * k * k
        res = {}
        ref = None
        for name, fmt in (('nchw', torch.contiguous_format), ('nhwc', torch.channels_last)):
            xi, wi = x.contiguous(memory_format=fmt), w.contiguous(memory_format=fmt)
            try:
                ms = timeit(lambda: F.conv2d(xi, wi, None, st, pad, d))
                res[name] = ms
                y = F.conv2d(xi, wi, None, st, pad, d)
                if ref is None:
                    ref = y
                else:
                    res[name + '_relerr'] = float((y - ref).abs().max() / ref.abs().max())
            except Exception as e:  # noqa: BLE001
                res[name] = 'fail %s' % type(e).__name__
        if d > 1:
            for name, fmt in (('s2b_nchw', torch.contiguous_format), ('s2b_nhwc', torch.channels_last)):
                # (B,C,H,W) -> (B*d*d, C, H/d, W/d)
                xs = x.view(B, cin, H // d, d, W // d, d).permute(0, 3, 5, 1, 2, 4).reshape(
                    B * d * d, cin, H // d, W // d).contiguous(memory_format=fmt)
                wi = w.contiguous(memory_format=fmt)
                try:
                    res[name] = timeit(lambda: F.conv2d(xs, wi, None, 1, 1, 1))
                    ys = F.conv2d(xs, wi, None, 1, 1, 1)
                    y = ys.view(B, d, d, cout, H // d, W // d).permute(0, 3, 4, 1, 5, 2).reshape(
                        B, cout, H, W)
                    res[name + '_relerr'] = float((y - ref).abs().max() / ref.abs().max())
                except Exception as e:  # noqa: BLE001
                    res[name] = 'fail %s' % type(e).__name__
        out = ['%dx%d->%d k%d s%d d%d' % (H, W, cout, k, st, d) + ' cin %d' % cin]
        for kk, v in res.items():
            if isinstance(v, float) and not kk.endswith('relerr'):
                out.append('%s %.3f ms (%.0f TF)' % (kk, v, flops / v / 1e9))
            else:
                out.append('%s %s' % (kk, v if isinstance(v, str) else '%.2e' % v))
        print(' | '.join(out), flush=True)


if __name__ == '__main__':
    main()
