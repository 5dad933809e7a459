"""libspalign's bf16 implicit-GEMM 3x3 convolution (spa_conv3x3_bf16) against MIOpen (torch F.conv2d) on the
heavy DRN layer shapes: numerics (float32 reference of the same bf16 operands) and TFLOP/s.
    python tools/conv_bench.py [--batch 30] [--quick]"""
import argparse
import importlib
import os
import sys

os.environ.setdefault('MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD', '0')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=30)
ap.add_argument('--quick', action='store_true')
ap.add_argument('--tile', type=int, default=4, choices=[2, 4], help='--dtype wino: F(2x2,3x3) or F(4x4,3x3)')
ap.add_argument('--dtype', default='bf16', choices=['bf16', 'fp32', 'wino'], help='fp32: spa_conv3x3_f32 against MIOpen float32')
a = ap.parse_args()
engine = importlib.import_module('superpixel-align_amd.engine')
eng = engine.Engine()
torch.backends.cudnn.benchmark = True
torch.manual_seed(0)


def run32(B, Cin, Cout, H, W, dil, res, reps=3):
    x = torch.randn((B, Cin, H, W), device='cuda').contiguous(memory_format=torch.channels_last)
    w = torch.randn((Cout, Cin, 3, 3), device='cuda') * (2.0 / (9 * Cin)) ** 0.5
    w_cl = w.contiguous(memory_format=torch.channels_last)
    bias = torch.randn((Cout,), device='cuda')
    r = torch.randn((B, Cout, H, W), device='cuda').contiguous(memory_format=torch.channels_last) if res else None
    wt = w.permute(0, 2, 3, 1).reshape(Cout, 9, Cin).contiguous()
    y = eng.conv3x3_f32(x, wt, bias, r, True, dil)
    nb = min(B, 2)
    ref = F.conv2d(x[:nb].double(), w.double(), bias.double(), 1, dil, dil)
    if res:
        ref = ref + r[:nb].double()
    ref = torch.relu(ref)
    err = (y[:nb].double() - ref).abs().max().item()
    scale = ref.abs().max().item()
    mi = torch.relu(F.conv2d(x[:nb], w_cl, bias, 1, dil, dil) + (r[:nb] if res else 0))
    err_mi = (mi.double() - ref).abs().max().item()

    def t(fn):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    ms_own = t(lambda: eng.conv3x3_f32(x, wt, bias, r, True, dil))
    ms_ref = t(lambda: F.conv2d(x, w_cl, None, 1, dil, dil))
    ms_full = t(lambda: eng.bias_act_(F.conv2d(x, w_cl, None, 1, dil, dil), bias, r, True))
    fl = 2.0 * B * H * W * Cout * 9 * Cin
    print('fp32 B %d  %4d -> %4d  %dx%d dil %d res %d | err vs float64 %.2e of scale (MIOpen %.2e) | own %.3f ms %.1f TF (%.3f of 157.3) | '
          'MIOpen conv only %.3f ms %.1f TF, + epilogue pass %.3f ms' % (B, Cin, Cout, H, W, dil, int(res), err / scale, err_mi / scale,
                                                                       ms_own, fl / ms_own / 1e9, fl / ms_own / 1e9 / 157.3,
                                                                       ms_ref, fl / ms_ref / 1e9, ms_full))
    return err / scale


def run32_1x1(B, Cin, Cout, H, W, reps=3):
    x = torch.randn((B, Cin, H, W), device='cuda').contiguous(memory_format=torch.channels_last)
    w = torch.randn((Cout, Cin, 1, 1), device='cuda') * (2.0 / Cin) ** 0.5
    bias = torch.randn((Cout,), device='cuda')
    wt = w.reshape(Cout, 1, Cin).contiguous()
    y = eng.conv3x3_f32(x, wt, bias, None, False, 1)
    nb = min(B, 2)
    ref = F.conv2d(x[:nb].double(), w.double(), bias.double())
    err = (y[:nb].double() - ref).abs().max().item() / ref.abs().max().item()
    def t(fn):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    ms_own = t(lambda: eng.conv3x3_f32(x, wt, bias, None, False, 1))
    w_cl = w.contiguous(memory_format=torch.channels_last)
    ms_full = t(lambda: eng.bias_act_(F.conv2d(x, w_cl), bias, None, False))
    fl = 2.0 * B * H * W * Cout * Cin
    print('fp32 1x1 B %d %4d -> %4d %dx%d | err vs float64 %.2e | own %.3f ms %.1f TF | MIOpen + epilogue pass %.3f ms'
          % (B, Cin, Cout, H, W, err, ms_own, fl / ms_own / 1e9, ms_full))


def run_wino(B, Cin, Cout, H, W, dil, res, reps=3, tile=2):
    x = torch.relu(torch.randn((B, Cin, H, W), device='cuda')).contiguous(memory_format=torch.channels_last)
    w = torch.randn((Cout, Cin, 3, 3), device='cuda') * (2.0 / (9 * Cin)) ** 0.5
    bias = torch.randn((Cout,), device='cuda')
    r = torch.randn((B, Cout, H, W), device='cuda').contiguous(memory_format=torch.channels_last) if res else None
    wt = w.permute(0, 2, 3, 1).reshape(Cout, 9, Cin).contiguous()
    u = eng.winograd_weights(w, tile)
    y = eng.conv3x3_wino_f32(x, u, bias, r, True, dil)
    nb = min(B, 2)
    ref = F.conv2d(x[:nb].double(), w.double(), bias.double(), 1, dil, dil)
    if res:
        ref = ref + r[:nb].double()
    ref = torch.relu(ref)
    scale = ref.abs().max().item()
    err = (y[:nb].double() - ref).abs().max().item() / scale
    yd = eng.conv3x3_f32(x, wt, bias, r, True, dil)
    err_d = (yd[:nb].double() - ref).abs().max().item() / scale

    def t(fn):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    ms_w = t(lambda: eng.conv3x3_wino_f32(x, u, bias, r, True, dil))
    ms_d = t(lambda: eng.conv3x3_f32(x, wt, bias, r, True, dil))
    fl = 2.0 * B * H * W * Cout * 9 * Cin
    print('winograd F(%dx%d,3x3) B %d %4d -> %4d %dx%d dil %d res %d | err vs float64: winograd %.2e, direct %.2e of scale | winograd %.3f ms '
          '(%.0f effective TF) | direct %.3f ms (%.0f TF)' % (tile, tile, B, Cin, Cout, H, W, dil, int(res), err, err_d, ms_w, fl / ms_w / 1e9,
                                                              ms_d, fl / ms_d / 1e9))


def run(B, Cin, Cout, H, W, dil, res, reps=5):
    x = torch.randn((B, Cin, H, W), device='cuda').to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn((Cout, Cin, 3, 3), device='cuda') * (2.0 / (9 * Cin)) ** 0.5).to(torch.bfloat16)
    w_cl = w.contiguous(memory_format=torch.channels_last)
    bias = torch.randn((Cout,), device='cuda')
    r = torch.randn((B, Cout, H, W), device='cuda').to(torch.bfloat16).contiguous(memory_format=torch.channels_last) if res else None
    wt = w.permute(0, 2, 3, 1).reshape(Cout, 9, Cin).contiguous()
    y = eng.conv3x3_bf16(x, wt, bias, r, True, dil)
    # numerics on a slice of the batch (float32 convolution of the same bf16 values)
    nb = min(B, 2)
    ref = F.conv2d(x[:nb].float(), w.float(), bias, 1, dil, dil)
    if res:
        ref = ref + r[:nb].float()
    ref = torch.relu(ref)
    err = (y[:nb].float() - ref).abs().max().item()
    scale = ref.abs().max().item()
    # timing
    def t(fn):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    ms_own = t(lambda: eng.conv3x3_bf16(x, wt, bias, r, True, dil))
    ms_ref = t(lambda: F.conv2d(x, w_cl, None, 1, dil, dil))
    fl = 2.0 * B * H * W * Cout * 9 * Cin
    print('B %d  %4d -> %4d  %dx%d dil %d res %d | max err %.3g (scale %.3g, rel %.2e) | own %.3f ms %.0f TF | MIOpen conv only '
          '%.3f ms %.0f TF' % (B, Cin, Cout, H, W, dil, int(res), err, scale, err / scale, ms_own, fl / ms_own / 1e9,
                               ms_ref, fl / ms_ref / 1e9))
    return err / scale


if a.dtype == 'wino':
    run_wino(2, 64, 64, 16, 40, 1, False, tile=a.tile)
    run_wino(1, 128, 256, 24, 300, 2, True, tile=a.tile)
    run_wino(2, 32, 64, 21, 301, 3, True, tile=a.tile)           # odd sizes: partial tiles on every sub-grid
    run_wino(1, 64, 128, 7, 9, 4, False, tile=a.tile)
    if not a.quick:
        B = a.batch
        run_wino(B, 64, 64, 256, 512, 1, True, tile=a.tile)
        for Cin, Cout, dil, res in [(128, 128, 1, True), (128, 256, 2, False), (256, 256, 2, True), (256, 512, 4, False),
                                    (512, 512, 4, True), (512, 512, 2, False), (512, 512, 1, False)]:
            run_wino(B, Cin, Cout, 128, 256, dil, res, tile=a.tile)
elif a.dtype == 'fp32':
    run32(2, 64, 256, 16, 40, 1, False)
    run32(1, 128, 256, 24, 300, 2, True)
    run32(2, 32, 64, 20, 300, 3, True)
    run32_1x1(2, 64, 128, 20, 300)
    if not a.quick:
        run32_1x1(a.batch, 128, 256, 128, 256)
        run32_1x1(a.batch, 256, 512, 128, 256)
    if not a.quick:
        B = a.batch
        for Cin, Cout, dil, res, H, W in [(64, 64, 1, True, 256, 512), (128, 128, 1, True, 128, 256), (128, 256, 2, False, 128, 256),
                                          (256, 256, 2, True, 128, 256), (256, 512, 4, False, 128, 256), (512, 512, 4, True, 128, 256),
                                          (512, 512, 2, False, 128, 256), (512, 512, 1, False, 128, 256)]:
            run32(B, Cin, Cout, H, W, dil, res)
elif a.quick:
    run(2, 64, 256, 16, 40, 1, False)
    run(1, 128, 256, 24, 300, 2, True)
else:
    B = a.batch
    run(2, 64, 256, 16, 40, 1, False)
    run(1, 128, 256, 24, 300, 2, True)
    for Cin, Cout, dil, res in [(128, 256, 2, False), (256, 256, 2, True), (256, 512, 4, False), (512, 512, 4, True),
                                (512, 512, 2, False), (512, 512, 1, False)]:
        run(B, Cin, Cout, 128, 256, dil, res)
    # the narrow channel tiles (64 / 128 output channels): HBM-bound layers
    run(2, 64, 64, 20, 300, 1, True)
    run(2, 128, 128, 9, 70, 1, False)
    run(B, 64, 64, 256, 512, 1, True)
    run(B, 128, 128, 128, 256, 1, True)
