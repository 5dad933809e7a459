"""GPU tests of libspalign's own DRN convolutions (models/drn.py:230-285 layers): the float32-MFMA implicit-GEMM
kernel in its 3x3 and 1x1 forms and the Winograd F(2x2,3x3) path, each against a float64 convolution of the same
operands (torch, the floating-point reference of this tier) — the tolerance is the summation-order rounding of a
float32 convolution, far inside the 1e-4 of the feature contract — and the bf16 kernel against a float32 convolution
of the same bf16 values."""
import importlib
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip('torch')
F = torch.nn.functional


@pytest.fixture(scope='module')
def eng():
    engine = importlib.import_module('superpixel-align_amd.engine')
    e = engine.Engine()
    yield e
    e.close()


def _operands(B, Cin, Cout, H, W, k, res, seed):
    g = torch.Generator(device='cuda').manual_seed(seed)
    x = torch.relu(torch.randn((B, Cin, H, W), device='cuda', generator=g)).contiguous(memory_format=torch.channels_last)
    w = torch.randn((Cout, Cin, k, k), device='cuda', generator=g) * (2.0 / (k * k * Cin)) ** 0.5
    bias = torch.randn((Cout,), device='cuda', generator=g)
    r = torch.randn((B, Cout, H, W), device='cuda', generator=g).contiguous(memory_format=torch.channels_last) if res else None
    return x, w, bias, r


def _ref64(x, w, bias, r, relu, dil):
    y = F.conv2d(x.double(), w.double(), bias.double(), 1, dil * (w.shape[2] // 2), dil)
    if r is not None:
        y = y + r.double()
    return torch.relu(y) if relu else y


# (B, Cin, Cout, H, W, dilation, residual, relu): every channel tile (64 / 128 / 256), both pixel tiles, partial
# tiles in x, rows whose dilated taps leave the image, the three dilations of the network
CASES = [(2, 32, 64, 20, 300, 1, True, True), (1, 64, 64, 17, 130, 1, False, True), (2, 64, 128, 9, 70, 3, True, False),
         (1, 128, 256, 24, 300, 2, True, True), (1, 256, 512, 8, 260, 4, False, True), (3, 96, 192, 5, 33, 1, False, False)]


@pytest.mark.parametrize('B,Cin,Cout,H,W,dil,res,relu', CASES)
def test_conv3x3_f32_matches_float64(eng, B, Cin, Cout, H, W, dil, res, relu):
    x, w, bias, r = _operands(B, Cin, Cout, H, W, 3, res, 1)
    wt = w.permute(0, 2, 3, 1).reshape(Cout, 9, Cin).contiguous()
    y = eng.conv3x3_f32(x, wt, bias, r, relu, dil)
    ref = _ref64(x, w, bias, r, relu, dil)
    assert y.is_contiguous(memory_format=torch.channels_last)
    scale = float(ref.abs().max())
    assert float((y.double() - ref).abs().max()) <= 4e-6 * scale


@pytest.mark.parametrize('B,Cin,Cout,H,W', [(2, 64, 128, 20, 300), (1, 128, 256, 13, 257), (2, 256, 512, 6, 40)])
def test_conv1x1_f32_matches_float64(eng, B, Cin, Cout, H, W):
    """the 1x1 projection of a BasicBlock (models/drn.py:195-203): the GEMM form of the same kernel"""
    x, w, bias, _ = _operands(B, Cin, Cout, H, W, 1, False, 2)
    y = eng.conv3x3_f32(x, w.reshape(Cout, 1, Cin).contiguous(), bias, None, False, 1)
    ref = _ref64(x, w, bias, None, False, 1)
    assert float((y.double() - ref).abs().max()) <= 4e-6 * float(ref.abs().max())


@pytest.mark.parametrize('tile', [2, 4])
@pytest.mark.parametrize('B,Cin,Cout,H,W,dil,res,relu', CASES + [(1, 64, 128, 7, 9, 4, False, True), (2, 32, 64, 21, 301, 3, True, True)])
def test_conv3x3_winograd_f32_matches_float64(eng, B, Cin, Cout, H, W, dil, res, relu, tile):
    """Winograd F(2x2,3x3) and F(4x4,3x3) on the sub-grids of the dilation, odd sizes (partial tiles on every
    sub-grid) included.  F(2x2) is as close to a float64 convolution as the direct float32 kernel; a single F(4x4)
    layer is ~3x further (1e-5 of scale allowed) — through the whole network that does not show
    (test_winograd_layers_inside_the_network)."""
    x, w, bias, r = _operands(B, Cin, Cout, H, W, 3, res, 3)
    u = eng.winograd_weights(w, tile)
    assert tuple(u.shape) == ((tile + 2) ** 2, Cout, Cin)
    y = eng.conv3x3_wino_f32(x, u, bias, r, relu, dil)
    ref = _ref64(x, w, bias, r, relu, dil)
    scale = float(ref.abs().max())
    assert float((y.double() - ref).abs().max()) <= (4e-6 if tile == 2 else 1e-5) * scale
    # and it is the direct kernel's result to rounding
    yd = eng.conv3x3_f32(x, w.permute(0, 2, 3, 1).reshape(Cout, 9, Cin).contiguous(), bias, r, relu, dil)
    assert float((y - yd).abs().max()) <= (6e-6 if tile == 2 else 1.2e-5) * scale


SPLIT_CASES = [(1, 128, 256, 24, 300, 2, True, True), (1, 256, 512, 8, 260, 4, False, True), (2, 128, 128, 21, 301, 3, True, False),
               (1, 512, 512, 9, 70, 1, True, True), (3, 160, 384, 5, 33, 1, False, False)]


@pytest.mark.parametrize('B,Cin,Cout,H,W,dil,res,relu', SPLIT_CASES)
def test_conv3x3_winograd_split_planes_matches_float64(eng, B, Cin, Cout, H, W, dil, res, relu):
    """F(4x4,3x3) with the 36 GEMMs on the 16-bit matrix cores: every float32 operand as two half-precision planes,
    three products (spa_conv3x3_wino4_f16s).  Same tolerance as the float32-operand form, and the tracked maxima are
    the tensors' own."""
    x, w, bias, r = _operands(B, Cin, Cout, H, W, 3, res, 5)
    x = x * 3.7
    u2, cs = eng.winograd_weights_split(w)
    am = eng.amax(x)
    assert float(am.view(torch.float32)) == float(x.abs().max())
    y, am_out = eng.conv3x3_wino_f16s(x, u2, cs, bias, r, relu, dil, amax_in=am)
    ref = _ref64(x, w, bias, r, relu, dil)
    scale = float(ref.abs().max())
    assert float((y.double() - ref).abs().max()) <= 1e-5 * scale
    assert float(am_out.view(torch.float32)) == float(y.abs().max())
    # against the float32-operand form of the same algorithm: rounding level
    y32 = eng.conv3x3_wino_f32(x, eng.winograd_weights(w, 4), bias, r, relu, dil)
    assert float((y - y32).abs().max()) <= 1.5e-5 * scale


def test_split_planes_survive_extreme_ranges(eng):
    """the scale comes from the largest magnitude: a tensor of 1e4-sized activations next to 1e-6-sized ones, an all-zero
    tensor and a bound far above the actual maximum all stay finite and as accurate as float32 allows"""
    x, w, bias, r = _operands(1, 128, 128, 16, 40, 3, False, 6)
    u2, cs = eng.winograd_weights_split(w)
    xs = x.clone()
    xs[:, :, :8] *= 1e4
    xs[:, :, 8:] *= 1e-6
    y, _ = eng.conv3x3_wino_f16s(xs, u2, cs, bias, None, False, 1)
    ref = _ref64(xs, w, bias, None, False, 1)
    assert torch.isfinite(y).all()
    assert float((y.double() - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    # the small half of the image alone is resolved relative to ITS scale when it is its own tensor
    y2, _ = eng.conv3x3_wino_f16s((x * 1e-6).contiguous(memory_format=torch.channels_last), u2, cs, torch.zeros_like(bias), None, False, 1)
    ref2 = _ref64(x * 1e-6, w, torch.zeros_like(bias), None, False, 1)
    assert float((y2.double() - ref2).abs().max()) <= 1e-5 * float(ref2.abs().max())
    z = torch.zeros_like(x)
    y0, am0 = eng.conv3x3_wino_f16s(z, u2, cs, bias, None, True, 2)
    assert torch.equal(y0, torch.relu(bias).view(1, -1, 1, 1).expand_as(y0).contiguous(memory_format=torch.channels_last))
    # a loose bound (what a producer that only knows an upper bound would hand over): 2^10 x the maximum
    loose = (x.abs().max() * 1024.0).reshape(1).view(torch.int32)
    y3, _ = eng.conv3x3_wino_f16s(x, u2, cs, bias, None, False, 1, amax_in=loose)
    ref3 = _ref64(x, w, bias, None, False, 1)
    assert float((y3.double() - ref3).abs().max()) <= 1e-5 * float(ref3.abs().max())


@pytest.mark.parametrize('B,Cin,Cout,H,W,dil,res,relu,taps', [(2, 64, 64, 20, 300, 1, True, True, 9), (1, 64, 128, 17, 130, 2, False, True, 9),
                                                                (1, 128, 256, 13, 257, 1, False, False, 1), (2, 256, 512, 6, 40, 1, False, False, 1),
                                                                (1, 32, 64, 9, 600, 4, True, False, 9)])
def test_conv3x3_split_planes_direct_matches_float64(eng, B, Cin, Cout, H, W, dil, res, relu, taps):
    """the direct kernel on the 16-bit matrix cores (64-channel layers, 1x1 projections): spa_conv3x3_f16s / spa_conv1x1_f16s"""
    k = 3 if taps == 9 else 1
    x, w, bias, r = _operands(B, Cin, Cout, H, W, k, res, 7)
    wt = w.permute(0, 2, 3, 1).reshape(Cout, taps, Cin).contiguous()
    wt2, inv_t = eng.split_planes(wt)
    y, am = eng.conv3x3_f16s(x, wt2, inv_t, bias, r, relu, dil)
    ref = _ref64(x, w, bias, r, relu, dil)
    scale = float(ref.abs().max())
    assert float((y.double() - ref).abs().max()) <= 4e-6 * scale
    assert float(am.view(torch.float32)) == float(y.abs().max())


@pytest.mark.parametrize('B,C,H,W,dil,res,relu', [(3, 64, 33, 512, 1, True, True), (2, 64, 40, 300, 4, True, False), (1, 64, 9, 37, 2, False, True),
                                                    (2, 64, 7, 1024, 1, False, True), (3, 128, 21, 256, 1, True, True), (2, 128, 50, 130, 2, False, True),
                                                    (2, 128, 64, 257, 3, True, True), (2, 128, 33, 100, 1, True, False), (1, 128, 3, 5, 1, False, True),
                                                    (30, 64, 64, 512, 1, True, True)])
def test_planes_in_lds_kernel_is_bit_identical_with_the_kernel_it_replaces(eng, B, C, H, W, dil, res, relu):
    """round 6: k_conv3x3_p16 (spa_convp.hip: planes built once per staged segment, in place in LDS, one barrier per group of three
    taps) feeds every accumulator the same matrix instructions in the same order as k_conv3x3_f32<SPLIT>: same bits, same tracked
    maximum, on partial tiles, image borders, every dilation — and the same bits again on every repetition (its waits are counted
    by hand: a race would show as a run that differs)"""
    x, w, bias, r = _operands(B, C, C, H, W, 3, res, 11)
    wt2, inv_t = eng.split_planes(w.permute(0, 2, 3, 1).reshape(C, 9, C).contiguous())
    am = eng.amax(x)
    try:
        eng.debug_set(1, 0)
        y0, a0 = eng.conv3x3_f16s(x, wt2, inv_t, bias, r, relu, dil, amax_in=am)
        eng.debug_set(1, 1)
        for rep in range(6 if B < 30 else 3):
            y1, a1 = eng.conv3x3_f16s(x, wt2, inv_t, bias, r, relu, dil, amax_in=am)
            assert torch.equal(y0, y1), 'repetition %d' % rep
            assert int(a0) == int(a1)
    finally:
        eng.debug_set(1, 1)
    ref = _ref64(x[:1], w, bias, None if r is None else r[:1], relu, dil)
    assert float((y1[:1].double() - ref).abs().max()) <= 4e-6 * float(ref.abs().max())
    assert eng.status() == 0


@pytest.mark.parametrize('B,Cin,Cout,Hi,Wi', [(2, 32, 64, 37, 301), (1, 64, 128, 20, 270), (1, 64, 64, 9, 130), (2, 32, 128, 16, 256),
                                              # maps narrower than one pixel tile (224 x 224 inputs: 112 -> 56 -> 28)
                                              (3, 32, 64, 112, 112), (3, 64, 128, 56, 56), (2, 64, 128, 29, 27)])
def test_conv3x3_stride2_with_projection_matches_float64(eng, B, Cin, Cout, Hi, Wi):
    """the stride-2 opening convolution of layers 3/4 and the block's 1x1 stride-2 projection in one pass
    (spa_conv3x3_s2_f16s): both outputs against float64 convolutions, odd input sizes included"""
    g = torch.Generator(device='cuda').manual_seed(9)
    x = (torch.relu(torch.randn((B, Cin, Hi, Wi), device='cuda', generator=g)) * 2.3).contiguous(memory_format=torch.channels_last)
    w = torch.randn((Cout, Cin, 3, 3), device='cuda', generator=g) * (2.0 / (9 * Cin)) ** 0.5
    wd = torch.randn((Cout, Cin, 1, 1), device='cuda', generator=g) * (2.0 / Cin) ** 0.5
    b = torch.randn((2 * Cout,), device='cuda', generator=g)
    wc = torch.zeros((2 * Cout, 9, Cin), device='cuda')
    wc[:Cout] = w.permute(0, 2, 3, 1).reshape(Cout, 9, Cin)
    wc[Cout:, 4] = wd.reshape(Cout, Cin)
    wt2, inv_t = eng.split_planes(wc)
    y, y2, am = eng.conv3x3_s2_f16s(x, wt2, inv_t, b, Cout, True)
    r1 = torch.relu(F.conv2d(x.double(), w.double(), b[:Cout].double(), 2, 1))
    r2 = F.conv2d(x.double(), wd.double(), b[Cout:].double(), 2, 0)
    assert y.shape == r1.shape and y2.shape == r2.shape
    assert float((y.double() - r1).abs().max()) <= 4e-6 * float(r1.abs().max())
    assert float((y2.double() - r2).abs().max()) <= 4e-6 * float(r2.abs().max())
    assert float(am.view(torch.float32)) == float(y.abs().max())
    # without the projection: csplit = Cout
    if Cout % 128 == 0:
        wt1, inv1 = eng.split_planes(wc[:Cout].contiguous())
        y1, none, _ = eng.conv3x3_s2_f16s(x, wt1, inv1, b[:Cout].contiguous(), Cout, False)
        assert none is None
        r = F.conv2d(x.double(), w.double(), b[:Cout].double(), 2, 1)
        assert float((y1.double() - r).abs().max()) <= 4e-6 * float(r.abs().max())


@pytest.mark.parametrize('B,Cin,Cout,Hi,Wi', [(2, 32, 64, 37, 301), (1, 64, 128, 20, 270), (3, 32, 64, 112, 112), (2, 64, 128, 29, 27)])
def test_conv3x3_stride2_with_projection_float32_instructions(eng, B, Cin, Cout, Hi, Wi):
    """round 6: the strict float32 network's opener + projection (spa_conv3x3_s2_f32, float32 matrix instructions): both outputs
    against float64 convolutions, odd input sizes and maps narrower than a pixel tile included"""
    g = torch.Generator(device='cuda').manual_seed(19)
    x = (torch.relu(torch.randn((B, Cin, Hi, Wi), device='cuda', generator=g)) * 2.3).contiguous(memory_format=torch.channels_last)
    w = torch.randn((Cout, Cin, 3, 3), device='cuda', generator=g) * (2.0 / (9 * Cin)) ** 0.5
    wp = torch.randn((Cout, Cin, 1, 1), device='cuda', generator=g) * (2.0 / Cin) ** 0.5
    b = torch.randn((2 * Cout,), device='cuda', generator=g)
    wt = torch.zeros((2 * Cout, 9, Cin), device='cuda')
    wt[:Cout] = w.permute(0, 2, 3, 1).reshape(Cout, 9, Cin)
    wt[Cout:, 4] = wp.reshape(Cout, Cin)
    y, y2 = eng.conv3x3_s2_f32(x, wt.contiguous(), b, Cout, True)
    ref = torch.relu(F.conv2d(x.double(), w.double(), b[:Cout].double(), 2, 1))
    ref2 = F.conv2d(x.double(), wp.double(), b[Cout:].double(), 2, 0)
    assert y.shape == ref.shape and y2.shape == ref2.shape
    assert float((y.double() - ref).abs().max()) <= 3e-6 * float(ref.abs().max())
    assert float((y2.double() - ref2).abs().max()) <= 3e-6 * float(ref2.abs().max())
    assert eng.status() == 0


@pytest.mark.parametrize('B,H,W', [(2, 37, 301), (1, 224, 224), (3, 8, 9)])
def test_drn_layer2_float32(eng, B, H, W):
    """round 6: layer 2 of DRN-D in plain float32 (spa_drn_layer2_f32, the strict float32 network) against a float64 convolution"""
    g = torch.Generator(device='cuda').manual_seed(23)
    x = torch.relu(torch.randn((B, 16, H, W), device='cuda', generator=g)).contiguous(memory_format=torch.channels_last)
    w = torch.randn((32, 16, 3, 3), device='cuda', generator=g) * (2.0 / 144) ** 0.5
    b = torch.randn((32,), device='cuda', generator=g)
    y = eng.drn_layer2_f32(x, w.permute(2, 3, 1, 0).reshape(9, 16, 32).contiguous(), b)
    ref = torch.relu(F.conv2d(x.double(), w.double(), b.double(), 2, 1))
    assert y.shape == ref.shape
    assert float((y.double() - ref).abs().max()) <= 2e-6 * float(ref.abs().max())


@pytest.mark.parametrize('B,H,W', [(2, 37, 61), (1, 64, 130), (1, 9, 257), (3, 8, 32)])
def test_drn_layer2_split_planes_matches_float64(eng, B, H, W):
    """layer 2 of DRN-D (conv3x3 16 -> 32, stride 2, padding 1, ReLU) on the 16-bit matrix cores: spa_drn_layer2_f16s against
    a float64 convolution, odd sizes and partial tiles included; the tracked maximum is the output's"""
    g = torch.Generator(device='cuda').manual_seed(21)
    x = (torch.relu(torch.randn((B, 16, H, W), device='cuda', generator=g)) * 1.7).contiguous(memory_format=torch.channels_last)
    w = torch.randn((32, 16, 3, 3), device='cuda', generator=g) * (2.0 / 144) ** 0.5
    b = torch.randn((32,), device='cuda', generator=g)
    wp, inv_t = eng.layer2_planes(w)
    y = eng.drn_layer2_f16s(x, wp, inv_t, b)
    ref = torch.relu(F.conv2d(x.double(), w.double(), b.double(), 2, 1))
    assert y.shape == ref.shape
    assert float((y.double() - ref).abs().max()) <= 3e-6 * float(ref.abs().max())
    assert float(y._spa_amax.view(torch.float32)) == float(y.abs().max())


def test_bias_act_tracks_the_maximum_it_stores(eng):
    """spa_bias_act_amax: the epilogue pass behind a MIOpen convolution also hands the next (split-plane) convolution its scale"""
    g = torch.Generator(device='cuda').manual_seed(13)
    for relu, res in ((True, False), (False, True)):
        y = torch.randn((2, 32, 37, 61), device='cuda', generator=g).contiguous(memory_format=torch.channels_last)
        r = torch.randn_like(y) if res else None
        b = torch.randn((32,), device='cuda', generator=g)
        ref = y + b.view(1, -1, 1, 1) + (r if res else 0)
        ref = torch.relu(ref) if relu else ref
        out = eng.bias_act_(y, b, r, relu, track_amax=True)
        assert torch.equal(out, ref)
        assert float(out._spa_amax.view(torch.float32)) == float(ref.abs().max())


def test_winograd_layers_inside_the_network(eng):
    """DRN-D-22 float32 with and without the Winograd / own-convolution paths: the map the pipeline pools (index 7)
    agrees to 2e-5 of its scale (north star: 1e-4), and the Winograd path really ran."""
    drn = importlib.import_module('superpixel-align_amd.drn')
    synth = importlib.import_module('superpixel-align_amd.synth')
    m = drn.create_drn('drn_d_22', device='cuda', dtype=torch.float32)
    x = synth.synth_batch([3, 4], 256, 512)
    E = drn._EPILOGUE
    saved = (E['winograd'], E['own_conv32'])
    split_saved = E['split_gemm']
    try:
        E['own_conv32'] = True
        E['winograd'], E['wino_launches'] = 4, 0
        E['split_gemm'], E['gemm16_launches'], E['gemm16n_launches'], E['conv16_launches'] = True, 0, 0, 0
        _, a4s = m.batch_predict(x, need=[7])
        assert E['wino_launches'] == 13 and E['gemm16_launches'] + E['gemm16n_launches'] == 13 and E['conv16_launches'] >= 3
        # run to run the same bits (every convolution of the float32 forward is libspalign's at every shape since round 5; before,
        # the stride-2 openers and the 1x1 projections of narrow maps were MIOpen's, whose results differ in the last bit from run
        # to run): 64 x 2048 pixels fill every kernel's pixel tiles
        xw = synth.synth_batch([5, 6], 64, 2048)
        _, ws = m.batch_predict(xw, need=[7])
        _, ws2 = m.batch_predict(xw, need=[7])
        assert torch.equal(ws[7], ws2[7])
        E['split_gemm'], E['wino_launches'] = False, 0
        _, a4 = m.batch_predict(x, need=[7])
        assert E['wino_launches'] == 13                   # layers 4-8: both channel counts >= 128
        E['winograd'], E['wino_launches'] = 2, 0
        _, a2 = m.batch_predict(x, need=[7])
        assert E['wino_launches'] == 9                    # layers 5-8: both channel counts >= 256
        E['winograd'] = 0
        _, b = m.batch_predict(x, need=[7])
        E['own_conv32'] = False
        _, c = m.batch_predict(x, need=[7])
    finally:
        E['winograd'], E['own_conv32'] = saved
        E['split_gemm'] = split_saved
    # float64 network on the CPU: the yardstick for all four
    m64 = drn.create_drn('drn_d_22', device='cpu', dtype=torch.float64)
    _, r = m64.batch_predict(x, need=[7])
    r = r[7].cuda()
    scale = float(r.abs().max())
    err = {k: float((v[7].double() - r).abs().max()) / scale for k, v in (('F(4x4,3x3) two half-precision planes', a4s), ('F(4x4,3x3)', a4), ('F(2x2,3x3)', a2), ('direct', b), ('MIOpen', c))}
    print('map 7 vs the float64 network, of scale:', err)
    assert max(err.values()) <= 1e-5                      # north star: 1e-4
    assert err['F(4x4,3x3)'] <= 2.0 * err['MIOpen'] + 1e-6
    assert err['F(4x4,3x3) two half-precision planes'] <= 2.0 * err['MIOpen'] + 1e-6


@pytest.mark.parametrize('B,Cin,Cout,H,W,dil,res', [(2, 64, 256, 16, 40, 1, False), (1, 128, 256, 24, 300, 2, True), (2, 64, 64, 20, 300, 1, True)])
def test_conv3x3_bf16_matches_float32_of_the_same_values(eng, B, Cin, Cout, H, W, dil, res):
    x, w, bias, r = _operands(B, Cin, Cout, H, W, 3, res, 4)
    xb, wb = x.to(torch.bfloat16).contiguous(memory_format=torch.channels_last), w.to(torch.bfloat16)
    rb = r.to(torch.bfloat16).contiguous(memory_format=torch.channels_last) if res else None
    y = eng.conv3x3_bf16(xb, wb.permute(0, 2, 3, 1).reshape(Cout, 9, Cin).contiguous(), bias, rb, True, dil)
    ref = F.conv2d(xb.float(), wb.float(), bias, 1, dil, dil)
    if res:
        ref = ref + rb.float()
    ref = torch.relu(ref)
    assert float((y.float() - ref).abs().max()) <= 1e-2 * float(ref.abs().max())      # one rounding to bf16 at the end


@pytest.mark.parametrize('Cin,Cout,taps,stride,dil,res,relu,B,H,W', [
    (16, 32, 9, 2, 1, False, True, 2, 37, 61),         # layer 2 of arch D (odd sizes: the last input row / column is padding)
    (32, 64, 9, 2, 1, False, True, 1, 64, 130),        # opener of layer 3 ...
    (32, 64, 1, 2, 1, False, False, 1, 64, 130),       # ... and its 1x1 stride-2 projection (no ReLU)
    (64, 128, 9, 2, 1, False, True, 2, 19, 67),        # opener of layer 4, ragged strips
    (64, 128, 1, 2, 1, False, False, 2, 19, 67),
    (128, 256, 1, 1, 1, False, False, 1, 9, 257),      # projections of layers 5 / 6
    (256, 512, 1, 1, 1, False, False, 2, 5, 33),
    (16, 16, 9, 1, 1, True, True, 2, 37, 61),          # arch C: layer 1's BasicBlock, layer 2's (stride 2, projection, 32 -> 32)
    (16, 32, 1, 2, 1, False, False, 1, 8, 32),
    (32, 32, 9, 1, 1, True, True, 2, 19, 67),
    (64, 64, 9, 1, 2, True, True, 1, 11, 70),          # a dilated stride-1 layer the heavy kernel would take: same answer here
])
def test_light_bf16_convolutions_match_float32_of_the_same_values(eng, Cin, Cout, taps, stride, dil, res, relu, B, H, W):
    """spa_conv_bf16_light (csrc/spa_convl.hip: the stride-2 / thin / 1x1 layers of the bf16 network, models/drn.py:134-151,
    195-203) against torch's float32 convolution of the SAME bf16 operands: the only differences allowed are the float32
    accumulation order and the final rounding to bf16 (one bf16 ulp of the largest output, 2^-8)."""
    g = torch.Generator(device='cuda').manual_seed(Cin * 7 + Cout + H + taps)
    k = 3 if taps == 9 else 1
    x = torch.randn((B, Cin, H, W), device='cuda', generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn((Cout, Cin, k, k), device='cuda', generator=g) * (2.0 / (taps * Cin)) ** 0.5).to(torch.bfloat16)
    bias = torch.randn((Cout,), device='cuda', generator=g)
    Ho, Wo = (H + stride - 1) // stride, (W + stride - 1) // stride
    r = torch.randn((B, Cout, Ho, Wo), device='cuda', generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last) if res else None
    y = eng.conv_bf16_light(x, w.permute(0, 2, 3, 1).reshape(Cout, taps, Cin).contiguous(), bias, r, relu, stride, dil)
    ref = F.conv2d(x.float(), w.float(), bias, stride, dil if taps == 9 else 0, dil)
    assert y.dtype == torch.bfloat16 and y.shape == ref.shape
    if res:
        ref = ref + r.float()
    if relu:
        ref = torch.relu(ref)
    assert float((y.float() - ref).abs().max()) <= float(ref.abs().max()) * 2.0 ** -8
    eng.raise_on_status()


def test_light_bf16_convolution_refuses_shapes_it_does_not_take(eng):
    """spa_conv_bf16_light fails loudly (no fallback): 48 input channels, a 3x3 layer with 128 input channels (spa_conv3x3_bf16's),
    24 output channels"""
    eng_mod = importlib.import_module('superpixel-align_amd.engine')
    for Cin, Cout, taps in ((48, 64, 9), (128, 128, 9), (16, 24, 9)):
        k = 3 if taps == 9 else 1
        x = torch.zeros((1, Cin, 8, 16), device='cuda', dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
        w = torch.zeros((Cout, taps, Cin), device='cuda', dtype=torch.bfloat16)
        with pytest.raises(eng_mod.SpalignError):
            eng.conv_bf16_light(x, w, torch.zeros((Cout,), device='cuda'), None, True, 1, 1)


@pytest.mark.parametrize('arch', ['drn_d_22', 'drn_c_26'])
def test_bf16_network_runs_on_own_kernels(eng, arch):
    """BASELINE configs[4]: the bf16 network without a library convolution (the light layers on spa_conv_bf16_light, the
    rest on spa_conv3x3_bf16 and the stem kernel), and as close to the float32 network as with the library's kernels."""
    drn = importlib.import_module('superpixel-align_amd.drn')
    synth = importlib.import_module('superpixel-align_amd.synth')
    m32 = drn.create_drn(arch, device='cuda', dtype=torch.float32, seed=3)
    m = drn.create_drn(arch, device='cuda', dtype=torch.bfloat16, seed=3)
    x = synth.synth_batch([3, 4], 128, 320)
    E = drn._EPILOGUE
    _, ref = m32.batch_predict(x, need=[7])
    E['library_convs'] = 0
    _, own = m.batch_predict(x, need=[7])
    assert own[7].dtype == torch.bfloat16
    assert E['library_convs'] == 0                     # (arch C: conv1 + layer1's first convolution are the fused bf16 stem)
    saved = E['own_conv']
    try:
        E['own_conv'] = False
        _, lib = m.batch_predict(x, need=[7])
    finally:
        E['own_conv'] = saved
    scale = float(ref[7].abs().max())
    e_own = float((own[7].float() - ref[7]).abs().max()) / scale
    e_lib = float((lib[7].float() - ref[7]).abs().max()) / scale
    print('%s bf16 map 7 vs the float32 network, of scale: own kernels %.3e, library %.3e' % (arch, e_own, e_lib))
    assert e_own <= max(1.5 * e_lib, 2e-2)


@pytest.mark.parametrize('Cin,Cout,stride,proj,res,relu,B,H,W', [(16, 16, 1, False, True, True, 2, 37, 61), (16, 16, 1, False, False, False, 1, 8, 32),
                                                                 (16, 32, 2, True, False, True, 2, 37, 61), (16, 32, 2, False, False, True, 1, 64, 130),
                                                                 (32, 32, 1, False, True, True, 2, 19, 67), (16, 32, 1, False, False, True, 1, 9, 33)])
def test_thin_convolutions_of_drn_c_match_float64(eng, Cin, Cout, stride, proj, res, relu, B, H, W):
    """spa_conv_small_f16s (csrc/spa_convs.hip): the 16- and 32-channel 3x3 convolutions at the top of DRN-C
    (models/drn.py:134-170: layer1's BasicBlock, layer2's stride-2 BasicBlock with its 1x1 projection) on the 16-bit matrix cores
    with two half-precision planes per operand, against float64 convolutions — odd sizes, partial tiles, both outputs of the
    opener + projection form, the tracked maximum."""
    g = torch.Generator(device='cuda').manual_seed(17)
    x = (torch.relu(torch.randn((B, Cin, H, W), device='cuda', generator=g)) * 1.9).contiguous(memory_format=torch.channels_last)
    w = torch.randn((Cout, Cin, 3, 3), device='cuda', generator=g) * (2.0 / (9 * Cin)) ** 0.5
    wd = torch.randn((32, Cin, 1, 1), device='cuda', generator=g) * (2.0 / Cin) ** 0.5 if proj else None
    bias = torch.randn((Cout + (32 if proj else 0),), device='cuda', generator=g)
    Ho, Wo = (H + stride - 1) // stride, (W + stride - 1) // stride
    r = torch.randn((B, Cout, Ho, Wo), device='cuda', generator=g).contiguous(memory_format=torch.channels_last) if res else None
    wp, inv_t = eng.small_planes(w, wd)
    y, y2 = eng.conv_small_f16s(x, wp, inv_t, bias, Cout, stride, 32 if proj else 0, r, relu)
    ref = F.conv2d(x.double(), w.double(), bias[:Cout].double(), stride, 1)
    if res:
        ref = ref + r.double()
    if relu:
        ref = torch.relu(ref)
    assert y.shape == ref.shape and y.is_contiguous(memory_format=torch.channels_last)
    assert float((y.double() - ref).abs().max()) <= 3e-6 * float(ref.abs().max())
    assert float(y._spa_amax.view(torch.float32)) == float(y.abs().max())
    if proj:
        ref2 = F.conv2d(x.double(), wd.double(), bias[Cout:].double(), stride, 0)
        assert y2.shape == ref2.shape
        assert float((y2.double() - ref2).abs().max()) <= 3e-6 * float(ref2.abs().max())
    else:
        assert y2 is None


def test_strict_float32_network_runs_on_own_kernels(eng):
    """round 6 (verdict r5 #7): with float32 matrix instructions (`--fp32_mfma_gemm`, SPA_SPLIT_GEMM=0) DRN-D-22 no longer hands layer 2
    and the stride-2 openers (+ projections) to the library: 0 library convolutions, and the map agrees with the float64 network"""
    drn = importlib.import_module('superpixel-align_amd.drn')
    synth = importlib.import_module('superpixel-align_amd.synth')
    E = drn._EPILOGUE
    saved = E['split_gemm']
    try:
        E['split_gemm'] = False
        model = drn.create_drn('drn_d_22', device='cuda', dtype=torch.float32)
        x = synth.synth_batch([3, 4], 128, 256)
        E['library_convs'] = 0
        _, maps = model.batch_predict(x, need=[2, 3, 7])
        assert E['library_convs'] == 0
    finally:
        E['split_gemm'] = saved
    ref_model = drn.create_drn('drn_d_22', device='cpu', fold_bn=False).double()
    with torch.no_grad():
        ref = ref_model.forward_maps(drn.DRN.normalise(torch.from_numpy(x)).double())
    for i in (2, 3, 7):
        r = ref[i].cuda()
        assert float((maps[i].double() - r).abs().max()) <= 2e-5 * float(r.abs().max()), i


def test_drn_c_front_runs_on_own_kernels(eng):
    """DRN-C-26, the reference's default backbone (batch_spalign_kmeans.py:524-526): with the fused stem that also stores conv1's
    output and the thin convolutions above, NO convolution of the float32 forward is MIOpen's at a size that fills the kernels'
    pixel tiles, maps 0, 1 and 7 agree with the float64 network, and the forward is reproducible bit for bit."""
    drn = importlib.import_module('superpixel-align_amd.drn')
    synth = importlib.import_module('superpixel-align_amd.synth')
    m = drn.create_drn('drn_c_26', device='cuda', dtype=torch.float32)
    assert m._front_c is not None
    x = synth.synth_batch([7, 8], 64, 2048)
    calls = [0]
    orig = drn.F.conv2d

    def counting(*a, **k):
        calls[0] += 1
        return orig(*a, **k)
    drn.F.conv2d = counting
    try:
        _, a = m.batch_predict(x, need=[0, 1, 7])
        _, b = m.batch_predict(x, need=[0, 1, 7])
    finally:
        drn.F.conv2d = orig
    assert calls[0] == 0
    assert all(torch.equal(a[i], b[i]) for i in (0, 1, 7))
    m64 = drn.create_drn('drn_c_26', device='cpu', dtype=torch.float64)
    _, r = m64.batch_predict(x, need=[0, 1, 7])
    for i in (0, 1, 7):
        ref = r[i].cuda()
        err = float((a[i].double() - ref).abs().max()) / float(ref.abs().max())
        assert err <= 1e-5, (i, err)
