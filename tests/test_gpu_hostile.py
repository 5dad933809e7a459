"""Is the split-plane float32 network (two half-precision planes per operand, three products: csrc/spa_gemm16.hip,
spa_conv32.hip, spa_stem.hip) still float32-equivalent on data that is NOT friendly?  (VERDICT r3, weak #2.)

The planes keep 22 significand bits after ONE power-of-two scale per tensor, so an element far below the tensor's
maximum keeps absolute, not relative precision.  Trained, BatchNorm-folded weights have per-channel scales spread over
orders of magnitude and activations have heavy tails; random-init weights and ReLU-of-Gaussian activations have neither.
These tests build that data on purpose and compare the default path with the float64 result AND with the same kernels on
float32 matrix instructions (`SPA_SPLIT_GEMM=0` / bench.py --fp32_mfma_gemm): the split path must be within 2x of the
float32-instruction path's error (+ one float32 rounding of slack) and inside the feature contract (1e-4, north star).

  (a) a DRN-D-22 whose consecutive layers are rescaled per channel by g = 10^U(-1.5, 1.5): BatchNorm gamma and beta times
      g, the next convolution's input-channel weights divided by g — the same function (ReLU is positively homogeneous),
      activations whose channels span 1e3 in magnitude and folded weights that span 1e-3 .. 1e3 to match;
  (b) single layers fed a unit-scale bulk with a few 1e3-sized outliers, error measured on the outputs the outliers do
      NOT reach, relative to the bulk's own scale;
  (c) the hostile model written as a .pth with exactly the key set and shapes of the reference's drn_d_22 module
      (models/drn_pytorch.py:280-284; tests/golden/drn_state_dict_keys.json from oracle/gen_golden_drn_keys.py) and
      read back through DRN.load_pth / create_drn(weights=...)."""
import importlib
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip('torch')
F = torch.nn.functional
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope='module')
def eng():
    engine = importlib.import_module('superpixel-align_amd.engine')
    e = engine.Engine()
    yield e
    e.close()


def _hostile_state_dict(x_norm, seed=0, spread=1.5):
    """state_dict (float32, reference key set incl. fc) of a DRN-D-22 with calibrated BatchNorm statistics and the
    per-channel rescaling of the module docstring; built on the CPU in float64."""
    drn = importlib.import_module('superpixel-align_amd.drn')
    g = torch.Generator().manual_seed(seed)
    m = drn.DRN('drn_d_22', bn_eps=drn.CHAINER_BN_EPS, with_fc=True).double()
    bns = [mod for mod in m.modules() if isinstance(mod, torch.nn.BatchNorm2d)]
    with torch.no_grad():
        for bn in bns:
            bn.weight.copy_(torch.rand(bn.weight.shape, generator=g, dtype=torch.float64) + 0.5)
            bn.bias.copy_(torch.randn(bn.bias.shape, generator=g, dtype=torch.float64) * 0.2)
            bn.momentum = 1.0                       # one training-mode pass: running statistics = this batch's
        m.train()
        m.forward_maps(x_norm.double())
        m.eval()

        def rescale(bn, consumers):
            s = torch.pow(10.0, (torch.rand(bn.weight.shape, generator=g, dtype=torch.float64) * 2 - 1) * spread)
            bn.weight.mul_(s)
            bn.bias.mul_(s)
            for conv in consumers:
                conv.weight.div_(s.view(1, -1, 1, 1))
        rescale(m.layer0[1], [m.layer1[0]])
        rescale(m.layer1[1], [m.layer2[0]])
        rescale(m.layer2[1], [m.layer3[0].conv1, m.layer3[0].downsample[0]])
        for name in ('layer3', 'layer4', 'layer5', 'layer6'):
            for blk in getattr(m, name):
                rescale(blk.bn1, [blk.conv2])
        rescale(m.layer7[1], [m.layer8[0]])
    sd = {k: v.float() for k, v in m.state_dict().items() if not k.endswith('num_batches_tracked')}
    return sd, m


def test_hostile_channel_scales_through_the_whole_network_and_a_reference_pth(eng, tmp_path):
    drn = importlib.import_module('superpixel-align_amd.drn')
    synth = importlib.import_module('superpixel-align_amd.synth')
    x = synth.synth_batch([5, 6], 256, 512)
    x_norm = drn.DRN.normalise(torch.from_numpy(x))
    sd, m64 = _hostile_state_dict(x_norm)
    # the folded weights really are hostile: per-output-channel and per-input-channel scales over >= 3 decades
    w = sd['layer5.1.conv2.weight'].double()
    per_in = w.abs().amax(dim=(0, 2, 3))
    assert float(per_in.max() / per_in.min()) > 300.0
    # (c) the reference's key set, through a file and load_pth
    keys = json.load(open(os.path.join(HERE, 'golden', 'drn_state_dict_keys.json')))['drn_d_22']
    assert {k: list(v.shape) for k, v in sd.items()} == keys
    path = str(tmp_path / 'drn_d_22-hostile.pth')
    torch.save(sd, path)
    E = drn._EPILOGUE
    saved = (E['split_gemm'], E['winograd'], E['own_conv32'])
    try:
        E['winograd'], E['own_conv32'] = 4, True
        E['split_gemm'] = True
        model = drn.create_drn('drn_d_22', weights=path, device='cuda', dtype=torch.float32)
        E['gemm16_launches'] = E['conv16_launches'] = 0
        _, split = model.batch_predict(x, need=[2, 4, 7])
        assert E['gemm16_launches'] > 0 and E['conv16_launches'] > 0       # the planes really ran
        E['split_gemm'] = False
        model32 = drn.create_drn('drn_d_22', weights=path, device='cuda', dtype=torch.float32)
        _, f32 = model32.batch_predict(x, need=[2, 4, 7])
    finally:
        E['split_gemm'], E['winograd'], E['own_conv32'] = saved
    # float64 yardstick: the SAME float32 weights (what the file holds), BatchNorm evaluated unfolded in float64
    ref_model = drn.DRN('drn_d_22', with_fc=True).double()
    ref_model.load_state_dict({k: v.double() for k, v in sd.items()}, strict=False)
    ref_model.eval()
    with torch.no_grad():
        ref = ref_model.forward_maps(x_norm.double())
    report = {}
    for i in (2, 4, 7):
        r = ref[i].cuda()
        scale = float(r.abs().max())
        es = float((split[i].double() - r).abs().max()) / scale
        e3 = float((f32[i].double() - r).abs().max()) / scale
        report[i] = (es, e3)
        assert es <= 1e-4, (i, es)                                   # the feature contract
        assert es <= 2.0 * e3 + 2e-7, (i, es, e3)                    # float32-equivalent: within 2x of float32 instructions
    print('hostile DRN-D-22, error of scale (split planes, float32 instructions) per map:', report)
    # and channel by channel: the small channels of a map are as accurate relative to THEIR scale
    r = ref[2].cuda()
    ch_scale = r.abs().amax(dim=(0, 2, 3)).clamp_min(1e-30)
    es_c = ((split[2].double() - r).abs().amax(dim=(0, 2, 3)) / ch_scale).max()
    e3_c = ((f32[2].double() - r).abs().amax(dim=(0, 2, 3)) / ch_scale).max()
    print('per-channel worst case, map 2: split %.2e, float32 instructions %.2e' % (float(es_c), float(e3_c)))
    assert float(es_c) <= 2.0 * float(e3_c) + 1e-6


def _bulk_error(y, ref, touched):
    keep = ~touched
    scale = float(ref[keep.expand_as(ref)].abs().max())
    return float(((y.double() - ref).abs() * keep).max()) / scale


@pytest.mark.parametrize('Cin,Cout,dil,form', [(256, 256, 2, 'wino'), (512, 512, 1, 'wino'), (64, 64, 1, 'direct'), (128, 128, 2, 'direct'),
                                               (256, 512, 1, '1x1')])
def test_heavy_tailed_activations_single_layers(eng, Cin, Cout, dil, form):
    """(b): unit-scale bulk + 1e3-sized outliers (1 in 20 000 elements).  On the outputs no outlier reaches, the error
    relative to the bulk's scale stays float32-class and within 2x of the float32-instruction kernel's."""
    g = torch.Generator(device='cuda').manual_seed(31)
    B, H, W = 2, 40, 72
    x = torch.relu(torch.randn((B, Cin, H, W), device='cuda', generator=g))
    hot = torch.zeros((B, Cin, H, W), dtype=torch.bool, device='cuda')
    n_hot = 8                                                    # "a few 1e3 outliers per tensor"
    idx = torch.randint(0, hot.numel(), (n_hot,), device='cuda', generator=g)
    hot.view(-1)[idx] = True
    x = torch.where(hot, x * 1e3 + 1e3, x).contiguous(memory_format=torch.channels_last)
    k = 1 if form == '1x1' else 3
    w = torch.randn((Cout, Cin, k, k), device='cuda', generator=g) * (2.0 / (k * k * Cin)) ** 0.5
    bias = torch.randn((Cout,), device='cuda', generator=g)
    ref = F.conv2d(x.double(), w.double(), bias.double(), 1, dil * (k // 2), dil)
    # outputs an outlier reaches.  Direct forms: its receptive field.  Winograd F(4x4,3x3): every output of a tile whose
    # 6 x 6 input patch holds the outlier (the transforms spread it over the tile's 36 products and cancel it again in
    # float32 — in EITHER arithmetic: reported below, not asserted), i.e. up to 5 sub-grid pixels away
    reach = 5 if form == 'wino' else k // 2
    any_hot = hot.any(dim=1, keepdim=True).double()
    kk = 2 * reach + 1
    touched = F.conv2d(any_hot, torch.ones((1, 1, kk, kk), device='cuda', dtype=torch.float64), None, 1, dil * reach, dil) > 0
    assert 0.0 < float(touched.double().mean()) < 0.9
    if form == 'wino':
        u2, cs = eng.winograd_weights_split(w)
        y, _ = eng.conv3x3_wino_f16s(x, u2, cs, bias, None, False, dil)
        y32 = eng.conv3x3_wino_f32(x, eng.winograd_weights(w, 4), bias, None, False, dil)
    else:
        taps = k * k
        wt = w.permute(0, 2, 3, 1).reshape(Cout, taps, Cin).contiguous()
        wt2, inv_t = eng.split_planes(wt)
        y, _ = eng.conv3x3_f16s(x, wt2, inv_t, bias, None, False, dil)
        y32 = eng.conv3x3_f32(x, wt, bias, None, False, dil)
    assert torch.isfinite(y).all()
    es, e3 = _bulk_error(y, ref, touched), _bulk_error(y32, ref, touched)
    glob = float((y.double() - ref).abs().max()) / float(ref.abs().max())
    near = touched.expand_as(ref)
    bulk_scale = float(ref[~near].abs().max())
    es_near = float(((y.double() - ref).abs() * near).max()) / bulk_scale
    e3_near = float(((y32.double() - ref).abs() * near).max()) / bulk_scale
    print('%s %d->%d: bulk error split %.2e, float32 instructions %.2e; global %.2e; beside an outlier, of the bulk scale: split %.2e, '
          'float32 instructions %.2e' % (form, Cin, Cout, es, e3, glob, es_near, e3_near))
    assert es <= 2.0 * e3 + 2e-6 and es <= 2e-5 and glob <= 1e-5
    assert es_near <= 2.0 * e3_near + 2e-6


def test_state_dict_key_set_is_the_reference_modules(eng):
    """both architectures: our module names / shapes (fc included) are the reference's PyTorch modules', so the published
    checkpoints load with strict=True"""
    drn = importlib.import_module('superpixel-align_amd.drn')
    keys = json.load(open(os.path.join(HERE, 'golden', 'drn_state_dict_keys.json')))
    for name in ('drn_d_22', 'drn_c_26'):
        m = drn.DRN(name, with_fc=True)
        own = {k: list(v.shape) for k, v in m.state_dict().items() if not k.endswith('num_batches_tracked')}
        assert own == keys[name]
