"""Dilated Residual Network feature extractor for the label-generation path (PyTorch-ROCm).

Role in the hot path: `model.batch_predict(imgs)` -> 8 intermediate maps, of which the caller
keeps `maps[i] for i in --use_feature_maps` (default [7] = layer8, 512 x H/8 x W/8)
(reference: batch_spalign_kmeans.py:431-435, models/drn.py:230-325).

What is kept from the reference, because results depend on it
  * architecture C-26 (the reference's create_model, batch_spalign_kmeans.py:524-526) and
    D-22 (BASELINE north star); layer plan [1,1,2,2,2,2,1,1], channels 16..512, dilations
    2 (layer5) / 4 (layer6) / 2 (layer7) / 1 (layer8), layer7/8 of arch C without residual
    (models/drn.py:158-165), arch D plain conv stacks for layers 1,2,7,8 (:134-145,166-170);
  * the map list uses the CHAINER convention: maps[i] is the output of layer(i+1) for both
    archs — Chainer's arch D does not export layer0 (models/drn.py:238-243) although the
    PyTorch file does (models/drn_pytorch.py:216-218);
  * BatchNorm eps 2e-5: load_npz into a fresh Chainer model restores parameters and running
    statistics only, so the published labels were produced with Chainer's default eps, not the
    1e-5 of the PyTorch checkpoint;
  * input normalisation arithmetic of DRN.batch_predict (models/drn.py:319-321): x/255 in
    float32, then (x - mean) and (x / std) evaluated in float64 and stored as float32.

What is designed for MI355X instead of translated
  * inference only: BatchNorm is folded into the preceding convolution at load time
    (`fold_bn=True`), halving the number of memory-bound elementwise passes over the
    16..512-channel full-resolution activations; the unused 1x1 `fc` head
    (models/drn.py:275-276, discarded at batch_spalign_kmeans.py:431) is not evaluated;
  * activations are kept channels-last (NHWC) end to end: that is MIOpen's preferred layout
    for the MFMA implicit-GEMM kernels and it is exactly the layout the pooling kernels of
    libspalign want (C contiguous per feature pixel), so no transpose separates the two;
  * optional bfloat16 execution (`dtype=torch.bfloat16`, BASELINE config 5) with float32
    accumulation inside the convolutions; the pooling kernels read bf16 directly.
Module names follow the upstream checkpoint (conv1/bn1/layerN.M.convK/downsample.0|1/fc) so
`drn_c_26-ddedf421.pth` / `drn_d_22-4bd2f8ea.pth` load with load_state_dict, and
`models/drn_c_26.npz` (Chainer) loads through `load_chainer_npz`.
"""
import math
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)
CHAINER_BN_EPS = 2e-5

_PLANS = {
    # name: (arch, blocks per layer)
    'drn_c_26': ('C', (1, 1, 2, 2, 2, 2, 1, 1)),
    'drn_d_22': ('D', (1, 1, 2, 2, 2, 2, 1, 1)),
}
_CHANNELS = (16, 32, 64, 128, 256, 512, 512, 512)


def _conv(cin, cout, k, stride=1, dilation=1):
    pad = dilation * (k // 2)
    return nn.Conv2d(cin, cout, k, stride=stride, padding=pad, dilation=dilation, bias=False)


# set by DRN.prepare() on a GPU: libspalign's fused bias/residual/ReLU; 'bytes' accumulates the algorithmic
# HBM bytes of its launches (read y + write y [+ read residual]) for bench.py's roofline entry
_EPILOGUE = {'engine': None, 'bytes': 0, 'launches': 0, 'own_conv': True, 'own_conv32': True, 'winograd': 4, 'wino_saved_flops': 0.0,
             'conv_flops': 0.0, 'light_flops': 0.0, 'light_bytes': 0.0, 'light_launches': 0, 'gemm_flops': 0.0, 'gemm_launches': 0, 'gemm_bytes': 0.0,
             'gemmn_flops': 0.0, 'gemmn_launches': 0, 'gemmn_bytes': 0.0, 'wino_direct_flops': 0.0, 'wino_in_bytes': 0.0,
             'wino_out_bytes': 0.0, 'wino_launches': 0,
             # F(4x4,3x3) GEMMs on the 16-bit matrix cores at float32 accuracy (two half-precision planes per operand, three
             # products: csrc/spa_gemm16.hip); SPA_SPLIT_GEMM=0 keeps the float32 matrix instructions
             'split_gemm': os.environ.get('SPA_SPLIT_GEMM', '1') != '0',
             'gemm16_flops': 0.0, 'gemm16_launches': 0, 'gemm16_bytes': 0.0,
             'gemm16n_flops': 0.0, 'gemm16n_launches': 0, 'gemm16n_bytes': 0.0, 'conv16_flops': 0.0, 'conv16_launches': 0, 'conv16_bytes': 0.0,
             # ... and per kernel instantiation (one `kernels` entry = one rocprofv3 row): c16[kind] = [flops, bytes, launches]
             'c16': {},
             # the layer's OWN bytes of the 256 x 256-tile Winograd layers (X, Y, the residual, the weight planes once): what a
             # convolution that kept V and M on chip would move — bench.py's `algorithmic_bytes` of the GEMM entry
             'gemm16_layer_bytes': 0.0,
             # convolutions handed to MIOpen (F.conv2d) by conv_bias_act: a forward that has any is not captured into a graph
             'library_convs': 0}


def _c16(kind, flops, nbytes, launches=1):
    """per-instantiation counters of the split-plane direct kernels (bench.py prices each against its own rocprofv3 row)"""
    c = _EPILOGUE['c16'].setdefault(kind, [0.0, 0.0, 0])
    c[0] += flops
    c[1] += nbytes
    c[2] += launches


def _epi_snapshot():
    """The numeric counters of _EPILOGUE (bench.py's FLOP / byte accounting), for the graph replay below."""
    snap = {k: v for k, v in _EPILOGUE.items() if isinstance(v, (int, float)) and not isinstance(v, bool)}
    snap['c16'] = {k: list(v) for k, v in _EPILOGUE['c16'].items()}
    return snap


def _epi_restore(snap):
    for k, v in snap.items():
        if k != 'c16':
            _EPILOGUE[k] = v
    _EPILOGUE['c16'] = {k: list(v) for k, v in snap['c16'].items()}


def _epi_delta(before):
    d = {k: _EPILOGUE[k] - v for k, v in before.items() if k != 'c16'}
    d['c16'] = {k: [a - b for a, b in zip(v, before['c16'].get(k, [0.0, 0.0, 0]))] for k, v in _EPILOGUE['c16'].items()}
    return d


def _epi_add(delta):
    for k, v in delta.items():
        if k != 'c16':
            _EPILOGUE[k] += v
    for k, v in delta['c16'].items():
        c = _EPILOGUE['c16'].setdefault(k, [0.0, 0.0, 0])
        for i in range(3):
            c[i] += v[i]


# The forward of a SMALL batch is bound by its ~100 launches, not by the GPU (30 images of 224 x 224, the reference's operating
# point: 9.3 ms of wall clock for 6 ms of kernels; every layer is a few python calls and a ctypes call): such shapes are captured
# once into a HIP graph (torch.cuda.CUDAGraph: the libspalign launches go to torch's current stream, which is the capturing one)
# and replayed — the same kernels with the same arguments, hence the same bits.  SPA_DRN_GRAPH: 1 = every shape, 0 = never,
# unset = batches of at most SPA_DRN_GRAPH_PIXELS pixels (default 4 Mi: full-size batches are GPU bound and keep their
# per-kernel timers).
def _graph_wanted(x):
    mode = os.environ.get('SPA_DRN_GRAPH', '')
    if mode == '0' or not x.is_cuda:
        return False
    if mode == '1':
        return True
    return x.shape[0] * x.shape[2] * x.shape[3] <= int(os.environ.get('SPA_DRN_GRAPH_PIXELS', str(4 << 20)))


def conv_bias_act(conv, bn, x, residual=None, relu=True):
    """relu?(bn(conv(x)) [+ residual]).  With BatchNorm folded (bn is Identity, conv carries the
    bias) and the tensors on the GPU in channels-last storage, the bias add, the residual add and
    the ReLU run as ONE in-place pass of libspalign (spa_bias_act) behind the MIOpen convolution
    instead of two or three separate elementwise kernels."""
    eng = _EPILOGUE['engine']
    if (eng is not None and x.is_cuda and isinstance(bn, nn.Identity) and conv.bias is not None
            and x.dtype in (torch.float32, torch.bfloat16)):
        packed = getattr(conv, '_spa_packed', None)
        if (packed is not None and x.dtype == torch.bfloat16 and _EPILOGUE['own_conv']
                and x.is_contiguous(memory_format=torch.channels_last)
                and (residual is None or residual.is_contiguous(memory_format=torch.channels_last))):
            # the heavy 3x3 (dilated) layers: libspalign's bf16 implicit GEMM with bias / residual / ReLU
            # in its epilogue (no separate elementwise pass, one rounding to bf16)
            _EPILOGUE['conv_flops'] += 2.0 * x.shape[0] * x.shape[2] * x.shape[3] * conv.out_channels * 9 * conv.in_channels
            return eng.conv3x3_bf16(x, packed[0], packed[1], residual, relu, conv.dilation[0])
        light = getattr(conv, '_spa_light', None)
        if (light is not None and x.dtype == torch.bfloat16 and _EPILOGUE['own_conv'] and os.environ.get('SPA_BF16_LIGHT', '1') != '0'
                and x.is_contiguous(memory_format=torch.channels_last)
                and (residual is None or residual.is_contiguous(memory_format=torch.channels_last))):
            # the light layers of the bf16 network (stride 2, 16 / 32 channels, 1x1 projections): libspalign's plain bf16 kernel
            opx = x.shape[0] * -(-x.shape[2] // conv.stride[0]) * -(-x.shape[3] // conv.stride[1])
            _EPILOGUE['light_flops'] += 2.0 * opx * conv.out_channels * light[0].shape[1] * conv.in_channels
            _EPILOGUE['light_bytes'] += 2.0 * (x.shape[0] * x.shape[2] * x.shape[3] * conv.in_channels
                                               + opx * conv.out_channels * (2 if residual is not None else 1))
            _EPILOGUE['light_launches'] += 1
            return eng.conv_bf16_light(x, light[0], light[1], residual, relu, conv.stride[0], conv.dilation[0])
        tile = _EPILOGUE['winograd']                   # 4 (default), 2 or 0/False
        wino = getattr(conv, '_spa_wino', {}).get(4 if tile == 4 else 2) if tile else None
        # with the split-plane kernels the direct form wins below 256 input channels (30 x 128 x 256 pixels, ms Winograd /
        # direct: 128 -> 128 1.10 / 0.99, 128 -> 256 1.85 / 1.61, 256 -> 256 2.65 / 3.09, 256 -> 512 4.28 / 5.35)
        if (wino is not None and _EPILOGUE['split_gemm'] and conv.in_channels < int(os.environ.get('SPA_WINO_MIN_CIN', '256'))
                and getattr(conv, '_spa_packed16', None) is not None):
            bn_ = 256 if conv.out_channels % 256 == 0 else 128             # the direct kernel's pixel tile (rule below)
            if -(-x.shape[3] // bn_) * bn_ <= 1.25 * x.shape[3]:
                wino = None
        if (wino is not None and x.dtype == torch.float32 and _EPILOGUE['own_conv32']
                and x.is_contiguous(memory_format=torch.channels_last)
                and (residual is None or residual.is_contiguous(memory_format=torch.channels_last))):
            # Winograd: F(4x4,3x3) multiplies 36/144 = 1/4 of the direct form's products and expands the activations
            # 2.25x (V, M); F(2x2,3x3) 16/36 and 4x.  Executed and direct-equivalent FLOPs are both counted.
            frac, expand = (0.25, 2.25) if wino[0].shape[0] == 36 else (16.0 / 36.0, 4.0)
            direct = 2.0 * x.shape[0] * x.shape[2] * x.shape[3] * conv.out_channels * 9 * conv.in_channels
            _EPILOGUE['wino_direct_flops'] += direct
            _EPILOGUE['wino_saved_flops'] += direct * (1.0 - frac)
            px = x.shape[0] * x.shape[2] * x.shape[3]
            split = wino[0].shape[0] == 36 and _EPILOGUE['split_gemm'] and conv._spa_wino.get('4s')
            in_bytes = 4.0 * px * (1 + expand) * conv.in_channels            # k_wino_in reads X and writes V
            mm_bytes = 4.0 * px * expand * (conv.in_channels + conv.out_channels)     # V read, M written
            out_bytes = 4.0 * px * ((2 if residual is not None else 1) + expand) * conv.out_channels   # M [+ R] read, Y written
            key = 'gemm' if conv.out_channels % 256 == 0 else 'gemmn'       # the 256 x 256 instance / the narrow tiles
            if split:
                key = key.replace('gemm', 'gemm16')
                if key == 'gemm16':
                    _EPILOGUE['gemm16_layer_bytes'] += 4.0 * px * (conv.in_channels + (2 if residual is not None else 1) * conv.out_channels) \
                        + 4.0 * 36 * conv.in_channels * conv.out_channels
            _EPILOGUE[key + '_flops'] += direct * frac
            _EPILOGUE[key + '_launches'] += 1
            _EPILOGUE[key + '_bytes'] += mm_bytes
            _EPILOGUE['wino_in_bytes'] += in_bytes
            _EPILOGUE['wino_out_bytes'] += out_bytes
            _EPILOGUE['wino_launches'] += 1
            if split:
                # the bound on max |x| that scales V travels with the tensor object from the call that produced it
                y, am = eng.conv3x3_wino_f16s(x, split[0], split[1], split[2], residual, relu, conv.dilation[0],
                                              amax_in=getattr(x, '_spa_amax', None))
                y._spa_amax = am
                return y
            return eng.conv3x3_wino_f32(x, wino[0], wino[1], residual, relu, conv.dilation[0])
        packed32 = getattr(conv, '_spa_packed32', None)
        # the direct kernel tiles an image ROW into 256 (128) pixels: on narrow maps most of a tile is padding
        # (28 pixels at the reference's 224 x 224 operating point: 403 images/s against 2 113 with MIOpen there), so
        # it is used where a row fills its tiles to 80 % (the Winograd path above flattens the tiles and does not care)
        if packed32 is not None and not (packed32[0].shape[1] == 1 and getattr(conv, '_spa_packed16', None) is not None
                                         and _EPILOGUE['split_gemm']):
            # (the split-plane 1x1 form takes the image as one row: no such limit, engine.conv3x3_f16s)
            # (... unless the layer is so small that a library convolution's launches cost more than the empty part of the tiles:
            # the 64-channel layers of a 224 x 224 input; the forward then has no library call left and runs as a captured graph)
            bn = 256 if conv.out_channels % 256 == 0 else 128
            if -(-x.shape[3] // bn) * bn > 1.25 * x.shape[3] and x.shape[0] * x.shape[2] * x.shape[3] > (1 << 20):
                packed32 = None
        if (packed32 is not None and x.dtype == torch.float32 and _EPILOGUE['own_conv32']
                and x.is_contiguous(memory_format=torch.channels_last)
                and (residual is None or residual.is_contiguous(memory_format=torch.channels_last))):
            # the same layers of the float32 network: libspalign's float32-MFMA implicit GEMM, epilogue fused
            fl = 2.0 * x.shape[0] * x.shape[2] * x.shape[3] * conv.out_channels * packed32[0].shape[1] * conv.in_channels
            p16 = getattr(conv, '_spa_packed16', None)
            if p16 is not None and _EPILOGUE['split_gemm']:
                # ... on the 16-bit matrix cores at float32 accuracy (two half-precision planes per operand)
                by = 4.0 * x.shape[0] * x.shape[2] * x.shape[3] * (conv.in_channels + conv.out_channels * (2 if residual is not None else 1))
                _EPILOGUE['conv16_flops'] += fl
                _EPILOGUE['conv16_bytes'] += by
                _EPILOGUE['conv16_launches'] += 1
                _c16('1x1' if packed32[0].shape[1] == 1 else ('256' if conv.out_channels % 256 == 0 else ('128' if conv.out_channels % 128 == 0 else '64')), fl, by)
                y, am = eng.conv3x3_f16s(x, p16[0], p16[1], packed32[1], residual, relu, conv.dilation[0],
                                         amax_in=getattr(x, '_spa_amax', None))
                y._spa_amax = am
                return y
            if packed32[0].shape[1] == 1:           # the GEMM form of the kernel (1x1 projection)
                key = 'gemm' if conv.out_channels % 256 == 0 and residual is None else 'gemmn'
                _EPILOGUE[key + '_flops'] += fl
                _EPILOGUE[key + '_launches'] += 1
                _EPILOGUE[key + '_bytes'] += 4.0 * x.shape[0] * x.shape[2] * x.shape[3] * (conv.in_channels + conv.out_channels)
            else:
                _EPILOGUE['conv_flops'] += fl
            return eng.conv3x3_f32(x, packed32[0], packed32[1], residual, relu, conv.dilation[0])
        l2 = getattr(conv, '_spa_layer2', None)
        if (l2 is not None and _EPILOGUE['split_gemm'] and _EPILOGUE['own_conv32'] and x.dtype == torch.float32 and relu
                and residual is None and x.is_contiguous(memory_format=torch.channels_last)):
            # layer 2 of arch D (16 -> 32 channels, stride 2): its own kernel on the 16-bit matrix cores
            fl2 = 2.0 * x.shape[0] * ((x.shape[2] + 1) // 2) * ((x.shape[3] + 1) // 2) * 32 * 9 * 16
            by2 = 4.0 * x.shape[0] * (x.shape[2] * x.shape[3] * 16 + ((x.shape[2] + 1) // 2) * ((x.shape[3] + 1) // 2) * 32)
            _EPILOGUE['conv16_flops'] += fl2
            _EPILOGUE['conv16_bytes'] += by2
            _EPILOGUE['conv16_launches'] += 1
            _c16('front', fl2, by2)
            return eng.drn_layer2_f16s(x, l2[0], l2[1], l2[2], amax_in=getattr(x, '_spa_amax', None))
        l232 = getattr(conv, '_spa_layer2_32', None)
        if (l232 is not None and not _EPILOGUE['split_gemm'] and _EPILOGUE['own_conv32'] and x.dtype == torch.float32 and relu
                and residual is None and x.is_contiguous(memory_format=torch.channels_last)):
            # the strict float32 network (float32 instructions everywhere): layer 2 as fmaf chains on the vector pipe
            _EPILOGUE['conv_flops'] += 2.0 * x.shape[0] * ((x.shape[2] + 1) // 2) * ((x.shape[3] + 1) // 2) * 32 * 9 * 16
            return eng.drn_layer2_f32(x, l232[0], l232[1])
        _EPILOGUE['library_convs'] += 1               # (MIOpen picks its algorithm per call context: such a forward is not captured)
        if os.environ.get('SPA_DRN_TRACE_LIBRARY'):
            import sys
            sys.stderr.write('[drn] library convolution: %d -> %d, kernel %s stride %s dilation %s on %s %s\n' % (
                conv.in_channels, conv.out_channels, tuple(conv.kernel_size), tuple(conv.stride), tuple(conv.dilation), tuple(x.shape), x.dtype))
        y = F.conv2d(x, conv.weight, None, conv.stride, conv.padding, conv.dilation)
        vec = 4 if y.dtype == torch.float32 else 8
        if (y.is_contiguous(memory_format=torch.channels_last) and y.shape[1] % vec == 0
                and (residual is None or residual.is_contiguous(memory_format=torch.channels_last))):
            _EPILOGUE['bytes'] += y.numel() * y.element_size() * (3 if residual is not None else 2)
            _EPILOGUE['launches'] += 1
            return eng.bias_act_(y, conv.bias, residual, relu, track_amax=_EPILOGUE['split_gemm'])
        y = y + conv.bias.view(1, -1, 1, 1)
    else:
        _EPILOGUE['library_convs'] += 1
        y = bn(conv(x))
    if residual is not None:
        y = y + residual
    return F.relu_(y) if relu else y


class BasicBlock(nn.Module):
    """conv-bn-relu-conv-bn (+ residual) - relu; `residual=False` for layers 7/8 of arch C."""

    def __init__(self, cin, cout, stride, dilation, residual, eps, project):
        super().__init__()
        self.conv1 = _conv(cin, cout, 3, stride, dilation[0])
        self.bn1 = nn.BatchNorm2d(cout, eps=eps)
        self.conv2 = _conv(cout, cout, 3, 1, dilation[1])
        self.bn2 = nn.BatchNorm2d(cout, eps=eps)
        self.downsample = None
        if project:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride=stride, bias=False),
                                            nn.BatchNorm2d(cout, eps=eps))
        self.residual = residual

    def forward(self, x):
        s2 = getattr(self, '_spa_s2', None)
        eng = _EPILOGUE['engine']
        if (s2 is not None and eng is not None and _EPILOGUE['split_gemm'] and _EPILOGUE['own_conv32'] and x.is_cuda
                and x.dtype == torch.float32 and x.is_contiguous(memory_format=torch.channels_last)
                and (128 * -(-((x.shape[3] + 1) // 2) // 128) <= 1.25 * ((x.shape[3] + 1) // 2)
                     # (narrow maps — 224 x 224 inputs, the reference's operating point — leave most of a 128-pixel tile empty; the
                     # layer is then so small that this costs less than the launches of a library convolution and its epilogue
                     # pass, and the forward stays on libspalign's kernels, which is what lets it run as a captured graph)
                     or x.shape[0] * ((x.shape[2] + 1) // 2) * ((x.shape[3] + 1) // 2) <= (1 << 20))):
            # the stride-2 opening convolution and the 1x1 stride-2 projection in ONE pass over x (csrc/spa_conv32.hip)
            fl2 = 2.0 * x.shape[0] * ((x.shape[2] + 1) // 2) * ((x.shape[3] + 1) // 2) * s2[3] * 10 * x.shape[1]
            by2 = 4.0 * x.shape[0] * (x.shape[2] * x.shape[3] * x.shape[1] + ((x.shape[2] + 1) // 2) * ((x.shape[3] + 1) // 2) * 2 * s2[3])
            _EPILOGUE['conv16_flops'] += fl2
            _EPILOGUE['conv16_bytes'] += by2
            _EPILOGUE['conv16_launches'] += 1
            _c16('front', fl2, by2)
            y, res, am = eng.conv3x3_s2_f16s(x, s2[0], s2[1], s2[2], s2[3], True, amax_in=getattr(x, '_spa_amax', None))
            y._spa_amax = am
            return conv_bias_act(self.conv2, self.bn2, y, res, True)
        s232 = getattr(self, '_spa_s2_32', None)
        if (s232 is not None and eng is not None and not _EPILOGUE['split_gemm'] and _EPILOGUE['own_conv32'] and x.is_cuda
                and x.dtype == torch.float32 and x.is_contiguous(memory_format=torch.channels_last)):
            # the strict float32 network: the same one-pass opener + projection with float32 matrix instructions
            _EPILOGUE['conv_flops'] += 2.0 * x.shape[0] * ((x.shape[2] + 1) // 2) * ((x.shape[3] + 1) // 2) * s232[2] * 10 * x.shape[1]
            y, res = eng.conv3x3_s2_f32(x, s232[0], s232[1], s232[2], True)
            return conv_bias_act(self.conv2, self.bn2, y, res, True)
        y = conv_bias_act(self.conv1, self.bn1, x, None, True)
        res = None
        if self.residual:
            res = x if self.downsample is None else conv_bias_act(self.downsample[0], self.downsample[1],
                                                                  x, None, False)
        return conv_bias_act(self.conv2, self.bn2, y, res, True)


class DRN(nn.Module):
    def __init__(self, name='drn_c_26', bn_eps=CHAINER_BN_EPS, num_classes=1000, with_fc=False):
        super().__init__()
        arch, plan = _PLANS[name]
        self.name, self.arch = name, arch
        ch = _CHANNELS
        self._cin = ch[0]
        eps = bn_eps
        stem = [_conv(3, ch[0], 7), nn.BatchNorm2d(ch[0], eps=eps)]
        if arch == 'C':
            self.conv1, self.bn1 = stem
            self.layer1 = self._blocks(ch[0], plan[0], 1, 1, True, True, eps)
            self.layer2 = self._blocks(ch[1], plan[1], 2, 1, True, True, eps)
        else:
            self.layer0 = nn.Sequential(*stem, nn.ReLU(inplace=True))
            self.layer1 = self._plain(ch[0], plan[0], 1, 1, eps)
            self.layer2 = self._plain(ch[1], plan[1], 2, 1, eps)
        self.layer3 = self._blocks(ch[2], plan[2], 2, 1, True, True, eps)
        self.layer4 = self._blocks(ch[3], plan[3], 2, 1, True, True, eps)
        self.layer5 = self._blocks(ch[4], plan[4], 1, 2, False, True, eps)
        self.layer6 = self._blocks(ch[5], plan[5], 1, 4, False, True, eps)
        if arch == 'C':
            self.layer7 = self._blocks(ch[6], plan[6], 1, 2, False, False, eps)
            self.layer8 = self._blocks(ch[7], plan[7], 1, 1, False, False, eps)
        else:
            self.layer7 = self._plain(ch[6], plan[6], 1, 2, eps)
            self.layer8 = self._plain(ch[7], plan[7], 1, 1, eps)
        self.fc = nn.Conv2d(ch[7], num_classes, 1) if with_fc else None
        self.folded = False
        self.compute_dtype = torch.float32
        self.use_fused_stem = True      # arch D, float32, folded BN, on the GPU: libspalign's stem kernel
        self._stem = None
        for m in self.modules():           # the reference's random init (models/drn.py:176-184)
            if isinstance(m, nn.Conv2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2.0 / n))
        self.eval()

    # -- builders --------------------------------------------------------------------------
    def _blocks(self, cout, count, stride, dilation, new_level, residual, eps):
        cin = self._cin
        project = stride != 1 or cin != cout
        first = (1, 1) if dilation == 1 else ((dilation // 2 if new_level else dilation), dilation)
        mods = [BasicBlock(cin, cout, stride, first, residual, eps, project)]
        for _ in range(1, count):
            mods.append(BasicBlock(cout, cout, 1, (dilation, dilation), residual, eps, False))
        self._cin = cout
        return nn.Sequential(*mods)

    def _plain(self, cout, count, stride, dilation, eps):
        mods = []
        for i in range(count):
            mods += [_conv(self._cin, cout, 3, stride if i == 0 else 1, dilation),
                     nn.BatchNorm2d(cout, eps=eps), nn.ReLU(inplace=True)]
            self._cin = cout
        return nn.Sequential(*mods)

    # -- inference-time transformations ---------------------------------------------------------
    @torch.no_grad()
    def fold_batchnorm(self):
        """Fold every BatchNorm into the convolution in front of it (eval-mode identity)."""
        def fold(conv, bn):
            scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
            conv.weight.mul_(scale.reshape(-1, 1, 1, 1))
            bias = bn.bias - bn.running_mean * scale
            conv.bias = nn.Parameter(bias if conv.bias is None else conv.bias * scale + bias)
            return nn.Identity()

        def walk(seq):
            mods = list(seq.children())
            for i, m in enumerate(mods[:-1]):
                if isinstance(m, nn.Conv2d) and isinstance(mods[i + 1], nn.BatchNorm2d):
                    seq[i + 1] = fold(m, mods[i + 1])
        if self.folded:
            return self
        if self.arch == 'C':
            self.bn1 = fold(self.conv1, self.bn1)
        for m in self.modules():
            if isinstance(m, BasicBlock):
                m.bn1 = fold(m.conv1, m.bn1)
                m.bn2 = fold(m.conv2, m.bn2)
                if m.downsample is not None:
                    walk(m.downsample)
            elif isinstance(m, nn.Sequential):
                walk(m)
        self.folded = True
        return self

    def prepare(self, device='cuda', dtype=torch.float32, fold_bn=True):
        """Move to the GPU in the layout/dtype the hot path runs in."""
        if fold_bn:
            self.fold_batchnorm()
        self.compute_dtype = dtype
        self.to(device=device, dtype=dtype, memory_format=torch.channels_last)
        self.eval()
        self.__dict__.pop('_graphs', None)              # captured forwards hold the OLD packed operands' addresses
        self._stem = None
        if torch.device(device).type == 'cuda':
            from .engine import default_engine, Engine        # fused glue kernels of libspalign
            _EPILOGUE['engine'] = default_engine()
            for m in self.modules():
                if isinstance(m, nn.Conv2d):
                    m._spa_packed = None
                    m._spa_packed32 = None
                    m._spa_packed16 = None
                    m._spa_layer2 = None
                    m._spa_layer2_32 = None
                    if (dtype == torch.float32 and self.folded and m.kernel_size == (3, 3) and m.stride == (2, 2)
                            and m.padding == (1, 1) and m.dilation == (1, 1) and m.groups == 1 and m.in_channels == 16
                            and m.out_channels == 32 and m.bias is not None):
                        m._spa_layer2 = Engine.layer2_planes(m.weight) + (m.bias.detach().float().contiguous(),)
                        # (the strict float32 network's form: (tap, input channel, output channel) float32)
                        m._spa_layer2_32 = (m.weight.detach().float().permute(2, 3, 1, 0).reshape(9, 16, 32).contiguous(),
                                            m.bias.detach().float().contiguous())
                    m._spa_wino = {}
                    # Winograd where it wins (measured, 30 x 128 x 256 pixels, ms direct / F(2x2) / F(4x4)):
                    #   512 -> 512  33.2 / 19.8 / 11.9     256 -> 512  16.8 / 11.6 / 7.0     256 -> 256  8.5 / 6.6 / 4.2
                    #   128 -> 256   4.3 /  4.1 /  2.6     128 -> 128   2.4 /  2.4 / 1.6      64 -> 64 (256 x 512)  2.6 / - / 2.7
                    # F(4x4,3x3) from 128 input channels up, F(2x2,3x3) (kept as an option) from 256
                    if (dtype == torch.float32 and self.folded and m.kernel_size == (3, 3) and m.stride == (1, 1)
                            and m.padding == m.dilation and m.dilation[0] == m.dilation[1] and m.groups == 1
                            and m.in_channels % 32 == 0 and m.out_channels % 64 == 0 and m.bias is not None
                            and m.in_channels >= 128 and m.out_channels >= 128):
                        bias32 = m.bias.detach().float().contiguous()
                        m._spa_wino[4] = (Engine.winograd_weights(m.weight, 4), bias32)
                        if m.out_channels % 128 == 0:
                            m._spa_wino['4s'] = Engine.winograd_weights_split(m.weight) + (bias32,)
                        if m.in_channels >= 256 and m.out_channels >= 256:
                            m._spa_wino[2] = (Engine.winograd_weights(m.weight, 2), bias32)
                    # operands of spa_conv3x3_f32: the same layers of the float32 network (Cin % 32 == 0)
                    if (dtype == torch.float32 and self.folded and m.kernel_size == (3, 3) and m.stride == (1, 1)
                            and m.padding == m.dilation and m.dilation[0] == m.dilation[1] <= 4 and m.groups == 1
                            and m.in_channels % 32 == 0 and m.out_channels % 64 == 0 and m.bias is not None):
                        wt = m.weight.detach().permute(0, 2, 3, 1).reshape(m.out_channels, 9, m.in_channels)
                        m._spa_packed32 = (wt.contiguous().float(), m.bias.detach().float().contiguous())
                        m._spa_packed16 = Engine.split_planes(m._spa_packed32[0])
                    # ... and the 1x1 stride-1 projections (layers 5 and 6): the same kernel, centre tap only
                    if (dtype == torch.float32 and self.folded and m.kernel_size == (1, 1) and m.stride == (1, 1)
                            and m.padding == (0, 0) and m.groups == 1 and m.in_channels % 32 == 0
                            and m.out_channels % 64 == 0 and m.bias is not None):
                        wt = m.weight.detach().reshape(m.out_channels, 1, m.in_channels)
                        m._spa_packed32 = (wt.contiguous().float(), m.bias.detach().float().contiguous())
                        m._spa_packed16 = Engine.split_planes(m._spa_packed32[0])
                    # operands of spa_conv3x3_bf16: 3x3, stride 1, padding = dilation, Cin % 64 == 0,
                    # Cout % 64 == 0 (layers 3-8 of the DRN: all 3x3 stride-1 layers from 64 channels up), bf16 network,
                    # BatchNorm folded
                    if (dtype == torch.bfloat16 and self.folded and m.kernel_size == (3, 3) and m.stride == (1, 1)
                            and m.padding == m.dilation and m.dilation[0] == m.dilation[1] <= 4 and m.groups == 1
                            and m.in_channels % 64 == 0 and m.out_channels % 64 == 0 and m.bias is not None):
                        wt = m.weight.detach().permute(0, 2, 3, 1).reshape(m.out_channels, 9, m.in_channels)
                        m._spa_packed = (wt.contiguous().to(torch.bfloat16), m.bias.detach().float().contiguous())
                    # operands of spa_conv_bf16_light: every other convolution of the bf16 network (stride 2, 16 / 32 input
                    # channels, 1x1 projections)
                    m._spa_light = None
                    if (dtype == torch.bfloat16 and self.folded and getattr(m, '_spa_packed', None) is None and m.groups == 1
                            and m.bias is not None and m.stride in ((1, 1), (2, 2))
                            and ((m.kernel_size == (3, 3) and m.padding == m.dilation and m.dilation[0] == m.dilation[1]
                                  and m.in_channels in (16, 32, 64))
                                 or (m.kernel_size == (1, 1) and m.padding == (0, 0) and m.in_channels in (16, 32, 64, 128, 256)))
                            and (m.out_channels % 64 == 0 and m.in_channels >= 32 or m.out_channels % 32 == 0 and m.in_channels <= 32
                                 or m.out_channels % 16 == 0 and m.in_channels == 16)):
                        taps = m.kernel_size[0] * m.kernel_size[1]
                        wt = m.weight.detach().permute(0, 2, 3, 1).reshape(m.out_channels, taps, m.in_channels)
                        m._spa_light = (wt.contiguous().to(torch.bfloat16), m.bias.detach().float().contiguous())
            for blk in self.modules():
                if isinstance(blk, BasicBlock):
                    blk._spa_s2 = None
                    blk._spa_s2_32 = None
                    c1, ds = blk.conv1, blk.downsample
                    if (dtype == torch.float32 and self.folded and blk.residual and ds is not None and c1.stride == (2, 2)
                            and c1.kernel_size == (3, 3) and c1.padding == (1, 1) and c1.dilation == (1, 1) and c1.bias is not None
                            and ds[0].kernel_size == (1, 1) and ds[0].stride == (2, 2) and ds[0].bias is not None
                            and c1.in_channels % 32 == 0 and c1.out_channels % 64 == 0):
                        co, ci = c1.out_channels, c1.in_channels
                        w = torch.zeros((2 * co, 9, ci), dtype=torch.float32, device=c1.weight.device)
                        w[:co] = c1.weight.detach().float().permute(0, 2, 3, 1).reshape(co, 9, ci)
                        w[co:, 4] = ds[0].weight.detach().float().reshape(co, ci)
                        wt2, inv_t = Engine.split_planes(w)
                        blk._spa_s2 = (wt2, inv_t, torch.cat([c1.bias.detach().float(), ds[0].bias.detach().float()]).contiguous(), co)
                        blk._spa_s2_32 = (w.contiguous(), blk._spa_s2[2], co)      # the strict float32 network's operands
            self._front_c = None
            if self.arch == 'C' and self.folded and dtype == torch.float32:
                # DRN-C's full-resolution front on libspalign's own kernels (csrc/spa_stem.hip, spa_convs.hip): conv1 + layer1's first
                # convolution as the fused stem (which also stores conv1's output, the block's residual), layer1's second convolution,
                # layer2's stride-2 opener together with its 1x1 projection, layer2's second convolution
                c0, b1, b2 = self.conv1, self.layer1[0], self.layer2[0]
                ok = (len(self.layer1) == 1 and len(self.layer2) == 1 and b1.residual and b1.downsample is None and b2.residual
                      and b2.downsample is not None and b2.conv1.stride == (2, 2) and b2.downsample[0].stride == (2, 2)
                      and all(c.bias is not None for c in (c0, b1.conv1, b1.conv2, b2.conv1, b2.conv2, b2.downsample[0])))
                if ok:
                    f32 = lambda t: t.detach().float().contiguous()
                    self._front_c = dict(
                        stem=(f32(c0.weight).reshape(16, 147).contiguous(), f32(c0.bias),
                              f32(b1.conv1.weight).permute(0, 2, 3, 1).reshape(16, 144).contiguous(), f32(b1.conv1.bias)),
                        l1c2=Engine.small_planes(b1.conv2.weight) + (f32(b1.conv2.bias),),
                        l2c1=Engine.small_planes(b2.conv1.weight, b2.downsample[0].weight)
                        + (torch.cat([f32(b2.conv1.bias), f32(b2.downsample[0].bias)]).contiguous(),),
                        l2c2=Engine.small_planes(b2.conv2.weight) + (f32(b2.conv2.bias),))
            self._stem_c16 = None
            if self.arch == 'C' and self.folded and dtype == torch.bfloat16:
                # bf16 DRN-C: conv1 + layer1's first convolution as the fused bf16 stem (which also stores conv1's output, the
                # block's residual); the rest of layers 1 / 2 runs on spa_conv_bf16_light through the modules
                c0, b1 = self.conv1, self.layer1[0]
                if (len(self.layer1) == 1 and b1.residual and b1.downsample is None and c0.bias is not None
                        and b1.conv1.bias is not None and b1.conv1.stride == (1, 1) and b1.conv1.dilation == (1, 1)):
                    self._stem_c16 = (c0.weight.detach().float().reshape(16, 147).contiguous(), c0.bias.detach().float().contiguous(),
                                      b1.conv1.weight.detach().float().permute(0, 2, 3, 1).reshape(16, 144).contiguous(),
                                      b1.conv1.bias.detach().float().contiguous())
            if self.arch == 'D' and self.folded and dtype in (torch.float32, torch.bfloat16):
                # operands of libspalign's fused stem kernel (normalise + layer0 + layer1)
                c0, c1 = self.layer0[0], self.layer1[0]
                self._stem = (c0.weight.detach().float().reshape(16, 147).contiguous(),
                              c0.bias.detach().float().contiguous(),
                              c1.weight.detach().float().permute(0, 2, 3, 1).reshape(16, 144).contiguous(),
                              c1.bias.detach().float().contiguous())
        return self

    # -- forward ---------------------------------------------------------------------------------
    def forward_maps(self, x, layer1_out=None, front_maps=None):
        """x: normalised (B,3,H,W) (or None with `layer1_out`, the fused stem's output, or `front_maps`, the outputs of the
        first len(front_maps) layers computed by libspalign's front kernels).  Returns the 8 maps of the Chainer convention."""
        def plain(seq, t):
            mods = list(seq.children())            # (conv, bn | Identity, relu) triples
            for i in range(0, len(mods), 3):
                t = conv_bias_act(mods[i], mods[i + 1], t, None, True)
            return t

        first = 1
        maps = []
        if front_maps is not None:
            maps = list(front_maps)
            x = maps[-1]
            first = len(maps) + 1
        elif layer1_out is not None:           # the fused stem already produced layer1's output
            x = layer1_out
            maps.append(x)
            first = 2
        elif self.arch == 'C':
            x = conv_bias_act(self.conv1, self.bn1, x, None, True)
        else:
            x = plain(self.layer0, x)
        for i in range(first, 9):
            layer = getattr(self, 'layer%d' % i)
            x = plain(layer, x) if (self.arch == 'D' and i in (1, 2, 7, 8)) else layer(x)
            maps.append(x)
            hook = self.__dict__.get('_layer_hook')
            if hook is not None:
                hook(i)                            # (LabelPipeline: work for another stream enqueued once layer i is in the queue)
        return maps

    def forward(self, x):
        maps = self.forward_maps(x)
        return (self.fc(maps[-1]) if self.fc is not None else None), maps

    @staticmethod
    def normalise(x):
        """DRN.batch_predict arithmetic (models/drn.py:319-321) without touching the input."""
        mean = torch.tensor(IMAGENET_MEAN, dtype=torch.float64, device=x.device).view(1, 3, 1, 1)
        std = torch.tensor(IMAGENET_STD, dtype=torch.float64, device=x.device).view(1, 3, 1, 1)
        x = x / 255.0
        x = (x.double() - mean).float()
        x = (x.double() / std).float()
        return x

    def _forward_chunk(self, xc):
        """One sub-batch: raw (b,3,H,W) float32 0..255 -> the 8 maps."""
        eng = _EPILOGUE['engine']
        fc = getattr(self, '_front_c', None)
        if (eng is not None and xc.is_cuda and fc is not None and self.use_fused_stem and _EPILOGUE['split_gemm']
                and _EPILOGUE['own_conv32'] and self.compute_dtype == torch.float32):
            # DRN-C: conv1 .. layer2 on libspalign's kernels (no MIOpen convolution, no separate epilogue pass)
            E = _EPILOGUE
            B, _, H, W = xc.shape
            a1, y0 = eng.drn_stem_d(xc.float().contiguous(), *fc['stem'], dtype=torch.float32, split=True, want_layer0=True)
            l1, _ = eng.conv_small_f16s(a1, fc['l1c2'][0], fc['l1c2'][1], fc['l1c2'][2], 16, 1, 0, y0, True, amax_in=a1._spa_amax)
            a2, proj = eng.conv_small_f16s(l1, fc['l2c1'][0], fc['l2c1'][1], fc['l2c1'][2], 32, 2, 32, None, True, amax_in=l1._spa_amax)
            l2, _ = eng.conv_small_f16s(a2, fc['l2c2'][0], fc['l2c2'][1], fc['l2c2'][2], 32, 1, 0, proj, True, amax_in=a2._spa_amax)
            Ho, Wo = a2.shape[2], a2.shape[3]
            fl2 = 2.0 * B * (H * W * 16 * 9 * 16 + Ho * Wo * (32 * 9 * 16 + 32 * 16 + 32 * 9 * 32))
            by2 = 4.0 * B * (H * W * (16 + 16 + 16) + H * W * 16 + Ho * Wo * (32 + 32) + Ho * Wo * (32 + 32 + 32))
            E['conv16_flops'] += fl2
            E['conv16_bytes'] += by2
            E['conv16_launches'] += 3
            _c16('front', fl2, by2, 3)
            return self.forward_maps(None, front_maps=[l1, l2])
        sc16 = getattr(self, '_stem_c16', None)
        if (eng is not None and xc.is_cuda and sc16 is not None and self.use_fused_stem and _EPILOGUE['own_conv']
                and self.compute_dtype == torch.bfloat16):
            a1, y0 = eng.drn_stem_d(xc.float().contiguous(), *sc16, dtype=torch.bfloat16, want_layer0=True)
            b1 = self.layer1[0]
            l1 = conv_bias_act(b1.conv2, b1.bn2, a1, y0, True)
            return self.forward_maps(None, front_maps=[l1])
        if eng is not None and xc.is_cuda and getattr(self, '_stem', None) is not None and self.use_fused_stem:
            l1 = eng.drn_stem_d(xc.float().contiguous(), *self._stem, dtype=self.compute_dtype,
                                split=_EPILOGUE['split_gemm'] and self.compute_dtype == torch.float32)
            return self.forward_maps(None, layer1_out=l1)
        if eng is not None and xc.is_cuda:
            xi = eng.drn_normalise(xc.float().contiguous(), self.compute_dtype)
        else:
            xi = self.normalise(xc.float())
            xi = xi.to(self.compute_dtype).contiguous(memory_format=torch.channels_last)
        return self.forward(xi)[1]

    @torch.no_grad()
    def batch_predict(self, x, sub_batch=None, need=None, streams=1):
        """x: (B,3,H,W) float32 RGB 0..255 (numpy or tensor). -> (logits or None, [8 maps]).

        need: indices of the maps the caller will use (the others come back as None: for the
        default --use_feature_maps 7 that avoids keeping / concatenating 38 GB of activations per
        batch of 30 full-size images).  streams: run that many parts of the batch on side streams
        — the memory-bound epilogue passes of one part then run under the MFMA-bound convolutions
        of another (2 parts of 15: the forward is 2 % shorter than with one part of 30, but in the
        full pipeline the label kernels that follow lose as much, so callers leave it at 1).
        Unlike the reference's --gpu -1 path the input array is never modified (the GPU path
        of the reference copies it too, which is the behaviour the launchers rely on)."""
        dev = next(self.parameters()).device
        x = torch.as_tensor(x)
        if x.device != dev:
            x = x.to(dev, non_blocking=True)
        assert x.ndim == 4 and x.shape[1] == 3
        B = x.shape[0]
        keep = set(range(8)) if need is None else set(int(i) for i in need)

        def pick(maps):
            return [m if i in keep else None for i, m in enumerate(maps)]

        parts = []
        if streams > 1 and x.is_cuda and B >= 2 * streams and not sub_batch:
            if len(getattr(self, '_side', ())) < streams:
                self._side = [torch.cuda.Stream(device=dev) for _ in range(streams)]
            cur = torch.cuda.current_stream(dev)
            per = (B + streams - 1) // streams
            for i in range(streams):
                lo, hi = i * per, min(B, (i + 1) * per)
                if lo >= hi:
                    break
                st = self._side[i]
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    xc = x[lo:hi]
                    xc.record_stream(st)
                    parts.append(pick(self._forward_chunk(xc)))
            for i in range(len(parts)):
                cur.wait_stream(self._side[i])
                for m in parts[i]:
                    if m is not None:
                        m.record_stream(cur)
        elif not sub_batch and _EPILOGUE['engine'] is not None and _graph_wanted(x) and not torch.is_grad_enabled():
            parts.append(self._graph_forward(x, keep, pick))
        else:
            sub = B if not sub_batch else sub_batch
            for s in range(0, B, sub):
                parts.append(pick(self._forward_chunk(x[s:s + sub])))
        maps = []
        for i in range(8):
            col = [p[i] for p in parts]
            maps.append(None if col[0] is None else (col[0] if len(col) == 1 else torch.cat(col, 0)))
        return None, maps

    def _graph_forward(self, x, keep, pick):
        """batch_predict's single-chunk forward as a captured graph (see _graph_wanted).  Falls back to the eager forward for a
        shape whose capture failed (and says so once)."""
        E = _EPILOGUE
        eng = E['engine']
        key = (tuple(x.shape), x.dtype, tuple(x.stride()), self.compute_dtype, tuple(sorted(keep)), E['split_gemm'], E['own_conv'],
               E['own_conv32'], E['winograd'], id(eng), self.use_fused_stem,
               tuple(os.environ.get(k) for k in ('SPA_WINO_MIN_CIN', 'SPA_GEMM16_STAGGER', 'SPA_CONV16_STAGGER', 'SPA_C32_LATE_PREFETCH')))
        cache = self.__dict__.setdefault('_graphs', {})
        # a captured launch holds the addresses of libspalign's workspaces (packed stem weights, the zero line, ...): when one of
        # them has been re-allocated since (another model or a larger shape on the same context), every graph is stale
        gen = eng.ws_generation()
        if self.__dict__.get('_graphs_gen') != gen:
            cache.clear()
            self._graphs_gen = gen
        ent = cache.get(key)
        if ent is None:
            # (a graph keeps its input and every activation of the forward: bound what a stream of odd shapes can pin — at most 8
            # graphs and SPA_DRN_GRAPH_BYTES (default 8 GiB) of pinned activations, estimated at 40 x the input)
            limit = int(os.environ.get('SPA_DRN_GRAPH_BYTES', str(8 << 30)))
            need = 40 * x.numel() * 4
            while cache and (len(cache) >= 8 or sum(e[4] for e in cache.values() if e) + need > limit):
                cache.pop(next(iter(cache)))
            ent = self._graph_capture(x, pick)
            if ent is not False:
                ent = ent + (need,)
            if eng.ws_generation() != gen:          # the capture's warm-up grew a workspace: graphs captured before it are stale
                cache.clear()
                self._graphs_gen = eng.ws_generation()
            cache[key] = ent
        if ent is False:
            return pick(self._forward_chunk(x))
        graph, g_in, g_out, delta, _ = ent
        g_in.copy_(x)
        graph.replay()
        _epi_add(delta)
        # (the graph's outputs are overwritten by the next replay: the caller gets its own copy)
        return [None if m is None else m.clone(memory_format=torch.preserve_format) for m in g_out]

    def _graph_capture(self, x, pick):
        eng = _EPILOGUE['engine']
        dev = x.device
        cur = torch.cuda.current_stream(dev)
        side = torch.cuda.Stream(device=dev)
        g_in = x.clone(memory_format=torch.preserve_format)
        snap = _epi_snapshot()
        prof_was = eng.prof_is_on()
        try:
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                for _ in range(2):                  # workspaces, kernel attributes, MIOpen's find: nothing of that inside the capture
                    self._forward_chunk(g_in)
            cur.wait_stream(side)
            torch.cuda.synchronize(dev)
            _epi_restore(snap)
            if prof_was:
                eng.prof_enable(False)              # (event records of the per-kernel timers do not belong into a graph)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side, capture_error_mode='thread_local'):
                g_out = pick(self._forward_chunk(g_in))
            delta = _epi_delta(snap)
            _epi_restore(snap)
            if delta.get('library_convs', 0) > 0 and os.environ.get('SPA_DRN_GRAPH') != '1':
                # MIOpen chooses its algorithm (and with it the summation order) differently inside a capture: the bits of an
                # eager forward are the contract, so a forward with library convolutions in it stays eager
                return False
            return graph, g_in, g_out, delta
        except Exception as e:                      # noqa: BLE001 - any capture failure means: run this shape eagerly
            import warnings
            warnings.warn('DRN forward of shape %s not captured into a graph (%s: %s); running it launch by launch'
                          % (tuple(x.shape), type(e).__name__, e))
            _epi_restore(snap)
            return False
        finally:
            if prof_was:
                eng.prof_enable(True)

    class _XP(object):
        """`model.xp` of the reference API (utils/apply_spalign_kmeans.py:30)."""
        @staticmethod
        def asarray(a):
            return torch.as_tensor(a)

    xp = _XP()

    # -- weights ---------------------------------------------------------------------------------
    def load_pth(self, path):
        sd = torch.load(path, map_location='cpu')
        sd = {k: v for k, v in sd.items() if self.fc is not None or not k.startswith('fc.')}
        self.load_state_dict(sd, strict=True)
        return self

    def load_chainer_npz(self, path):
        """models/drn_c_26.npz written by the reference's convert_pth2ch.py (chainer save_npz).

        Key grammar of chainer.serializers.save_npz: a Chain names its children by attribute ("layer3", "conv1",
        "downsample"), the reference's Sequential is a ChainList (models/sequential.py:9, :272-290) and numbers
        ONLY its links, in insertion order — F.relu entries are skipped, so Sequential(conv, bn, relu, conv, bn,
        relu) saves "0/W", "1/gamma", "2/W", "3/gamma" where torch's nn.Sequential has modules 0, 1, 3, 4.  A link
        holds W, b (Convolution2D), gamma, beta, avg_mean, avg_var, N (BatchNormalization)."""
        rename = {'W': 'weight', 'b': 'bias', 'gamma': 'weight', 'beta': 'bias',
                  'avg_mean': 'running_mean', 'avg_var': 'running_var'}
        own = self.state_dict()

        def resolve(parts):
            mod, out = self, []
            for p in parts:
                if isinstance(mod, nn.Sequential) and p.isdigit():
                    links = [i for i, m in enumerate(mod) if any(True for _ in m.parameters())]
                    if int(p) >= len(links):
                        raise KeyError('no link %s under %s' % (p, '.'.join(out)))
                    p = str(links[int(p)])
                out.append(p)
                mod = getattr(mod, p) if not p.isdigit() else mod[int(p)]
            return out
        with np.load(path) as z:
            for key in z.files:
                parts = key.split('/')
                if parts[-1] == 'N' or (parts[0] == 'fc' and self.fc is None):
                    continue
                try:
                    name = '.'.join(resolve(parts[:-1]) + [rename[parts[-1]]])
                except (AttributeError, IndexError, KeyError):
                    raise KeyError('unexpected entry %s in %s' % (key, path))
                if name not in own:
                    raise KeyError('unexpected entry %s in %s' % (key, path))
                own[name].copy_(torch.from_numpy(z[key]).reshape(own[name].shape))
        return self


def flops_per_image(name, H, W):
    """2*MACs of the convolutions (fc excluded) — BASELINE.md section 3."""
    per_full = {'drn_c_26': 1418.6e9, 'drn_d_22': 1089.5e9}[name]
    return per_full * (H * W) / (1024.0 * 2048.0)


def create_drn(name='drn_c_26', weights=None, device='cuda', dtype=torch.float32, fold_bn=True,
               bn_eps=CHAINER_BN_EPS, seed=0):
    """Factory behind create_model(): random init (seeded) unless a weight file is given."""
    gen_state = torch.random.get_rng_state()
    torch.manual_seed(seed)
    model = DRN(name, bn_eps=bn_eps)
    torch.random.set_rng_state(gen_state)
    if weights:
        if weights.endswith('.npz'):
            model.load_chainer_npz(weights)
        else:
            model.load_pth(weights)
    return model.prepare(device, dtype, fold_bn)
