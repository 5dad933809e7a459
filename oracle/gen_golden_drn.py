#!/usr/bin/env python3
"""ORACLE — TEST INFRASTRUCTURE ONLY.  DRN golden vectors.

Run in the BUILD container only:  python oracle/gen_golden_drn.py
Imports the reference's own PyTorch DRN definition (/root/reference/models/drn_pytorch.py —
the architecture source of truth that models/convert_pth2ch.py converts to Chainer), fills it
with weights that are a pure function of (parameter name, shape) — so the test can rebuild the
same weights without shipping 80 MB — patches BatchNorm eps to Chainer's 2e-5, runs one small
input and stores the eight maps of the CHAINER convention (models/drn.py:238-273: arch D does not
export layer0) as float32 digests + the full last map.
"""
import os
import sys
import zlib

import numpy as np
import torch

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, '/root/reference/models')


def det_fill(model):
    """weights/statistics as a function of the tensor's name: same in generator and test."""
    with torch.no_grad():
        for name, t in list(model.named_parameters()) + list(model.named_buffers()):
            if name.endswith('num_batches_tracked'):
                continue
            g = torch.Generator().manual_seed(zlib.crc32(name.encode()) & 0x7fffffff)
            if name.endswith('running_var'):
                t.copy_(torch.rand(t.shape, generator=g) * 0.5 + 0.75)
            elif name.endswith('running_mean'):
                t.copy_(torch.randn(t.shape, generator=g) * 0.1)
            elif t.ndim == 4:
                fan = t.shape[1] * t.shape[2] * t.shape[3]
                t.copy_(torch.randn(t.shape, generator=g) * (2.0 / fan) ** 0.5)
            elif name.endswith('weight'):
                t.copy_(torch.rand(t.shape, generator=g) * 0.5 + 0.75)
            else:
                t.copy_(torch.randn(t.shape, generator=g) * 0.1)


if __name__ == '__main__':
    import drn_pytorch as ref
    out = {}
    x = torch.from_numpy(np.random.RandomState(0).uniform(0, 255, (2, 3, 48, 64)).astype(np.float32))
    mean = torch.tensor([0.485, 0.456, 0.406], dtype=torch.float64).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225], dtype=torch.float64).view(1, 3, 1, 1)
    xn = x / 255.0
    xn = (xn.double() - mean).float()
    xn = (xn.double() / std).float()
    for name, ctor in (('drn_c_26', ref.drn_c_26), ('drn_d_22', ref.drn_d_22)):
        m = ctor(pretrained=False, out_map=True, out_middle=True)
        det_fill(m)
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.eps = 2e-5
        m.eval()
        with torch.no_grad():
            _, maps = m(xn)
        if name == 'drn_d_22':
            maps = maps[1:]          # Chainer convention: layer0 is not exported
        assert len(maps) == 8
        out[name + '_shapes'] = np.array([list(t.shape) for t in maps], np.int64)
        out[name + '_means'] = np.array([float(t.double().mean()) for t in maps])
        out[name + '_abs'] = np.array([float(t.double().abs().mean()) for t in maps])
        out[name + '_map7'] = maps[7].numpy()
        out[name + '_nparams'] = np.array(sum(p.numel() for n, p in m.named_parameters() if not n.startswith('fc.')))
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'drn_maps.npz'), x=x.numpy(), **out)
    print({k: (v.shape if hasattr(v, 'shape') else v) for k, v in out.items()})
