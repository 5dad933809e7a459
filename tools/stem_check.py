#!/usr/bin/env python3
"""Development aid: the fused DRN-D stem kernel against the MIOpen path (max error, time)."""
import importlib
import os
import sys
import time

os.environ.setdefault('MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD', '0')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

drn = importlib.import_module('superpixel-align_amd.drn')
torch.backends.cudnn.benchmark = True
B, H, W = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (4, 1024, 2048)))
model = drn.create_drn('drn_d_22', None, device='cuda', dtype=torch.float32)
x = torch.rand(B, 3, H, W, device='cuda') * 255
for fused in (False, True):
    model.use_fused_stem = fused
    _, maps = model.batch_predict(x)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(3):
        _, maps = model.batch_predict(x)
    torch.cuda.synchronize()
    print('fused stem %s: %.2f ms per forward' % (fused, (time.time() - t0) / 3 * 1e3))
    if fused:
        for i in (0, 1, 7):
            d = (maps[i].float() - ref[i].float()).abs().max().item()
            print('  map %d: max abs diff %.3e, scale %.3e -> %.2e of scale' % (i, d, ref[i].abs().max().item(),
                                                                               d / ref[i].abs().max().item()))
    else:
        ref = [m.clone() for m in maps]
