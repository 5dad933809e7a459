#!/usr/bin/env python3
"""Development aid: digests of the float32 stem's outputs (DRN-D: layer 1's map; DRN-C: conv1's map and layer 1's first
convolution) and of map 7 on seeded inputs, so that two builds can be compared bit for bit from two processes:
    SPA_LIB_PATH=$PWD/ab/libspalign_old.so python tools/stem_digest.py ; python tools/stem_digest.py"""
import hashlib, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
drn = importlib.import_module('superpixel-align_amd.drn')
synth = importlib.import_module('superpixel-align_amd.synth')
for arch, shape in (('drn_d_22', (2, 250, 517)), ('drn_c_26', (1, 128, 320)), ('drn_d_22', (3, 1024, 2048))):
    m = drn.create_drn(arch, device='cuda', dtype=torch.float32, seed=1)
    x = synth.synth_batch(list(range(shape[0])), shape[1], shape[2])
    _, maps = m.batch_predict(x, need=[0, 1, 7])
    torch.cuda.synchronize()
    print(arch, shape, ' '.join('%d:%s' % (i, hashlib.sha256(maps[i].float().cpu().numpy().tobytes()).hexdigest()[:12]) for i in sorted(maps) if maps[i] is not None) if isinstance(maps, dict)
          else ' '.join('%d:%s' % (i, hashlib.sha256(v.float().cpu().numpy().tobytes()).hexdigest()[:12]) for i, v in enumerate(maps) if v is not None))
