#!/usr/bin/env python3
"""ORACLE — TEST INFRASTRUCTURE ONLY.  Key set and tensor shapes of the reference's PyTorch checkpoints.

Run in the BUILD container only:  python oracle/gen_golden_drn_keys.py
Instantiates drn_d_22 and drn_c_26 of /root/reference/models/drn_pytorch.py (:262-284, the modules
`drn_d_22-4bd2f8ea.pth` / `drn_c_26-ddedf421.pth` load into) and stores every state_dict key with its shape —
without `num_batches_tracked`, which the 2017 checkpoints do not hold — in tests/golden/drn_state_dict_keys.json.
A test writes a .pth with exactly this key set and loads it through DRN.load_pth."""
import json
import os
import sys

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, '/root/reference/models')
import drn_pytorch as ref          # noqa: E402

out = {}
for name in ('drn_d_22', 'drn_c_26'):
    m = getattr(ref, name)(pretrained=False)
    out[name] = {k: list(v.shape) for k, v in m.state_dict().items() if not k.endswith('num_batches_tracked')}
path = os.path.join(ROOT, 'tests', 'golden', 'drn_state_dict_keys.json')
with open(path, 'w') as f:
    json.dump(out, f, indent=0, sort_keys=True)
print(path, {k: len(v) for k, v in out.items()})
