#!/usr/bin/env python3
"""ms per GraspPointCNN forward (HIP events, 10 repetitions) for the library in LG_LIB_PATH and the LG_CNN_* switches in the
environment (timing-ablation variants of tools/build_variants.sh produce wrong logits by design).
usage: python tools/cnn_time.py [patches]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import leafgrasp_amd as L  # noqa: E402
import synthetic_inputs as SI  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5120
dev = torch.device("cuda", 0)
sel = L.GraspPointSelector(dev, load_model=False)
sel.set_cnn_state_dict(SI.cnn_closed_form_params(seed=0))
x = torch.from_numpy(SI.synthetic_patches(64, seed=1)).to(dev).repeat((n + 63) // 64, 1, 1, 1)[:n].contiguous()
for _ in range(3):
    sel.cnn_forward(x)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    sel.cnn_forward(x)
e1.record()
torch.cuda.synchronize()
print(f"{os.path.basename(os.environ.get('LG_LIB_PATH', 'liblgrasp.so')):28s} patches={n} ms={e0.elapsed_time(e1) / 10:.3f}", flush=True)
