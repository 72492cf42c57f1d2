#!/usr/bin/env python3
"""Profiling driver: `reps` forward passes of GraspPointCNN on `patches` seeded patches through the C-ABI, variant chosen by
the LG_CNN_* switches in the environment (default Winograd F(4x4,3x3)).  usage: python tools/cnn_run.py [patches] [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import leafgrasp_amd as L  # noqa: E402
import synthetic_inputs as SI  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5120
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda", 0)
sel = L.GraspPointSelector(dev, load_model=False)
sel.set_cnn_state_dict(SI.cnn_closed_form_params(seed=0))
x = torch.from_numpy(SI.synthetic_patches(64, seed=1)).to(dev).repeat((n + 63) // 64, 1, 1, 1)[:n].contiguous()
for _ in range(reps):
    out = sel.cnn_forward(x)
torch.cuda.synchronize()
print("ok", float(out[0]))
