#!/usr/bin/env python3
"""Batched leaf selection only (lg_leaf_select_batch, B x 1080p label + depth frames resident in HBM): wall time per call and,
under rocprofv3 --kernel-trace --stats, the per-kernel times of the stage.  usage: python3 tools/leaf_batch.py [B] [calls]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import leafgrasp_amd as L  # noqa: E402
import synthetic_inputs as SI  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 10
H, W = 1080, 1920
scenes = [SI.synthetic_scene(H, W, 100 + i) for i in range(8)]
lab = torch.from_numpy(np.stack([scenes[i % 8][0] for i in range(B)]).astype(np.int16)).cuda()
dep = torch.from_numpy(np.stack([scenes[i % 8][1] for i in range(B)])).cuda()
ols = L.OptimalLeafSelector("cuda:0")
ols.set_camera_params(scenes[0][2])
for _ in range(3):
    ids = ols.select_optimal_leaves_batch(lab, dep)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(calls):
    ids = ols.select_optimal_leaves_batch(lab, dep)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / calls
print({"B": B, "ms_per_call": round(1e3 * dt, 4), "frames_per_s": round(B / dt, 1),
       "frac_of_6B_per_px_bound": round(B * H * W * 6 / dt / 8e12, 4), "ids": ids[:8]})
