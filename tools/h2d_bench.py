#!/usr/bin/env python3
"""PCIe-inclusive rate of the grasp-scoring path (SURVEY 8d "report separately with H2D ... included"; never bench.py's
`value`): every step first copies its B frames (depth f32 + mask u8 = 5 B/px) from pinned host memory, then runs
lg_select_grasp.  Also prints the copy-only rate.  Usage: python tools/h2d_bench.py [B]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import leafgrasp_amd as L  # noqa: E402
import synthetic_inputs as O  # seeded inputs only (closed-form weights / scenes / patches)

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
H, W = 1080, 1920
scenes = [O.synthetic_scene(H, W, 100 + i) for i in range(4)]
P = scenes[0][2]
m_h = torch.from_numpy(np.stack([(scenes[i % 4][0] == 1) for i in range(B)]).astype(np.uint8)).pin_memory()
d_h = torch.from_numpy(np.stack([scenes[i % 4][1] for i in range(B)])).pin_memory()
sel = L.GraspPointSelector("cuda:0", load_model=False)
sel.set_camera_params(P)
sel.set_cnn_state_dict(O.cnn_closed_form_params(0))
m_d, d_d = torch.empty_like(m_h, device="cuda"), torch.empty_like(d_h, device="cuda")


def step(copy=True, run=True):
    if copy:
        m_d.copy_(m_h, non_blocking=True)
        d_d.copy_(d_h, non_blocking=True)
    if run:
        sel.select_grasp_points_batch(m_d, d_d)


for _ in range(3):
    step()
torch.cuda.synchronize()
res = {}
for name, kw in (("copy_only", dict(run=False)), ("copy_then_score", dict()), ("score_only", dict(copy=False))):
    t0 = time.perf_counter()
    for _ in range(8):
        step(**kw)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 8
    res[name + "_frames_per_s"] = round(B / dt, 1)
    if name == "copy_only":
        res["h2d_GBps"] = round(B * H * W * 5 / dt / 1e9, 1)
print(res)
