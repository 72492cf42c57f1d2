#!/usr/bin/env python3
"""lg_orient_kernel on masks with more than one connected component: a 1080p leaf alone, with 40 specks beside it,
and two leaves of similar size, 64 frames per call.  Prints the kernel's time per call (HIP events inside the library).
usage (GPU box): python tools/orient_speckle_time.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import leafgrasp_amd as L  # noqa: E402
import synthetic_inputs as SI  # noqa: E402
from leafgrasp_amd._lib import lib  # noqa: E402

H, W, B = 1080, 1920, 64
labels, depth, P = SI.synthetic_scene(H, W, seed=101)
ids, counts = np.unique(labels[labels > 0], return_counts=True)
order = ids[np.argsort(-counts)]
leaf = labels == order[0]
rng = np.random.default_rng(0)
speck = leaf.copy()
for _ in range(40):
    y, x = rng.integers(5, H - 5), rng.integers(5, W - 5)
    speck[y:y + rng.integers(1, 4), x:x + rng.integers(1, 4)] = True
two = leaf | (labels == order[1])
sel = L.GraspPointSelector(torch.device("cuda:0"), load_model=False)
sel.set_camera_params(P)
d = torch.from_numpy(np.repeat(depth[None], B, 0)).cuda()
for name, m in (("one leaf", leaf), ("leaf + 40 specks", speck), ("two leaves", two)):
    mt = torch.from_numpy(np.repeat(m[None], B, 0)).cuda()
    sel.score_maps(mt, d)
    lib.lg_profile_enable(sel._h, 1)
    for _ in range(3):
        _, _, th = sel.score_maps(mt, d)
    n, ms = C.c_int(0), C.c_double(0)
    lib.lg_profile_read(sel._h, b"orient", C.byref(n), C.byref(ms))
    lib.lg_profile_enable(sel._h, 0)
    print(f"{name:18s}: lg_orient_kernel {ms.value / max(1, n.value):.3f} ms per {B} frames, theta {th[0]:.6f}", flush=True)
