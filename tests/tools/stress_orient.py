#!/usr/bin/env python3
"""Randomised comparison of the device orientation kernel (lg_orient_kernel) with the host analysis it replaces (lg_contour.cpp,
LG_HOST_ORIENT=1): N random masks -- blobs from thresholded smoothed noise, unions of rotated ellipses and bars, salt noise, thin
diagonal structures -- at random sizes; all five outputs must agree as float32 (theta within 2e-7).
usage (GPU box): python tests/tools/stress_orient.py [N=600] [seed=0]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 600
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
os.environ["LG_HOST_ORIENT"] = "1"
import leafgrasp_amd as L  # noqa: E402

host = L.GraspPointSelector("cuda:0", load_model=False)
del os.environ["LG_HOST_ORIENT"]
dev = L.GraspPointSelector("cuda:0", load_model=False)
rng = np.random.default_rng(seed)


def smooth(a, k):
    for _ in range(k):
        a = (a + np.roll(a, 1, 0) + np.roll(a, -1, 0) + np.roll(a, 1, 1) + np.roll(a, -1, 1)) / 5.0
    return a


bad = found = 0
for case in range(N):
    H, W = int(rng.integers(8, 420)), int(rng.integers(8, 640))
    kind = case % 5
    yy, xx = np.mgrid[0:H, 0:W]
    if kind == 0:
        m = smooth(rng.random((H, W)), int(rng.integers(1, 6))) > rng.uniform(0.48, 0.56)
    elif kind == 1:
        m = np.zeros((H, W), bool)
        for _ in range(int(rng.integers(1, 6))):
            cx, cy, a, b, t = rng.uniform(0, W), rng.uniform(0, H), rng.uniform(2, W / 3), rng.uniform(2, H / 3), rng.uniform(0, np.pi)
            u = (xx - cx) * np.cos(t) + (yy - cy) * np.sin(t)
            v = -(xx - cx) * np.sin(t) + (yy - cy) * np.cos(t)
            m |= (u / a) ** 2 + (v / b) ** 2 <= 1
    elif kind == 2:
        m = rng.random((H, W)) > rng.uniform(0.5, 0.995)
    elif kind == 3:
        m = np.zeros((H, W), bool)
        for _ in range(int(rng.integers(1, 5))):
            t, w = rng.uniform(0, np.pi), rng.uniform(0.5, 3)
            d = (xx - rng.uniform(0, W)) * np.sin(t) - (yy - rng.uniform(0, H)) * np.cos(t)
            m |= np.abs(d) < w
    else:
        m = (smooth(rng.random((H, W)), 3) > 0.5) & (((xx // int(rng.integers(2, 9))) + (yy // int(rng.integers(2, 9)))) % 2 == 0)
    m = m.astype(np.uint8)
    a, b = host.estimate_leaf_orientation(m), dev.estimate_leaf_orientation(m)
    if (a[0] is None) != (b[0] is None):
        bad += 1
        print("MISMATCH found", case, H, W, kind, a, b, flush=True)
        continue
    if a[0] is None:
        continue
    found += 1
    if abs(a[0] - b[0]) > 2e-7 or a[1:] != b[1:]:
        bad += 1
        print("MISMATCH", case, H, W, kind, a, b, flush=True)
print(f"cases={N} with_contour={found} mismatches={bad}", flush=True)
sys.exit(1 if bad else 0)
