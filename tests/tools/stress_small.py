#!/usr/bin/env python3
"""Degenerate-size / degenerate-mask GPU-vs-oracle sweep of lg_score_maps: frames of 8..80 x 8..300 pixels (smaller than a
plane tile or a sweep wave), single pixels, noise, rectangles, full-minus-one-pixel masks, bottom-half noise: distance map,
the two sdf maxima (windowed sweeps + frame-border norm), theta, every plane and the validity mask.
Usage: python tests/tools/stress_small.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import leafgrasp_amd as L
from oracle import lg_oracle as O
sel = L.GraspPointSelector("cuda:0", load_model=False)
rng = np.random.default_rng(5)
bad = 0
N = 150
for case in range(N):
    H = int(rng.integers(8, 80)); W = int(rng.integers(8, 300))
    kind = case % 5
    m = np.zeros((H, W), np.uint8)
    if kind == 0: m[rng.integers(0, H), rng.integers(0, W)] = 1
    elif kind == 1: m = (rng.random((H, W)) > 0.5).astype(np.uint8)
    elif kind == 2:
        y0, x0 = rng.integers(0, H), rng.integers(0, W); m[y0:y0 + rng.integers(1, H + 1), x0:x0 + rng.integers(1, W + 1)] = 1
    elif kind == 3: m[:] = 1; m[rng.integers(0, H), rng.integers(0, W)] = 0
    else: m[H // 2:, :] = (rng.random((H - H // 2, W)) > 0.2)
    depth = (0.4 + 0.1 * rng.random((H, W))).astype(np.float32)
    P = np.array([[300.0, 0, W / 2.1, 0], [0, 300.0, H / 1.9, 0], [0, 0, 1, 0]])
    sel.set_camera_params(P)
    maps, valid, theta = sel.score_maps(torch.from_numpy(m).cuda(), torch.from_numpy(depth).cuda())
    ref = O.RefGraspPointSelector(); ref.set_camera_params(P)
    sc = ref._calculate_all_scores(m, depth)
    msgs = []
    if not np.array_equal(maps["distance_map"].cpu().numpy(), sc["distance_map"]): msgs.append("dm")
    mi, mo, win = sel.dt_maxima(0)
    din, dout = O.distance_transform(m, 5), O.distance_transform(1 - m, 5)
    if mi != din.max() or mo != dout.max(): msgs.append(f"max {mi},{mo} vs {din.max()},{dout.max()} win {win}")
    if (theta is None) == (ref._last_angle is None) and (theta is None or abs(theta - ref._last_angle) < 1e-6):
        for k, v in sc.items():
            g = maps[k].cpu().numpy()
            if not np.allclose(g, v, rtol=1e-4, atol=1e-6): msgs.append(f"{k} {np.max(np.abs(g - v)):.3g}")
        if not np.array_equal(valid.cpu().numpy().astype(bool), ref._get_valid_regions(m, sc)): msgs.append("valid")
    else:
        msgs.append(f"theta {theta} vs {ref._last_angle}")
    if msgs:
        bad += 1; print(case, H, W, kind, msgs)
print(f"{N - bad}/{N} clean")
