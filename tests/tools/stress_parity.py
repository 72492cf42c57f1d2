#!/usr/bin/env python3
"""Randomised GPU-vs-oracle parity sweep (many seeds / odd sizes / multi-component masks).
Usage: python tests/tools/stress_parity.py [n_cases]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import leafgrasp_amd as L  # noqa: E402
from oracle import lg_oracle as O  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
sel = L.GraspPointSelector("cuda:0", load_model=False)
rng = np.random.default_rng(2024)
bad = 0
for case in range(n_cases):
    H = int(rng.integers(40, 420))
    W = int(rng.integers(40, 560))
    if case % 5 == 0:
        W = (W // 64) * 64 + 64          # exact multiples of the tile width
    if case % 7 == 0:
        W = W // 4 * 4 + int(rng.integers(1, 4))  # widths that are not multiples of 4
    labels, depth, P = O.synthetic_scene(H, W, 1000 + case)
    kind = case % 4
    if kind == 0:
        mask = (labels == 1)
    elif kind == 1:
        mask = (labels >= 1)                       # several blobs, holes possible, border contact
    elif kind == 2:
        mask = (labels == 1) | (rng.random((H, W)) > 0.995)   # specks: many tiny components
    else:
        mask = (labels == 2) | (labels == 3)
    mask = mask.astype(np.uint8)
    sel.set_camera_params(P)
    gs = int(rng.choice([1, 3, 7])) if case % 3 == 1 else 5      # the caller's ImageProcessor decides the smoothing (round 3)
    maps, valid, theta = sel.score_maps(torch.from_numpy(mask).cuda(), torch.from_numpy(depth).cuda(), L.ImageProcessor(H, W, 21, gs))
    ref = O.RefGraspPointSelector(gaussian_size=gs)
    ref.set_camera_params(P)
    sc = ref._calculate_all_scores(mask, depth)
    msgs = []
    if not np.array_equal(maps["distance_map"].cpu().numpy(), sc["distance_map"]):
        msgs.append("distance_map")
    if not np.array_equal(maps["stem_penalty"].cpu().numpy(), sc["stem_penalty"]):
        msgs.append("stem")
    if (theta is None) != (ref._last_angle is None) or (theta is not None and abs(theta - ref._last_angle) > 1e-6):
        msgs.append(f"theta {theta} vs {ref._last_angle}")
    else:
        for k, v in sc.items():
            g = maps[k].cpu().numpy()
            if not np.allclose(g, v, rtol=1e-4, atol=1e-6):
                msgs.append(f"{k} maxabs {np.max(np.abs(g - v)):.3g}")
        if not np.array_equal(valid.cpu().numpy().astype(bool), ref._get_valid_regions(mask, sc)):
            msgs.append("valid")
        # top-k / spacing combinations: the usual window takes the kernel's short path, the others its general form
        for kk, md in ((20, 10), ([5, 64, 20, 33][case % 4], [0, 3, 25, 40, 17][case % 5])):
            cg = sel._get_candidate_points(maps["traditional_score"], valid, kk, md)
            co = ref._get_candidate_points(maps["traditional_score"].cpu().numpy(), valid.cpu().numpy().astype(bool), kk, md)
            if cg != co:
                msgs.append(f"candidates(on the GPU planes, k={kk}, min_distance={md})")
    if msgs:
        bad += 1
        print(f"case {case}: H={H} W={W} kind={kind} gaussian={gs}: " + "; ".join(msgs))
print(f"{n_cases - bad}/{n_cases} cases clean")
sys.exit(1 if bad else 0)
