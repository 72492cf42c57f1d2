#!/usr/bin/env python3
"""End-to-end sweep: GPU select_grasp_point / select_optimal_leaf vs the float64 CPU oracle over many seeds.
Reports how often the float32 planes reorder near-tied candidates.  Usage: python tests/tools/stress_e2e.py [n]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import leafgrasp_amd as L  # noqa: E402
from oracle import lg_oracle as O  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
sel = L.GraspPointSelector("cuda:0", load_model=False)
ols = L.OptimalLeafSelector("cuda:0")
params = O.cnn_closed_form_params(0)
sel.set_cnn_state_dict(params)
cand_diff = grasp_diff = leaf_diff = 0
for i in range(n):
    H, W = [(480, 640), (540, 960), (600, 800)][i % 3]
    labels, depth, P = O.synthetic_scene(H, W, 3000 + i)
    sel.set_camera_params(P)
    ols.set_camera_params(P)
    rl = O.RefOptimalLeafSelector()
    rl.set_camera_params(P)
    lid_o = rl.select_optimal_leaf(labels, depth)
    lid_g = ols.select_optimal_leaf(torch.from_numpy(labels).cuda(), torch.from_numpy(depth).cuda())
    if lid_o != lid_g:
        leaf_diff += 1
        print(f"seed {i}: leaf {lid_g} vs oracle {lid_o}")
    lid = lid_o if lid_o is not None else 1
    mask = (labels == lid).astype(np.uint8)
    ref = O.RefGraspPointSelector(cnn=lambda x: O.cnn_forward(params, x))
    ref.set_camera_params(P)
    exp, dbg = ref.select_grasp_point(mask, depth, return_debug=True)
    got = sel.select_grasp_point(torch.from_numpy(mask.astype(bool)).cuda(), torch.from_numpy(depth).cuda(), None)
    maps, valid, _ = sel.score_maps(torch.from_numpy(mask).cuda(), torch.from_numpy(depth).cuda())
    cands = sel._get_candidate_points(maps["traditional_score"], valid, 20, 10)
    if cands != dbg.get("candidates"):
        cand_diff += 1
        oc = dbg.get("candidates") or []
        k = next((j for j, (a, b) in enumerate(zip(cands, oc)) if a != b), min(len(cands), len(oc)))
        tr = dbg["scores"]["traditional_score"]
        print(f"seed {i}: candidates diverge at rank {k}: gpu {cands[k] if k < len(cands) else None} "
              f"oracle {oc[k] if k < len(oc) else None}; oracle scores there "
              f"{[float(tr[y, x]) for (x, y) in (cands[k:k + 1] + oc[k:k + 1])]}")
    if got[0] != exp[0]:
        grasp_diff += 1
        print(f"seed {i}: grasp {got[0]} vs oracle {exp[0]}")
print(f"n={n}: leaf mismatches {leaf_diff}, candidate-list mismatches {cand_diff}, grasp-point mismatches {grasp_diff}")
