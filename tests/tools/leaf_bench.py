#!/usr/bin/env python3
"""Per-frame latency of the leaf-selection stage (lg_leaf_stats + host Pareto) and of the whole ROS-free node
sequence on one MI355X, next to the restated CPU path (oracle).  Usage: python tests/tools/leaf_bench.py [H W]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import leafgrasp_amd as L  # noqa: E402
from oracle import lg_oracle as O  # noqa: E402

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1080, 1920)
labels, depth, P = O.synthetic_scene(H, W, 4)
lab_d, dep_d = torch.from_numpy(labels).cuda(), torch.from_numpy(depth).cuda()
ols = L.OptimalLeafSelector("cuda:0")
ols.set_camera_params(P)
for _ in range(3):
    ols.select_optimal_leaf(lab_d, dep_d)
torch.cuda.synchronize()
t0 = time.perf_counter()
N = 20
for _ in range(N):
    lid = ols.select_optimal_leaf(lab_d, dep_d)
torch.cuda.synchronize()
gpu_leaf = (time.perf_counter() - t0) / N
hz = L.LeafGraspHarness(H, W, "cuda:0", load_model=False)
hz.camera_info_callback(P.reshape(-1))
hz.grasp_selector.set_cnn_state_dict(O.cnn_closed_form_params(0))
m16, d32 = labels.astype(np.uint16).reshape(-1), depth.reshape(-1)
for _ in range(3):
    hz.process(m16, d32)
t0 = time.perf_counter()
for _ in range(N):
    csv = hz.process(m16, d32)
node = (time.perf_counter() - t0) / N
# batched node sequence: leaf selection + grasp selection for B frames per call, tensors resident on the device
B = 128
scenes = [O.synthetic_scene(H, W, 100 + i) for i in range(4)]
lab_b = torch.from_numpy(np.stack([scenes[i % 4][0] for i in range(B)]).astype(np.int16)).cuda()
dep_b = torch.from_numpy(np.stack([scenes[i % 4][1] for i in range(B)])).cuda()
for _ in range(2):
    ols.select_optimal_leaves_batch(lab_b, dep_b)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    ids_b = ols.select_optimal_leaves_batch(lab_b, dep_b)
torch.cuda.synchronize()
leaf_batch = (time.perf_counter() - t0) / 5
for _ in range(2):
    hz.process_batch_device(lab_b, dep_b)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    csv_b = hz.process_batch_device(lab_b, dep_b)
torch.cuda.synchronize()
node_batch = (time.perf_counter() - t0) / 5
ref = O.RefOptimalLeafSelector()
ref.set_camera_params(P)
t0 = time.perf_counter()
rid = ref.select_optimal_leaf(labels, depth)
cpu_leaf = time.perf_counter() - t0
px = H * W
print({"HxW": (H, W), "leaf_id": lid, "oracle_leaf_id": rid,
       "gpu_select_optimal_leaf_ms": round(1e3 * gpu_leaf, 3),
       "gpu_leaf_GBps_(6B/px x 8 passes)": round(px * 6 * 8 / gpu_leaf / 1e9, 1),
       "cpu_oracle_select_optimal_leaf_ms": round(1e3 * cpu_leaf, 1),
       "node_harness_wire_to_csv_ms (H2D + leaf + grasp incl. CNN)": round(1e3 * node, 3), "csv": csv,
       f"batched_leaf_selection_B{B}_frames_per_s": round(B / leaf_batch, 1),
       f"batched_node_sequence_B{B}_frames_per_s (leaf + grasp incl. CNN, device-resident)": round(B / node_batch, 1),
       "batched_csv_found": sum(c is not None for c in csv_b)})
