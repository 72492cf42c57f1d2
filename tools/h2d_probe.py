#!/usr/bin/env python3
"""Why did the overlapped host-to-device leg of bench.py stop overlapping (BENCH_r02 5113 vs BENCH_r03 4273 frames/s)?
One variant per process: python3 tools/h2d_probe.py <variant>
  base      selector, then the copy stream; scoring on the thread's current (default) stream   (= bench.py round 3)
  cs_first  the copy stream is created BEFORE any selector exists
  own       scoring on a torch stream of its own instead of the default stream
  extra     three more selectors are created, used once and destroyed before the leg (what the `configs` leg leaves behind)
  extra_cs_first / extra_own: combinations"""
import gc
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import leafgrasp_amd as L  # noqa: E402
import synthetic_inputs as SI  # noqa: E402

variant = sys.argv[1] if len(sys.argv) > 1 else "base"
nh, H, W = 64, 1080, 1920
dev = torch.device("cuda:0")
scenes = [SI.synthetic_scene(H, W, 100 + i) for i in range(4)]
P = scenes[0][2]
params = SI.cnn_closed_form_params(0)
masks = np.stack([(scenes[i % 4][0] == 1) for i in range(nh)]).astype(np.uint8)
depths = np.stack([scenes[i % 4][1] for i in range(nh)])
h_mask, h_depth = torch.from_numpy(masks).pin_memory(), torch.from_numpy(depths).pin_memory()
cs = torch.cuda.Stream(device=dev) if "cs_first" in variant else None


def new_sel():
    s = L.GraspPointSelector(dev, load_model=False)
    s.set_camera_params(P)
    s.set_cnn_state_dict(params)
    return s


if "extra" in variant:
    dm, dd = torch.from_numpy(masks[:8]).to(dev), torch.from_numpy(depths[:8]).to(dev)
    for _ in range(3):
        x = new_sel()
        x.select_grasp_points_batch(dm.view(torch.bool), dd)
        x = None
    gc.collect()
sel = new_sel()
if cs is None:
    cs = torch.cuda.Stream(device=dev)
ss = torch.cuda.Stream(device=dev) if "own" in variant else torch.cuda.current_stream(dev)
bufs = [(torch.empty_like(h_mask, device=dev), torch.empty_like(h_depth, device=dev)) for _ in range(2)]
evs = [torch.cuda.Event(), torch.cuda.Event()]


def score(k):
    with torch.cuda.stream(ss):
        return sel.select_grasp_points_batch(bufs[k][0].view(torch.bool), bufs[k][1])


def issue_copy(k, stream):
    with torch.cuda.stream(stream):
        bufs[k][0].copy_(h_mask, non_blocking=True)
        bufs[k][1].copy_(h_depth, non_blocking=True)
        evs[k].record(stream)


issue_copy(0, ss); issue_copy(1, ss)
torch.cuda.synchronize(dev)
for _ in range(2):
    score(0)
steps = 6
torch.cuda.synchronize(dev)
t0 = time.perf_counter()
for _ in range(steps):          # serial: copy and scoring on one stream
    issue_copy(0, ss)
    score(0)
torch.cuda.synchronize(dev)
serial = nh * steps / (time.perf_counter() - t0)
t0 = time.perf_counter()
for _ in range(steps):
    issue_copy(0, ss)
torch.cuda.synchronize(dev)
copy_only = nh * steps / (time.perf_counter() - t0)
t0 = time.perf_counter()
for _ in range(steps):
    score(0)
torch.cuda.synchronize(dev)
score_only = nh * steps / (time.perf_counter() - t0)
n_o = 2 * steps
torch.cuda.synchronize(dev)
t0 = time.perf_counter()
issue_copy(0, cs)
for i in range(n_o):
    evs[i % 2].synchronize()
    if i + 1 < n_o:
        issue_copy((i + 1) % 2, cs)
    score(i % 2)
torch.cuda.synchronize(dev)
over = nh * n_o / (time.perf_counter() - t0)
print({"variant": variant, "serial": round(serial, 1), "overlapped": round(over, 1), "copy_only": round(copy_only, 1),
       "score_only": round(score_only, 1), "overlapped_over_serial": round(over / serial, 3)})
