"""Per-call latency of the reference-shaped single-frame API at 1080p (tensors resident on the device)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import synthetic_inputs as SI  # noqa: E402
import leafgrasp_amd as L  # noqa: E402

H, W = 1080, 1920
dev = torch.device("cuda:0")
labels, depth, P = SI.synthetic_scene(H, W, seed=100)
lab = torch.from_numpy(labels.astype(np.int16)).to(dev)
dep = torch.from_numpy(depth).to(dev)
sel = L.GraspPointSelector(dev, load_model=False)
sel.set_camera_params(P)
sel.set_cnn_state_dict(SI.cnn_closed_form_params(seed=0))
ols = L.OptimalLeafSelector(dev)
ols.set_camera_params(P)
ip = L.ImageProcessor(H, W, 21, 5)


def t(fn, n=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, r


ms_leaf, lid = t(lambda: ols.select_optimal_leaf(lab, dep))
mask = lab == lid
ms_grasp, res = t(lambda: sel.select_grasp_point(mask, dep, ip))
sel.clear_cnn()
ms_cv, _ = t(lambda: sel.select_grasp_point(mask, dep, ip))
print(f"select_optimal_leaf {ms_leaf:.3f} ms | select_grasp_point incl. CNN {ms_grasp:.3f} ms | CV only {ms_cv:.3f} ms | "
      f"leaf id {lid} grasp {res[0]}")
