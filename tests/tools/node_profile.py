"""Where the batched node sequence spends its time (host wall clock with device syncs between the stages)."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import synthetic_inputs as SI  # noqa: E402
import leafgrasp_amd as L  # noqa: E402

H, W, nb = 1080, 1920, 128
dev = torch.device("cuda:0")
scenes = [SI.synthetic_scene(H, W, seed=100 + s) for s in range(4)]
lab = torch.from_numpy(np.stack([scenes[i % 4][0] for i in range(nb)]).astype(np.int16)).to(dev)
dep = torch.from_numpy(np.stack([scenes[i % 4][1] for i in range(nb)])).to(dev)
hz = L.LeafGraspHarness(H, W, dev, load_model=False)
hz.camera_info_callback(np.asarray(scenes[0][2]).reshape(-1))
hz.grasp_selector.set_cnn_state_dict(SI.cnn_closed_form_params(seed=0))
hz.process_batch_device(lab, dep)
torch.cuda.synchronize()


def t(fn, n=3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, r


ms_all, _ = t(lambda: hz.process_batch_device(lab, dep))
ms_stats, st = t(lambda: hz.leaf_scorer.leaf_statistics_batch(lab, dep))
ms_leaf, ids = t(lambda: hz.leaf_scorer.select_optimal_leaves_batch(lab, dep))
idt = torch.tensor([i if i is not None else 0 for i in ids], dtype=lab.dtype, device=dev).reshape(-1, 1, 1)
ms_mask, opt = t(lambda: lab == idt)
ms_grasp, _ = t(lambda: hz.grasp_selector.select_grasp_points_batch(opt, dep))
print(f"whole {ms_all:.2f} ms | leaf stats (device + copy-back) {ms_stats:.2f} | leaf selection total {ms_leaf:.2f} | "
      f"mask compare {ms_mask:.2f} | grasp batch {ms_grasp:.2f}   -> {nb / ms_all * 1e3:.0f} frames/s")
pr = cProfile.Profile()
pr.enable()
hz.process_batch_device(lab, dep)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
for B2 in (128, 256):
    lab2 = lab.repeat(B2 // nb, 1, 1) if B2 > nb else lab
    dep2 = dep.repeat(B2 // nb, 1, 1) if B2 > nb else dep
    for ch in (1, 2, 4):
        hz.process_batch_device(lab2, dep2, chunks=ch)
        ms, _ = t(lambda: hz.process_batch_device(lab2, dep2, chunks=ch), n=4)
        print(f"B={B2} chunks={ch}: {ms:.2f} ms -> {B2 / ms * 1e3:.0f} frames/s", flush=True)
