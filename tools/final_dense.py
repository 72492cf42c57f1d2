#!/usr/bin/env python3
"""lg_final_kernel with EVERY tile on the stencil path: `reps` score_maps passes over `batch` 1080x1920 frames whose masks are
"every leaf" (labels >= 1) on a handle created with LG_NO_SKIP (1: tile-level constant path off; 3: also the wave-level
off-leaf shortcut off).  Prints the kernel's average duration (its own dispatch events) and the fraction of the 8 TB/s HBM
peak at 37.25 algorithmic bytes per pixel.  Also the profiling driver of tools/pmc_final_dense.sh.
usage: python tools/final_dense.py [batch] [reps] [no_skip]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 128
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
os.environ["LG_NO_SKIP"] = sys.argv[3] if len(sys.argv) > 3 else "1"
import leafgrasp_amd as L  # noqa: E402
from leafgrasp_amd._lib import lib  # noqa: E402
import synthetic_inputs as SI  # noqa: E402

H, W = 1080, 1920
dev = torch.device("cuda", 0)
scenes = [SI.synthetic_scene(H, W, seed=100 + s) for s in range(8)]
sel = L.GraspPointSelector(dev, load_model=False)
sel.set_camera_params(scenes[0][2])
m = torch.from_numpy(np.stack([scenes[i % 8][0] >= 1 for i in range(batch)]).astype(np.uint8)).to(dev)
d = torch.from_numpy(np.stack([scenes[i % 8][1] for i in range(batch)])).to(dev)
sel.score_maps(m, d)
lib.lg_profile_enable(sel._h, 2)
for _ in range(reps):
    sel.score_maps(m, d)
torch.cuda.synchronize()
n, ms = C.c_int(0), C.c_double(0.0)
lib.lg_profile_read(sel._h, b"final", C.byref(n), C.byref(ms))
avg = ms.value / max(1, n.value)
gbs = 37.25 * batch * H * W / (avg * 1e-3) / 1e9
print(f"no_skip={os.environ['LG_NO_SKIP']} batch={batch} launches={n.value} avg_ms={avg:.4f} achieved_GBps={gbs:.1f} frac_of_8TBps={gbs / 8000:.4f} "
      f"leaf_coverage={float(m.float().mean()):.3f}", flush=True)
