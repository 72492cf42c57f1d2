#!/usr/bin/env python3
"""tests/golden/collector_vectors.npz: outputs of the REFERENCE EnhancedGraspDataCollector
(scripts/utils/ml_grasp_optimizer/data_collector.py) for its cv2-free methods -- _check_boundaries, _extract_patches,
_rotate_tensor, _rotate_point, _add_sample, _generate_augmented_samples (seeded), save_samples (on-disk layout) -- on a
seeded synthetic scene whose score planes come from the oracle.  rospy is the logger stub, cv2 an import-only stub
(the tip / stem / edge helpers that need it are NOT run: those rows stay "parity unpinned").  Runnable only where
/root/reference exists; the fixture is data."""
import os
import random
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402
from oracle import lg_oracle as O  # noqa: E402  (synthetic inputs + score planes only)

MG._install_stubs()
os.environ["HOME"] = tempfile.mkdtemp(prefix="lg_collector_")
dc = MG._load_by_path("ref_data_collector", "/root/reference/scripts/utils/ml_grasp_optimizer/data_collector.py")

H, W = 135, 180
labels, depth, P = O.synthetic_scene(H, W, 2)
mask = (labels == 1).astype(np.uint8)
ref_sel = O.RefGraspPointSelector()
ref_sel.set_camera_params(P)
scores = {k: np.asarray(v) for k, v in ref_sel._calculate_all_scores(mask, depth).items()}
col = dc.EnhancedGraspDataCollector(patch_size=32, resume=False)
mt, dt = torch.from_numpy(mask.astype(bool)), torch.from_numpy(depth)

out = {"H": np.array(H), "W": np.array(W), "seed": np.array(2)}
ys, xs = np.nonzero(mask)
cy, cx = int(ys.mean()), int(xs.mean())
pts = [(cx, cy), (16, 16), (15, 40), (W - 16, H - 16), (W - 17, H - 17), (cx + 40, cy), (5, 5), (W - 1, H - 1),
       (cx - 12, cy + 7), (cx + 9, cy - 11), (int(xs.min()) + 2, int(ys[xs.argmin()])), (cx, int(ys.max()) + 10)]
out["pts"] = np.array(pts)
out["bounds_ok"] = np.array([col._check_boundaries(x, y, mask.shape, 16) for x, y in pts])
ok, dps, mps, sps = [], [], [], []
for x, y in pts:
    r = col._extract_patches(x, y, mt, dt, scores)
    ok.append(r is not None)
    if r is not None:
        dps.append(r[0].numpy()); mps.append(r[1].float().numpy()); sps.append(r[2].numpy())
out["extract_ok"] = np.array(ok)
out["extract_depth"], out["extract_mask"], out["extract_scores"] = np.stack(dps), np.stack(mps), np.stack(sps)
out["rot_points"] = np.array([col._rotate_point(p, a, 32) for p in ((16, 16), (3, 7), (30, 2), (0, 31)) for a in (90, 180, 270)])
out["rot_tensor"] = np.stack([col._rotate_tensor(torch.arange(16.0).reshape(4, 4), a).numpy() for a in (90, 180, 270)])

# seeded augmentation + bookkeeping + on-disk layout
random.seed(123)
torch.manual_seed(123)
d0, m0, s0 = col._extract_patches(cx, cy, mt, dt, scores)
assert col._add_sample(d0, m0, s0, 0.8125, (cx, cy), label=1, is_augmented=False)
col._generate_augmented_samples(d0, m0, s0, 0.8125, (cx, cy))
assert col._add_sample(d0, m0.float(), s0, 0.0, (cx + 1, cy), label=0, is_augmented=False)
out["aug_points"] = np.array([s["grasp_point"] for s in col.samples])
out["aug_total"] = np.array([s["total_score"] for s in col.samples])
out["aug_mask"] = np.stack([s["mask_patch"].float().numpy() for s in col.samples])
out["aug_scores"] = np.stack([s["score_patches"].numpy() for s in col.samples])
out["aug_depth"] = np.stack([s["depth_patch"].numpy() for s in col.samples])
out["stats"] = np.array([col.stats["positive_samples"], col.stats["augmented_samples"], col.stats["negative_samples"]])
col.save_samples()
saved = torch.load(os.path.join(col.data_dir, "training_data.pt"), weights_only=True)
out["saved_keys"] = np.array(sorted(saved.keys()))
out["saved_layout"] = np.array([f"{k}:{str(v.dtype)}:{tuple(v.shape)}" for k, v in sorted(saved.items())])
out["saved_files"] = np.array(sorted(os.listdir(col.data_dir)))
np.savez_compressed(os.path.join(HERE, "collector_vectors.npz"), **out)
print(out["bounds_ok"], out["extract_ok"], out["rot_points"].tolist(), out["stats"], out["saved_layout"], out["saved_files"])
