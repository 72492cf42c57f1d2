import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import leafgrasp_amd as L
from oracle import lg_oracle as O
H, W, case = 284, 537, 72
labels, depth, P = O.synthetic_scene(H, W, 1000 + case)
mask = (labels == 1).astype(np.uint8)
sel = L.GraspPointSelector("cuda:0", load_model=False); sel.set_camera_params(P)
maps, valid, theta = sel.score_maps(torch.from_numpy(mask).cuda(), torch.from_numpy(depth).cuda())
ref = O.RefGraspPointSelector(); ref.set_camera_params(P)
sdf, parts = ref.calculate_sdf_score(mask, return_parts=True)
g = maps["sdf_score"].cpu().numpy()
d = np.abs(g - sdf); y, x = np.unravel_index(d.argmax(), d.shape)
print("theta gpu/ref", theta, parts["angle"], "diff", theta - parts["angle"])
print("worst at", (x, y), "gpu", g[y, x], "ref", sdf[y, x], "cx,cy", ref.camera_cx, ref.camera_cy)
din = parts["dist_inside"]; dout = parts["dist_outside"]
print("din", din[y, x], "max|sdf|", np.max(np.abs(din - dout)), "n bad", int((d > 1e-6 + 1e-4 * np.abs(sdf)).sum()))
vx, vy = x - ref.camera_cx, y - ref.camera_cy; r = np.hypot(vx, vy)
a = parts["angle"]
print("r", r, "align ref", abs(vx / r * np.sin(a) - vy / r * np.cos(a)), "interior", np.exp(-((din[y, x] - 20) ** 2) / 800))
ys, xs = np.where(d > 1e-6 + 1e-4 * np.abs(sdf)); print("bad px sample", list(zip(xs[:8], ys[:8])))
