"""Per-call timeline of lg_select_grasp (LG_TRACE=1): when the sweeps, the planes, the top-k and the CNN of a 256-frame call finish, and
where the host waits.  usage (GPU box): python tools/trace_run.py"""
import os, sys, time
sys.path.insert(0, os.getcwd())
os.environ["LG_TRACE"] = "1"
import numpy as np, torch
import bench
import leafgrasp_amd as L, synthetic_inputs as SI
m, d, P, _ = bench.make_frames(256, 1080, 1920, workers=8)
dev = torch.device("cuda:0")
sel = L.GraspPointSelector(dev, load_model=False); sel.set_camera_params(P); sel.set_cnn_state_dict(SI.cnn_closed_form_params(0))
mt, dt = torch.from_numpy(m).to(dev), torch.from_numpy(d).to(dev)
for i in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = sel.select_grasp_points_batch(mt, dt)
    torch.cuda.synchronize(); print(f"call {i}: {1e3*(time.perf_counter()-t0):.3f} ms", file=sys.stderr)
