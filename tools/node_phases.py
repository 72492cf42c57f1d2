import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
import leafgrasp_amd as L, synthetic_inputs as SI
nb = 128
m, d, P, labels = bench.make_frames(nb, 1080, 1920, workers=8)
dev = torch.device("cuda:0")
lab = torch.from_numpy(np.stack([labels[i % len(labels)] for i in range(nb)])).to(dev)
dep = torch.from_numpy(d[:nb]).to(dev)
hz = L.LeafGraspHarness(1080, 1920, dev, load_model=False)
hz.camera_info_callback(np.asarray(P).reshape(-1))
hz.grasp_selector.set_cnn_state_dict(SI.cnn_closed_form_params(seed=0))
def sync(): torch.cuda.synchronize(dev)
for it in range(4):
    sync(); t0 = time.perf_counter()
    ids = hz.leaf_scorer.select_optimal_leaves_batch(lab, dep)
    sync(); t1 = time.perf_counter()
    idt = torch.tensor(ids, dtype=lab.dtype, device=dev).reshape(-1, 1, 1)
    optimal = lab == idt
    sync(); t2 = time.perf_counter()
    res = hz.grasp_selector.select_grasp_points_batch(optimal, dep, image_processor=hz.image_processor)
    sync(); t3 = time.perf_counter()
    out = [hz.format_result(p2, p3, pre) for (p2, p3, pre) in res if p2 is not None]
    t4 = time.perf_counter()
    sync(); t5 = time.perf_counter()
    hz.process_batch_device(lab, dep)
    sync(); t6 = time.perf_counter()
    print("leaf %.3f  mask %.3f  grasp %.3f  format %.3f  | whole %.3f ms" % (1e3*(t1-t0), 1e3*(t2-t1), 1e3*(t3-t2), 1e3*(t4-t3), 1e3*(t6-t5)))
