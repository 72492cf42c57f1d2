"""Which conv layer / which patches deviate: GraspPointCNN logits with one layer at a time on the Winograd F(4x4,3x3) kernel (LG_CNN_WINO_MASK)
against the all-direct forward, for 20 / 41 / 300 patches.  usage (GPU box): python tools/cnn_layer_bisect.py"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import leafgrasp_amd as L
import synthetic_inputs as SI
params = SI.cnn_closed_form_params(seed=0)
for n in (20, 41, 300):
    x = torch.from_numpy(SI.synthetic_patches(n, seed=5)).cuda()
    os.environ["LG_CNN_DIRECT"] = "1"
    sel = L.GraspPointSelector(torch.device("cuda:0"), load_model=False)
    sel.set_cnn_state_dict(params)
    want = sel.cnn_forward(x).cpu().numpy()
    del os.environ["LG_CNN_DIRECT"]
    for mask in (1, 2, 4, 8, 16, 32, 63):
        os.environ["LG_CNN_WINO_MASK"] = str(mask)
        sel.set_cnn_state_dict(params)
        got = sel.cnn_forward(x).cpu().numpy()
        bad = np.nonzero(~np.isclose(got, want, rtol=2e-5, atol=2e-6))[0]
        print(f"n={n} mask={mask}: bad patches {bad.tolist()[:40]} maxerr {np.abs(got-want).max():.3e}", flush=True)
    del os.environ["LG_CNN_WINO_MASK"]
