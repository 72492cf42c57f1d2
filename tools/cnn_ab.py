#!/usr/bin/env python3
"""Times GraspPointCNN forward (lg_cnn_forward through the Python mirror) for the library selected by LG_LIB_PATH /
the LG_CNN_* switches.  usage: python tools/cnn_ab.py [patches] -- prints ms per forward (HIP events, 10 reps)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import leafgrasp_amd as L
import synthetic_inputs as O  # seeded inputs only (closed-form weights / scenes / patches)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2560
sel = L.GraspPointSelector(torch.device("cuda", 0), load_model=False)
sel.set_cnn_state_dict(O.cnn_closed_form_params(seed=0))
x = torch.from_numpy(O.synthetic_patches(64, seed=1)).cuda().repeat((n + 63) // 64, 1, 1, 1)[:n].contiguous()
for _ in range(3):
    sel.cnn_forward(x)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    sel.cnn_forward(x)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f"{os.environ.get('LG_LIB_PATH', 'default').split('/')[-1]:28s} KC={os.environ.get('LG_CNN_WS_KC', '8')} "
      f"patches={n} ms={ms:.3f}  direct-equivalent TFLOP/s={312.83e6 * n / ms / 1e9:.1f}")
