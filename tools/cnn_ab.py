#!/usr/bin/env python3
"""GraspPointCNN forward through the C-ABI: per-variant time and deviation.

    python tools/cnn_ab.py [patches]      (default 5120 = 256 frames x 20 candidates)

Variants are chosen by the LG_CNN_* switches, which lg_cnn_load reads when the model is loaded:
default = Winograd F(4x4,3x3), LG_CNN_F23=1 = Winograd F(2x2,3x3), LG_CNN_DIRECT=1 = direct implicit GEMM.
Prints ms per forward (HIP events, 10 repetitions) and the largest |logit - direct| over 64 seeded patches.
Only seeded inputs are used here (synthetic_inputs.py); parity against the oracle lives in tests/."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import leafgrasp_amd as L  # noqa: E402
import synthetic_inputs as SI  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5120
dev = torch.device("cuda", 0)
params = SI.cnn_closed_form_params(seed=0)
x = torch.from_numpy(SI.synthetic_patches(64, seed=1)).to(dev).repeat((n + 63) // 64, 1, 1, 1)[:n].contiguous()
ref = None
for name, env in (("direct", {"LG_CNN_DIRECT": "1"}), ("wino F(2x2,3x3)", {"LG_CNN_F23": "1"}), ("wino F(4x4,3x3)", {})):
    for k in ("LG_CNN_DIRECT", "LG_CNN_F23", "LG_CNN_WINO_MASK"):
        os.environ.pop(k, None)
    os.environ.update(env)
    sel = L.GraspPointSelector(dev, load_model=False)
    sel.set_cnn_state_dict(params)
    for _ in range(3):
        out = sel.cnn_forward(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        out = sel.cnn_forward(x)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    out = out[:64].double().cpu()
    if ref is None:
        ref = out
    dev_max = float((out - ref).abs().max())
    print(f"{name:18s} patches={n} ms={ms:.3f}  direct-equivalent TFLOP/s={312.83e6 * n / ms / 1e9:7.1f}  "
          f"max|logit - direct|={dev_max:.3e}  finite={bool(torch.isfinite(out).all())}", flush=True)
    del sel
