"""Per-tensor gradient error of lg_train_step against the oracle in fp32 and fp64 (conditioning diagnosis)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import synthetic_inputs as S  # noqa: E402
from oracle import lg_oracle as O  # noqa: E402
from leafgrasp_amd.trainer import GraspTrainer, dropout_layout  # noqa: E402

for att, filt, n, seed in [("channel", (32, 64, 128), 8, 3), ("hybrid", (64, 128, 256), 6, 7), ("channel", (64, 128, 256), 16, 6),
                           ("none", (64, 128, 256, 512), 8, 2)]:
    params = S.cnn_closed_form_params(seed=seed, attention_type=att, filters=filt)
    x = S.synthetic_patches(n, seed=20 + seed)
    y = (np.random.default_rng(seed).random(n) < 0.4).astype(np.float32)
    y[0], y[1] = 0.0, 1.0
    rng = np.random.default_rng(seed)
    mk = [((rng.random((n, w)) >= p) / (1.0 - p)).astype(np.float32) for w, p in dropout_layout(filt)]
    tr = GraspTrainer(torch.device("cuda:0"), attention_type=att, encoder_filters=filt, max_batch=16)
    tr.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    r32 = O.cnn_train_step(params, x, y, masks=mk)
    r64 = O.cnn_train_step(params, x, y, masks=mk, dtype=torch.float64)
    loss, logits, gnorm = tr.train_step(x, y, masks=mk, return_logits=True)
    print(att, filt, n, "loss", loss, r32["loss"], r64["loss"], "gnorm", gnorm, r32["grad_norm"], r64["grad_norm"])
    print(" logits err gpu-64 %.2e  32-64 %.2e" % (np.abs(logits.cpu().numpy() - r64["logits"]).max(), np.abs(r32["logits"] - r64["logits"]).max()))
    g = tr.gradients()
    for k in r64["grads"]:
        d64 = np.linalg.norm(r64["grads"][k])
        e_gpu = np.linalg.norm(g[k].numpy() - r64["grads"][k]) / (d64 + 1e-30)
        e_32 = np.linalg.norm(r32["grads"][k] - r64["grads"][k]) / (d64 + 1e-30)
        print("  %-24s |g| %.3e  gpu-vs-f64 %.2e  torch32-vs-f64 %.2e" % (k, d64, e_gpu, e_32))
