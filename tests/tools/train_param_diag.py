#!/usr/bin/env python3
"""Where do the parameters after two Adam steps differ from the reference fixture?  Per case: share of sampled elements outside
1e-4, split by how far the element's reference gradient stands above its tensor's RMS (both steps), and by tensor."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.test_train_oracle import CASES, case_inputs, noisy_bias  # noqa: E402
from leafgrasp_amd.trainer import GraspTrainer, dropout_layout  # noqa: E402

tv = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "golden", "train_vectors.npz"))
for att, filt, n in CASES:
    tag = f"{att}_{len(filt)}x{filt[0]}"
    params, x, y = case_inputs(att, filt, n)
    names = [str(k) for k in tv[f"{tag}_names"]]
    pick = tv[f"{tag}_pick"]
    tr = GraspTrainer(torch.device("cuda:0"), attention_type=att, encoder_filters=filt, max_batch=16)
    tr.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    ones = [np.ones((n, w), np.float32) for w, _ in dropout_layout(filt)]
    tr.train_step(x, y, masks=ones)
    g_step0 = tr.gradients()
    tr.train_step(x, y, masks=ones)
    sd = tr.state_dict()
    ps = np.stack([sd[k].numpy().reshape(-1)[pick[i]] for i, k in enumerate(names)])
    gs0 = np.stack([g_step0[k].numpy().reshape(-1)[pick[i]] for i, k in enumerate(names)])
    ref_p, g0, g1 = tv[f"{tag}_psample"], tv[f"{tag}_gsample0"], tv[f"{tag}_gsample1"]
    noisy = np.array([noisy_bias(k) for k in names])
    rms0 = np.sqrt((g0.astype(np.float64) ** 2).mean(axis=1, keepdims=True)) + 1e-30
    rms1 = np.sqrt((g1.astype(np.float64) ** 2).mean(axis=1, keepdims=True)) + 1e-30
    bad = np.abs(ps - ref_p) > 2e-5 + 1e-4 * np.abs(ref_p)
    print(f"== {tag}: bad overall {bad[~noisy].mean():.4f}")
    for tau in (1e-4, 1e-3, 1e-2, 1e-1):
        clear = (np.abs(g0) >= tau * rms0) & (np.abs(g1) >= tau * rms1) & ~noisy[:, None]
        print(f"   tau {tau:g}: clear share {clear.mean():.3f}  bad among clear {bad[clear].mean():.4f}")
    sign_flip = (np.sign(gs0) != np.sign(g0)) & ~noisy[:, None]
    print(f"   step-0 gradient sign differs on {sign_flip.mean():.4f} of the elements; among bad {sign_flip[bad & ~noisy[:, None]].mean():.3f}; "
          f"bad with equal sign {np.mean(bad & ~sign_flip & ~noisy[:, None]):.4f}")
    dev = np.abs(ps - ref_p)
    print("   |dp| quantiles of bad elements:", np.quantile(dev[bad & ~noisy[:, None]], [0.1, 0.5, 0.9]) if bad.any() else None)
    per = [(names[i], float(bad[i].mean())) for i in range(len(names)) if bad[i].mean() > 0.05 and not noisy[i]]
    print("   tensors with > 5 % bad:", per[:12])
    rel0 = np.abs(gs0 - g0) / (rms0 + 1e-30)
    print("   step-0 gradient |d|/rms quantiles:", np.quantile(rel0[~noisy], [0.5, 0.9, 0.99, 1.0]))
