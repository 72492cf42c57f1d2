#!/usr/bin/env python3
"""tests/golden/cnn_variant_vectors.npz: logits of the REFERENCE GraspPointCNN for the non-default attention types
('channel', 'hybrid', 'none'; scripts/utils/ml_grasp_optimizer/model.py:30-60, 108-121) and the non-default encoder_filters
('lightweight' [32,64,128], 'deep' [64,128,256,512], 'wide' [128,256,512]; train_model_mlflow.py:177-182) with the closed-form weights of
oracle.lg_oracle.cnn_closed_form_params and the seeded patches of synthetic_patches.  Runnable only where
/root/reference exists (pure torch module, imported read-only); the fixture is data."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.join(HERE))
from make_golden import _load_by_path  # noqa: E402
from oracle import lg_oracle as O  # noqa: E402  (closed-form fill + synthetic inputs only)

model = _load_by_path("ref_model", "/root/reference/scripts/utils/ml_grasp_optimizer/model.py")
out = {"x_seed": np.array(5), "n": np.array(12)}
x = O.synthetic_patches(12, seed=5)
for att in ("channel", "hybrid", "none"):
    params = O.cnn_closed_form_params(seed=1, attention_type=att)
    net = model.GraspPointCNN(in_channels=9, attention_type=att)
    sd = net.state_dict()
    assert set(params) == {k for k in sd if not k.endswith("num_batches_tracked")}, att
    for k, v in params.items():
        assert tuple(sd[k].shape) == tuple(v.shape), k
        sd[k] = torch.from_numpy(v)
    net.load_state_dict(sd)
    net.eval()
    with torch.no_grad():
        out[f"logits_{att}"] = net(torch.from_numpy(x)).reshape(-1).numpy()
        out[f"logits_f64_{att}"] = net.double()(torch.from_numpy(x).double()).reshape(-1).numpy()
# encoder_filters variants, paired with attention types as in scripts/demo_mlflow_setup.py:44-49
for name, filt, att in (("lightweight", (32, 64, 128), "spatial"), ("deep", (64, 128, 256, 512), "hybrid"),
                        ("wide", (128, 256, 512), "none")):
    params = O.cnn_closed_form_params(seed=2, attention_type=att, filters=filt)
    net = model.GraspPointCNN(in_channels=9, attention_type=att, encoder_filters=list(filt))
    sd = net.state_dict()
    assert set(params) == {k for k in sd if not k.endswith("num_batches_tracked")}, name
    for k, v in params.items():
        assert tuple(sd[k].shape) == tuple(v.shape), k
        sd[k] = torch.from_numpy(v)
    net.load_state_dict(sd)
    net.eval()
    with torch.no_grad():
        out[f"logits_{name}"] = net(torch.from_numpy(x)).reshape(-1).numpy()
        out[f"logits_f64_{name}"] = net.double()(torch.from_numpy(x).double()).reshape(-1).numpy()
np.savez_compressed(os.path.join(HERE, "cnn_variant_vectors.npz"), **out)
print({k: (v.tolist() if v.size < 4 else v[:3].tolist()) for k, v in out.items()})
