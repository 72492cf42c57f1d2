#!/usr/bin/env python3
"""tests/golden/train_vectors.npz: two optimisation steps of the REFERENCE training loop body (scripts/train_model.py:247-265:
zero_grad / model(batch) / BCEWithLogitsLoss(pos_weight=2.0) / backward / clip_grad_norm_(1.0) / Adam(lr=5e-4,
weight_decay=0.01).step()) on the REFERENCE module (scripts/utils/ml_grasp_optimizer/model.py::GraspPointCNN) in train()
mode, from the closed-form weights of synthetic_inputs.cnn_closed_form_params on the seeded patches of synthetic_patches.
Dropout draws from torch's global generator, which no other implementation can replay: the instantiated module's Dropout
layers get p = 0 here (module configuration, the source is untouched); dropout itself is covered by explicit-mask tests
against the restatement that this fixture pins.  Stored: loss / logits / gradient norm per step, every BatchNorm running
statistic after the two steps, and for every parameter tensor its gradient norm per step plus 48 sampled entries of the
gradients and of the final parameters / Adam moments (fixed seeded index set).  Runnable only where /root/reference exists
(pure torch module, imported read-only); the fixture is data."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
from make_golden import _load_by_path  # noqa: E402
import synthetic_inputs as S  # noqa: E402

model = _load_by_path("ref_model", "/root/reference/scripts/utils/ml_grasp_optimizer/model.py")
torch.set_num_threads(4)
out = {}
CASES = (("spatial", (64, 128, 256), 8), ("none", (32, 64, 128), 6), ("hybrid", (64, 128, 256), 6), ("channel", (32, 64, 128), 8))
for att, filt, n in CASES:
    tag = f"{att}_{len(filt)}x{filt[0]}"
    params = S.cnn_closed_form_params(seed=3, attention_type=att, filters=filt)
    net = model.GraspPointCNN(in_channels=9, attention_type=att, encoder_filters=list(filt))
    sd = net.state_dict()
    for k, v in params.items():
        sd[k] = torch.from_numpy(v.copy())
    net.load_state_dict(sd)
    for m in net.modules():
        if isinstance(m, (torch.nn.Dropout, torch.nn.Dropout2d)):
            m.p = 0.0
    net.train()
    x = torch.from_numpy(S.synthetic_patches(n, seed=11))
    y = torch.tensor([(i * 5 + 1) % 3 == 0 for i in range(n)], dtype=torch.float32)
    criterion = torch.nn.BCEWithLogitsLoss(pos_weight=torch.tensor([2.0]))
    optimizer = torch.optim.Adam(net.parameters(), lr=0.0005, weight_decay=0.01)
    rng = np.random.default_rng(99)
    names = [k for k, _ in net.named_parameters()]
    pick = {k: rng.integers(0, p.numel(), 48) for k, p in net.named_parameters()}
    out[f"{tag}_labels"] = y.numpy()
    for step in range(2):
        optimizer.zero_grad()
        outputs = net(x)
        loss = criterion(outputs.squeeze(), y)
        loss.backward()
        total = torch.nn.utils.clip_grad_norm_(net.parameters(), max_norm=1.0)
        # .grad now holds the CLIPPED gradients; store the unclipped ones (grad / coef)
        coef = min(1.0 / (float(total) + 1e-6), 1.0)
        out[f"{tag}_loss{step}"] = np.array(loss.item())
        out[f"{tag}_logits{step}"] = outputs.detach().reshape(-1).numpy().copy()
        out[f"{tag}_gnorm{step}"] = np.array(float(total))
        out[f"{tag}_gtnorm{step}"] = np.array([float(p.grad.norm()) / coef for p in net.parameters()])
        out[f"{tag}_gsample{step}"] = np.stack([p.grad.reshape(-1)[pick[k]].numpy() / coef for k, p in net.named_parameters()])
        optimizer.step()
    out[f"{tag}_psample"] = np.stack([p.detach().reshape(-1)[pick[k]].numpy() for k, p in net.named_parameters()])
    out[f"{tag}_msample"] = np.stack([optimizer.state[p]["exp_avg"].reshape(-1)[pick[k]].numpy() for k, p in net.named_parameters()])
    out[f"{tag}_vsample"] = np.stack([optimizer.state[p]["exp_avg_sq"].reshape(-1)[pick[k]].numpy() for k, p in net.named_parameters()])
    out[f"{tag}_pick"] = np.stack([pick[k] for k in names])
    out[f"{tag}_names"] = np.array(names)
    sd = net.state_dict()
    out[f"{tag}_buffers"] = np.concatenate([sd[k].numpy().reshape(-1) for k in sd if "running_" in k])
    out[f"{tag}_nbt"] = np.array(int(sd["encoder.0.1.num_batches_tracked"]))
np.savez_compressed(os.path.join(HERE, "train_vectors.npz"), **out)
print({k: (v.tolist() if v.size < 4 else v.reshape(-1)[:3].tolist()) for k, v in out.items() if "sample" not in k and "pick" not in k and "names" not in k})

# ---- host-side helpers of the training script: EarlyStopping (train_model.py:11-39), normalize_data (:41-62),
# analyze_predictions (:64-99), run from the reference file itself (rospy / cv2 replaced by import-only stubs as in make_golden.py)
from make_golden import _install_stubs  # noqa: E402

_install_stubs()
sys.path.insert(0, "/root/reference/scripts")
tm = _load_by_path("ref_train_model", "/root/reference/scripts/train_model.py")


class _Model:
    def __init__(self):
        self.w = torch.zeros(1)

    def state_dict(self):
        return {"w": self.w.clone()}

    def load_state_dict(self, sd):
        self.w = sd["w"].clone()


hout = {}
rng = np.random.default_rng(7)
seqs, stops, bests, restored = [], [], [], []
for s_i in range(16):
    n = 60
    base = 1.0 / (1.0 + 0.15 * np.arange(n)) + 0.3
    noise = rng.normal(0, 0.004 * (1 + s_i % 4), n)
    seq = base + noise
    if s_i % 3 == 0:
        seq[20:] = seq[20] + np.abs(rng.normal(0, 0.0008, n - 20))      # plateau within min_delta
    es = tm.EarlyStopping(patience=15 if s_i % 2 == 0 else 5, min_delta=0.001, restore_best_weights=True)
    m = _Model()
    stop = -1
    for epoch, v in enumerate(seq):
        m.w = torch.tensor([float(epoch)])
        if es.step(float(v), epoch, m):
            stop = epoch
            break
    seqs.append(seq)
    stops.append(stop)
    bests.append([es.best_epoch, es.best_loss])
    restored.append(float(m.w[0]))
hout["es_seqs"], hout["es_stop"], hout["es_best"], hout["es_restored"] = np.array(seqs), np.array(stops), np.array(bests), np.array(restored)
d = torch.from_numpy(rng.random((6, 1, 8, 8)).astype(np.float32) * 0.4 + 0.3)      # any spatial size: global / per-channel statistics
sc = torch.from_numpy((rng.random((6, 7, 8, 8)) * np.arange(1, 8)[None, :, None, None]).astype(np.float32))
nd = tm.normalize_data(d, sc)
hout["nd_depth_in"], hout["nd_score_in"] = d.numpy(), sc.numpy()
hout["nd_depth"], hout["nd_score"] = nd["depth_patches"].numpy(), nd["score_patches"].numpy()
hout["nd_stats"] = np.concatenate([nd["stats"]["depth_mean"].reshape(-1).numpy(), nd["stats"]["depth_std"].reshape(-1).numpy(),
                                   nd["stats"]["score_mean"].reshape(-1).numpy(), nd["stats"]["score_std"].reshape(-1).numpy()])
outs = torch.from_numpy(rng.normal(0.3, 1.0, (40, 1)).astype(np.float32))
labs = torch.from_numpy((rng.random(40) < 0.45).astype(np.float32))
ap = tm.analyze_predictions(outs, labs)
hout["ap_outputs"], hout["ap_labels"] = outs.numpy(), labs.numpy()
hout["ap_metrics"] = np.array([ap["positive_accuracy"], ap["negative_accuracy"], ap["precision"], ap["recall"], ap["f1_score"],
                               ap["confusion_matrix"]["true_positive"], ap["confusion_matrix"]["false_positive"],
                               ap["confusion_matrix"]["false_negative"], ap["confusion_matrix"]["true_negative"]], np.float64)
np.savez_compressed(os.path.join(HERE, "train_host_vectors.npz"), **hout)
print("host helpers:", hout["es_stop"].tolist(), hout["ap_metrics"].tolist())
