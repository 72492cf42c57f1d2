#!/usr/bin/env python3
"""Throughput of the training step (lg_train_step) on synthetic patches: the reference's batch size (16,
train_model.py:207) and larger batches.  Inputs and labels are resident on the device, dropout masks are drawn on the
device, the loss is read back once at the end (steps are stream-ordered, no host round trip in between).
    python tools/train_bench.py [--batches 16,256,2048] [--steps 50]
Prints one JSON object per batch size: ms per step, samples/s, executed conv TFLOP/s (forward + backward-data +
backward-weights: 3 x 2 x 9 x Cin x Cout x H x W per layer and sample, minus the first layer's backward-data)."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import synthetic_inputs as S  # noqa: E402
from leafgrasp_amd._lib import lib  # noqa: E402
from leafgrasp_amd.trainer import GraspTrainer  # noqa: E402


def conv_flops_per_sample(filters):
    fl, c, wi = 0, 9, 32
    for b, f in enumerate(filters):
        for k, cin in enumerate((c, f)):
            per = 2 * 9 * cin * f * wi * wi
            fl += per * (2 if (b == 0 and k == 0) else 3)
        c, wi = f, wi // 2
    return fl


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="16,256,2048")
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--filters", default="64,128,256")
    ap.add_argument("--torch-baseline", action="store_true",
                    help="also time the same step written with torch.nn modules + torch.optim.Adam on the same device "
                         "(PyTorch-ROCm eager: MIOpen / rocBLAS kernels) -- what the reference's script would run here")
    ap.add_argument("--cpu-baseline", action="store_true",
                    help="also time the stock-torch step on the host cores (batch 16, a few steps)")
    a = ap.parse_args()
    filters = tuple(int(v) for v in a.filters.split(","))
    dev = torch.device("cuda:0")
    for n in [int(v) for v in a.batches.split(",")]:
        tr = GraspTrainer(dev, encoder_filters=filters, max_batch=n)
        base = S.synthetic_patches(min(n, 256), seed=1)
        x = torch.from_numpy(base).to(dev).repeat((n + base.shape[0] - 1) // base.shape[0], 1, 1, 1)[:n].contiguous()
        y = (torch.arange(n, device=dev) % 3 == 0).float()
        loss = C.c_float()

        def step(sync):
            rc = lib.lg_train_step(tr._h, x.data_ptr(), y.data_ptr(), n, None, 1234, C.byref(tr.hp), 1,
                                   C.byref(loss) if sync else None, None, None)
            assert rc == 0, lib.lg_train_last_error(tr._h)
        torch.cuda.synchronize()
        for _ in range(a.warmup):
            step(False)
        step(True)
        first = loss.value
        t0 = time.perf_counter()
        for _ in range(a.steps - 1):
            step(False)
        step(True)
        dt = (time.perf_counter() - t0) / a.steps
        fl = conv_flops_per_sample(filters) * n
        print(json.dumps({"batch": n, "filters": list(filters), "ms_per_step": round(dt * 1e3, 4),
                          "samples_per_s": round(n / dt, 1), "conv_tflops": round(fl / dt / 1e12, 2),
                          "loss_first": round(first, 4), "loss_last": round(loss.value, 4)}), flush=True)
        del tr
        if a.torch_baseline:
            print(json.dumps(torch_eager(n, filters, x, y, a.steps, a.warmup)), flush=True)
        if a.cpu_baseline and n == 16:
            r = torch_eager(n, filters, x.cpu(), y.cpu(), 5, 1)
            r["device"], r["threads"] = "cpu", torch.get_num_threads()
            print(json.dumps(r), flush=True)


def torch_eager(n, filters, x, y, steps, warmup):
    """The architecture of scripts/utils/ml_grasp_optimizer/model.py (spatial attention) and the loop body of
    scripts/train_model.py:247-265, written here with stock torch modules (measurement only)."""
    import torch.nn as nn
    blocks, c = [], 9
    for f in filters:
        blocks.append(nn.Sequential(nn.Conv2d(c, f, 3, padding=1), nn.BatchNorm2d(f), nn.ReLU(inplace=True),
                                    nn.Conv2d(f, f, 3, padding=1), nn.BatchNorm2d(f), nn.ReLU(inplace=True),
                                    nn.MaxPool2d(2), nn.Dropout2d(0.3)))
        c = f

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.encoder = nn.ModuleList(blocks)
            self.attention = nn.Sequential(nn.Conv2d(c, 1, 1), nn.Sigmoid())
            self.classifier = nn.Sequential(nn.Linear(c, c), nn.BatchNorm1d(c), nn.ReLU(inplace=True), nn.Dropout(0.5),
                                            nn.Linear(c, c // 2), nn.BatchNorm1d(c // 2), nn.ReLU(inplace=True), nn.Dropout(0.5),
                                            nn.Linear(c // 2, c // 4), nn.BatchNorm1d(c // 4), nn.ReLU(inplace=True), nn.Dropout(0.4),
                                            nn.Linear(c // 4, 1))

        def forward(self, h):
            for b in self.encoder:
                h = b(h)
            h = h * self.attention(h)
            return self.classifier(h.mean(dim=(2, 3)))

    net = Net().to(x.device).train()
    crit = nn.BCEWithLogitsLoss(pos_weight=torch.tensor([2.0], device=x.device))
    opt = torch.optim.Adam(net.parameters(), lr=0.0005, weight_decay=0.01)

    def step():
        opt.zero_grad()
        loss = crit(net(x).squeeze(1), y)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(net.parameters(), max_norm=1.0)
        opt.step()
        return loss
    for _ in range(warmup + 3):
        step()
    if x.is_cuda:
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    loss.item()
    dt = (time.perf_counter() - t0) / steps
    return {"torch_eager_batch": n, "ms_per_step": round(dt * 1e3, 4), "samples_per_s": round(n / dt, 1)}


if __name__ == "__main__":
    main()
