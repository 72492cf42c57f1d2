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


if __name__ == "__main__":
    main()
