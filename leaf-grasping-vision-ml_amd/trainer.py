"""GraspPointCNN training on the MI355X path.

Mirror of scripts/train_model.py: `normalize_data` (:41-62), the inner optimisation step (:247-265: zero_grad, train-mode
forward, BCEWithLogitsLoss(pos_weight=2.0), backward, clip_grad_norm_(1.0), Adam(lr=5e-4, weight_decay=0.01)), the epoch
loop with WeightedRandomSampler / ReduceLROnPlateau / EarlyStopping (:155-356) and `analyze_predictions` (:64-99).  The
step itself (forward, backward, optimizer) is lg_train_step of liblgrasp.so -- hand-written gfx950 kernels, no autograd,
no CPU fallback.  The model is scripts/utils/ml_grasp_optimizer/model.py::GraspPointCNN with any of its attention types
('spatial' = the training script's model) and encoder_filters; state dicts use the reference module's key names, so a checkpoint written here is
read by GraspPointSelector.load_ml_model (key 'model_state_dict') and by the reference itself.

Dropout: the reference draws its masks from torch's global generator; a different generator cannot reproduce that
stream, so `train_step` takes explicit keep masks (tests) or draws them on the device from `seed` (training)."""
import ctypes as C
import os

import numpy as np
import torch

from ._lib import LgError, LgTrainHparams, lib

_VP = C.c_void_p
_FP = C.POINTER(C.c_float)
_ATT = {"spatial": 0, "channel": 1, "hybrid": 2, "none": 3}   # include/leafgrasp.h LG_ATT_*


class EarlyStopping:
    """train_model.py:11-39, same constructor and step(): stop after `patience` epochs without an improvement of more than
    `min_delta`; on stopping put the best epoch's weights back.  `model` is anything with state_dict() / load_state_dict()
    (the reference passes its nn.Module, fit() passes the GraspTrainer)."""

    def __init__(self, patience=15, min_delta=0.001, restore_best_weights=True):
        self.patience, self.min_delta, self.restore_best_weights = patience, min_delta, restore_best_weights
        self.counter, self.best_loss, self.best_epoch, self.best_weights = 0, None, None, None

    def _snapshot(self, model):
        return {k: (v.cpu().clone() if hasattr(v, "clone") else v) for k, v in model.state_dict().items()}

    def step(self, val_loss, epoch, model):
        if self.best_loss is None:
            self.best_loss, self.best_epoch = val_loss, epoch
            if self.restore_best_weights:
                self.best_weights = self._snapshot(model)
        elif val_loss > self.best_loss - self.min_delta:
            self.counter += 1
            if self.counter >= self.patience:
                if self.restore_best_weights and self.best_weights is not None:
                    model.load_state_dict(self.best_weights)
                return True
        else:
            self.best_loss, self.best_epoch, self.counter = val_loss, epoch, 0
            if self.restore_best_weights:
                self.best_weights = self._snapshot(model)
        return False


class PlateauScheduler:
    """torch.optim.lr_scheduler.ReduceLROnPlateau(mode='min', factor, patience, min_lr) as the script configures it
    (train_model.py:223-229; threshold 1e-4 relative, no cooldown, eps 1e-8) driving GraspTrainer.lr."""

    def __init__(self, trainer, factor=0.5, patience=5, min_lr=1e-6, threshold=1e-4, eps=1e-8):
        self.trainer, self.factor, self.patience, self.min_lr, self.threshold, self.eps = trainer, factor, patience, min_lr, threshold, eps
        self.best, self.num_bad_epochs = float("inf"), 0

    def step(self, metric):
        if metric < self.best * (1.0 - self.threshold):
            self.best, self.num_bad_epochs = metric, 0
        else:
            self.num_bad_epochs += 1
        if self.num_bad_epochs > self.patience:
            new_lr = max(self.trainer.lr * self.factor, self.min_lr)
            if self.trainer.lr - new_lr > self.eps:
                self.trainer.lr = new_lr
            self.num_bad_epochs = 0


def normalize_data(depth_patches, score_patches):
    """train_model.py:41-62 -- global z-score of the depth patches, per-channel z-score of the score patches."""
    depth_mean, depth_std = depth_patches.mean(), depth_patches.std()
    score_mean = score_patches.mean(dim=(0, 2, 3), keepdim=True)
    score_std = score_patches.std(dim=(0, 2, 3), keepdim=True)
    return {"depth_patches": (depth_patches - depth_mean) / depth_std,
            "score_patches": (score_patches - score_mean) / score_std,
            "stats": {"depth_mean": depth_mean, "depth_std": depth_std, "score_mean": score_mean, "score_std": score_std}}


def prepare_features(data, device=None):
    """train_model.py:167-180 -- the training_data.pt dict (EnhancedGraspDataCollector's layout: 'depth_patches' [n,32,32],
    'mask_patches' [n,32,32], 'score_patches' [n,7,32,32], 'labels' [n]) -> (features [n,9,32,32], labels [n] float,
    normalization stats): depth and mask get their channel axis, depth / scores are z-scored by normalize_data, channels
    are concatenated as [depth, mask, scores]."""
    to = (lambda t: t.to(device)) if device is not None else (lambda t: t)
    depth = to(torch.as_tensor(data["depth_patches"]).float()).unsqueeze(1)
    mask = to(torch.as_tensor(data["mask_patches"]).float()).unsqueeze(1)
    scores = to(torch.as_tensor(data["score_patches"]).float())
    labels = to(torch.as_tensor(data["labels"]).float())
    normalized = normalize_data(depth, scores)
    features = torch.cat([normalized["depth_patches"], mask, normalized["score_patches"]], dim=1)
    return features, labels, normalized["stats"]


def analyze_predictions(outputs, labels, threshold=0.5):
    """train_model.py:64-99 (same keys; note the reference thresholds the LOGITS at 0.5 here)."""
    predicted = (outputs.squeeze() > threshold).float()
    correct_pos = ((predicted == 1) & (labels == 1)).sum().item()
    correct_neg = ((predicted == 0) & (labels == 0)).sum().item()
    total_pos, total_neg = (labels == 1).sum().item(), (labels == 0).sum().item()
    tp, fp, fn, tn = correct_pos, total_neg - correct_neg, total_pos - correct_pos, correct_neg
    precision = tp / (tp + fp) if tp + fp > 0 else 0
    recall = tp / (tp + fn) if tp + fn > 0 else 0
    f1 = 2 * precision * recall / (precision + recall) if precision + recall > 0 else 0
    return {"positive_accuracy": (correct_pos / total_pos if total_pos else 0) * 100,
            "negative_accuracy": (correct_neg / total_neg if total_neg else 0) * 100,
            "precision": precision * 100, "recall": recall * 100, "f1_score": f1 * 100,
            "confusion_matrix": {"true_positive": tp, "false_positive": fp, "false_negative": fn, "true_negative": tn}}


def parameter_layout(filters, attention_type="spatial", in_channels=9):
    """[(state_dict key, shape)] in model.parameters() order, and the same for the BatchNorm buffers
    (running_mean, running_var per BatchNorm in module order) -- the flat vectors of lg_train_set_state."""
    params, buffers = [], []
    c = in_channels
    for b, f in enumerate(filters):
        for conv, bn, cin in ((0, 1, c), (3, 4, f)):
            params += [(f"encoder.{b}.{conv}.weight", (f, cin, 3, 3)), (f"encoder.{b}.{conv}.bias", (f,)),
                       (f"encoder.{b}.{bn}.weight", (f,)), (f"encoder.{b}.{bn}.bias", (f,))]
            buffers += [(f"encoder.{b}.{bn}.running_mean", (f,)), (f"encoder.{b}.{bn}.running_var", (f,))]
        c = f
    F = filters[-1]
    if attention_type == "spatial":
        params += [("attention.0.weight", (1, F, 1, 1)), ("attention.0.bias", (1,))]
    elif attention_type == "channel":           # model.py:37-44
        params += [("attention.1.weight", (F // 16, F, 1, 1)), ("attention.1.bias", (F // 16,)),
                   ("attention.3.weight", (F, F // 16, 1, 1)), ("attention.3.bias", (F,))]
    elif attention_type == "hybrid":            # model.py:45-58
        params += [("spatial_attention.0.weight", (1, F, 1, 1)), ("spatial_attention.0.bias", (1,)),
                   ("channel_attention.1.weight", (F // 16, F, 1, 1)), ("channel_attention.1.bias", (F // 16,)),
                   ("channel_attention.3.weight", (F, F // 16, 1, 1)), ("channel_attention.3.bias", (F,))]
    for idx, i, o in ((0, F, F), (4, F, F // 2), (8, F // 2, F // 4), (12, F // 4, 1)):
        params += [(f"classifier.{idx}.weight", (o, i)), (f"classifier.{idx}.bias", (o,))]
        if idx != 12:
            params += [(f"classifier.{idx + 1}.weight", (o,)), (f"classifier.{idx + 1}.bias", (o,))]
            buffers += [(f"classifier.{idx + 1}.running_mean", (o,)), (f"classifier.{idx + 1}.running_var", (o,))]
    return params, buffers


def dropout_layout(filters):
    """[(width, p)] of the dropout layers in module order: Dropout2d(0.3) per encoder block (model.py:25),
    Dropout(0.5), Dropout(0.5), Dropout(0.4) in the classifier (:70-80).  A mask block is [N][width]."""
    F = filters[-1]
    return [(f, 0.3) for f in filters] + [(F, 0.5), (F // 2, 0.5), (F // 4, 0.4)]


class GraspTrainer:
    """One GraspPointCNN + Adam optimizer living on the device behind lg_train_*."""

    def __init__(self, device, attention_type="spatial", encoder_filters=(64, 128, 256), max_batch=16, lr=0.0005,
                 weight_decay=0.01, betas=(0.9, 0.999), eps=1e-8, max_grad_norm=1.0, pos_weight=2.0, seed=42):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("GraspTrainer needs a HIP device ('cuda'): no CPU fallback")
        if attention_type not in _ATT:
            raise ValueError(f"attention_type must be one of {sorted(_ATT)}, not {attention_type!r}")
        self.attention_type, self.filters = attention_type, tuple(int(f) for f in encoder_filters)
        self.hp = LgTrainHparams(lr, betas[0], betas[1], eps, weight_decay, max_grad_norm, pos_weight)
        self.seed, self.max_batch = int(seed), int(max_batch)
        self._params, self._buffers = parameter_layout(self.filters, attention_type)
        self._h = _VP()
        arr = (C.c_int32 * len(self.filters))(*self.filters)
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        rc = lib.lg_train_create(idx, len(self.filters), arr, _ATT[attention_type], self.max_batch, C.byref(self._h))
        if rc != 0:
            raise LgError(f"lg_train_create failed ({rc})")
        n_p, n_b, mrow = C.c_int64(), C.c_int64(), C.c_int64()
        lib.lg_train_sizes(self._h, C.byref(n_p), C.byref(n_b), C.byref(mrow))
        self.n_params, self.n_buffers, self.mask_row = n_p.value, n_b.value, mrow.value
        assert self.n_params == sum(int(np.prod(s)) for _, s in self._params)
        assert self.n_buffers == sum(int(np.prod(s)) for _, s in self._buffers)
        assert self.mask_row == sum(w for w, _ in dropout_layout(self.filters))
        self.num_batches_tracked = 0
        self.initialize_weights(seed)

    def __del__(self):
        try:
            if self._h:
                lib.lg_train_destroy(self._h)
                self._h = _VP()
        except Exception:  # noqa: BLE001
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise LgError(f"{what} failed ({rc}): {lib.lg_train_last_error(self._h).decode()}")

    # ------------------------------------------------------------------ state
    def initialize_weights(self, seed=42):
        """GraspPointCNN._initialize_weights (model.py:87-99): kaiming_normal_ (fan_out / relu for convs, default for
        linears), zero biases, BatchNorm weight 1 / bias 0, running statistics 0 / 1."""
        g = torch.Generator().manual_seed(int(seed))
        sd = {}
        for name, shape in self._params:
            if len(shape) == 4:      # conv: fan_out = cout * kh * kw
                std = (2.0 / (shape[0] * shape[2] * shape[3])) ** 0.5
                sd[name] = torch.randn(shape, generator=g) * std
            elif len(shape) == 2:    # linear: fan_in, gain sqrt(2) (kaiming_normal_ defaults: a=0, leaky_relu)
                sd[name] = torch.randn(shape, generator=g) * (2.0 / shape[1]) ** 0.5
            else:
                is_bn_w = name.endswith(".weight")
                sd[name] = torch.ones(shape) if is_bn_w else torch.zeros(shape)
        for name, shape in self._buffers:
            sd[name] = torch.ones(shape) if name.endswith("running_var") else torch.zeros(shape)
        self.load_state_dict(sd)

    @staticmethod
    def _flat(sd, layout):
        return np.ascontiguousarray(np.concatenate([np.asarray(
            sd[k].detach().cpu().numpy() if hasattr(sd[k], "detach") else sd[k], dtype=np.float32).reshape(-1)
            for k, _ in layout]))

    def _unflat(self, vec, layout):
        out, o = {}, 0
        for k, shape in layout:
            n = int(np.prod(shape))
            out[k] = torch.from_numpy(vec[o:o + n].reshape(shape).copy())
            o += n
        return out

    def load_state_dict(self, sd, optimizer_state=None):
        """Module state dict (reference key names).  optimizer_state = {'exp_avg': {key: t}, 'exp_avg_sq': {...}, 'step': n}
        replaces the Adam state as well; None leaves it as it is (a fresh trainer starts from zeros / step 0)."""
        for k, shape in self._params + self._buffers:
            if k not in sd or tuple(sd[k].shape) != tuple(shape):
                raise ValueError(f"state_dict entry {k!r} missing or of wrong shape (expected {shape})")
        p, b = self._flat(sd, self._params), self._flat(sd, self._buffers)
        if optimizer_state is not None:
            m = self._flat(optimizer_state["exp_avg"], self._params)
            v = self._flat(optimizer_state["exp_avg_sq"], self._params)
            step = int(optimizer_state["step"])
        else:   # nn.Module.load_state_dict leaves the optimizer alone: keep the moments and the step count
            cur = self._get(m=True, v=True)
            m, v, step = cur["exp_avg"], cur["exp_avg_sq"], int(cur["step"])
        nbt = [sd[k] for k in sd if k.endswith("num_batches_tracked")]
        self.num_batches_tracked = int(nbt[0]) if nbt else 0
        self._check(lib.lg_train_set_state(self._h, p.ctypes.data_as(_FP), b.ctypes.data_as(_FP),
                                           m.ctypes.data_as(_FP) if m is not None else None,
                                           v.ctypes.data_as(_FP) if v is not None else None, step), "lg_train_set_state")

    def _get(self, params=False, buffers=False, m=False, v=False, grads=False):
        outs = {}
        ptr = {}
        for key, want, n in (("params", params, self.n_params), ("buffers", buffers, self.n_buffers),
                             ("exp_avg", m, self.n_params), ("exp_avg_sq", v, self.n_params), ("grads", grads, self.n_params)):
            if want:
                outs[key] = np.empty(n, np.float32)
                ptr[key] = outs[key].ctypes.data_as(_FP)
            else:
                ptr[key] = None
        step = C.c_int64()
        self._check(lib.lg_train_get_state(self._h, ptr["params"], ptr["buffers"], ptr["exp_avg"], ptr["exp_avg_sq"],
                                           ptr["grads"], C.byref(step)), "lg_train_get_state")
        outs["step"] = step.value
        return outs

    def state_dict(self):
        """The reference module's state_dict (CPU tensors), num_batches_tracked included."""
        o = self._get(params=True, buffers=True)
        sd = self._unflat(o["params"], self._params)
        sd.update(self._unflat(o["buffers"], self._buffers))
        for k, _ in self._buffers:
            if k.endswith("running_var"):
                sd[k.replace("running_var", "num_batches_tracked")] = torch.tensor(self.num_batches_tracked)
        return sd

    def gradients(self):
        """Unclipped gradients of the last step, keyed like the parameters."""
        return self._unflat(self._get(grads=True)["grads"], self._params)

    def optimizer_state(self):
        o = self._get(m=True, v=True)
        return {"exp_avg": self._unflat(o["exp_avg"], self._params),
                "exp_avg_sq": self._unflat(o["exp_avg_sq"], self._params), "step": o["step"]}

    def torch_optimizer_state_dict(self):
        """The optimizer state in torch.optim.Adam.state_dict() layout ('state' by parameter index in model.parameters()
        order, one param group): what the reference stores as 'optimizer_state_dict' (train_model.py:326) and what
        torch.optim.Adam(model.parameters(), ...).load_state_dict accepts."""
        o = self.optimizer_state()
        state = {i: {"step": torch.tensor(float(o["step"])), "exp_avg": o["exp_avg"][k], "exp_avg_sq": o["exp_avg_sq"][k]}
                 for i, (k, _) in enumerate(self._params)} if o["step"] > 0 else {}
        group = {"lr": float(self.hp.lr), "betas": (float(self.hp.beta1), float(self.hp.beta2)), "eps": float(self.hp.eps),
                 "weight_decay": float(self.hp.weight_decay), "amsgrad": False, "maximize": False, "foreach": None,
                 "capturable": False, "differentiable": False, "fused": None, "params": list(range(len(self._params)))}
        return {"state": state, "param_groups": [group]}

    def load_torch_optimizer_state_dict(self, osd, module_state_dict=None):
        """Inverse of torch_optimizer_state_dict (resume from a checkpoint written by the reference or by fit())."""
        st = osd["state"]
        step = int(st[0]["step"]) if st else 0
        zeros = {k: torch.zeros(shape) for k, shape in self._params}
        m = {k: (st[i]["exp_avg"] if i in st else zeros[k]) for i, (k, _) in enumerate(self._params)}
        v = {k: (st[i]["exp_avg_sq"] if i in st else zeros[k]) for i, (k, _) in enumerate(self._params)}
        g = osd["param_groups"][0]
        self.hp.lr, (self.hp.beta1, self.hp.beta2) = float(g["lr"]), (float(g["betas"][0]), float(g["betas"][1]))
        self.hp.eps, self.hp.weight_decay = float(g["eps"]), float(g["weight_decay"])
        self.load_state_dict(module_state_dict if module_state_dict is not None else self.state_dict(),
                             {"exp_avg": m, "exp_avg_sq": v, "step": step})

    @property
    def lr(self):
        return self.hp.lr

    @lr.setter
    def lr(self, value):
        self.hp.lr = float(value)

    # ------------------------------------------------------------------ the step
    def _masks_to_device(self, masks, N):
        """masks: list of [N, width] keep masks (0 or 1/(1-p)) per dropout layer, module order."""
        lay = dropout_layout(self.filters)
        if len(masks) != len(lay):
            raise ValueError(f"expected {len(lay)} dropout masks")
        flat = []
        for mk, (w, _) in zip(masks, lay):
            mk = torch.as_tensor(mk, dtype=torch.float32)
            if tuple(mk.shape) != (N, w):
                raise ValueError(f"dropout mask of shape {tuple(mk.shape)}, expected {(N, w)}")
            flat.append(mk.reshape(-1))
        return torch.cat(flat).to(self.device).contiguous()

    def train_step(self, batch_x, batch_y, masks=None, apply_update=True, return_logits=False):
        """One iteration of train_model.py:247-265.  batch_x [N,9,32,32], batch_y [N] in {0,1}.  Returns the loss
        (float), or (loss, logits [N] device tensor, total gradient norm before clipping) with return_logits."""
        x = torch.as_tensor(batch_x).to(self.device, torch.float32).contiguous()
        y = torch.as_tensor(batch_y).to(self.device, torch.float32).contiguous()
        N = x.shape[0]
        if tuple(x.shape) != (N, 9, 32, 32) or tuple(y.shape) != (N,):
            raise ValueError("batch_x must be [N,9,32,32] and batch_y [N]")
        if not 2 <= N <= self.max_batch:
            raise ValueError(f"batch size {N} outside [2, max_batch={self.max_batch}] (BatchNorm in train mode needs N > 1)")
        mk = self._masks_to_device(masks, N) if masks is not None else None
        logits = torch.empty(N, dtype=torch.float32, device=self.device)
        loss, gnorm = C.c_float(), C.c_float()
        torch.cuda.current_stream(self.device).synchronize()   # the library works on its own stream
        self._check(lib.lg_train_step(self._h, x.data_ptr(), y.data_ptr(), N, mk.data_ptr() if mk is not None else None,
                                      self.seed, C.byref(self.hp), 1 if apply_update else 0, C.byref(loss), C.byref(gnorm),
                                      logits.data_ptr()), "lg_train_step")
        self.num_batches_tracked += 1
        if return_logits:
            return loss.value, logits, gnorm.value
        return loss.value

    # ------------------------------------------------------------------ data parallel (one process per GPU)
    def gradient_tensor(self):
        """The library's flat gradient vector as a torch tensor (no copy): what the ranks all-reduce."""
        ptr, n = _FP(), C.c_int64()
        self._check(lib.lg_train_grad_buffer(self._h, C.byref(ptr), C.byref(n)), "lg_train_grad_buffer")

        class _Dev:   # __cuda_array_interface__ v2: torch wraps the device memory without owning it
            __cuda_array_interface__ = {"shape": (n.value,), "typestr": "<f4", "version": 2,
                                        "data": (C.cast(ptr, C.c_void_p).value, False)}
        return torch.as_tensor(_Dev(), device=self.device)

    def train_step_ddp(self, batch_x, batch_y, dist, masks=None):
        """Data-parallel step: every rank runs forward + backward on ITS shard of the batch, the flat gradient vector is
        averaged over the ranks with one all-reduce (RCCL when the process group's backend is 'nccl'), then every rank
        applies the same clip + Adam update -- the semantics of torch's DistributedDataParallel around the loop body of
        train_model.py:247-265 (mean of the per-rank mean losses; BatchNorm statistics per rank).  Returns the rank's loss."""
        loss = self.train_step(batch_x, batch_y, masks=masks, apply_update=False)   # synchronous: gradients are complete
        g = self.gradient_tensor()
        dist.all_reduce(g)
        g /= dist.get_world_size()
        torch.cuda.current_stream(self.device).synchronize()
        self._check(lib.lg_train_apply(self._h, C.byref(self.hp), None), "lg_train_apply")
        return loss

    # ------------------------------------------------------------------ evaluation (model.eval(): running statistics)
    def predict_logits(self, features, selector=None, batch=4096):
        """Eval-mode logits through the inference path (lg_cnn_forward with the current weights)."""
        from .grasp_point_selector import GraspPointSelector
        sel = selector or GraspPointSelector(self.device)
        sel.set_cnn_state_dict(self.state_dict())
        x = torch.as_tensor(features).to(self.device, torch.float32)
        return torch.cat([sel.cnn_forward(x[i:i + batch]) for i in range(0, x.shape[0], batch)])

    def bce_with_logits(self, logits, labels):
        """nn.BCEWithLogitsLoss(pos_weight) mean (validation loss, train_model.py:285)."""
        pw = torch.tensor([self.hp.pos_weight], device=logits.device)
        return torch.nn.functional.binary_cross_entropy_with_logits(logits, labels.to(logits), pos_weight=pw)

    # ------------------------------------------------------------------ epoch loop (train_model.py:155-356)
    def fit(self, features, labels, num_epochs=150, batch_size=16, val_fraction=0.2, save_dir=None, patience=15,
            min_delta=0.001, sched_factor=0.5, sched_patience=5, min_lr=1e-6, normalization_stats=None, log=print):
        """80/20 split, weighted sampling with replacement, ReduceLROnPlateau(min, 0.5, 5, min_lr 1e-6),
        EarlyStopping(15, 0.001, restore best weights), best_model.pth with the reference's checkpoint keys.
        The last incomplete batch of an epoch is used when it has at least 2 samples (a batch of 1 raises in the
        reference: BatchNorm1d in train mode)."""
        gen = torch.Generator().manual_seed(self.seed)
        feats = torch.as_tensor(features, dtype=torch.float32).to(self.device)
        labs = torch.as_tensor(labels, dtype=torch.float32).to(self.device)
        n = labs.shape[0]
        perm = torch.randperm(n, generator=gen).to(self.device)
        n_tr = int((1.0 - val_fraction) * n)
        tr_i, va_i = perm[:n_tr], perm[n_tr:]
        tr_x, tr_y, va_x, va_y = feats[tr_i], labs[tr_i], feats[va_i], labs[va_i]
        pos_weight = (tr_y == 0).sum() / (tr_y == 1).sum()
        w = torch.ones_like(tr_y)
        w[tr_y == 0] = 1.0 / pos_weight
        from .grasp_point_selector import GraspPointSelector
        sel = GraspPointSelector(self.device)
        best_val = float("inf")
        early_stopping = EarlyStopping(patience=patience, min_delta=min_delta, restore_best_weights=True)
        scheduler = PlateauScheduler(self, factor=sched_factor, patience=sched_patience, min_lr=min_lr)
        train_losses, val_losses, metrics_history = [], [], []
        for epoch in range(num_epochs):
            idx = torch.multinomial(w.cpu(), n_tr, replacement=True, generator=gen).to(self.device)
            tot, nb, correct = 0.0, 0, 0
            for s in range(0, n_tr, batch_size):
                bi = idx[s:s + batch_size]
                if bi.numel() < 2:
                    continue
                loss, logits, _ = self.train_step(tr_x[bi], tr_y[bi], return_logits=True)
                tot, nb = tot + loss, nb + 1
                correct += ((torch.sigmoid(logits) > 0.5).float() == tr_y[bi]).sum().item()
            train_losses.append(tot / max(nb, 1))
            vl = self.predict_logits(va_x, selector=sel)
            # the reference averages per-batch means over len(val_loader) (:306)
            vb = [self.bce_with_logits(vl[s:s + batch_size], va_y[s:s + batch_size]).item()
                  for s in range(0, va_y.shape[0], batch_size)]
            val_loss = float(np.mean(vb)) if vb else float("nan")
            val_losses.append(val_loss)
            metrics = analyze_predictions(vl, va_y)
            metrics_history.append(metrics)
            scheduler.step(val_loss)
            if log:
                log(f"Epoch [{epoch + 1}/{num_epochs}]: train {train_losses[-1]:.4f} val {val_loss:.4f} "
                    f"acc {100.0 * correct / max(n_tr, 1):.2f}% lr {self.lr:.6f} f1 {metrics['f1_score']:.2f}%")
            if val_loss < best_val:
                best_val = val_loss
                if save_dir:
                    os.makedirs(save_dir, exist_ok=True)
                    torch.save({"epoch": epoch, "model_state_dict": self.state_dict(),
                                "optimizer_state_dict": self.torch_optimizer_state_dict(), "val_loss": best_val,
                                "metrics": metrics,
                                "normalization_stats": normalization_stats, "train_losses": train_losses,
                                "val_losses": val_losses, "metrics_history": metrics_history},
                               os.path.join(save_dir, "best_model.pth"))
            if early_stopping.step(val_loss, epoch, self):   # restores the best epoch's module state (:21-39)
                if log:
                    log(f"Early stopping triggered! Best epoch was {early_stopping.best_epoch + 1}")
                break
        return {"train_losses": train_losses, "val_losses": val_losses, "metrics_history": metrics_history,
                "best_val_loss": best_val, "best_epoch": early_stopping.best_epoch}


def train_grasp_model(data_path=None, save_dir=None, device="cuda:0", num_epochs=150, log=print, **trainer_kwargs):
    """scripts/train_model.py::train_grasp_model (:155-395) on this path: torch.manual_seed(42), load
    ~/leaf_grasp_output/ml_training_data/training_data.pt (EnhancedGraspDataCollector's file; loaded with weights_only=True),
    normalise, 80/20 split, weighted sampling, Adam / ReduceLROnPlateau / EarlyStopping, best_model.pth + final_model.pth in
    ~/leaf_grasp_output/ml_models.  The plots of the script (:101-153) are not produced.  Returns the history dict of fit()."""
    torch.manual_seed(42)
    data_path = data_path or os.path.expanduser("~/leaf_grasp_output/ml_training_data/training_data.pt")
    save_dir = save_dir or os.path.expanduser("~/leaf_grasp_output/ml_models")
    data = torch.load(data_path, map_location="cpu", weights_only=True)
    if log:
        log(f"Loaded {len(data['labels'])} samples")
    features, labels, stats = prepare_features(data)
    if log:
        log(f"Combined features shape: {tuple(features.shape)}")
    trainer = GraspTrainer(torch.device(device), **trainer_kwargs)
    hist = trainer.fit(features, labels, num_epochs=num_epochs, save_dir=save_dir, normalization_stats=stats, log=log)
    torch.save({"epoch": len(hist["val_losses"]) - 1, "model_state_dict": trainer.state_dict(),
                "optimizer_state_dict": trainer.torch_optimizer_state_dict(),
                "val_loss": hist["val_losses"][-1] if hist["val_losses"] else None,
                "metrics": hist["metrics_history"][-1] if hist["metrics_history"] else None,
                "normalization_stats": stats, "train_losses": hist["train_losses"], "val_losses": hist["val_losses"],
                "metrics_history": hist["metrics_history"]}, os.path.join(save_dir, "final_model.pth"))
    if log:
        log(f"Best validation loss: {hist['best_val_loss']:.4f}")
    hist["trainer"] = trainer
    return hist


if __name__ == "__main__":
    train_grasp_model()
