"""GraspPointCNN weights -> lg_cnn_weights (include/leafgrasp.h).  State-dict layout of
scripts/utils/ml_grasp_optimizer/model.py::GraspPointCNN(in_channels=9, attention_type, encoder_filters [64,128,256]);
checkpoint key 'model_state_dict' (grasp_point_selector.py:48-49).  The attention variant ('spatial' = the node's
default, 'channel', 'hybrid', 'none'; model.py:30-60) is recognised from the keys.  encoder_filters: the four configurations of the reference's sweep
(train_model_mlflow.py:177-182), recognised from the conv shapes."""
import ctypes as C

import numpy as np

from ._lib import LgCnnWeights

_FP = C.POINTER(C.c_float)

ATT_SPATIAL, ATT_CHANNEL, ATT_HYBRID, ATT_NONE = 0, 1, 2, 3   # include/leafgrasp.h LG_ATT_*
# encoder_filters of the reference's sweep (train_model_mlflow.py:177-182, mlflow_experiment_configs.py:86-112)
SUPPORTED_FILTERS = ((32, 64, 128), (64, 128, 256), (64, 128, 256, 512), (128, 256, 512))


def encoder_filters_of(state_dict):
    """encoder_filters from the conv shapes of the state dict (block b: encoder.{b}.0.weight is [f_b, cin, 3, 3])."""
    f = []
    while f"encoder.{len(f)}.0.weight" in state_dict:
        f.append(int(state_dict[f"encoder.{len(f)}.0.weight"].shape[0]))
    return tuple(f)


def expected_shapes(filters, in_channels=9):
    shapes = {}
    c = in_channels
    for b, f in enumerate(filters):
        for conv, bn, cin in ((0, 1, c), (3, 4, f)):
            shapes[f"encoder.{b}.{conv}.weight"] = (f, cin, 3, 3)
            shapes[f"encoder.{b}.{conv}.bias"] = (f,)
            for s_ in ("weight", "bias", "running_mean", "running_var"):
                shapes[f"encoder.{b}.{bn}.{s_}"] = (f,)
        c = f
    F = filters[-1]
    for p in ("attention.0", "spatial_attention.0"):
        shapes[f"{p}.weight"] = (1, F, 1, 1)
        shapes[f"{p}.bias"] = (1,)
    for p in ("attention", "channel_attention"):
        shapes[f"{p}.1.weight"] = (F // 16, F, 1, 1)
        shapes[f"{p}.1.bias"] = (F // 16,)
        shapes[f"{p}.3.weight"] = (F, F // 16, 1, 1)
        shapes[f"{p}.3.bias"] = (F,)
    for idx, i, o in ((0, F, F), (4, F, F // 2), (8, F // 2, F // 4), (12, F // 4, 1)):
        shapes[f"classifier.{idx}.weight"] = (o, i)
        shapes[f"classifier.{idx}.bias"] = (o,)
        if idx != 12:
            for s_ in ("weight", "bias", "running_mean", "running_var"):
                shapes[f"classifier.{idx + 1}.{s_}"] = (o,)
    return shapes


EXPECTED_SHAPES = expected_shapes((64, 128, 256))   # the node's default model


def pack_state_dict(state_dict):
    """-> (LgCnnWeights, keepalive list).  Accepts torch tensors or numpy arrays; validates shapes."""
    keep = []
    filters = encoder_filters_of(state_dict)
    if filters not in SUPPORTED_FILTERS:
        raise ValueError(f"GraspPointCNN encoder_filters {list(filters)} not supported (reference configurations: "
                         f"{[list(f) for f in SUPPORTED_FILTERS]}, in_channels=9)")
    shapes = expected_shapes(filters)

    def arr(key):
        if key not in state_dict:
            raise KeyError(f"GraspPointCNN state_dict is missing '{key}'")
        v = state_dict[key]
        if hasattr(v, "detach"):
            v = v.detach().cpu().numpy()
        a = np.ascontiguousarray(v, dtype=np.float32)
        if tuple(a.shape) != shapes[key]:
            raise ValueError(f"'{key}' has shape {tuple(a.shape)}, expected {shapes[key]} "
                             f"(encoder_filters {list(filters)}, in_channels=9)")
        keep.append(a)
        return a.ctypes.data_as(_FP)

    w = LgCnnWeights()
    w.n_blocks = len(filters)
    for i, f in enumerate(filters):
        w.filters[i] = f
    li = 0
    for b in range(len(filters)):
        for conv, bn in ((0, 1), (3, 4)):
            w.conv_w[li] = arr(f"encoder.{b}.{conv}.weight")
            w.conv_b[li] = arr(f"encoder.{b}.{conv}.bias")
            w.bn_g[li] = arr(f"encoder.{b}.{bn}.weight")
            w.bn_b[li] = arr(f"encoder.{b}.{bn}.bias")
            w.bn_m[li] = arr(f"encoder.{b}.{bn}.running_mean")
            w.bn_v[li] = arr(f"encoder.{b}.{bn}.running_var")
            li += 1
    def channel(prefix):
        w.ca_w1, w.ca_b1 = arr(f"{prefix}.1.weight"), arr(f"{prefix}.1.bias")
        w.ca_w2, w.ca_b2 = arr(f"{prefix}.3.weight"), arr(f"{prefix}.3.bias")

    if "spatial_attention.0.weight" in state_dict:      # 'hybrid'  (model.py:45-58)
        w.attention_type = ATT_HYBRID
        w.att_w, w.att_b = arr("spatial_attention.0.weight"), arr("spatial_attention.0.bias")
        channel("channel_attention")
    elif "attention.1.weight" in state_dict:            # 'channel' (model.py:37-44)
        w.attention_type = ATT_CHANNEL
        channel("attention")
    elif "attention.0.weight" in state_dict:            # 'spatial' (model.py:32-36)
        w.attention_type = ATT_SPATIAL
        w.att_w, w.att_b = arr("attention.0.weight"), arr("attention.0.bias")
    else:                                               # 'none'
        w.attention_type = ATT_NONE
    for i, idx in enumerate((0, 4, 8, 12)):
        w.fc_w[i] = arr(f"classifier.{idx}.weight")
        w.fc_b[i] = arr(f"classifier.{idx}.bias")
        if idx != 12:
            w.fbn_g[i] = arr(f"classifier.{idx + 1}.weight")
            w.fbn_b[i] = arr(f"classifier.{idx + 1}.bias")
            w.fbn_m[i] = arr(f"classifier.{idx + 1}.running_mean")
            w.fbn_v[i] = arr(f"classifier.{idx + 1}.running_var")
    w.bn_eps = 1e-5
    return w, keep
