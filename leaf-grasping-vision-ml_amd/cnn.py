"""GraspPointCNN weights -> lg_cnn_weights (include/leafgrasp.h).  State-dict layout of
scripts/utils/ml_grasp_optimizer/model.py::GraspPointCNN(in_channels=9, attention_type, encoder_filters [64,128,256]);
checkpoint key 'model_state_dict' (grasp_point_selector.py:48-49).  The attention variant ('spatial' = the node's
default, 'channel', 'hybrid', 'none'; model.py:30-60) is recognised from the keys.  Other encoder_filters are rejected."""
import ctypes as C

import numpy as np

from ._lib import LgCnnWeights

_FP = C.POINTER(C.c_float)

EXPECTED_SHAPES = {}
_c = 9
for _b, _f in enumerate((64, 128, 256)):
    for _conv, _bn, _cin in ((0, 1, _c), (3, 4, _f)):
        EXPECTED_SHAPES[f"encoder.{_b}.{_conv}.weight"] = (_f, _cin, 3, 3)
        EXPECTED_SHAPES[f"encoder.{_b}.{_conv}.bias"] = (_f,)
        for _s in ("weight", "bias", "running_mean", "running_var"):
            EXPECTED_SHAPES[f"encoder.{_b}.{_bn}.{_s}"] = (_f,)
    _c = _f
ATT_SPATIAL, ATT_CHANNEL, ATT_HYBRID, ATT_NONE = 0, 1, 2, 3   # include/leafgrasp.h LG_ATT_*
for _p in ("attention.0", "spatial_attention.0"):
    EXPECTED_SHAPES[f"{_p}.weight"] = (1, 256, 1, 1)
    EXPECTED_SHAPES[f"{_p}.bias"] = (1,)
for _p in ("attention", "channel_attention"):
    EXPECTED_SHAPES[f"{_p}.1.weight"] = (16, 256, 1, 1)
    EXPECTED_SHAPES[f"{_p}.1.bias"] = (16,)
    EXPECTED_SHAPES[f"{_p}.3.weight"] = (256, 16, 1, 1)
    EXPECTED_SHAPES[f"{_p}.3.bias"] = (256,)
for _li, (_idx, _i, _o) in enumerate(((0, 256, 256), (4, 256, 128), (8, 128, 64), (12, 64, 1))):
    EXPECTED_SHAPES[f"classifier.{_idx}.weight"] = (_o, _i)
    EXPECTED_SHAPES[f"classifier.{_idx}.bias"] = (_o,)
    if _idx != 12:
        for _s in ("weight", "bias", "running_mean", "running_var"):
            EXPECTED_SHAPES[f"classifier.{_idx + 1}.{_s}"] = (_o,)


def pack_state_dict(state_dict):
    """-> (LgCnnWeights, keepalive list).  Accepts torch tensors or numpy arrays; validates shapes."""
    keep = []

    def arr(key):
        if key not in state_dict:
            raise KeyError(f"GraspPointCNN state_dict is missing '{key}'")
        v = state_dict[key]
        if hasattr(v, "detach"):
            v = v.detach().cpu().numpy()
        a = np.ascontiguousarray(v, dtype=np.float32)
        if tuple(a.shape) != EXPECTED_SHAPES[key]:
            raise ValueError(f"'{key}' has shape {tuple(a.shape)}, expected {EXPECTED_SHAPES[key]} "
                             "(only encoder_filters [64,128,256] with in_channels=9 is supported)")
        keep.append(a)
        return a.ctypes.data_as(_FP)

    w = LgCnnWeights()
    li = 0
    for b in range(3):
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
