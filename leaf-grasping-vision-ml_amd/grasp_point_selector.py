"""GraspPointSelector -- drop-in mirror of scripts/utils/grasp_point_selector.py::GraspPointSelector.

Same constructor, attributes (`camera_cx`, `camera_cy`, `f_norm`), method names, argument order, return
shapes and error convention (never raise across `select_grasp_point`: log and return (None, None, None),
grasp_point_selector.py:251-253).  All per-pixel work runs in liblgrasp.so (hand-written gfx950 HIP) through
the C-ABI in include/leafgrasp.h; tensors stay on the GPU and are handed over as raw device pointers.
There is no CPU fallback: constructing a selector without a HIP device raises.
"""
import ctypes as C
import math
import os

import numpy as np
import torch

from . import _lib
from ._lib import LG_NUM_MAPS, MAP_NAMES, LgGraspResult, check, lib
from ._log import logerr, loginfo, logwarn
from .cnn import pack_state_dict
from .image_processor import flatness_config

# LgGraspResult rows as a numpy record (select_grasp_points_batch reads whole columns)
_RESULT_DTYPE = np.dtype([(n, np.int32 if t is C.c_int else np.float32) for n, t in LgGraspResult._fields_])
assert _RESULT_DTYPE.itemsize == C.sizeof(LgGraspResult)

_VP = C.c_void_p


def _device_index(device):
    d = torch.device(device) if not isinstance(device, torch.device) else device
    if d.type != "cuda":
        raise RuntimeError(
            f"leafgrasp_amd needs a HIP device (got device '{d}'): the grasp-scoring path has no CPU fallback")
    return d.index if d.index is not None else torch.cuda.current_device()


class GraspPointSelector:
    def __init__(self, device, load_model=True):
        self.device = torch.device(device) if not isinstance(device, torch.device) else device
        # scoring weights / parameters kept as attributes like the reference (grasp_point_selector.py:17-33)
        self.flatness_weight = 0.25
        self.isolation_weight = 0.4
        self.edge_weight = 0.2
        self.accessibility_weight = 0.15
        self.min_flat_area_size = 15
        self.min_edge_distance = 20
        self.isolation_radius = 50
        self.camera_cx = 707
        self.camera_cy = 494
        self.f_norm = None
        self.erosion_kernel_size = 21
        self.erosion_iterations = 2
        self.params = _lib.default_params()
        self._h = _VP()
        self._dev_index = _device_index(self.device)
        check(None, lib.lg_create(self._dev_index, C.byref(self._h)), "lg_create")
        self.ml_predictor = None  # truthy object when CNN weights are loaded (reference attribute name)
        self._keep = None
        if load_model:
            self.load_ml_model()

    def __del__(self):
        try:
            if getattr(self, "_h", None) is not None and self._h.value:
                lib.lg_destroy(self._h)
                self._h = _VP()
        except Exception:  # noqa: BLE001
            pass

    # ------------------------------------------------------------------ model (grasp_point_selector.py:43-57)
    def load_ml_model(self, model_path=None):
        """Load trained CNN weights from ~/leaf_grasp_output/ml_models/best_model.pth (key
        'model_state_dict'); absent file => traditional scoring only, exactly like the reference."""
        try:
            model_path = model_path or os.path.expanduser("~/leaf_grasp_output/ml_models/best_model.pth")
            if os.path.exists(model_path):
                checkpoint = torch.load(model_path, map_location="cpu", weights_only=True)
                self.set_cnn_state_dict(checkpoint["model_state_dict"])
                loginfo("Loaded ML grasp model")
            else:
                logwarn("No ML model found, will use traditional scoring only")
                self.ml_predictor = None
        except Exception as e:  # noqa: BLE001
            logerr(f"Error loading ML model: {str(e)}")
            self.ml_predictor = None

    def set_cnn_state_dict(self, state_dict):
        """Install GraspPointCNN weights (state_dict of the reference module; BN is folded in the library)."""
        w, keep = pack_state_dict(state_dict)
        check(self._h, lib.lg_cnn_load(self._h, C.byref(w)), "lg_cnn_load")
        self._keep = keep
        self.ml_predictor = "lg_cnn"

    def clear_cnn(self):
        lib.lg_cnn_unload(self._h)
        self.ml_predictor = None

    # ------------------------------------------------------------------ camera (:145-150)
    def set_camera_params(self, projection_matrix):
        self.f_norm = projection_matrix[0, 0]
        self.camera_cx = projection_matrix[0, 2]
        self.camera_cy = projection_matrix[1, 2]
        self.baseline = -projection_matrix[0, 3] / self.f_norm

    def _sync_params(self, image_processor=None):
        if self.f_norm is None:
            # reference: np.full(..., None) in calculate_approach_vector_score raises -> caught -> None triple
            raise RuntimeError("camera parameters not set (f_norm is None): call set_camera_params first")
        p = self.params
        p.cx, p.cy, p.f = float(self.camera_cx), float(self.camera_cy), float(self.f_norm)
        p.min_edge_distance = float(self.min_edge_distance)
        # _calculate_flatness_map smooths with the CALLER's ImageProcessor (:635-657 -> image_processor.py:56-64): its
        # Gaussian size goes into lg_params (1 / 3 / 5 / 7; anything else is refused by the library, an even size fails in
        # the reference as well).  None (not possible in the reference) = the node's size 5.
        p.gaussian_size = 5 if image_processor is None else flatness_config(image_processor)
        return p

    # ------------------------------------------------------------------ tensors
    def _prep_inputs(self, leaf_mask, depth_tensor):
        if isinstance(leaf_mask, np.ndarray):
            leaf_mask = torch.from_numpy(leaf_mask)
        if isinstance(depth_tensor, np.ndarray):
            depth_tensor = torch.from_numpy(depth_tensor)
        is_bool = leaf_mask.dtype == torch.bool
        m = leaf_mask.to(self.device)
        if is_bool:
            m = m.contiguous().view(torch.uint8)  # bool storage is one 0/1 byte: reinterpret, no kernel
        elif m.dtype != torch.uint8:
            m = (m != 0).to(torch.uint8)
        m = m.contiguous()
        d = depth_tensor.to(self.device, torch.float32).contiguous()
        if m.dim() == 2:
            m, d = m.unsqueeze(0), d.unsqueeze(0)
        if m.shape != d.shape or m.dim() != 3:
            raise ValueError(f"mask {tuple(m.shape)} and depth {tuple(d.shape)} must both be [H,W] or [B,H,W]")
        return m, d, is_bool

    def _stream(self):
        return _VP(torch.cuda.current_stream(self.device).cuda_stream)

    def _alloc_maps(self, B, H, W, names=MAP_NAMES):
        maps = torch.empty((LG_NUM_MAPS, B, H, W), dtype=torch.float32, device=self.device)
        ptrs = (_VP * LG_NUM_MAPS)()
        for i, n in enumerate(MAP_NAMES):
            ptrs[i] = maps[i].data_ptr() if n in names else None
        return maps, ptrs

    # ------------------------------------------------------------------ score planes (:256-288)
    def score_maps(self, leaf_mask, depth_tensor, image_processor=None):
        """All eight planes + validity mask, device resident.
        Returns (dict name -> float32 tensor [B,H,W] or [H,W], valid uint8 tensor, theta list)."""
        m, d, _ = self._prep_inputs(leaf_mask, depth_tensor)
        B, H, W = m.shape
        p = self._sync_params(image_processor)
        maps, ptrs = self._alloc_maps(B, H, W)
        valid = torch.empty((B, H, W), dtype=torch.uint8, device=self.device)
        theta = (C.c_float * B)()
        with torch.cuda.device(self.device):
            check(self._h, lib.lg_score_maps(self._h, d.data_ptr(), m.data_ptr(), B, H, W, C.byref(p), C.byref(ptrs),
                                             valid.data_ptr(), theta, self._stream()), "lg_score_maps")
        squeeze = (leaf_mask.dim() if hasattr(leaf_mask, "dim") else leaf_mask.ndim) == 2
        out = {n: (maps[i, 0] if squeeze else maps[i]) for i, n in enumerate(MAP_NAMES)}
        th = [None if math.isnan(t) else float(t) for t in theta]
        return out, (valid[0] if squeeze else valid), (th[0] if squeeze else th)

    def dt_maxima(self, frame=0):
        """Inspection: (max d_in, max d_out) of `frame` in the last call, as float32 like cv2's planes
        (grasp_point_selector.py:529-533), and the sweep window (x0, x1, y0, y1) the distance transforms ran on."""
        import ctypes as C
        out, win = (C.c_uint32 * 2)(), (C.c_int32 * 4)()
        rc = lib.lg_debug_dt_max(self._h, int(frame), out, win)
        if rc != 0:
            raise RuntimeError(lib.lg_last_error(self._h).decode())
        scale = np.float32(1.0 / 65536.0)
        return np.float32(out[0]) * scale, np.float32(out[1]) * scale, tuple(win)

    def dt_form(self, frame=0):
        """Inspection: (searched, d_out sweeps skipped) of `frame` in the last call -- whether its d_in came from the row search
        or from the two sweeps is decided per batch on the device."""
        import ctypes as C
        form = (C.c_int32 * 2)()
        rc = lib.lg_debug_dt_form(self._h, int(frame), form)
        if rc != 0:
            raise RuntimeError(lib.lg_last_error(self._h).decode())
        return bool(form[0]), bool(form[1])

    def _calculate_all_scores(self, leaf_mask_np, depth_tensor, image_processor=None):
        """Reference signature (:256): numpy uint8 mask in, dict of numpy planes out."""
        out, _, _ = self.score_maps(leaf_mask_np, depth_tensor, image_processor)
        return {k: v.cpu().numpy() for k, v in out.items()}

    def _get_valid_regions(self, leaf_mask_np, scores):  # :282-288 (host helper for callers holding a scores dict)
        return ((scores["distance_map"] > self.min_edge_distance) & (np.asarray(leaf_mask_np) > 0)
                & (scores["stem_penalty"] < 0.8))

    # ------------------------------------------------------------------ candidates (:447-482)
    def _get_candidate_points(self, score_map, valid_regions, top_k=20, min_distance=10):
        try:
            sm = torch.as_tensor(np.asarray(score_map, dtype=np.float32) if not torch.is_tensor(score_map) else score_map)
            vr = torch.as_tensor(np.asarray(valid_regions) if not torch.is_tensor(valid_regions) else valid_regions)
            sm = sm.to(self.device, torch.float32).contiguous()
            vr = (vr.to(self.device) != 0).to(torch.uint8).contiguous()
            if sm.dim() == 2:
                sm, vr = sm.unsqueeze(0), vr.unsqueeze(0)
            B, H, W = sm.shape
            xy = torch.zeros((B, top_k, 2), dtype=torch.int32, device=self.device)
            n = torch.zeros((B,), dtype=torch.int32, device=self.device)
            with torch.cuda.device(self.device):
                check(self._h, lib.lg_topk_nms(self._h, sm.data_ptr(), vr.data_ptr(), B, H, W, int(top_k),
                                               int(min_distance), xy.data_ptr(), n.data_ptr(), self._stream()),
                      "lg_topk_nms")
            xy, n = xy.cpu().numpy(), n.cpu().numpy()
            res = [[(int(xy[b, i, 0]), int(xy[b, i, 1])) for i in range(int(n[b]))] for b in range(B)]
            return res[0] if B == 1 else res
        except Exception as e:  # noqa: BLE001
            logerr(f"Error getting candidate points: {str(e)}")
            return []

    # ------------------------------------------------------------------ patches + CNN (:59-143)
    def gather_patches(self, leaf_mask, depth_tensor, maps, points):
        """[len(points), 9, 32, 32] float32 device tensor of normalised patches (single frame)."""
        m, d, _ = self._prep_inputs(leaf_mask, depth_tensor)
        _, H, W = m.shape
        k = len(points)
        xy = torch.tensor(points, dtype=torch.int32, device=self.device).reshape(1, k, 2).contiguous()
        n = torch.tensor([k], dtype=torch.int32, device=self.device)
        ptrs = (_VP * LG_NUM_MAPS)()
        keep = []
        for i, name in enumerate(MAP_NAMES):
            t = maps[name]
            t = torch.as_tensor(t).to(self.device, torch.float32).contiguous()
            keep.append(t)
            ptrs[i] = t.data_ptr()
        patches = torch.empty((1, k, 9, 32, 32), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            check(self._h, lib.lg_gather_patches(self._h, d.data_ptr(), m.data_ptr(), C.byref(ptrs), 1, H, W, k,
                                                 xy.data_ptr(), n.data_ptr(), patches.data_ptr(), self._stream()),
                  "lg_gather_patches")
            torch.cuda.current_stream(self.device).synchronize()
        return patches[0]

    def cnn_forward(self, patches):
        """GraspPointCNN logits for [N,9,32,32] float32 patches (device tensor)."""
        x = torch.as_tensor(patches).to(self.device, torch.float32).contiguous()
        N = x.shape[0]
        logits = torch.empty((N,), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            check(self._h, lib.lg_cnn_forward(self._h, x.data_ptr(), N, logits.data_ptr(), self._stream()),
                  "lg_cnn_forward")
        return logits

    def get_ml_score(self, leaf_mask, depth_tensor, scores, point):
        """Reference signature (:59): CNN score of one point, None when no model / on error."""
        try:
            if self.ml_predictor is None:
                return None
            x, y = point
            m = torch.as_tensor(leaf_mask)
            H, W = m.shape[-2:]
            if m.dtype == torch.bool and (x < 16 or y < 16 or x + 16 > W or y + 16 > H):
                logwarn(f"Failed to extract mask patch at ({x}, {y})")
                return None
            patches = self.gather_patches(leaf_mask, depth_tensor, scores, [(int(x), int(y))])
            logit = float(self.cnn_forward(patches)[0])
            s = 1.0 / (1.0 + math.exp(-logit))
            return float(np.tanh(s * 3.0) * 0.5 + 0.5)
        except Exception as e:  # noqa: BLE001
            logerr(f"Error in ML scoring: {str(e)}")
            return None

    # ------------------------------------------------------------------ the path (:184-253)
    def select_grasp_points_batch(self, leaf_masks, depth_tensors, return_maps=False, image_processor=None):
        """B frames at once ([B,H,W] mask + depth).  Returns a list of (xy, XYZ, preXYZ) triples."""
        m, d, is_bool = self._prep_inputs(leaf_masks, depth_tensors)
        B, H, W = m.shape
        p = self._sync_params(image_processor)
        p.mask_is_bool = 1 if is_bool else 0
        maps = valid = None
        ptrs = None
        vptr = None
        if return_maps:
            maps, ptrs = self._alloc_maps(B, H, W)
            valid = torch.empty((B, H, W), dtype=torch.uint8, device=self.device)
            vptr = valid.data_ptr()
        res = (LgGraspResult * B)()
        with torch.cuda.device(self.device):
            check(self._h, lib.lg_select_grasp(self._h, d.data_ptr(), m.data_ptr(), B, H, W, C.byref(p),
                                               C.byref(ptrs) if ptrs is not None else None, vptr, res,
                                               self._stream()), "lg_select_grasp")
        # one structured view of the result rows instead of a ctypes attribute access per field and frame (0.12 -> 0.04 ms per 256
        # frames; float32 -> Python float conversions are the same values either way)
        a = np.frombuffer(res, dtype=_RESULT_DTYPE, count=B)
        cols = [a[n].tolist() for n in ("found", "x", "y", "X", "Y", "Z", "has_pre", "pX", "pY", "pZ")]
        out = [((x, y), (X, Y, Z), (pX, pY, pZ) if hp else None) if f else (None, None, None)
               for f, x, y, X, Y, Z, hp, pX, pY, pZ in zip(*cols)]
        self.last_results = res
        if return_maps:
            return out, {n: maps[i] for i, n in enumerate(MAP_NAMES)}, valid
        return out

    def select_grasp_points_for_leaves(self, label_tensors, leaf_ids, depth_tensors, image_processor=None):
        """select_grasp_points_batch(label_tensors == leaf_ids[:, None, None], depth_tensors) -- the node's `optimal_mask =
        mask_tensor == optimal_leaf_id` and select_grasp_point (leaf_grasp_node_v3.py:118-125) -- with the comparison folded into
        the library's first pass over the frames (lg_select_grasp_labels).  label_tensors: [B,H,W] int16 on this device; leaf_ids:
        B ints, None for a frame without a leaf (the node does not call select_grasp_point for it: its triple is (None, None,
        None)).  The mask counts as the torch.bool tensor the node passes."""
        lab = label_tensors
        if not (torch.is_tensor(lab) and lab.dtype == torch.int16 and lab.is_cuda and lab.dim() == 3 and lab.is_contiguous()):
            raise ValueError("label_tensors must be a contiguous [B,H,W] int16 tensor on the device")
        d = depth_tensors.to(self.device, torch.float32).contiguous()
        if d.shape != lab.shape:
            raise ValueError(f"labels {tuple(lab.shape)} and depth {tuple(d.shape)} must both be [B,H,W]")
        B, H, W = lab.shape
        p = self._sync_params(image_processor)
        p.mask_is_bool = 1
        ids = (C.c_int32 * B)(*[(-(1 << 30)) if i is None else int(i) for i in leaf_ids])
        res = (LgGraspResult * B)()
        with torch.cuda.device(self.device):
            check(self._h, lib.lg_select_grasp_labels(self._h, d.data_ptr(), lab.data_ptr(), ids, B, H, W, C.byref(p), None, None, res,
                                                      self._stream()), "lg_select_grasp_labels")
        for b, i in enumerate(leaf_ids):   # the node never calls select_grasp_point for a frame without a leaf (:113-116)
            if i is None:
                res[b].found = 0
        a = np.frombuffer(res, dtype=_RESULT_DTYPE, count=B)
        cols = [a[n].tolist() for n in ("found", "x", "y", "X", "Y", "Z", "has_pre", "pX", "pY", "pZ")]
        self.last_results = res
        return [((x, y), (X, Y, Z), (pX, pY, pZ) if hp else None) if f else (None, None, None)
                for f, x, y, X, Y, Z, hp, pX, pY, pZ in zip(*cols)]

    def select_grasp_point(self, leaf_mask, depth_tensor, image_processor=None, pcl_data=None):
        """Select optimal grasp point using combined traditional and ML approach (reference :184)."""
        try:
            res = self.select_grasp_points_batch(leaf_mask, depth_tensor, image_processor=image_processor)[0]
            if res[0] is None:
                logwarn("No valid candidate points found")
            elif pcl_data is not None:
                # get_3d_grasp_point's point-cloud branch reads self.width, which the class never defines (:164-178): the
                # reference raises there and ends in its logged None triple
                raise AttributeError("'GraspPointSelector' object has no attribute 'width'")
            return res
        except Exception as e:  # noqa: BLE001
            logerr(f"Error in grasp point selection: {str(e)}")
            return None, None, None

    # ------------------------------------------------------------------ small host-side helpers kept for callers
    def get_3d_grasp_point(self, grasp_point_2d, depth_tensor, pcl_data=None):  # :152-180
        u, v = grasp_point_2d
        depth_value = depth_tensor[v, u].item()
        return ((depth_value * (u - self.camera_cx)) / self.f_norm,
                (depth_value * (v - self.camera_cy)) / self.f_norm, depth_value)

    def _project_point_to_2d(self, point_3d):  # :821-826
        x, y, z = point_3d
        return (int((x * self.f_norm / z) + self.camera_cx), int((y * self.f_norm / z) + self.camera_cy))

    def estimate_leaf_orientation(self, leaf_mask_np):  # :718-752 (LeafVisualizer calls this, visualizer.py:77)
        """(angle_rad, major_axis, minor_axis, (cx, cy)) or four Nones."""
        try:
            m = torch.as_tensor(np.asarray(leaf_mask_np) if not torch.is_tensor(leaf_mask_np) else leaf_mask_np)
            m = (m.to(self.device) != 0).to(torch.uint8).contiguous()
            H, W = m.shape
            out = (C.c_float * 5)()
            found = C.c_int(0)
            with torch.cuda.device(self.device):
                check(self._h, lib.lg_leaf_orientation(self._h, m.data_ptr(), H, W, out, C.byref(found),
                                                       self._stream()), "lg_leaf_orientation")
            if not found.value:
                return None, None, None, None
            return float(out[0]), float(out[1]), float(out[2]), (float(out[3]), float(out[4]))
        except Exception as e:  # noqa: BLE001
            logerr(f"Error in leaf orientation estimation: {str(e)}")
            return None, None, None, None
