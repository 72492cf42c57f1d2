"""OptimalLeafSelector -- drop-in mirror of scripts/utils/leaf_scorer.py::OptimalLeafSelector.

The L full-frame passes per leaf of the reference (boolean masks, fancy indexing, np.median, the global
distance field) are replaced by lg_leaf_stats (include/leafgrasp.h): a fixed number of streaming HIP passes
that return per-label area / centroid sums / depth sums / exact median depth / border flag and the two
clutter extrema.  The remaining arithmetic is O(#leaves) and stays on the host in NumPy with the
reference's dtypes (float32 medians and means, float64 scores).

Semantics note (DESIGN.md "Clutter field"): the reference calls skfmm.distance (scikit-fmm 2022.3.26, not
available) and only consumes the arg-extrema; this build defines the field as the exact Euclidean distance to
the nearest leaf pixel (arg-max with first-occurrence ties).  PARITY UNPINNED vs scikit-fmm.
"""
import ctypes as C

import math

import numpy as np
import torch

from ._lib import LG_ERR_INVALID, LgLeafStat, check, lib
from ._log import logerr, loginfo, logwarn
from .grasp_point_selector import _device_index

_VP = C.c_void_p
_MAX_LEAVES = 64      # result capacity of the first attempt; a frame with more labels (up to 1024) is re-read with room for them
_MAX_LABELS = 1024    # distinct labels per frame the library handles (lg_leaf.hip: LGL_MAXL)
# numpy view of lg_leaf_stat (include/leafgrasp.h; ctypes mirror: _lib.LgLeafStat)
_STAT_DTYPE = np.dtype([("id", "<i4"), ("area", "<i4"), ("touches_border", "<i4"), ("pad_", "<i4"), ("sum_x", "<f8"),
                        ("sum_y", "<f8"), ("sum_depth", "<f8"), ("sum_ray", "<f8"), ("median_depth", "<f4"), ("pad2_", "<f4")])
assert _STAT_DTYPE.itemsize == C.sizeof(LgLeafStat)


def pareto_mask_max(scores):
    """paretoset(scores, sense=['max']*k): non-dominated rows, first of identical rows kept
    (paretoset 1.2.3 is not installed; leaf_scorer.py:158,163).  Plain-Python comparisons on the float64 values (the
    per-frame host cost of the batched path is this function and its caller, not the device passes)."""
    rows = scores if isinstance(scores, list) else np.asarray(scores, np.float64).tolist()   # lists of Python floats as they are
    n = len(rows)
    keep = np.ones(n, bool)
    for i in range(n):
        ri = rows[i]
        for j in range(n):
            if i == j:
                continue
            rj = rows[j]
            ge = True
            gt = False
            for a, b in zip(rj, ri):
                if a < b:
                    ge = False
                    break
                if a > b:
                    gt = True
            if ge and (gt or j < i):
                keep[i] = False
                break
    return keep


class OptimalLeafSelector:
    def __init__(self, device):
        self.device = torch.device(device) if not isinstance(device, torch.device) else device
        self.camera_cx = None  # leaf_scorer.py:14-17
        self.camera_cy = None
        self.f_norm = None
        self._h = _VP()
        check(None, lib.lg_create(_device_index(self.device), C.byref(self._h)), "lg_create")

    def __del__(self):
        try:
            if getattr(self, "_h", None) is not None and self._h.value:
                lib.lg_destroy(self._h)
                self._h = _VP()
        except Exception:  # noqa: BLE001
            pass

    def set_camera_params(self, projection_matrix):  # :19-23
        self.f_norm = projection_matrix[0, 0]
        self.camera_cx = projection_matrix[0, 2]
        self.camera_cy = projection_matrix[1, 2]

    def get_tall_leaves(self):  # :205-207
        return self._tall_leaves if hasattr(self, "_tall_leaves") else []

    # ------------------------------------------------------------------ device pass
    def leaf_statistics(self, mask_tensor, depth_tensor):
        """-> (list of dict per label id ascending, (min_global (y,x), max_global (y,x)))."""
        lab = torch.as_tensor(mask_tensor).to(self.device)
        if lab.dtype != torch.int16:
            lab = lab.to(torch.int16)  # wire dtype of /leaves_masks (leaf_grasp_node_v3.py:188)
        lab = lab.contiguous()
        dep = torch.as_tensor(depth_tensor).to(self.device, torch.float32).contiguous()
        H, W = lab.shape
        cap = _MAX_LEAVES
        while True:
            stats = (LgLeafStat * cap)()
            n = C.c_int(0)
            ext = (C.c_int32 * 4)()
            with torch.cuda.device(self.device):
                rc = lib.lg_leaf_stats(self._h, lab.data_ptr(), dep.data_ptr(), H, W,
                                       float(self.camera_cx), float(self.camera_cy), float(self.f_norm),
                                       stats, cap, C.byref(n), ext,
                                       _VP(torch.cuda.current_stream(self.device).cuda_stream))
            if rc == LG_ERR_INVALID and cap < n.value <= _MAX_LABELS:   # more labels than the result array holds: once more
                cap = n.value
                continue
            check(self._h, rc, "lg_leaf_stats")
            break
        out = []
        for i in range(n.value):
            s = stats[i]
            out.append(dict(id=int(s.id), area=int(s.area), touches_border=bool(s.touches_border),
                            sum_x=float(s.sum_x), sum_y=float(s.sum_y), sum_depth=float(s.sum_depth),
                            sum_ray=float(s.sum_ray), median_depth=np.float32(s.median_depth)))
        return out, ((int(ext[0]), int(ext[1])), (int(ext[2]), int(ext[3]))), (H, W)

    def leaf_statistics_batch(self, mask_tensors, depth_tensors):
        """B frames [B,H,W] per call (lg_leaf_stats_batch) -> list of (stats, extrema, (H, W)) or None for a frame the
        library could not handle (more than 1024 labels)."""
        lab = torch.as_tensor(mask_tensors).to(self.device)
        if lab.dtype != torch.int16:
            lab = lab.to(torch.int16)
        lab = lab.contiguous()
        dep = torch.as_tensor(depth_tensors).to(self.device, torch.float32).contiguous()
        B, H, W = lab.shape
        cap = _MAX_LEAVES
        while True:
            stats = (LgLeafStat * (cap * B))()
            n = (C.c_int * B)()
            ext = (C.c_int32 * (4 * B))()
            status = (C.c_int * B)()
            with torch.cuda.device(self.device):
                check(self._h, lib.lg_leaf_stats_batch(self._h, lab.data_ptr(), dep.data_ptr(), B, H, W,
                                                       float(self.camera_cx), float(self.camera_cy), float(self.f_norm),
                                                       stats, cap, n, ext, status,
                                                       _VP(torch.cuda.current_stream(self.device).cuda_stream)),
                      "lg_leaf_stats_batch")
            need = max([n[b] for b in range(B) if status[b] == LG_ERR_INVALID], default=0)
            if cap < need <= _MAX_LABELS:       # some frame has more labels than the result rows hold: once more with room
                cap = need
                continue
            break
        # one structured view over the whole result instead of a ctypes attribute read per field (8 fields x ~8 leaves x B
        # frames were 2 of the 3 ms of host time per 128 frames)
        arr = np.frombuffer(stats, dtype=_STAT_DTYPE).reshape(B, cap)
        nn, st, ex = np.frombuffer(n, np.int32), np.frombuffer(status, np.int32), np.frombuffer(ext, np.int32).reshape(B, 4).tolist()
        out = []
        for b in range(B):
            if st[b] != 0:
                out.append(None)
                continue
            a = arr[b, :nn[b]]
            med = a["median_depth"]
            fs = [dict(id=i, area=ar, touches_border=bool(tb), sum_x=sx, sum_y=sy, sum_depth=sd, sum_ray=sr, median_depth=med[k])
                  for k, (i, ar, tb, sx, sy, sd, sr) in enumerate(zip(a["id"].tolist(), a["area"].tolist(),
                                                                    a["touches_border"].tolist(), a["sum_x"].tolist(),
                                                                    a["sum_y"].tolist(), a["sum_depth"].tolist(),
                                                                    a["sum_ray"].tolist()))]
            e = ex[b]
            out.append((fs, ((e[0], e[1]), (e[2], e[3])), (H, W)))
        return out

    def select_optimal_leaves_batch(self, mask_tensors, depth_tensors):
        """select_optimal_leaf for B frames in ONE library call (lg_leaf_select_batch: the device passes, then tall-leaf split,
        scores, Pareto filter and weighted pick on the host inside the library -- the same arithmetic as
        _select_from_statistics, without a Python loop over frames and leaves) -> list of leaf ids (or None)."""
        try:
            lab = torch.as_tensor(mask_tensors).to(self.device)
            if lab.dtype != torch.int16:
                lab = lab.to(torch.int16)
            lab = lab.contiguous()
            dep = torch.as_tensor(depth_tensors).to(self.device, torch.float32).contiguous()
            B, H, W = lab.shape
            tall_cap = 64
            ids = (C.c_int32 * B)()
            n_tall = (C.c_int32 * B)()
            tall = (C.c_int32 * (B * tall_cap))()
            with torch.cuda.device(self.device):
                check(self._h, lib.lg_leaf_select_batch(self._h, lab.data_ptr(), dep.data_ptr(), B, H, W, float(self.camera_cx),
                                                        float(self.camera_cy), float(self.f_norm), ids, n_tall, tall, tall_cap,
                                                        _VP(torch.cuda.current_stream(self.device).cuda_stream)),
                      "lg_leaf_select_batch")
        except Exception as e:  # noqa: BLE001
            logerr(f"Error in leaf selection: {str(e)}")
            return [None] * len(mask_tensors)
        # (one conversion per array: a ctypes element access per frame and tall leaf cost 0.1 ms per 128 frames)
        ids_l = np.frombuffer(ids, dtype=np.int32, count=B).tolist()
        nt_l = np.frombuffer(n_tall, dtype=np.int32, count=B).tolist()
        tall_l = np.frombuffer(tall, dtype=np.int32, count=B * tall_cap).reshape(B, tall_cap).tolist()
        out, tall_b = [], []
        for b, (i, nt) in enumerate(zip(ids_l, nt_l)):
            if i == -2 or nt > tall_cap:   # many labels / leaves: the general path for this frame
                out.append(self.select_optimal_leaf(lab[b], dep[b]))
                tall_b.append(self.get_tall_leaves() if out[-1] is not None else [])
                continue
            out.append(i if i >= 0 else None)
            tall_b.append(tall_l[b][:nt] if i >= 0 else [])
        self._tall_leaves_batch = tall_b
        return out

    def select_from_statistics_batch(self, per_frame):
        """The host half of select_optimal_leaves_batch: per-frame candidate scores, Pareto set and weighted pick from the
        statistics of leaf_statistics_batch (kept separate so that a caller can run it beside the next chunk's device pass)."""
        out, tall = [], []
        for fr in per_frame:
            if fr is None:
                logerr("Error in leaf selection: unsupported frame (more than 1024 labels)")
                out.append(None)
            else:
                out.append(self._select_from_statistics(*fr))
            tall.append(self.get_tall_leaves() if out[-1] is not None else [])
        self._tall_leaves_batch = tall
        return out

    # ------------------------------------------------------------------ the selection (:25-203)
    def select_optimal_leaf(self, mask_tensor, depth_tensor, return_debug=False):
        """Enhanced leaf selection with tall leaf consideration."""
        try:
            return self._select_from_statistics(*self.leaf_statistics(mask_tensor, depth_tensor), return_debug=return_debug)
        except Exception as e:  # noqa: BLE001  (:201-203)
            logerr(f"Error in leaf selection: {str(e)}")
            return None

    def _select_from_statistics(self, stats, extrema, shape, return_debug=False):
        try:
            (min_global, max_global), (H, W) = extrema, shape
            # torch.unique(mask)[1:] skips the smallest value (the background 0 when present, :32)
            total = sum(s["area"] for s in stats)
            if total == H * W and stats:
                stats = stats[1:]
            depth_list = [s["median_depth"] for s in stats if s["area"] > 0]
            if not depth_list:
                return None
            depth_mean = np.mean(np.array(depth_list))  # float32, like the reference (:53-54)
            tall_leaves = [s["id"] for s in stats if s["median_depth"] < depth_mean]  # :58-62
            loginfo(f"Found {len(tall_leaves)} tall leaves (average depth: {depth_mean:.3f}m)")
            candidates = []
            for s in stats:
                area = s["area"]
                if area < 10000:  # :79-81
                    continue
                centroid = (s["sum_x"] / area, s["sum_y"] / area)  # :84-88
                # (math.sqrt == np.sqrt bit for bit: both correctly rounded; np.exp below is kept, libm's exp may differ in the last ulp)
                dist_to_min = math.sqrt((centroid[0] - min_global[1]) ** 2 + (centroid[1] - min_global[0]) ** 2)
                dist_to_max = math.sqrt((centroid[0] - max_global[1]) ** 2 + (centroid[1] - max_global[0]) ** 2)
                total_dist = dist_to_min + dist_to_max
                clutter_score = dist_to_min / total_dist if total_dist > 0 else 0  # :91-101
                mean_depth = np.float32(s["sum_depth"] / area)  # np.mean of float32 depths (:105-106)
                # mean over pixels of sqrt(X^2+Y^2+Z^2) with X=(mean_depth*(x-cx))/f ... = mean_depth/f * mean ray (:109-115)
                mean_distance = float(mean_depth) / float(self.f_norm) * (s["sum_ray"] / area)
                distance_score = np.exp(-mean_distance / 0.3)  # :117
                if s["touches_border"]:  # :277-306
                    visibility_score = 0.0
                else:
                    d = math.sqrt((centroid[0] - W / 2) ** 2 + (centroid[1] - H / 2) ** 2)
                    visibility_score = 1.0 - d / math.sqrt((W / 2) ** 2 + (H / 2) ** 2)
                candidates.append({
                    "leaf_id": s["id"],
                    # Python floats: the same IEEE doubles as the reference's float64 array, without an ndarray per leaf
                    "scores": (float(clutter_score), float(distance_score), float(visibility_score)),
                    "raw_scores": {"clutter": clutter_score, "distance": mean_distance, "visibility": visibility_score},
                    "is_tall": s["id"] in tall_leaves,
                    "centroid": centroid,
                })
            if not candidates:
                logwarn("No valid leaf candidates found")
                return None
            try:
                tall_c = [c for c in candidates if c["is_tall"]]
                reg_c = [c for c in candidates if not c["is_tall"]]
                if tall_c:  # :150-160 (scores * 1.1: the same double products as the array expression)
                    pm = pareto_mask_max([[v * 1.1 for v in c["scores"]] for c in tall_c])
                    pareto = [c for i, c in enumerate(tall_c) if pm[i]]
                else:
                    pm = pareto_mask_max([list(c["scores"]) for c in reg_c])
                    pareto = [c for i, c in enumerate(reg_c) if pm[i]]
                if not pareto:
                    pareto = tall_c if tall_c else reg_c
                weights = np.array([0.35, 0.35, 0.3])  # :170
                best_score, best_leaf = float("-inf"), None
                self._tall_leaves = tall_leaves
                for c in pareto:
                    sc = c["scores"]   # np.sum of three float64 products adds them left to right
                    ws = 0.35 * float(sc[0]) + 0.35 * float(sc[1]) + 0.3 * float(sc[2])
                    if ws > best_score:
                        best_score, best_leaf = ws, c["leaf_id"]
                if return_debug:
                    for c in candidates:
                        c["scores"] = np.array(c["scores"], dtype=np.float64)
                    return best_leaf, dict(candidates=candidates, tall=tall_leaves, extrema=(min_global, max_global),
                                           depth_list=depth_list)
                return best_leaf
            except Exception as e:  # noqa: BLE001  (:198-202)
                logerr(f"Error in Pareto optimization: {str(e)}")
                if candidates:
                    return max(candidates, key=lambda x: np.mean(np.array(x["scores"])))["leaf_id"]
                return None
        except Exception as e:  # noqa: BLE001  (:201-203)
            logerr(f"Error in leaf selection: {str(e)}")
            return None

    def _calculate_visibility_score(self, leaf_mask):  # :277-306, host helper kept for callers
        leaf_mask = np.asarray(leaf_mask.cpu() if torch.is_tensor(leaf_mask) else leaf_mask).astype(bool)
        h, w = leaf_mask.shape
        ys, xs = np.where(leaf_mask)
        if len(ys) == 0:
            return 0.0
        if leaf_mask[0, :].sum() + leaf_mask[-1, :].sum() + leaf_mask[:, 0].sum() + leaf_mask[:, -1].sum() > 0:
            return 0.0
        d = np.sqrt((np.mean(xs) - w / 2) ** 2 + (np.mean(ys) - h / 2) ** 2)
        return 1.0 - d / np.sqrt((w / 2) ** 2 + (h / 2) ** 2)
