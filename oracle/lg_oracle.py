"""
CPU ORACLE for the grasp-scoring hot path -- TEST INFRASTRUCTURE, NOT THE PRODUCT.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product (leaf-grasping-vision-ml_amd/) never imports it and
fails loudly when its HIP library is missing.

What it is: a NumPy (+ plain C, oracle/lg_oracle.c) restatement of the reference's
per-pixel scoring path, written from the reference source text, each function
citing the reference file:line it follows (paths relative to /root/reference).
dtypes follow the reference step by step (float64 planes, float32 distance
transforms / flatness) so the timing of the cpu_baseline leg is the same work.

Parity status
  * pinned by reference-generated golden vectors (tests/golden/, made by
    tests/golden/make_golden.py importing the reference in the build container):
    accessibility, approach, flatness, valid regions, candidate points, patch
    extraction, GraspPointCNN forward, ML post-transform, camera/3-D math,
    visibility score, HybridSelector / ConfidenceManager.
  * "parity unpinned" (third-party arithmetic absent from this image, no fixture
    in the reference): every cv2.* boundary (chamfer distanceTransform, ellipse
    dilate, findContours/minAreaRect) -- restated in lg_oracle.c from OpenCV
    4.10's published algorithms; skfmm.distance (replaced by an exact Euclidean
    distance transform, see clutter_extrema); paretoset (restated O(n^2)).
"""
from __future__ import annotations

import ctypes
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build_c(force: bool = False) -> str:
    """Compile oracle/lg_oracle.c -> oracle/_build/liblg_oracle.so (gcc)."""
    out_dir = os.path.join(_HERE, "_build")
    so = os.path.join(out_dir, "liblg_oracle.so")
    src = os.path.join(_HERE, "lg_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        os.makedirs(out_dir, exist_ok=True)
        subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", so, src, "-lm"])
    return so


def _lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build_c())
        u8p = ctypes.POINTER(ctypes.c_uint8)
        L.lg_chamfer_dt.argtypes = [u8p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                    ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_uint32)]
        L.lg_chamfer_dt_ex.argtypes = [u8p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_uint,
                                       ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_uint32)]
        L.lg_chamfer_dt.restype = None
        L.lg_ellipse_se.argtypes = [ctypes.c_int, u8p]
        L.lg_dilate.argtypes = [u8p, ctypes.c_int, ctypes.c_int, u8p, ctypes.c_int, u8p]
        L.lg_leaf_orientation.argtypes = [u8p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
        L.lg_leaf_orientation.restype = ctypes.c_int
        L.lg_leaf_contour_points.argtypes = [u8p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int32), ctypes.c_int]
        L.lg_leaf_contour_points.restype = ctypes.c_int
        L.lg_greedy_nms.argtypes = [ctypes.POINTER(ctypes.c_int64), ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                                    ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int32)]
        L.lg_greedy_nms.restype = ctypes.c_int
        L.lg_fmm_distance.argtypes = [u8p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
        L.lg_fmm_distance.restype = None
        L.lg_fmm_distance_tie.argtypes = [u8p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.c_int]
        L.lg_fmm_distance_tie.restype = None
        _LIB = L
    return _LIB


def _u8(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))


# --------------------------------------------------------------------------- cv2 stand-ins
INIT_DIST0 = (2 ** 31 - 1) >> 2   # OpenCV's border initialiser: INT_MAX >> 2, or INT_MAX in later revisions (lg_oracle.c)


def distance_transform(src_u8, mask_size=5, return_fix=False, init_dist0=INIT_DIST0):
    """cv2.distanceTransform(src, cv2.DIST_L2, mask_size) -> float32 [H,W]  (lg_oracle.c)."""
    a, p = _u8(src_u8)
    H, W = a.shape
    out = np.empty((H, W), np.float32)
    fix = np.empty((H, W), np.uint32)
    _lib().lg_chamfer_dt_ex(p, H, W, int(mask_size), int(init_dist0), out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                            fix.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)))
    return (out, fix) if return_fix else out


def ellipse_se(k):
    """cv2.getStructuringElement(cv2.MORPH_ELLIPSE, (k, k)) -> uint8 [k,k]."""
    se = np.empty((k, k), np.uint8)
    _lib().lg_ellipse_se(int(k), se.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)))
    return se


def dilate(src_u8, se):
    """cv2.dilate(src, se) for 0/1 images."""
    a, p = _u8(src_u8)
    s, sp = _u8(se)
    out = np.empty_like(a)
    _lib().lg_dilate(p, a.shape[0], a.shape[1], sp, s.shape[0], out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)))
    return out


def leaf_orientation_raw(mask_u8):
    """findContours(EXTERNAL, NONE) -> max contourArea -> minAreaRect; see lg_oracle.c."""
    a, p = _u8(mask_u8)
    out = (ctypes.c_double * 8)()
    ok = _lib().lg_leaf_orientation(p, a.shape[0], a.shape[1], out)
    return None if not ok else [out[i] for i in range(7)]


def leaf_contour_points(mask_u8):
    """Points of max(findContours(EXTERNAL, CHAIN_APPROX_NONE), key=contourArea), tracing order -> int32 [n,2] (x,y)."""
    a, p = _u8(mask_u8)
    cap = 4 * int(a.sum()) + 16
    out = np.empty((cap, 2), np.int32)
    n = _lib().lg_leaf_contour_points(p, a.shape[0], a.shape[1], out.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), cap)
    return out[:n].copy()


# --------------------------------------------------------------------------- ml_grasp_optimizer/data_collector.py
# Restatement of EnhancedGraspDataCollector's geometry helpers (cv2 calls replaced by the stand-ins above: "parity
# unpinned" vs real OpenCV; the pure-torch / pure-Python helpers are pinned by tests/golden/collector_vectors.npz).
SCORE_KEYS = ("sdf_score", "approach_score", "flatness_map", "isolation_map", "distance_map", "accessibility_map",
              "stem_penalty")   # required_scores, data_collector.py:138-140


def collector_extract_patches(x, y, mask, depth, scores, patch_size=32, k=0):
    """_extract_patches (:91-173) + _rotate_tensor (:395-398): raw slices [y-h:y+h, x-h:x+h], rotated k quarter turns.
    Returns (depth [P,P], mask [P,P] float, scores [7,P,P]) or None like the reference's validation."""
    h = patch_size // 2
    H, W = mask.shape
    if y < h or y >= H - h or x < h or x >= W - h:      # _check_boundaries :83-89 (note: >= H - h excludes y = H - h)
        return None
    d = np.asarray(depth)[y - h:y + h, x - h:x + h].astype(np.float32)
    m = (np.asarray(mask)[y - h:y + h, x - h:x + h] != 0).astype(np.float32)
    if not np.all(np.isfinite(d)) or not m.any():
        return None
    sc = np.stack([np.asarray(scores[kname])[y - h:y + h, x - h:x + h].astype(np.float32) for kname in SCORE_KEYS])
    if not np.all(np.isfinite(sc)):
        return None
    return np.rot90(d, k).copy(), np.rot90(m, k).copy(), np.rot90(sc, k, axes=(-2, -1)).copy()


def collector_rotate_point(point, angle, size):
    """_rotate_point (:400-420), float64 like the reference."""
    x, y = point
    c = size // 2
    a = np.radians(angle)
    x -= c
    y -= c
    nx = x * np.cos(a) - y * np.sin(a)
    ny = x * np.sin(a) + y * np.cos(a)
    return int(nx + c), int(ny + c)


def collector_tip_points(mask_u8):
    """_get_tip_points (:426-443): local maxima of the distance transform under a 5x5 box, on the mask; sorted by
    distance (descending, stable: ties keep row-major order), top quarter."""
    m = (np.asarray(mask_u8) != 0).astype(np.uint8)
    dist = distance_transform(m, 5)
    H, W = m.shape
    pad = np.full((H + 4, W + 4), -np.inf, np.float32)
    pad[2:-2, 2:-2] = dist
    mx = np.max(np.stack([pad[dy:dy + H, dx:dx + W] for dy in range(5) for dx in range(5)]), axis=0)
    ys, xs = np.where((mx == dist) & (m > 0))
    pts = list(zip(xs.tolist(), ys.tolist()))
    pts.sort(key=lambda p: dist[p[1], p[0]], reverse=True)
    return pts[:max(1, len(pts) // 4)]


def collector_stem_points(mask_u8):
    """_get_stem_points (:445-460): bottom quarter of the mask eroded twice by the 5x5 ellipse."""
    m = (np.asarray(mask_u8) != 0).astype(np.uint8)
    H = m.shape[0]
    m[:int(0.75 * H)] = 0
    se = ellipse_se(5)
    for _ in range(2):   # erode = complement of the dilation of the complement (symmetric SE; the frame does not erode)
        m = (1 - dilate(1 - m, se)).astype(np.uint8)
    ys, xs = np.where(m > 0)
    return list(zip(xs.tolist(), ys.tolist()))


def collector_edge_points(mask_u8):
    """_get_edge_points (:462-490): points of the largest outer contour whose turning angle is below 45 degrees."""
    pts = leaf_contour_points((np.asarray(mask_u8) != 0).astype(np.uint8)).astype(np.int64)
    n = len(pts)
    out = []
    for i in range(n):
        prev, cur, nxt = pts[i - 1], pts[i], pts[(i + 1) % n]
        v1, v2 = prev - cur, nxt - cur
        cross = v1[0] * v2[1] - v1[1] * v2[0]            # np.cross of 2-vectors (deprecated form in the reference)
        ang = abs(np.arctan2(cross, v1[0] * v2[0] + v1[1] * v2[1]))
        if ang < np.pi / 4:
            out.append((int(cur[0]), int(cur[1])))
    return out


# --------------------------------------------------------------------------- image_processor.py
def gaussian_kernel(size=5):
    """ImageProcessor._create_gaussian_kernel (scripts/utils/image_processor.py:25-32)."""
    sigma = size / 6.0
    center = size // 2
    x, y = np.meshgrid(np.arange(size), np.arange(size))
    kernel = np.exp(-((x - center) ** 2 + (y - center) ** 2) / (2 * sigma ** 2))
    kernel = kernel / kernel.sum()
    return kernel.astype(np.float32)


SOBEL_X = np.array([[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]], np.float32)  # image_processor.py:19
SOBEL_Y = SOBEL_X.T.copy()  # image_processor.py:21


def _xcorr_valid_f32(padded, k):
    """F.conv2d (cross-correlation, no padding) of one float32 plane, float32 accumulate."""
    kh, kw = k.shape
    H = padded.shape[0] - kh + 1
    W = padded.shape[1] - kw + 1
    acc = np.zeros((H, W), np.float32)
    for i in range(kh):
        for j in range(kw):
            if k[i, j] != 0:
                acc += np.float32(k[i, j]) * padded[i:i + H, j:j + W]
    return acc


def smooth_depth(depth_f32, gaussian_size=5):
    """ImageProcessor.smooth_depth (image_processor.py:56-64): reflect pad g//2, 5x5 Gaussian."""
    g = gaussian_kernel(gaussian_size)
    p = gaussian_size // 2
    padded = np.pad(np.asarray(depth_f32, np.float32), p, mode="reflect")
    return _xcorr_valid_f32(padded, g)


def flatness_map(depth_times_mask_f32, gaussian_size=5):
    """GraspPointSelector._calculate_flatness_map (grasp_point_selector.py:635-657)."""
    g = smooth_depth(depth_times_mask_f32, gaussian_size)
    padded = np.pad(g, 1, mode="reflect")
    dx = _xcorr_valid_f32(padded, SOBEL_X)
    dy = _xcorr_valid_f32(padded, SOBEL_Y)
    mag = np.sqrt(dx * dx + dy * dy, dtype=np.float32)
    return np.exp(-mag * np.float32(5.0), dtype=np.float32)


# --------------------------------------------------------------------------- grasp_point_selector.py
class RefGraspPointSelector:
    """Restatement of scripts/utils/grasp_point_selector.py::GraspPointSelector (CV maps, candidates,
    CNN rescoring loop, 3-D back-projection, pre-grasp).  numpy in, numpy out."""

    def __init__(self, cnn=None, gaussian_size=5, init_dist0=INIT_DIST0):
        self.init_dist0 = init_dist0   # OpenCV's INIT_DIST0 (library-revision dependent, see distance_transform)
        self.camera_cx = 707  # grasp_point_selector.py:29-31
        self.camera_cy = 494
        self.f_norm = None
        self.min_edge_distance = 20  # :25
        self.gaussian_size = gaussian_size  # leaf_grasp_node_v3.py:37
        self.cnn = cnn  # callable [B,9,32,32] float32 -> [B] logits, or None (= no best_model.pth, :52-54)

    # :145-150
    def set_camera_params(self, P):
        self.f_norm = P[0, 0]
        self.camera_cx = P[0, 2]
        self.camera_cy = P[1, 2]
        self.baseline = -P[0, 3] / self.f_norm

    # :502-524
    def _calculate_accessibility_score(self, m):
        height, width = m.shape
        y_grid, x_grid = np.ogrid[:height, :width]
        dist = np.sqrt((x_grid - self.camera_cx) ** 2 + (y_grid - self.camera_cy) ** 2)
        max_dist = np.sqrt(width ** 2 + height ** 2)
        amap = 1 - (dist / max_dist)
        angle = np.arctan2(y_grid - self.camera_cy, x_grid - self.camera_cx)
        return (0.7 * amap + 0.3 * np.cos(angle)) * m

    # :569-593  (depth argument is unused by the reference)
    def calculate_approach_vector_score(self, m):
        height, width = m.shape
        y, x = np.indices((height, width))
        vx = x - self.camera_cx
        vy = y - self.camera_cy
        norms = np.sqrt(vx * vx + vy * vy + float(self.f_norm) ** 2)
        norms[norms == 0] = 1
        return np.abs(self.f_norm / norms) * m

    # :718-752
    def estimate_leaf_orientation(self, m):
        r = leaf_orientation_raw(m)
        if r is None:
            return None, None, None, None
        return r[0], r[1], r[2], (r[3], r[4])

    # :526-567
    def calculate_sdf_score(self, m, return_parts=False):
        m = np.ascontiguousarray(m, np.uint8)
        dist_inside = distance_transform(m, 5, init_dist0=self.init_dist0)
        dist_outside = distance_transform(1 - m, 5, init_dist0=self.init_dist0)
        sdf = dist_inside - dist_outside
        optimal_distance = 20
        interior = np.exp(-((dist_inside - optimal_distance) ** 2) / (2 * optimal_distance ** 2))
        sdf = sdf / np.max(np.abs(sdf))
        y, x = np.indices(m.shape)
        vx = (x - self.camera_cx).astype(np.float64)
        vy = (y - self.camera_cy).astype(np.float64)
        norms = np.sqrt(vx * vx + vy * vy)
        norms[norms == 0] = 1
        vx = vx / norms
        vy = vy / norms
        angle, _, _, _ = self.estimate_leaf_orientation(m)
        if angle is not None:
            # np.abs(np.cross(v, (cos a, sin a))) for 2-vectors = |vx*sin a - vy*cos a|  (:556-558)
            align = np.abs(vx * np.sin(angle) - vy * np.cos(angle))
        else:
            align = np.ones_like(sdf)
        final = (0.4 * interior + 0.4 * align + 0.2 * sdf) * m
        if return_parts:
            return final, dict(dist_inside=dist_inside, dist_outside=dist_outside, angle=angle)
        return final

    # :595-633
    def _calculate_isolation_score(self, m):
        height, width = m.shape
        kernel_close = ellipse_se(30)
        kernel_wide = ellipse_se(40)
        current = m.astype(np.uint8)
        all_leaves = (m > 0).astype(np.uint8)
        other = all_leaves - current
        ic = dilate(other, kernel_close)
        dc = distance_transform(1 - ic, 3, init_dist0=self.init_dist0)
        sc = dc / (np.max(dc) + 1e-6)
        iw = dilate(other, kernel_wide)
        dw = distance_transform(1 - iw, 3, init_dist0=self.init_dist0)
        sw = dw / (np.max(dw) + 1e-6)
        iso = (0.7 * sc + 0.3 * sw).astype(np.float32)  # float32 array * python float stays float32
        yc = np.linspace(1.0, 0.2, height)[:, np.newaxis]
        hp = np.tile(yc, (1, width))
        return iso * hp * current

    # :688-701
    def _calculate_stem_penalty(self, m):
        bottom = np.zeros_like(m)
        h, w = m.shape
        third = h // 3
        bottom[-third:, :] = 1
        masked_bottom = m & bottom
        stem = dilate(masked_bottom, ellipse_se(30)) & m
        return stem.astype(np.float32)

    # :256-280
    def _calculate_all_scores(self, m, depth_f32):
        m = np.ascontiguousarray(m, np.uint8)
        sdf, parts = self.calculate_sdf_score(m, return_parts=True)
        scores = {
            "sdf_score": sdf,
            "approach_score": self.calculate_approach_vector_score(m),
            "flatness_map": flatness_map(np.asarray(depth_f32, np.float32) * m.astype(np.float32),
                                         self.gaussian_size),
            "isolation_map": self._calculate_isolation_score(m),
            "distance_map": distance_transform(m, 5, init_dist0=self.init_dist0),  # :266 (recomputed by the reference)
            "accessibility_map": self._calculate_accessibility_score(m),
            "stem_penalty": self._calculate_stem_penalty(m).astype(np.float32),
        }
        scores["traditional_score"] = (
            0.4 * scores["approach_score"] + 0.3 * scores["sdf_score"]
            + 0.2 * scores["flatness_map"] + 0.1 * scores["accessibility_map"]
        ) * (1 - scores["stem_penalty"])
        self._last_angle = parts["angle"]
        return scores

    # :282-288
    def _get_valid_regions(self, m, scores):
        return (scores["distance_map"] > self.min_edge_distance) & (m > 0) & (scores["stem_penalty"] < 0.8)

    # :447-482.  tie_rule="total": score desc then flat index desc (the build's documented
    # total order); tie_rule="numpy": the reference's np.argsort()[::-1].
    def _get_candidate_points(self, score_map, valid, top_k=20, min_distance=10, tie_rule="total"):
        vs = np.asarray(score_map * valid)
        H, W = vs.shape
        flat = vs.ravel() + 0.0  # -0.0 -> +0.0
        if tie_rule == "numpy":
            order = np.argsort(flat)[::-1]
        else:   # NaN first (np.argsort sorts NaN last, the reference reverses the order), then score desc, then flat index desc
            nan = np.isnan(flat)
            order = np.lexsort((-np.arange(flat.size), -np.where(nan, 0.0, flat), ~nan))
        order = np.ascontiguousarray(order, np.int64)
        out = np.zeros((top_k, 2), np.int32)
        n = _lib().lg_greedy_nms(order.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), order.size, H, W,
                                 int(top_k), int(min_distance), out.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
        return [(int(out[i, 0]), int(out[i, 1])) for i in range(n)]

    # :392-445  (numpy branch; replicate padding at the image border)
    @staticmethod
    def _extract_local_patch(arr, x, y, size=32):
        half = size // 2
        h, w = arr.shape
        x1, x2 = max(0, x - half), min(w, x + half)
        y1, y2 = max(0, y - half), min(h, y + half)
        patch = arr[y1:y2, x1:x2].copy()
        pl, pr = half - (x - x1), half - (x2 - x)
        pt, pb = half - (y - y1), half - (y2 - y)
        if min(pl, pr, pt, pb) < 0:
            return None
        if pl + pr + pt + pb > 0:
            patch = np.pad(patch, ((pt, pb), (pl, pr)), mode="edge")
        return patch if patch.shape == (size, size) else None

    # :59-143 feature assembly (9 channels, per-patch min-max when max>min; mask channel raw)
    def patch_features(self, m, depth_f32, scores, point):
        x, y = point
        d = self._extract_local_patch(np.asarray(depth_f32, np.float32), x, y).astype(np.float32)
        mk = self._extract_local_patch(np.asarray(m), x, y).astype(np.float32)
        if d.max() > d.min():
            d = (d - d.min()) / (d.max() - d.min())
        chans = [d, mk]
        for name in ("sdf_score", "approach_score", "flatness_map", "isolation_map", "distance_map",
                     "accessibility_map", "stem_penalty"):
            p = self._extract_local_patch(scores[name], x, y).astype(np.float32)
            if p.max() > p.min():
                p = (p - p.min()) / (p.max() - p.min())
            chans.append(p)
        return np.stack(chans).astype(np.float32)

    @staticmethod
    def ml_post(logit):
        """:133-136  sigmoid -> tanh(3 s)/2 + 1/2"""
        s = 1.0 / (1.0 + math.exp(-float(logit)))
        return float(np.tanh(s * 3.0) * 0.5 + 0.5)

    # :152-180 (pcl_data branch is never taken: :166-167)
    def get_3d_grasp_point(self, pt, depth_f32):
        u, v = pt
        z = float(depth_f32[v, u])
        return (z * (u - self.camera_cx) / self.f_norm, z * (v - self.camera_cy) / self.f_norm, z)

    # :821-826
    def _project_point_to_2d(self, p):
        x, y, z = p
        return (int((x * self.f_norm / z) + self.camera_cx), int((y * self.f_norm / z) + self.camera_cy))

    # :754-819
    def calculate_pre_grasp_point(self, g3, m):
        try:
            g = np.array(g3, dtype=np.float64)
            with np.errstate(all="ignore"):
                direction = g / np.linalg.norm(g)
            dil = dilate(m, ellipse_se(31))
            min_d, max_d, step = 0.05, 0.10, 0.01
            for dist in np.arange(min_d, max_d, step):
                t = (g3[0] - direction[0] * dist, g3[1] - direction[1] * dist, g3[2])
                with np.errstate(all="ignore"):
                    u, v = self._project_point_to_2d(t)   # int(nan) / int(inf) raise, as in the reference
                if not (0 <= u < m.shape[1] and 0 <= v < m.shape[0]):
                    continue
                if dil[v, u] == 0:
                    if np.linalg.norm(np.array(t) - g) >= min_d:
                        return t
            return (g3[0] - direction[0] * max_d, g3[1] - direction[1] * max_d, g3[2])
        except Exception:   # :817-819: logged, None (a grasp point whose depth is NaN, infinite or 0 has no pre-grasp point)
            return None

    # :184-253
    def select_grasp_point(self, m, depth_f32, tie_rule="total", mask_is_bool=True, return_debug=False):
        m = np.ascontiguousarray(m, np.uint8)
        depth_f32 = np.asarray(depth_f32, np.float32)
        scores = self._calculate_all_scores(m, depth_f32)
        valid = self._get_valid_regions(m, scores)
        cands = self._get_candidate_points(scores["traditional_score"], valid, 20, 10, tie_rule)
        if not cands:
            return (None, None, None) if not return_debug else ((None, None, None), {})
        best = cands[0]
        best_score = scores["traditional_score"][best[1], best[0]]
        ml_scores = []
        if self.cnn is not None and len(cands) > 1:
            H, W = m.shape
            for (x, y) in cands:
                trad = scores["traditional_score"][y, x]
                # Appendix B.7: with a torch *bool* mask the reference cannot replicate-pad a
                # border-overlapping patch (F.pad has no bool kernel) -> ml score None -> skipped.
                if mask_is_bool and (x < 16 or y < 16 or x + 16 > W or y + 16 > H):
                    ml_scores.append(None)
                    continue
                feat = self.patch_features(m, depth_f32, scores, (x, y))
                ml = self.ml_post(self.cnn(feat[None])[0])
                ml_scores.append(ml)
                conf = 1.0 - abs(ml - 0.5) * 2
                w = min(0.3, conf * 0.6)
                comb = (1.0 - w) * trad + w * ml
                if comb > best_score:
                    best_score = comb
                    best = (x, y)
        g3 = self.get_3d_grasp_point(best, depth_f32)
        pre = self.calculate_pre_grasp_point(g3, m)
        res = (best, g3, pre)
        if return_debug:
            return res, dict(scores=scores, valid=valid, candidates=cands, ml_scores=ml_scores)
        return res


# --------------------------------------------------------------------------- model.py
def cnn_forward(params, x, dtype=None):
    """GraspPointCNN.forward in eval mode (model.py:101-128); the attention variant follows from the keys.
    torch (CPU) functional restatement; x [B,9,32,32] -> logits [B]."""
    import torch
    import torch.nn.functional as F

    dt = dtype or torch.float32
    p = {k: torch.as_tensor(v).to(dt) for k, v in params.items()}
    h = torch.as_tensor(x).to(dt)
    n_blocks = 0
    while f"encoder.{n_blocks}.0.weight" in p:
        n_blocks += 1
    with torch.no_grad():
        for b in range(n_blocks):
            for conv, bn in ((0, 1), (3, 4)):
                h = F.conv2d(h, p[f"encoder.{b}.{conv}.weight"], p[f"encoder.{b}.{conv}.bias"], padding=1)
                h = F.batch_norm(h, p[f"encoder.{b}.{bn}.running_mean"], p[f"encoder.{b}.{bn}.running_var"],
                                 p[f"encoder.{b}.{bn}.weight"], p[f"encoder.{b}.{bn}.bias"], False, 0.0, 1e-5)
                h = F.relu(h)
            h = F.max_pool2d(h, 2)
        def chan(prefix):   # AdaptiveAvgPool2d(1) -> 1x1 conv -> ReLU -> 1x1 conv -> Sigmoid   (model.py:37-44)
            g = h.mean(dim=(2, 3), keepdim=True)
            z = F.relu(F.conv2d(g, p[f"{prefix}.1.weight"], p[f"{prefix}.1.bias"]))
            return torch.sigmoid(F.conv2d(z, p[f"{prefix}.3.weight"], p[f"{prefix}.3.bias"]))

        if "attention.0.weight" in p:            # 'spatial' (the node's default, grasp_point_selector.py:40)
            h = h * torch.sigmoid(F.conv2d(h, p["attention.0.weight"], p["attention.0.bias"]))
        elif "attention.1.weight" in p:          # 'channel'
            h = h * chan("attention")
        elif "spatial_attention.0.weight" in p:  # 'hybrid': x * spatial(x) * channel(x)   (model.py:115-119)
            sp = torch.sigmoid(F.conv2d(h, p["spatial_attention.0.weight"], p["spatial_attention.0.bias"]))
            h = h * sp * chan("channel_attention")
        h = h.mean(dim=(2, 3))
        for idx in (0, 4, 8):
            h = F.linear(h, p[f"classifier.{idx}.weight"], p[f"classifier.{idx}.bias"])
            h = F.batch_norm(h, p[f"classifier.{idx + 1}.running_mean"], p[f"classifier.{idx + 1}.running_var"],
                             p[f"classifier.{idx + 1}.weight"], p[f"classifier.{idx + 1}.bias"], False, 0.0, 1e-5)
            h = F.relu(h)
        h = F.linear(h, p["classifier.12.weight"], p["classifier.12.bias"])
    return h.reshape(-1).to(torch.float64).numpy()


def cnn_train_step(params, x, y, masks=None, opt_state=None, lr=0.0005, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01,
                   max_grad_norm=1.0, pos_weight=2.0, apply_update=True, dtype=None):
    """One iteration of scripts/train_model.py:247-265 restated with torch (CPU, fp32) functional ops + autograd:
    GraspPointCNN.forward in TRAIN mode (model.py:101-128: batch-statistics BatchNorm with momentum 0.1, Dropout2d(0.3)
    after every encoder block, Dropout(0.5/0.5/0.4) in the classifier), BCEWithLogitsLoss(pos_weight) (:221,251),
    clip_grad_norm_(max_norm) (:256), Adam(lr, weight_decay as L2) (:222,258).
    params: module state dict (numpy / torch; any attention type, recognised from the keys).  masks: list of [N, width] keep masks already
    scaled by 1/(1-p), one per dropout layer in module order (None = no dropout: every mask 1).
    opt_state: {'exp_avg': {k: arr}, 'exp_avg_sq': {...}, 'step': int} or None (fresh optimizer).
    Returns dict(loss, logits, grad_norm, grads, params (updated state dict incl. running statistics), opt_state)."""
    import torch
    import torch.nn.functional as F

    dt = dtype or torch.float32   # float64: the conditioning reference of the tests
    p = {k: torch.as_tensor(np.asarray(v)).to(dt).clone() for k, v in params.items()
         if not k.endswith("num_batches_tracked")}
    names = [k for k in p if "running_" not in k]
    for k in names:
        p[k].requires_grad_(True)
    h = torch.as_tensor(np.asarray(x)).to(dt)
    yt = torch.as_tensor(np.asarray(y)).to(dt)
    n_blocks = 0
    while f"encoder.{n_blocks}.0.weight" in p:
        n_blocks += 1
    mi = 0

    def mask(width):
        nonlocal mi
        mk = None if masks is None else torch.as_tensor(np.asarray(masks[mi])).to(dt)
        mi += 1
        assert mk is None or tuple(mk.shape) == (h.shape[0], width)
        return mk

    for b in range(n_blocks):
        for conv, bn in ((0, 1), (3, 4)):
            h = F.conv2d(h, p[f"encoder.{b}.{conv}.weight"], p[f"encoder.{b}.{conv}.bias"], padding=1)
            h = F.batch_norm(h, p[f"encoder.{b}.{bn}.running_mean"], p[f"encoder.{b}.{bn}.running_var"],
                             p[f"encoder.{b}.{bn}.weight"], p[f"encoder.{b}.{bn}.bias"], True, 0.1, 1e-5)
            h = F.relu(h)
        h = F.max_pool2d(h, 2)
        mk = mask(h.shape[1])                       # Dropout2d: one Bernoulli draw per (sample, channel)
        if mk is not None:
            h = h * mk[:, :, None, None]
    def chan(prefix):   # AdaptiveAvgPool2d(1) -> 1x1 conv -> ReLU -> 1x1 conv -> Sigmoid   (model.py:37-44)
        g = h.mean(dim=(2, 3), keepdim=True)
        z = F.relu(F.conv2d(g, p[f"{prefix}.1.weight"], p[f"{prefix}.1.bias"]))
        return torch.sigmoid(F.conv2d(z, p[f"{prefix}.3.weight"], p[f"{prefix}.3.bias"]))

    if "attention.0.weight" in p:            # 'spatial'
        h = h * torch.sigmoid(F.conv2d(h, p["attention.0.weight"], p["attention.0.bias"]))
    elif "attention.1.weight" in p:          # 'channel'
        h = h * chan("attention")
    elif "spatial_attention.0.weight" in p:  # 'hybrid': x * spatial(x) * channel(x)   (model.py:115-119)
        sp = torch.sigmoid(F.conv2d(h, p["spatial_attention.0.weight"], p["spatial_attention.0.bias"]))
        h = h * sp * chan("channel_attention")
    h = h.mean(dim=(2, 3))
    for idx in (0, 4, 8):
        h = F.linear(h, p[f"classifier.{idx}.weight"], p[f"classifier.{idx}.bias"])
        h = F.batch_norm(h, p[f"classifier.{idx + 1}.running_mean"], p[f"classifier.{idx + 1}.running_var"],
                         p[f"classifier.{idx + 1}.weight"], p[f"classifier.{idx + 1}.bias"], True, 0.1, 1e-5)
        h = F.relu(h)
        mk = mask(h.shape[1])
        if mk is not None:
            h = h * mk
    logits = F.linear(h, p["classifier.12.weight"], p["classifier.12.bias"]).reshape(-1)
    loss = F.binary_cross_entropy_with_logits(logits, yt, pos_weight=torch.tensor([float(pos_weight)], dtype=dt))
    grads = torch.autograd.grad(loss, [p[k] for k in names])
    grads = {k: g.detach() for k, g in zip(names, grads)}
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).to(dt)
    coef = min(float(max_grad_norm) / (float(total) + 1e-6), 1.0) if max_grad_norm and max_grad_norm > 0 else 1.0
    st = opt_state or {"exp_avg": {}, "exp_avg_sq": {}, "step": 0}
    step = int(st["step"]) + (1 if apply_update else 0)
    new_p = {k: v.detach().clone() for k, v in p.items()}
    new_m, new_v = {}, {}
    if apply_update:
        b1, b2 = betas
        bc1, bc2 = 1.0 - b1 ** step, 1.0 - b2 ** step
        for k in names:
            g = grads[k] * coef + weight_decay * new_p[k]
            m = torch.as_tensor(np.asarray(st["exp_avg"].get(k, np.zeros(g.shape, np.float32)))).to(dt)
            v = torch.as_tensor(np.asarray(st["exp_avg_sq"].get(k, np.zeros(g.shape, np.float32)))).to(dt)
            m = b1 * m + (1.0 - b1) * g
            v = b2 * v + (1.0 - b2) * g * g
            new_p[k] = new_p[k] - (lr / bc1) * (m / (v.sqrt() / math.sqrt(bc2) + eps))
            new_m[k], new_v[k] = m, v
    return {"loss": float(loss.detach()), "logits": logits.detach().numpy().astype(np.float64), "grad_norm": float(total),
            "grads": {k: g.numpy() for k, g in grads.items()}, "params": {k: v.numpy() for k, v in new_p.items()},
            "opt_state": {"exp_avg": {k: v.numpy() for k, v in new_m.items()},
                          "exp_avg_sq": {k: v.numpy() for k, v in new_v.items()}, "step": step}}


# --------------------------------------------------------------------------- leaf_scorer.py
def visibility_score(leaf_mask):
    """OptimalLeafSelector._calculate_visibility_score (scripts/utils/leaf_scorer.py:277-306)."""
    h, w = leaf_mask.shape
    ys, xs = np.where(leaf_mask)
    if len(ys) == 0:
        return 0.0
    border = np.sum(leaf_mask[0, :]) + np.sum(leaf_mask[-1, :]) + np.sum(leaf_mask[:, 0]) + np.sum(leaf_mask[:, -1])
    if border > 0:
        return 0.0
    cx, cy = np.mean(xs), np.mean(ys)
    d = np.sqrt((cx - w / 2) ** 2 + (cy - h / 2) ** 2)
    return 1.0 - d / np.sqrt((w / 2) ** 2 + (h / 2) ** 2)


def clutter_extrema(labels, method="edt"):
    """leaf_scorer.py:66-71.  The reference runs skfmm.distance (scikit-fmm 2022.3.26, absent here)
    on phi = 0 on any leaf / 1 elsewhere and only consumes argmin / argmax.  Build semantics
    (DESIGN.md): method="edt" = exact Euclidean distance to the nearest leaf pixel; argmin = first leaf
    pixel, argmax = first maximum in row-major order.  method="fmm" = the restated second-order fast
    marching field (fmm_distance), used to measure how far the semantics can move the arg-max.
    PARITY UNPINNED vs scikit-fmm."""
    from scipy import ndimage

    leaf = np.asarray(labels) >= 1
    field = ndimage.distance_transform_edt(~leaf) if method == "edt" else fmm_distance(leaf)
    mn = np.unravel_index(field.argmin(), field.shape)
    mx = np.unravel_index(field.argmax(), field.shape)
    return mn, mx


def fmm_distance(leaf_bool, tie=0):
    """Restated second-order fast marching (what skfmm.distance does at leaf_scorer.py:69); see lg_oracle.c.
    Only used to quantify the EDT-vs-FMM deviation of the consumed arg-max.  tie: order in which equal heap keys leave
    (0 = this heap's own, 1 / 2 = increasing / decreasing flat index) -- a measuring knob.  PARITY UNPINNED."""
    a, p = _u8(np.asarray(leaf_bool).astype(np.uint8))
    out = np.empty(a.shape, np.float64)
    _lib().lg_fmm_distance_tie(p, a.shape[0], a.shape[1], out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), int(tie))
    return out


def pareto_max(scores):
    """paretoset(scores, sense=['max']*3) (paretoset 1.2.3, absent here): non-dominated rows,
    only the first of identical rows kept.  O(n^2) restatement.  PARITY UNPINNED."""
    s = np.asarray(scores, np.float64)
    n = len(s)
    keep = np.ones(n, bool)
    for i in range(n):
        for j in range(n):
            if i == j:
                continue
            if np.all(s[j] >= s[i]) and (np.any(s[j] > s[i]) or j < i):
                keep[i] = False
                break
    return keep


class RefOptimalLeafSelector:
    """Restatement of scripts/utils/leaf_scorer.py::OptimalLeafSelector.select_optimal_leaf (:25-203)."""

    def __init__(self, field="edt"):
        self.camera_cx = self.camera_cy = self.f_norm = None
        self._tall_leaves = []
        self.field = field  # "edt" (build semantics) or "fmm" (restated scikit-fmm behaviour)

    def set_camera_params(self, P):  # :19-23
        self.f_norm, self.camera_cx, self.camera_cy = P[0, 0], P[0, 2], P[1, 2]

    def get_tall_leaves(self):  # :205-207
        return self._tall_leaves

    def select_optimal_leaf(self, labels_i16, depth_f32, return_debug=False):
        mask_np = np.asarray(labels_i16)
        depth_np = np.asarray(depth_f32, np.float32)
        leaf_ids = np.unique(mask_np)[1:]  # torch.unique(mask)[1:]  (:32)
        depth_list, leaf_masks = [], []
        for lid in leaf_ids:
            lm = mask_np == lid
            leaf_masks.append(lm)
            d = depth_np[lm]
            if len(d) > 0:
                depth_list.append(np.median(d))
        if not depth_list:
            return None
        depth_mean = np.mean(np.array(depth_list))
        tall = [int(leaf_ids[i]) for i, d in enumerate(depth_list) if d < depth_mean]
        mn, mx = clutter_extrema(mask_np, self.field)
        cands = []
        for idx, lid in enumerate(leaf_ids):
            lm = leaf_masks[idx]
            if np.sum(lm) < 10000:
                continue
            ys, xs = np.where(lm)
            c = (np.mean(xs), np.mean(ys))
            dmin = np.sqrt((c[0] - mn[1]) ** 2 + (c[1] - mn[0]) ** 2)
            dmax = np.sqrt((c[0] - mx[1]) ** 2 + (c[1] - mx[0]) ** 2)
            tot = dmin + dmax
            clutter = dmin / tot if tot > 0 else 0
            mean_depth = np.mean(depth_np[lm])
            X = (mean_depth * (xs - self.camera_cx)) / self.f_norm
            Y = (mean_depth * (ys - self.camera_cy)) / self.f_norm
            Z = np.full_like(X, mean_depth)
            mean_distance = np.mean(np.sqrt(X ** 2 + Y ** 2 + Z ** 2))
            dist_score = np.exp(-mean_distance / 0.3)
            vis = visibility_score(lm)
            cands.append(dict(leaf_id=int(lid), scores=np.array([clutter, dist_score, vis], np.float64),
                              mean_distance=float(mean_distance), is_tall=int(lid) in tall))
        if not cands:
            return None
        tall_c = [c for c in cands if c["is_tall"]]
        reg_c = [c for c in cands if not c["is_tall"]]
        if tall_c:
            sc = np.stack([c["scores"] for c in tall_c]) * 1.1
            pm = pareto_max(sc)
            pc = [c for i, c in enumerate(tall_c) if pm[i]]
        else:
            sc = np.stack([c["scores"] for c in reg_c])
            pm = pareto_max(sc)
            pc = [c for i, c in enumerate(reg_c) if pm[i]]
        if not pc:
            pc = tall_c if tall_c else reg_c
        w = np.array([0.35, 0.35, 0.3])
        best, best_s = None, float("-inf")
        self._tall_leaves = tall
        for c in pc:
            s = np.sum(w * c["scores"])
            if s > best_s:
                best_s, best = s, c["leaf_id"]
        if return_debug:
            return best, dict(candidates=cands, tall=tall, extrema=(mn, mx), depth_list=depth_list)
        return best


# --------------------------------------------------------------------------- synthetic inputs
# The seeded scene / patch / closed-form-weight generators live in synthetic_inputs.py (repo root): they are INPUTS shared
# by bench.py, the tools and the tests, not part of the checker.  Re-exported here for the tests' convenience.
import sys as _sys  # noqa: E402

_sys.path.insert(0, os.path.dirname(_HERE)) if os.path.dirname(_HERE) not in _sys.path else None
from synthetic_inputs import (CNN_FILTERS, _hash_unit, cnn_closed_form_params, cnn_param_shapes,  # noqa: E402,F401
                              synthetic_patches, synthetic_scene)
