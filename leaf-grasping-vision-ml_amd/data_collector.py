"""EnhancedGraspDataCollector -- self-supervised sample harvesting on the device-resident score planes.

Mirror of scripts/utils/ml_grasp_optimizer/data_collector.py (class, method names, arguments, return values, on-disk
format of `training_data.pt`).  What runs where:
  * patch extraction (_extract_patches :91-173) and the rot90 of the augmentation (:250-266): HIP kernel
    `lg_harvest_patches` -- raw 32x32 windows of depth / mask / the seven score planes, validity flags from the kernel;
  * negative-sample regions: `lg_negative_masks` (5x5 local maxima of distance_map on the mask = _get_tip_points :426-443,
    double 5x5-ellipse erosion of the bottom quarter = _get_stem_points :445-460) and `lg_leaf_contour`
    (cv2.findContours EXTERNAL/NONE + max contourArea of _get_edge_points :462-490; curvature test in NumPy);
  * bookkeeping, random choices (`random.uniform` / `random.sample` in the reference's order), saving: Python.
Differences a maintainer should know: patch_size must be 32 (the kernel's window); tensors must live on the HIP device;
`scores['distance_map']` is used as the distance transform of the mask (the reference recomputes the same transform with
cv2.distanceTransform); samples are kept on the CPU; `load_existing_data` uses torch.load(weights_only=True).
"""
import ctypes as C
import os
import random
import shutil

import numpy as np
import torch

from . import _log as rospy
from ._lib import LG_NUM_MAPS, MAP_INDEX, check, lib
from .grasp_point_selector import _device_index

_VP = C.c_void_p
REQUIRED_SCORES = ("sdf_score", "approach_score", "flatness_map", "isolation_map", "distance_map", "accessibility_map",
                   "stem_penalty")   # data_collector.py:138-140 (= channels 2..8 of the CNN input)


class EnhancedGraspDataCollector:
    def __init__(self, patch_size=32, resume=True, data_dir=None, device="cuda:0"):
        if patch_size != 32:
            raise ValueError("the harvesting kernel extracts 32x32 patches (the reference default)")
        self.patch_size = patch_size
        self.samples = []
        self.device = torch.device(device)
        self._h = _VP()
        check(None, lib.lg_create(_device_index(self.device), C.byref(self._h)), "lg_create")
        self.data_dir = data_dir or os.path.expanduser("~/leaf_grasp_output/ml_training_data")   # :16
        if not resume and os.path.exists(self.data_dir):
            shutil.rmtree(self.data_dir)
            rospy.loginfo(f"Existing data at {self.data_dir} cleared because resume is set to False.")
        os.makedirs(self.data_dir, exist_ok=True)
        self.stats = {"positive_samples": 0, "negative_samples": 0, "augmented_samples": 0}
        if resume:
            self.load_existing_data()
            rospy.loginfo(f"Resumed with {len(self.samples)} existing samples")
            self._log_collection_progress()
        else:
            rospy.loginfo("Starting fresh data collection. Existing data (if any) was cleared.")
        rospy.loginfo(f"Enhanced data collector initialized. Saving to: {self.data_dir}")

    def __del__(self):
        try:
            if getattr(self, "_h", None) is not None and self._h.value:
                lib.lg_destroy(self._h)
                self._h = _VP()
        except Exception:  # noqa: BLE001
            pass

    # ------------------------------------------------------------------ persistence (:43-79, :500-598)
    def load_existing_data(self):
        try:
            save_path = os.path.join(self.data_dir, "training_data.pt")
            if os.path.exists(save_path):
                rospy.loginfo(f"Found existing data at {save_path}")
                data = torch.load(save_path, weights_only=True)
                for i in range(len(data["labels"])):
                    self.samples.append({
                        "depth_patch": data["depth_patches"][i], "mask_patch": data["mask_patches"][i],
                        "score_patches": data["score_patches"][i], "total_score": data["total_scores"][i].item(),
                        "grasp_point": tuple(data["grasp_points"][i].tolist()), "label": data["labels"][i].item(),
                        "is_augmented": data["is_augmented"][i].item()})
                self.stats["positive_samples"] = sum(1 for s in self.samples if s["label"] == 1 and not s["is_augmented"])
                self.stats["augmented_samples"] = sum(1 for s in self.samples if s["label"] == 1 and s["is_augmented"])
                self.stats["negative_samples"] = sum(1 for s in self.samples if s["label"] == 0)
                rospy.loginfo("Loaded existing data")
            else:
                rospy.loginfo("No existing data found. Starting fresh collection.")
        except Exception as e:  # noqa: BLE001
            rospy.logerr(f"Error loading existing data: {str(e)}")
            rospy.logerr("Starting fresh collection.")
            self.samples = []
            self.stats = {"positive_samples": 0, "negative_samples": 0, "augmented_samples": 0}

    def save_samples(self):
        try:
            if not self.samples:
                rospy.logwarn("No samples to save")
                return
            save_path = os.path.join(self.data_dir, "training_data.pt")
            backup_path = save_path + ".backup"
            if os.path.exists(save_path):
                shutil.copy2(save_path, backup_path)
            try:
                data = {
                    "depth_patches": torch.stack([s["depth_patch"] for s in self.samples]),
                    "mask_patches": torch.stack([s["mask_patch"] for s in self.samples]),
                    "score_patches": torch.stack([s["score_patches"] for s in self.samples]),
                    "labels": torch.tensor([s["label"] for s in self.samples]),
                    "total_scores": torch.tensor([s["total_score"] for s in self.samples]),
                    "grasp_points": torch.tensor([s["grasp_point"] for s in self.samples]),
                    "is_augmented": torch.tensor([s["is_augmented"] for s in self.samples]),
                }
                quality = {
                    "depth_range": [data["depth_patches"].min().item(), data["depth_patches"].max().item()],
                    "mask_coverage": (data["mask_patches"] > 0).float().mean().item(),
                    "positive_ratio": (data["labels"] == 1).float().mean().item(),
                    "augmented_ratio": data["is_augmented"].float().mean().item(),
                    "score_statistics": {"mean": data["total_scores"].mean().item(), "std": data["total_scores"].std().item(),
                                         "min": data["total_scores"].min().item(), "max": data["total_scores"].max().item()},
                }
                torch.save(data, save_path)
                with open(os.path.join(self.data_dir, "collection_metadata.txt"), "w") as f:
                    f.write("=== Data Collection Statistics ===\n")
                    f.write(f"Original positive samples: {self.stats['positive_samples']}\n")
                    f.write(f"Augmented positive samples: {self.stats['augmented_samples']}\n")
                    f.write(f"Negative samples: {self.stats['negative_samples']}\n")
                    f.write(f"Total samples: {len(self.samples)}\n\n")
                    f.write("=== Tensor Shapes ===\n")
                    for key, tensor in data.items():
                        f.write(f"{key}: {tensor.shape}\n")
                    f.write("\n=== Quality Metrics ===\n")
                    f.write(f"Depth range: {quality['depth_range']}\n")
                    f.write(f"Mask coverage: {quality['mask_coverage']:.3f}\n")
                    f.write(f"Positive ratio: {quality['positive_ratio']:.3f}\n")
                    f.write(f"Augmented ratio: {quality['augmented_ratio']:.3f}\n")
                    f.write("\nScore Statistics:\n")
                    for key, value in quality["score_statistics"].items():
                        f.write(f"{key}: {value:.3f}\n")
                rospy.loginfo(f"Saved {len(self.samples)} samples to {save_path}")
                if os.path.exists(backup_path):
                    os.remove(backup_path)
            except Exception as e:  # noqa: BLE001
                rospy.logerr(f"Error during data preparation and saving: {str(e)}")
                if os.path.exists(backup_path):
                    shutil.copy2(backup_path, save_path)
                    rospy.loginfo("Restored from backup")
                raise
            with open(os.path.join(self.data_dir, "collection_progress.txt"), "w") as f:
                f.write(f"last_frame: {self.stats['positive_samples']}\n")
        except Exception as e:  # noqa: BLE001
            rospy.logerr(f"Error saving samples: {str(e)}")

    # ------------------------------------------------------------------ device plumbing
    def _stream(self):
        return _VP(torch.cuda.current_stream(self.device).cuda_stream)

    def _device_inputs(self, leaf_mask, depth_tensor, scores):
        """u8 mask, f32 depth and the seven f32 planes on the device (numpy / CPU inputs are uploaded)."""
        m = torch.as_tensor(leaf_mask).to(self.device)
        m = (m != 0).to(torch.uint8).contiguous()
        d = torch.as_tensor(depth_tensor).to(self.device, torch.float32).contiguous()
        planes = []
        for name in REQUIRED_SCORES:
            planes.append(torch.as_tensor(scores[name]).to(self.device, torch.float32).contiguous())
        return m, d, planes

    def _harvest(self, m, d, planes, points, rots=None):
        """-> depth [n,32,32], mask [n,32,32], scores [n,7,32,32] (device tensors) and flags [n] (CPU ints)."""
        n = len(points)
        H, W = m.shape
        xy = torch.tensor(points, dtype=torch.int32, device=self.device).reshape(n, 2).contiguous()
        rot = torch.tensor(rots, dtype=torch.int32, device=self.device) if rots is not None else None
        od = torch.empty((n, 32, 32), dtype=torch.float32, device=self.device)
        om = torch.empty((n, 32, 32), dtype=torch.float32, device=self.device)
        osc = torch.empty((n, 7, 32, 32), dtype=torch.float32, device=self.device)
        fl = torch.empty((n,), dtype=torch.int32, device=self.device)
        ptrs = (_VP * LG_NUM_MAPS)()
        for i, p in enumerate(planes):
            ptrs[i] = p.data_ptr()
        with torch.cuda.device(self.device):
            check(self._h, lib.lg_harvest_patches(self._h, d.data_ptr(), m.data_ptr(), C.byref(ptrs), H, W, n, xy.data_ptr(),
                                                  rot.data_ptr() if rot is not None else None, od.data_ptr(),
                                                  om.data_ptr(), osc.data_ptr(), fl.data_ptr(), self._stream()),
                  "lg_harvest_patches")
        return od, om, osc, fl.cpu().tolist()

    # ------------------------------------------------------------------ reference helpers
    def _check_boundaries(self, x, y, shape, half_size):   # :83-89
        if y < half_size or y >= shape[0] - half_size or x < half_size or x >= shape[1] - half_size:
            rospy.logwarn(f"Point ({x},{y}) too close to edge. Image shape: {shape}")
            return False
        return True

    @staticmethod
    def _flags_ok(flag):
        if flag & 8:
            rospy.logwarn("Patch extraction out of bounds")
        elif flag & 1:
            rospy.logwarn("Invalid depth values in patch")
        elif flag & 2:
            rospy.logwarn("Empty mask patch")
        elif flag & 4:
            rospy.logwarn("Invalid values in score patch")
        return flag == 0

    def _extract_patches(self, x, y, leaf_mask, depth_tensor, scores):   # :91-173
        try:
            if not torch.is_tensor(depth_tensor) or not torch.is_tensor(leaf_mask):
                rospy.logwarn("Invalid input tensors")
                return None
            for name in REQUIRED_SCORES:
                if name not in scores:
                    rospy.logwarn(f"Missing required score: {name}")
                    return None
            m, d, planes = self._device_inputs(leaf_mask, depth_tensor, scores)
            od, om, osc, fl = self._harvest(m, d, planes, [(int(x), int(y))])
            if not self._flags_ok(fl[0]):
                return None
            return od[0].cpu(), om[0].cpu(), osc[0].cpu()
        except Exception as e:  # noqa: BLE001
            rospy.logerr(f"Error extracting patches: {str(e)}")
            return None

    def _rotate_tensor(self, tensor, angle):   # :395-398
        return torch.rot90(tensor, angle // 90, dims=(-2, -1))

    def _rotate_point(self, point, angle, size):   # :400-420
        x, y = point
        center = size // 2
        a = np.radians(angle)
        x -= center
        y -= center
        new_x = x * np.cos(a) - y * np.sin(a)
        new_y = x * np.sin(a) + y * np.cos(a)
        return (int(new_x + center), int(new_y + center))

    def _add_sample(self, depth_patch, mask_patch, score_patches, total_score, grasp_point, label, is_augmented):   # :350-393
        try:
            if not all(isinstance(x, torch.Tensor) for x in [depth_patch, mask_patch]):
                rospy.logwarn("Invalid tensor types in sample")
                return False
            if not all(x.shape == (self.patch_size, self.patch_size) for x in [depth_patch, mask_patch]):
                rospy.logwarn("Invalid patch shapes")
                return False
            if mask_patch.dtype == torch.bool:
                mask_patch = mask_patch.float()
            self.samples.append({"depth_patch": depth_patch, "mask_patch": mask_patch, "score_patches": score_patches,
                                 "total_score": float(total_score), "grasp_point": tuple(map(int, grasp_point)),
                                 "label": int(label), "is_augmented": bool(is_augmented)})
            if label == 1:
                self.stats["augmented_samples" if is_augmented else "positive_samples"] += 1
            else:
                self.stats["negative_samples"] += 1
            return True
        except Exception as e:  # noqa: BLE001
            rospy.logwarn(f"Error adding sample: {str(e)}")
            return False

    def _log_collection_progress(self):   # :492-498
        rospy.loginfo("\n=== Data Collection Progress ===")
        rospy.loginfo(f"Original positive samples: {self.stats['positive_samples']}")
        rospy.loginfo(f"Augmented positive samples: {self.stats['augmented_samples']}")
        rospy.loginfo(f"Negative samples: {self.stats['negative_samples']}")
        rospy.loginfo(f"Total samples: {len(self.samples)}")

    # ------------------------------------------------------------------ negative-sample regions (:426-490)
    def _negative_masks(self, m, dist):
        H, W = m.shape
        tip = torch.empty((H, W), dtype=torch.uint8, device=self.device)
        stem = torch.empty((H, W), dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            check(self._h, lib.lg_negative_masks(self._h, dist.data_ptr(), m.data_ptr(), H, W, tip.data_ptr(),
                                                 stem.data_ptr(), self._stream()), "lg_negative_masks")
        return tip, stem

    def _get_tip_points(self, mask, dist=None):
        """Local maxima of the distance transform (5x5), on the mask, sorted by distance, top quarter.  `dist` is the
        chamfer-5 distance transform of `mask` (scores['distance_map']); computed on the fly when omitted."""
        try:
            m = (torch.as_tensor(mask).to(self.device) != 0).to(torch.uint8).contiguous()
            if dist is None:
                dist = self._distance_map(m)
            dist = torch.as_tensor(dist).to(self.device, torch.float32).contiguous()
            tip, _ = self._negative_masks(m, dist)
            idx = torch.nonzero(tip)                      # row-major, like np.where
            vals = dist[idx[:, 0], idx[:, 1]].cpu().tolist()
            pts = [(int(x), int(y)) for y, x in idx.cpu().tolist()]
            order = sorted(range(len(pts)), key=lambda i: vals[i], reverse=True)   # stable, like list.sort(reverse=True)
            pts = [pts[i] for i in order]
            return pts[:max(1, len(pts) // 4)]
        except Exception as e:  # noqa: BLE001
            rospy.logwarn(f"Error getting tip points: {str(e)}")
            return []

    def _get_stem_points(self, mask):
        try:
            m = (torch.as_tensor(mask).to(self.device) != 0).to(torch.uint8).contiguous()
            dist = torch.zeros(m.shape, dtype=torch.float32, device=self.device)
            _, stem = self._negative_masks(m, dist)
            return [(int(x), int(y)) for y, x in torch.nonzero(stem).cpu().tolist()]
        except Exception as e:  # noqa: BLE001
            rospy.logwarn(f"Error getting stem points: {str(e)}")
            return []

    def _get_edge_points(self, mask):
        try:
            m = (torch.as_tensor(mask).to(self.device) != 0).to(torch.uint8).contiguous()
            H, W = m.shape
            n = C.c_int(0)
            cap = 4 * (H + W) + 16
            while True:
                buf = np.empty((cap, 2), np.int32)
                with torch.cuda.device(self.device):
                    check(self._h, lib.lg_leaf_contour(self._h, m.data_ptr(), H, W, buf.ctypes.data, cap, C.byref(n),
                                                       self._stream()), "lg_leaf_contour")
                if n.value <= cap:
                    break
                cap = n.value
            c = buf[:n.value].astype(np.int64)
            if len(c) == 0:
                return []
            v1 = np.roll(c, 1, axis=0) - c
            v2 = np.roll(c, -1, axis=0) - c
            ang = np.abs(np.arctan2(v1[:, 0] * v2[:, 1] - v1[:, 1] * v2[:, 0], (v1 * v2).sum(axis=1)))
            return [(int(p[0]), int(p[1])) for p in c[ang < np.pi / 4]]
        except Exception as e:  # noqa: BLE001
            rospy.logwarn(f"Error getting edge points: {str(e)}")
            return []

    def _distance_map(self, m_u8):
        """cv2.distanceTransform(mask, DIST_L2, 5) through lg_score_maps (only the distance plane is requested)."""
        H, W = m_u8.shape
        dist = torch.empty((H, W), dtype=torch.float32, device=self.device)
        trad = torch.empty((H, W), dtype=torch.float32, device=self.device)
        ptrs = (_VP * LG_NUM_MAPS)()
        ptrs[MAP_INDEX["distance_map"]] = dist.data_ptr()
        ptrs[MAP_INDEX["traditional_score"]] = trad.data_ptr()
        depth = torch.zeros((H, W), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            check(self._h, lib.lg_score_maps(self._h, depth.data_ptr(), m_u8.data_ptr(), 1, H, W, None, C.byref(ptrs), None,
                                             None, self._stream()), "lg_score_maps")
        return dist

    # ------------------------------------------------------------------ collection (:175-348)
    def collect_sample(self, leaf_mask, depth_tensor, rgb_image, scores, grasp_point_2d, total_score):
        try:
            if not torch.is_tensor(depth_tensor) or not torch.is_tensor(leaf_mask):
                rospy.logerr("Invalid input tensors")
                return False
            x, y = map(int, grasp_point_2d)
            if x < 0 or y < 0 or x >= leaf_mask.shape[1] or y >= leaf_mask.shape[0]:
                rospy.logerr(f"Invalid grasp point coordinates: ({x}, {y})")
                return False
            if not all(s in scores for s in REQUIRED_SCORES):
                rospy.logerr("Missing required score maps")
                return False
            half = self.patch_size // 2
            rospy.loginfo(f"Collecting positive sample at point ({x}, {y}) with score {total_score:.3f}")
            if not self._check_boundaries(x, y, leaf_mask.shape, half):
                return False
            m, d, planes = self._device_inputs(leaf_mask, depth_tensor, scores)
            # the positive patch and its three rotations in one launch
            od, om, osc, fl = self._harvest(m, d, planes, [(x, y)] * 4, rots=[0, 1, 2, 3])
            if not self._flags_ok(fl[0]):
                return False
            if not self._add_sample(od[0].cpu(), om[0].cpu(), osc[0].cpu(), total_score, grasp_point_2d, label=1,
                                    is_augmented=False):
                return False
            self._generate_augmented_samples(od[0], om[0], osc[0], total_score, grasp_point_2d, _rotated=(od, om, osc))
            self._collect_validated_negative_samples(leaf_mask, depth_tensor, scores, _dev=(m, d, planes))
            self._log_collection_progress()
            if (self.stats["positive_samples"] + self.stats["negative_samples"]) % 5 == 0:
                self.save_samples()
            return True
        except Exception as e:  # noqa: BLE001
            rospy.logerr(f"Error in sample collection: {str(e)}")
            return False

    def _generate_augmented_samples(self, depth_patch, mask_patch, score_patches, total_score, grasp_point_2d, _rotated=None):
        """rot90 x {90,180,270}, 1-2 % depth noise, score * U(0.95,1)  (:250-293).  The `random` calls come in the
        reference's order (uniform, [randn], uniform per angle); the Gaussian noise is drawn with torch on the patch's
        device, so it is reproducible per device generator like the reference's randn_like."""
        try:
            for k, angle in enumerate((90, 180, 270), start=1):
                try:
                    if _rotated is not None:
                        rot_depth, rot_mask, rot_scores = _rotated[0][k], _rotated[1][k], _rotated[2][k]
                    else:
                        rot_depth = self._rotate_tensor(depth_patch.float(), angle)
                        rot_mask = self._rotate_tensor(mask_patch.float(), angle)
                        rot_scores = torch.stack([self._rotate_tensor(s.float(), angle) for s in score_patches])
                    rot_mask = (rot_mask > 0.5).float()
                    noise_factor = random.uniform(0.01, 0.02)
                    depth_noise = torch.randn_like(rot_depth) * (noise_factor * rot_depth.mean())
                    noisy_depth = torch.clamp(rot_depth + depth_noise, min=0.0)
                    new_point = self._rotate_point(grasp_point_2d, angle, self.patch_size)
                    ok = self._add_sample(noisy_depth.cpu(), rot_mask.cpu(), rot_scores.cpu(),
                                          total_score * random.uniform(0.95, 1.0), new_point, label=1, is_augmented=True)
                    if not ok:
                        rospy.logwarn(f"Failed to add augmented sample for angle {angle}")
                except Exception as e:  # noqa: BLE001
                    rospy.logwarn(f"Error in augmentation for angle {angle}: {str(e)}")
        except Exception as e:  # noqa: BLE001
            rospy.logwarn(f"Error in augmentation: {str(e)}")

    def _collect_validated_negative_samples(self, leaf_mask, depth_tensor, scores, _dev=None):
        """Up to 3 negatives per positive from tip / stem / edge points, at most 10 attempts (:295-348)."""
        try:
            m, d, planes = _dev if _dev is not None else self._device_inputs(leaf_mask, depth_tensor, scores)
            max_attempts, max_negative_samples = 10, 3
            attempts = collected = 0
            # the three point lists do not change between attempts (the reference recomputes identical lists)
            tip_points = self._get_tip_points(m, planes[REQUIRED_SCORES.index("distance_map")])
            stem_points = self._get_stem_points(m)
            edge_points = self._get_edge_points(m)
            while collected < max_negative_samples and attempts < max_attempts:
                try:
                    negative_points = []
                    if tip_points:
                        negative_points.extend(random.sample(tip_points, min(1, len(tip_points))))
                    if stem_points:
                        negative_points.extend(random.sample(stem_points, min(1, len(stem_points))))
                    if edge_points:
                        negative_points.extend(random.sample(edge_points, min(1, len(edge_points))))
                    if negative_points:
                        od, om, osc, fl = self._harvest(m, d, planes, negative_points)
                        for i, (px, py) in enumerate(negative_points):
                            if collected >= max_negative_samples:
                                break
                            if fl[i] == 0 and self._add_sample(od[i].cpu(), om[i].cpu(), osc[i].cpu(), 0.0, (px, py),
                                                               label=0, is_augmented=False):
                                collected += 1
                    attempts += 1
                except Exception as e:  # noqa: BLE001
                    rospy.logwarn(f"Error processing negative samples: {str(e)}")
                    attempts += 1
        except Exception as e:  # noqa: BLE001
            rospy.logerr(f"Error in negative sample collection: {str(e)}")
