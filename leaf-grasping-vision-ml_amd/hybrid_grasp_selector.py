"""HybridGraspSelector -- the facade the reference documents but never ships
(README.md:65-69,202-204; vla_system/README.md:46-49: `HybridGraspSelector().select_grasp_point(image,
candidates)`).  It is the composition implemented procedurally in scripts/leaf_grasp_node_vla.py:97-221:

    geometric scores -> (optional) VLA scores -> ConfidenceManager -> HybridSelector.select_best_candidate
    -> GraspPointSelector.select_grasp_point on the winner's mask            (the HIP hot path)

`candidates` are dicts as built by `generate_candidates` (leaf_grasp_node_vla.py:148-182): 'leaf_id', 'x',
'y', 'geometric_score', 'clutter_score', 'distance_score', 'visibility_score', 'mask'.
"""
import numpy as np
import torch

from ._log import loginfo, logwarn
from .confidence_manager import ConfidenceManager
from .grasp_point_selector import GraspPointSelector
from .hybrid_selector import HybridSelector
from .image_processor import ImageProcessor
from .leaf_scorer import OptimalLeafSelector


class HybridGraspSelector:
    def __init__(self, device=None, vla_scorer=None, load_model=True, min_leaf_area=3500):
        self.device = torch.device(device if device is not None else "cuda:0")
        self.grasp_selector = GraspPointSelector(self.device, load_model=load_model)   # traditional_selector (:62)
        self.leaf_scorer = OptimalLeafSelector(self.device)                            # (:65)
        self.hybrid_selector = HybridSelector(self.device)                             # (:71)
        self.confidence_manager = ConfidenceManager()                                  # (:72)
        self.vla_scorer = vla_scorer   # any object with evaluate_candidates(image, candidates, instruction)
        self.vla_enabled = vla_scorer is not None
        self.min_leaf_area = min_leaf_area  # leaf_grasp_node_vla.py:32
        self.image_processor = None
        self.last_selection = None

    def set_camera_params(self, projection_matrix):  # camera_info_callback (:89-95)
        self.grasp_selector.set_camera_params(projection_matrix)
        self.leaf_scorer.set_camera_params(projection_matrix)

    # leaf_grasp_node_vla.py:148-182.  The reference calls a non-existent
    # OptimalLeafSelector._calculate_all_scores (SURVEY headline 4); the evident intent -- a per-leaf
    # geometric score -- is taken from the selector's own clutter/distance/visibility scores with its
    # 0.35/0.35/0.30 weights (leaf_scorer.py:170).
    def generate_candidates(self, mask_tensor, depth_tensor, top_n=5):
        mask_tensor = torch.as_tensor(mask_tensor).to(self.device)
        best, dbg = self.leaf_scorer.select_optimal_leaf(mask_tensor, depth_tensor, return_debug=True) or (None, None)
        if dbg is None:
            return []
        w = np.array([0.35, 0.35, 0.3])
        out = []
        for c in dbg["candidates"]:
            # the selector's own area floor (10000, leaf_scorer.py:80) subsumes the node's 3500 (:156)
            sc = c["scores"]
            out.append({
                "leaf_id": c["leaf_id"], "x": float(c["centroid"][0]), "y": float(c["centroid"][1]),
                "geometric_score": float(np.sum(w * sc)), "clutter_score": float(sc[0]),
                "distance_score": float(sc[1]), "visibility_score": float(sc[2]),
                "mask": mask_tensor == c["leaf_id"],
            })
        out.sort(key=lambda x: x["geometric_score"], reverse=True)
        return out[:top_n]

    def select_grasp_point(self, image, candidates, depth_tensor=None,
                           instruction="Select the best leaf for grasping"):
        """Returns (grasp_point_2d, grasp_point_3d, pre_grasp_point) for the hybrid-selected leaf, or
        (None, None, None).  `image` is the left camera image handed to the VLA scorer (may be None)."""
        if not candidates:
            logwarn("No valid candidates found")
            return None, None, None
        geometric_scores = [c["geometric_score"] for c in candidates]
        best = None
        if self.vla_enabled and image is not None:
            try:  # leaf_grasp_node_vla.py:116-137
                vla_scores = self.vla_scorer.evaluate_candidates(image, candidates, instruction)
                conf = self.confidence_manager.calculate_confidence(vla_scores, geometric_scores)
                best = self.hybrid_selector.select_best_candidate(candidates, geometric_scores, vla_scores, conf)
                loginfo(f"Selection strategy: {self.hybrid_selector.get_selection_strategy(conf)}")
            except Exception as e:  # noqa: BLE001
                logwarn(f"VLA processing failed, using traditional CV: {e}")
                best = None
        if best is None:
            best = max(candidates, key=lambda x: x["geometric_score"])
        self.last_selection = best
        if depth_tensor is None:
            depth_tensor = best.get("depth")
        if depth_tensor is None:
            logwarn("No depth tensor supplied")
            return None, None, None
        if self.image_processor is None:
            H, W = best["mask"].shape[-2:]
            self.image_processor = ImageProcessor(H, W, 21, 5)
        return self.grasp_selector.select_grasp_point(best["mask"], depth_tensor, self.image_processor, pcl_data=None)
