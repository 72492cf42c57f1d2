"""ROS-free equivalent of the working node scripts/leaf_grasp_node_v3.py: decodes the
wire arrays of msg/masks.msg (uint16[] imageData, read as int16 like the node, :188) and msg/depth.msg
(float32[] imageData, :199), runs leaf selection then grasp selection in the node's order (:102-158) and
formats the /optimal_leaf_grasp CSV (:160-178).  Height/width are parameters instead of the node's
hard-coded 1080x1440 (:32-33)."""
import numpy as np
import torch

from .grasp_point_selector import GraspPointSelector
from .image_processor import ImageProcessor
from .leaf_scorer import OptimalLeafSelector


class LeafGraspHarness:
    def __init__(self, height=1080, width=1440, device="cuda:0", load_model=True):
        self.height, self.width = height, width
        self.device = torch.device(device)
        self.kernel_size = 21              # leaf_grasp_node_v3.py:35
        self.gaussian_kernel_size = 5      # :37
        self.grasp_selector = GraspPointSelector(self.device, load_model=load_model)
        self.image_processor = ImageProcessor(height, width, self.kernel_size, self.gaussian_kernel_size)
        self.leaf_scorer = OptimalLeafSelector(self.device)
        self.latest_mask = None
        self.latest_depth = None
        self.leaf_grasp_done = False       # the /leaf_grasp_done ROS parameter (:28,108,157)
        self.last_leaf_id = None

    def camera_info_callback(self, P):     # :93-100
        P = np.asarray(P, dtype=np.float64).reshape(3, 4)
        self.grasp_selector.set_camera_params(P)
        self.leaf_scorer.set_camera_params(P)

    def mask_callback(self, image_data):   # :185-194
        a = np.asarray(image_data)
        a = a.astype(np.uint16).view(np.int16) if a.dtype != np.int16 else a
        self.latest_mask = torch.from_numpy(np.ascontiguousarray(a)).reshape(self.height, self.width)

    def depth_callback(self, image_data):  # :196-205
        self.latest_depth = torch.from_numpy(np.asarray(image_data, dtype=np.float32)).reshape(self.height, self.width)

    @staticmethod
    def format_result(grasp_point_2d, grasp_point_3d, pre_grasp_point):  # publish_results (:170-176)
        if pre_grasp_point is not None:
            return (f"{grasp_point_2d[0]},{grasp_point_2d[1]},"
                    f"{grasp_point_3d[0]},{grasp_point_3d[1]},{grasp_point_3d[2]},"
                    f"{pre_grasp_point[0]},{pre_grasp_point[1]},{pre_grasp_point[2]}")
        return f"{grasp_point_2d[0]},{grasp_point_2d[1]},{grasp_point_3d[0]},{grasp_point_3d[1]},{grasp_point_3d[2]}"

    def select_optimal_leaf(self):         # :102-158
        """Returns the CSV string the node would publish on /optimal_leaf_grasp, or None."""
        if self.latest_mask is None or self.latest_depth is None:
            return None
        self.leaf_grasp_done = False
        try:
            depth_tensor = self.latest_depth.to(self.device)
            mask_tensor = self.latest_mask.to(self.device)
            leaf_id = self.leaf_scorer.select_optimal_leaf(mask_tensor, depth_tensor)
            self.last_leaf_id = leaf_id
            if leaf_id is None:
                return None
            optimal_mask = mask_tensor == leaf_id
            p2, p3, pre = self.grasp_selector.select_grasp_point(optimal_mask, depth_tensor, self.image_processor,
                                                                 pcl_data=None)
            if p2 is None:
                return None
            return self.format_result(p2, p3, pre)
        finally:
            self.leaf_grasp_done = True

    def process(self, mask_data, depth_data):
        self.mask_callback(mask_data)
        self.depth_callback(depth_data)
        return self.select_optimal_leaf()

    def process_batch(self, mask_frames, depth_frames):
        """B frames of wire data ([B, H*W] or [B, H, W]) through the node's sequence with ONE leaf-selection pass sequence
        (lg_leaf_stats_batch) and ONE grasp-selection call (lg_select_grasp) for all frames: the batched form of
        select_optimal_leaf (:102-158).  Returns the list of CSV strings (None where the node would publish nothing)."""
        m = np.asarray(mask_frames)
        m = m.astype(np.uint16).view(np.int16) if m.dtype != np.int16 else m
        B = m.shape[0]
        mask_t = torch.from_numpy(np.ascontiguousarray(m)).reshape(B, self.height, self.width).to(self.device)
        depth_t = torch.from_numpy(np.ascontiguousarray(np.asarray(depth_frames, dtype=np.float32))).reshape(
            B, self.height, self.width).to(self.device)
        return self.process_batch_device(mask_t, depth_t)

    def process_batch_device(self, mask_t, depth_t):
        """process_batch for label / depth tensors already on the device ([B,H,W] int16 / float32)."""
        B = mask_t.shape[0]
        ids = self.leaf_scorer.select_optimal_leaves_batch(mask_t, depth_t)
        self.last_leaf_ids = ids
        out = [None] * B
        keep = [b for b in range(B) if ids[b] is not None]
        if not keep:
            return out
        idt = torch.tensor([ids[b] for b in keep], dtype=mask_t.dtype, device=self.device).reshape(-1, 1, 1)
        sel = torch.tensor(keep, device=self.device)
        optimal = mask_t.index_select(0, sel) == idt                    # optimal_mask = (mask_tensor == optimal_leaf_id)
        res = self.grasp_selector.select_grasp_points_batch(optimal, depth_t.index_select(0, sel))
        for b, (p2, p3, pre) in zip(keep, res):
            if p2 is not None:
                out[b] = self.format_result(p2, p3, pre)
        return out
