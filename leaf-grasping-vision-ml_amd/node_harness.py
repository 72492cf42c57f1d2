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
        self._pool = None                  # worker thread of the chunked batch path

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

    @staticmethod
    def format_results(results, n):
        """format_result for the first n rows of an lg_grasp_result array (lg_format_grasp_results): a list of n strings, None
        where a frame has no result."""
        import ctypes as C

        from ._lib import LgError, lib
        buf = C.create_string_buffer(256 * max(1, n))
        used = C.c_int64(0)
        rc = lib.lg_format_grasp_results(results, n, buf, len(buf), C.byref(used))
        if rc != 0:
            raise LgError(f"lg_format_grasp_results failed with status {rc}")
        return [ln if ln else None for ln in C.string_at(buf, used.value).decode().split("\n")[:n]]

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

    def process_batch_device(self, mask_t, depth_t, chunks=None):
        """process_batch for label / depth tensors already on the device ([B,H,W] int16 / float32).
        `chunks` > 1 walks the batch in pieces: while the device computes the leaf statistics of chunk k+1 (worker thread;
        the C call releases the GIL), this thread does the host-side selection of chunk k and issues its grasp pass (its
        own handle and streams).  Same results (frames are independent), but measured SLOWER at 1080p -- 128 frames: 21.5
        ms in one piece, 23.7 in two, 27.2 in four; 256 frames: 41.1 / 43.6 / 46.9 -- both stages are HBM-bound, so
        running them side by side gains nothing and the smaller launches cost: the default is one piece."""
        B = mask_t.shape[0]
        n = chunks if chunks else 1
        n = max(1, min(n, B))
        bounds = [(B * k) // n for k in range(n + 1)]
        out, ids_all = [None] * B, [None] * B
        if n == 1:
            self._grasp_chunk(mask_t, depth_t, self.leaf_scorer.select_optimal_leaves_batch(mask_t, depth_t), 0, out, ids_all)
            self.last_leaf_ids = ids_all
            return out
        if self._pool is None:
            from concurrent.futures import ThreadPoolExecutor
            self._pool = ThreadPoolExecutor(max_workers=1)

        def stats(k):
            try:
                return self.leaf_scorer.leaf_statistics_batch(mask_t[bounds[k]:bounds[k + 1]], depth_t[bounds[k]:bounds[k + 1]])
            except Exception as e:  # noqa: BLE001   (select_optimal_leaves_batch's convention: log, no result)
                from ._log import logerr
                logerr(f"Error in leaf selection: {str(e)}")
                return None
        fut = self._pool.submit(stats, 0)
        for k in range(n):
            per_frame = fut.result()
            if k + 1 < n:
                fut = self._pool.submit(stats, k + 1)
            lo, hi = bounds[k], bounds[k + 1]
            ids = [None] * (hi - lo) if per_frame is None else self.leaf_scorer.select_from_statistics_batch(per_frame)
            self._grasp_chunk(mask_t[lo:hi], depth_t[lo:hi], ids, lo, out, ids_all)
        self.last_leaf_ids = ids_all
        return out

    def _grasp_chunk(self, mask_t, depth_t, ids, offset, out, ids_all):
        """optimal_mask = (mask_tensor == optimal_leaf_id) and select_grasp_point for the frames of one chunk (:118-125)."""
        B = mask_t.shape[0]
        ids_all[offset:offset + B] = ids
        keep = [b for b in range(B) if ids[b] is not None]
        if not keep:
            return
        if mask_t.dtype == torch.int16 and mask_t.is_contiguous() and depth_t.is_contiguous():
            # the comparison inside the library's first pass over the labels (frames without a leaf get an empty mask there)
            self.grasp_selector.select_grasp_points_for_leaves(mask_t, ids, depth_t, image_processor=self.image_processor)
            rows = range(B)
        else:
            idt = torch.tensor([ids[b] for b in keep], dtype=mask_t.dtype, device=self.device).reshape(-1, 1, 1)
            if len(keep) == B:
                optimal, dep = mask_t == idt, depth_t
            else:
                sel = torch.tensor(keep, device=self.device)
                optimal, dep = mask_t.index_select(0, sel) == idt, depth_t.index_select(0, sel)
            self.grasp_selector.select_grasp_points_batch(optimal, dep, image_processor=self.image_processor)
            rows = keep
        # the messages of the whole chunk in one native call (format_result's strings, character for character: Python's float
        # repr costs 0.24 ms per 128 frames -- a third of a millisecond the device waits for)
        for b, line in zip(rows, self.format_results(self.grasp_selector.last_results, len(rows))):
            out[offset + b] = line
