"""leafgrasp_amd -- MI355X-native implementation of the per-pixel grasp-scoring hot path of
Srecharan/Leaf-Grasping-Vision-ML behind the reference's own selector interface.

Importing this package loads liblgrasp.so (hand-written gfx950 HIP behind the C-ABI of
include/leafgrasp.h).  There is no CPU / PyTorch fallback: a missing library raises ImportError here,
and constructing a selector without a HIP device raises RuntimeError.
"""
from ._lib import LIB_PATH, MAP_NAMES, LgError, lib  # noqa: F401  (loads the library, fails loudly)
from ._lib import LgParams as ScoreParams  # every constant of the path; defaults = the reference values
from ._lib import default_params  # noqa: F401
from .confidence_manager import ConfidenceManager
from .data_collector import EnhancedGraspDataCollector
from .grasp_point_selector import GraspPointSelector
from .hybrid_grasp_selector import HybridGraspSelector
from .hybrid_selector import HybridSelector
from .image_processor import ImageProcessor
from .leaf_scorer import OptimalLeafSelector
from .node_harness import LeafGraspHarness

__all__ = ["GraspPointSelector", "ImageProcessor", "HybridSelector", "ConfidenceManager", "ScoreParams",
           "default_params", "MAP_NAMES", "LIB_PATH", "LgError", "OptimalLeafSelector", "HybridGraspSelector",
           "LeafGraspHarness", "EnhancedGraspDataCollector"]
