"""HybridSelector -- mirror of vla_system/hybrid_selector.py (CV / VLA score fusion, host side)."""
from typing import Dict, List

import numpy as np

from ._log import loginfo
from .confidence_manager import ConfidenceManager


class HybridSelector:
    def __init__(self, device=None):  # hybrid_selector.py:8-10
        self.device = device
        self.confidence_manager = ConfidenceManager()

    def select_best_candidate(self, candidates: List[Dict], geometric_scores: List[float],
                              vla_scores: List[float], vla_confidence: float) -> Dict:  # :12-34
        if not candidates:
            return None
        weights = self._calculate_weights(vla_confidence)
        hybrid_scores = self._compute_hybrid_scores(geometric_scores, vla_scores, weights)
        best_idx = np.argmax(hybrid_scores)
        best = candidates[best_idx].copy()
        best["hybrid_score"] = hybrid_scores[best_idx]
        best["vla_weight"] = weights["vla"]
        best["geometric_weight"] = weights["geometric"]
        loginfo(f"Selected candidate {best_idx} with hybrid score {hybrid_scores[best_idx]:.3f}")
        loginfo(f"Weights - VLA: {weights['vla']:.2f}, Geometric: {weights['geometric']:.2f}")
        return best

    def _calculate_weights(self, vla_confidence: float) -> Dict[str, float]:  # :36-51
        if vla_confidence > 0.8:
            vla_weight = 0.6
        elif vla_confidence > 0.5:
            vla_weight = 0.3
        elif vla_confidence > 0.2:
            vla_weight = 0.1
        else:
            vla_weight = 0.0
        return {"vla": vla_weight, "geometric": 1.0 - vla_weight}

    def _compute_hybrid_scores(self, geometric_scores, vla_scores, weights) -> List[float]:  # :53-66
        g = self._normalize_scores(np.array(geometric_scores))
        v = self._normalize_scores(np.array(vla_scores))
        return (weights["geometric"] * g + weights["vla"] * v).tolist()

    def _normalize_scores(self, scores: np.ndarray) -> np.ndarray:  # :68-78
        if len(scores) == 0:
            return scores
        mn, mx = np.min(scores), np.max(scores)
        if mx - mn < 1e-6:
            return np.ones_like(scores) * 0.5
        return (scores - mn) / (mx - mn)

    def get_selection_strategy(self, vla_confidence: float) -> str:  # :80-87
        if vla_confidence > 0.8:
            return "VLA_DOMINANT"
        elif vla_confidence > 0.5:
            return "BALANCED"
        elif vla_confidence > 0.2:
            return "GEOMETRIC_DOMINANT"
        return "GEOMETRIC_ONLY"
