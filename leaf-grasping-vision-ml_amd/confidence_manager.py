"""ConfidenceManager -- mirror of vla_system/confidence_manager.py (host-side float64 arithmetic on <= 20
scores).  confidence = clip(0.4*consistency + 0.3*(1-variance) + 0.3*magnitude)."""
from typing import List

import numpy as np


class ConfidenceManager:
    def __init__(self):
        self.confidence_history = []
        self.max_history = 10  # confidence_manager.py:9

    def calculate_confidence(self, vla_scores: List[float], geometric_scores: List[float]) -> float:  # :11-29
        if not len(vla_scores) or not len(geometric_scores):
            return 0.0
        consistency = self._calculate_consistency(vla_scores, geometric_scores)
        variance = self._calculate_variance(vla_scores)
        magnitude = self._calculate_magnitude(vla_scores)
        confidence = np.clip(0.4 * consistency + 0.3 * (1 - variance) + 0.3 * magnitude, 0.0, 1.0)
        self._update_history(confidence)
        return confidence

    def _calculate_consistency(self, vla_scores, geometric_scores) -> float:  # :31-46
        v = np.array(vla_scores)
        g = np.array(geometric_scores)
        if len(v) < 2:
            return 0.5
        vn = (v - np.min(v)) / (np.max(v) - np.min(v) + 1e-6)
        gn = (g - np.min(g)) / (np.max(g) - np.min(g) + 1e-6)
        with np.errstate(invalid="ignore", divide="ignore"):
            corr = np.corrcoef(vn, gn)[0, 1]
        if np.isnan(corr):
            return 0.5
        return (corr + 1) / 2

    def _calculate_variance(self, scores) -> float:  # :48-55
        if len(scores) < 2:
            return 1.0
        return np.clip(np.var(scores) / (np.mean(scores) + 1e-6), 0.0, 1.0)

    def _calculate_magnitude(self, scores) -> float:  # :57-64
        if not len(scores):
            return 0.0
        mx = np.max(scores)
        rng = np.max(scores) - np.min(scores)
        return np.clip(mx * (1 + rng / 2), 0.0, 1.0)

    def _update_history(self, confidence: float):  # :66-69
        self.confidence_history.append(confidence)
        if len(self.confidence_history) > self.max_history:
            self.confidence_history.pop(0)

    def get_running_confidence(self) -> float:  # :71-75
        if not self.confidence_history:
            return 0.0
        return np.mean(self.confidence_history[-5:])

    def is_stable(self, threshold: float = 0.1) -> bool:  # :77-82
        if len(self.confidence_history) < 3:
            return False
        return np.std(self.confidence_history[-3:]) < threshold
