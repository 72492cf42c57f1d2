"""LLaVA candidate scorer for BASELINE config 5 -- mirror of vla_system/llava_processor.py::LLaVAProcessor
on stock PyTorch-ROCm (bf16).  Weights are loaded from a LOCAL path only (there is no network; the reference's
hub name "llava-hf/llava-v1.6-mistral-7b-hf", llava_processor.py:20, cannot be resolved here).  Without a
model every candidate scores 0.5, exactly the reference's fallback (llava_processor.py:35-36)."""
from typing import Dict, List

import numpy as np
import torch

from ._log import loginfo, logwarn


class LLaVAScorer:
    def __init__(self, device="cuda:0", model_path=None, dtype=torch.bfloat16):
        self.device = device
        self.model = None
        self.processor = None
        if model_path:
            self.load_model(model_path, dtype)

    def load_model(self, model_path, dtype=torch.bfloat16):  # llava_processor.py:18-31
        try:
            from transformers import LlavaNextForConditionalGeneration, LlavaNextProcessor

            self.processor = LlavaNextProcessor.from_pretrained(model_path, local_files_only=True)
            self.model = LlavaNextForConditionalGeneration.from_pretrained(
                model_path, torch_dtype=dtype, low_cpu_mem_usage=True, local_files_only=True).to(self.device)
            loginfo(f"LLaVA model loaded on {self.device}")
        except Exception as e:  # noqa: BLE001
            logwarn(f"Failed to load LLaVA model: {e}")
            self.model = None

    def evaluate_candidates(self, image: np.ndarray, candidates: List[Dict],
                            instruction: str = "Select the best leaf for grasping") -> List[float]:  # :33-52
        if self.model is None:
            return [0.5] * len(candidates)
        try:
            from PIL import Image

            pil = Image.fromarray(np.ascontiguousarray(image[..., ::-1]))  # BGR -> RGB (cv2.cvtColor, :39)
            scores = [self._evaluate_single_candidate(pil, self._create_evaluation_prompt(c, instruction))
                      for c in candidates]
            return self._normalize_scores(scores)
        except Exception as e:  # noqa: BLE001
            logwarn(f"VLA evaluation failed: {e}")
            return [0.5] * len(candidates)

    def _create_evaluation_prompt(self, candidate: Dict, instruction: str) -> str:  # :54-77
        return f"""<|im_start|>system
You are an expert robotic vision system evaluating leaf grasp candidates.
<|im_end|>
<|im_start|>user
<image>
Task: {instruction}

Candidate details:
- Position: ({candidate.get('x', 0)}, {candidate.get('y', 0)})
- Geometric score: {candidate.get('geometric_score', 0.5):.3f}
- Clutter score: {candidate.get('clutter_score', 0.5):.3f}
- Distance score: {candidate.get('distance_score', 0.5):.3f}

Rate this candidate from 0.0 to 1.0 for grasping suitability. Consider:
1. Leaf isolation and accessibility
2. Surface quality for stable grasping
3. Positioning relative to other leaves

Respond with only a decimal number between 0.0 and 1.0.
<|im_end|>
<|im_start|>assistant
"""

    def _evaluate_single_candidate(self, image, prompt: str) -> float:  # :79-101
        try:
            # (by keyword: the reference's positional (prompt, image) is the transformers 4.3x order; 5.x takes (images, text))
            inputs = self.processor(text=prompt, images=image, return_tensors="pt").to(self.device)
            with torch.no_grad():
                output = self.model.generate(**inputs, max_new_tokens=10, do_sample=False, temperature=0.1)  # :85-90
            response = self.processor.decode(output[0], skip_special_tokens=True).split("assistant")[-1].strip()
            try:
                return float(np.clip(float(response), 0.0, 1.0))
            except Exception:  # noqa: BLE001
                return 0.5
        except Exception as e:  # noqa: BLE001
            logwarn(f"Single candidate evaluation failed: {e}")
            return 0.5

    def _normalize_scores(self, scores: List[float]) -> List[float]:  # :103-112
        if not scores:
            return []
        s = np.array(scores)
        if np.std(s) < 1e-6:
            return [0.5] * len(s)
        return ((s - np.min(s)) / (np.max(s) - np.min(s))).tolist()

    def get_confidence(self, scores: List[float]) -> float:  # :114-122
        if not scores:
            return 0.0
        s = np.array(scores)
        return float(np.clip(np.max(s) * (1 + (np.max(s) - np.min(s))), 0.0, 1.0))
