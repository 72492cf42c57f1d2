#!/usr/bin/env python3
"""
Round-3 additions to the reference-generated fixtures (same method as make_golden.py: the reference is IMPORTED
read-only from /root/reference with a rospy logger stub and import-only cv2 / skfmm / paretoset stubs).

  * ImageProcessor(H, W, 21, size).smooth_depth and GraspPointSelector._calculate_flatness_map for Gaussian sizes
    1, 3, 7 (scripts/utils/image_processor.py:25-32,56-64; grasp_point_selector.py:635-657) on the frame of
    reference_vectors.npz -- the planes a caller gets when it hands select_grasp_point an ImageProcessor other than
    the node's size 5 -- and smooth_depth's (H+1) x (W+1) result for the even size 4.
  * LLaVAProcessor's live path WITHOUT weights (vla_system/llava_processor.py:33-122): the class is instantiated without
    running load_model (no from_pretrained call is ever made: there is no network and no weights on this filesystem) and
    given a scripted stand-in for the (processor, model) pair, so that processor(prompt, image) -> generate(max_new_tokens,
    do_sample) -> decode -> split("assistant") -> float -> clip / 0.5 runs exactly as written.  Recorded: the prompt text,
    the per-candidate raw scores, _normalize_scores, get_confidence, the generate() keyword arguments.

Writes tests/golden/reference_vectors_r3.npz (data only).  Runnable only where /root/reference exists.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import REF, REPO, _install_stubs, _load_by_path  # noqa: E402
sys.path.insert(0, os.path.join(REPO, "tests"))
from scripted_llava import ScriptedModel, ScriptedProcessor  # noqa: E402  (the stand-in pair, shared with the tests)

# decoded texts a LLaVA-style chat model could return for one candidate (skip_special_tokens leaves the role words in);
# "!raise" makes the scripted generate() raise instead
RESPONSES = [
    "system\nYou are an expert ...\nuser\n\nTask: ...\nassistant\n0.73",
    "user ... assistant 1.7",
    "assistant\n-0.2",
    "assistant no number here",
    "assistant 0.5 because the leaf is isolated",
    "my assistant says: assistant 0.25 ",
    "!raise",
    "0.9",
    "assistant\n.5",
    "assistant 1e-1",
]


def main():
    _install_stubs()
    from scripts.utils.grasp_point_selector import GraspPointSelector
    from scripts.utils.image_processor import ImageProcessor
    out = {}
    base = np.load(os.path.join(HERE, "reference_vectors.npz"))
    mask, depth = base["mask"], base["depth"]
    H, W = mask.shape
    sel = GraspPointSelector(torch.device("cpu"))
    sel.set_camera_params(base["P"])
    dm = torch.from_numpy(depth) * torch.from_numpy(mask).float()
    for size in (1, 3, 7):
        ip = ImageProcessor(H, W, 21, size)
        out[f"gaussian_{size}"] = ip.get_kernel("gaussian", torch.device("cpu")).numpy()
        out[f"smooth_{size}"] = ip.smooth_depth(dm, torch.device("cpu")).numpy()
        out[f"flatness_{size}"] = sel._calculate_flatness_map(dm, ip).cpu().numpy()
    ip4 = ImageProcessor(H, W, 21, 4)
    out["gaussian_4"] = ip4.get_kernel("gaussian", torch.device("cpu")).numpy()
    out["smooth_4"] = ip4.smooth_depth(dm, torch.device("cpu")).numpy()
    assert out["smooth_4"].shape == (H + 1, W + 1)

    # ---- LLaVAProcessor with a scripted (processor, model) pair; __init__ / load_model are NOT run
    sys.modules["cv2"].cvtColor = lambda img, code: np.ascontiguousarray(img[..., ::-1])   # evaluate_candidates :39 only
    sys.modules["cv2"].COLOR_BGR2RGB = 4
    pkg = types.ModuleType("vla_system_ref")
    pkg.__path__ = [os.path.join(REF, "vla_system")]
    sys.modules["vla_system_ref"] = pkg
    lp = _load_by_path("vla_system_ref.llava_processor", os.path.join(REF, "vla_system", "llava_processor.py"))
    proc = lp.LLaVAProcessor.__new__(lp.LLaVAProcessor)
    proc.device = "cpu"
    proc.processor, proc.model = ScriptedProcessor(RESPONSES), ScriptedModel(RESPONSES)
    rng = np.random.default_rng(17)
    cands = [{"leaf_id": i + 1, "x": float(np.round(rng.uniform(0, 1440), 2)), "y": float(np.round(rng.uniform(0, 1080), 2)),
              "geometric_score": float(rng.random()), "clutter_score": float(rng.random()),
              "distance_score": float(rng.random())} for i in range(len(RESPONSES))]
    cands[3].pop("clutter_score")          # .get defaults of _create_evaluation_prompt (:64-66)
    cands[4].pop("x")
    image = rng.integers(0, 255, (24, 32, 3), dtype=np.uint8)
    out["llava_responses"] = np.array(RESPONSES)
    out["llava_cand_keys"] = np.array(["leaf_id", "x", "y", "geometric_score", "clutter_score", "distance_score"])
    out["llava_cands"] = np.array([[c.get(k, np.nan) for k in out["llava_cand_keys"]] for c in cands], np.float64)
    out["llava_image"] = image
    instruction = "Select the best leaf for grasping"
    out["llava_instruction"] = np.array(instruction)
    out["llava_prompts"] = np.array([proc._create_evaluation_prompt(c, instruction) for c in cands])
    from PIL import Image
    pil = Image.fromarray(image[..., ::-1].copy())
    raw = [float(proc._evaluate_single_candidate(pil, p)) for p in out["llava_prompts"]]
    out["llava_raw"] = np.array(raw)
    out["llava_generate_kwargs"] = np.array(sorted(f"{k}={v}" for k, v in proc.model.kwargs[0].items()))
    proc.processor, proc.model = ScriptedProcessor(RESPONSES), ScriptedModel(RESPONSES)
    out["llava_eval"] = np.array(proc.evaluate_candidates(image, cands, instruction))
    assert list(proc.processor.prompts) == list(out["llava_prompts"])
    script2 = ["assistant 0.62", "assistant\n0.4", "assistant 0.55"]   # min-max normalisation that is not the identity
    proc.processor, proc.model = ScriptedProcessor(script2), ScriptedModel(script2)
    out["llava_responses2"] = np.array(script2)
    out["llava_eval2"] = np.array(proc.evaluate_candidates(image, cands[:3], instruction))
    out["llava_conf_of_eval2"] = np.array(float(proc.get_confidence(list(out["llava_eval2"]))))
    out["llava_norm_of_raw"] = np.array(proc._normalize_scores(raw))
    out["llava_conf_of_eval"] = np.array(float(proc.get_confidence(list(out["llava_eval"]))))
    # helper edge cases (:103-122)
    out["llava_norm_const"] = np.array(proc._normalize_scores([0.4, 0.4, 0.4]))
    out["llava_norm_empty_len"] = np.array(len(proc._normalize_scores([])))
    out["llava_conf_empty"] = np.array(float(proc.get_confidence([])))
    out["llava_conf_cases_in"] = np.array([[0.2, 0.3, 0.25], [0.9, 0.1, 0.5], [0.5, 0.5, 0.5], [1.0, 0.0, 0.3]])
    out["llava_conf_cases_out"] = np.array([float(proc.get_confidence(list(r))) for r in out["llava_conf_cases_in"]])
    # no model -> 0.5 for everyone (:35-36); a failure before the loop (bad image) -> 0.5 for everyone (:50-52)
    proc.model = None
    out["llava_eval_no_model"] = np.array(proc.evaluate_candidates(image, cands[:3], instruction))
    proc.processor, proc.model = ScriptedProcessor(RESPONSES), ScriptedModel(RESPONSES)
    out["llava_eval_bad_image"] = np.array(proc.evaluate_candidates("not an image", cands[:4], instruction))

    path = os.path.join(HERE, "reference_vectors_r3.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "with", len(out), "arrays")
    for k in ("llava_raw", "llava_eval", "llava_conf_of_eval", "llava_generate_kwargs", "llava_eval_bad_image"):
        print(k, out[k].tolist())


if __name__ == "__main__":
    main()
