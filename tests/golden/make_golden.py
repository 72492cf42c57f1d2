#!/usr/bin/env python3
"""
Generate tests/golden/*.npz by IMPORTING the reference (read-only, /root/reference) in the build
container.  Only the importable, third-party-free functions are run: rospy is
replaced by a logger stub (the reference's own pattern, vla_system/demos/test_vla_simple.py:10-15)
and cv2 / skfmm / paretoset by empty import-only stubs, so any function that needs them cannot
produce a vector here (those rows stay "parity unpinned").

The fixtures are DATA (inputs + the reference's outputs).  The reference never travels to the GPU
box; this script is only runnable where /root/reference exists.

    python tests/golden/make_golden.py
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)


class _Rospy(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("log"):
            return lambda *a, **k: None
        raise AttributeError(name)


def _install_stubs():
    sys.modules["rospy"] = _Rospy("rospy")
    for m in ("cv2", "skfmm", "paretoset"):
        sys.modules[m] = types.ModuleType(m)
    sys.modules["paretoset"].paretoset = None
    sys.path.insert(0, REF)


def _load_by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def ellipse_mask(H, W, cx, cy, a, b, ang_deg):
    yy, xx = np.mgrid[0:H, 0:W]
    t = np.deg2rad(ang_deg)
    u = (xx - cx) * np.cos(t) + (yy - cy) * np.sin(t)
    v = -(xx - cx) * np.sin(t) + (yy - cy) * np.cos(t)
    return ((u / a) ** 2 + (v / b) ** 2 <= 1.0).astype(np.uint8)


def main():
    _install_stubs()
    from scripts.utils.grasp_point_selector import GraspPointSelector
    from scripts.utils.image_processor import ImageProcessor
    from scripts.utils.ml_grasp_optimizer.model import GraspPointCNN
    from scripts.utils.leaf_scorer import OptimalLeafSelector
    from oracle import lg_oracle as O  # only for the closed-form CNN fill + synthetic inputs

    out = {}
    H, W = 96, 128
    P = np.array([[1750.68 * W / 1440, 0, 707.87 * W / 1440, -200.0],
                  [0, 1749.7 * H / 1080, 494.07 * H / 1080, 0], [0, 0, 1, 0]], np.float64)
    sel = GraspPointSelector(torch.device("cpu"))
    sel.set_camera_params(P)
    ip = ImageProcessor(H, W, 21, 5)

    rng = np.random.default_rng(1234)
    mask = ellipse_mask(H, W, 70.3, 50.2, 40, 22, 30.0)
    yy, xx = np.mgrid[0:H, 0:W]
    depth = (0.45 + 1e-3 * xx + 5e-4 * yy + rng.normal(0, 0.002, (H, W))).astype(np.float32)
    out["P"] = P
    out["mask"] = mask
    out["depth"] = depth
    out["cam"] = np.array([sel.f_norm, sel.camera_cx, sel.camera_cy, sel.baseline], np.float64)

    # --- closed-form planes (grasp_point_selector.py:502-524, 569-593)
    out["accessibility"] = sel._calculate_accessibility_score(mask)
    out["approach"] = sel.calculate_approach_vector_score(mask, None)

    # --- flatness (grasp_point_selector.py:635-657 + image_processor.py:56-64)
    dm = torch.from_numpy(depth) * torch.from_numpy(mask).float()
    out["flatness"] = sel._calculate_flatness_map(dm, ip).cpu().numpy()
    out["smooth"] = ip.smooth_depth(dm, torch.device("cpu")).numpy()
    out["gaussian"] = ip.get_kernel("gaussian", torch.device("cpu")).numpy()
    out["sobel_x"] = ip.get_kernel("sobel_x", torch.device("cpu")).numpy()
    out["sobel_y"] = ip.get_kernel("sobel_y", torch.device("cpu")).numpy()
    # second flatness case: bigger, leaf boundary + noise, odd-ish size
    H2, W2 = 135, 180
    mask2 = ellipse_mask(H2, W2, 90.0, 70.0, 60, 30, 115.0)
    yy2, xx2 = np.mgrid[0:H2, 0:W2]
    depth2 = (0.52 - 2e-4 * xx2 + 1e-4 * yy2 + rng.normal(0, 0.002, (H2, W2))).astype(np.float32)
    ip2 = ImageProcessor(H2, W2, 21, 5)
    out["mask2"] = mask2
    out["depth2"] = depth2
    out["flatness2"] = sel._calculate_flatness_map(
        torch.from_numpy(depth2) * torch.from_numpy(mask2).float(), ip2).cpu().numpy()

    # --- valid regions (grasp_point_selector.py:282-288) on synthetic score planes
    dist_map = (rng.random((H, W)) * 60).astype(np.float32)
    stem = (rng.random((H, W)) > 0.9).astype(np.float32)
    out["vr_dist"] = dist_map
    out["vr_stem"] = stem
    out["valid_regions"] = sel._get_valid_regions(mask, {"distance_map": dist_map, "stem_penalty": stem})

    # --- candidate points (grasp_point_selector.py:447-482); distinct-valued maps => no tie dependence
    sm = np.random.default_rng(7).random((64, 96))
    out["cand_sm"] = sm
    out["cand_sm_out"] = np.array(sel._get_candidate_points(sm, np.ones_like(sm, bool), top_k=6, min_distance=10))
    sm2 = np.random.default_rng(11).random((H2, W2))
    g = np.exp(-(((xx2 - 95) / 50.0) ** 2 + ((yy2 - 70) / 30.0) ** 2))
    sm2 = 0.8 * g + 0.2 * sm2
    valid2 = np.zeros((H2, W2), bool)
    valid2[6:-6, 9:-9] = True  # big enough for >= 20 spaced strictly-positive candidates
    out["cand_sm2"] = sm2
    out["cand_valid2"] = valid2
    out["cand_sm2_out"] = np.array(sel._get_candidate_points(sm2, valid2, top_k=20, min_distance=10))

    # --- patch extraction (grasp_point_selector.py:392-445), numpy + torch float, interior + border
    plane = rng.random((H, W))
    pts = [(64, 48), (5, 7), (127, 95), (120, 3), (16, 16), (112, 80)]
    out["patch_plane"] = plane
    out["patch_pts"] = np.array(pts)
    out["patch_np"] = np.stack([sel._extract_local_patch(plane, x, y, 32) for x, y in pts])
    out["patch_torch"] = np.stack([sel._extract_local_patch(torch.from_numpy(depth), x, y, 32).numpy() for x, y in pts])
    # Appendix B.7: bool tensors cannot be replicate-padded -> None at the border, fine inside
    bm = torch.from_numpy(mask.astype(bool))
    out["patch_bool_is_none"] = np.array([sel._extract_local_patch(bm, x, y, 32) is None for x, y in pts])

    # --- GraspPointCNN forward, closed-form weights (model.py:5-128)
    params = O.cnn_closed_form_params(seed=0)
    net = GraspPointCNN(in_channels=9)
    sd = net.state_dict()
    for k, v in params.items():
        assert tuple(sd[k].shape) == tuple(v.shape), k
        sd[k] = torch.from_numpy(v)
    net.load_state_dict(sd)
    net.eval()
    out["cnn_param_count"] = np.array(sum(p.numel() for p in net.parameters()))
    x = O.synthetic_patches(20, seed=5)
    with torch.no_grad():
        out["cnn_logits"] = net(torch.from_numpy(x)).reshape(-1).numpy()
        out["cnn_logits_f64"] = net.double()(torch.from_numpy(x).double()).reshape(-1).numpy()
    net.float()
    out["cnn_x_seed"] = np.array(5)

    # --- get_ml_score end to end with an injected model (grasp_point_selector.py:59-143)
    sel.ml_predictor = net
    sc_rng = np.random.default_rng(21)
    scores = {k: sc_rng.random((H, W)) for k in ("sdf_score", "approach_score", "isolation_map", "accessibility_map")}
    scores["flatness_map"] = sc_rng.random((H, W)).astype(np.float32)
    scores["distance_map"] = (sc_rng.random((H, W)) * 40).astype(np.float32)
    scores["stem_penalty"] = (sc_rng.random((H, W)) > 0.5).astype(np.float32)
    scores["stem_penalty"][30:70, 40:90] = 0.0  # a constant patch: max == min branch
    for k, v in scores.items():
        out["ml_" + k] = v
    ml_pts = [(64, 48), (60, 50), (100, 70)]
    fmask = torch.from_numpy(mask.astype(np.float32))
    out["ml_pts"] = np.array(ml_pts)
    out["ml_scores"] = np.array([sel.get_ml_score(fmask, torch.from_numpy(depth), scores, p) for p in ml_pts])
    out["ml_post_in"] = np.array([-2.0, 0.0, 2.0, 0.37])
    out["ml_post_out"] = np.array([np.tanh(float(torch.sigmoid(torch.tensor(v))) * 3.0) * 0.5 + 0.5
                                   for v in out["ml_post_in"]])

    # --- camera math (grasp_point_selector.py:145-180, 821-826)
    out["g3d"] = np.array(sel.get_3d_grasp_point((64, 48), torch.from_numpy(depth)))
    out["proj2d"] = np.array(sel._project_point_to_2d((0.013, -0.021, 0.47)))

    # --- visibility (leaf_scorer.py:277-306) at native size
    ols = OptimalLeafSelector(torch.device("cpu"))
    e = ellipse_mask(1080, 1440, 700, 500, 260, 140, 0).astype(bool)
    e2 = e.copy()
    e2[0, 5] = True
    out["vis"] = np.array([ols._calculate_visibility_score(e), ols._calculate_visibility_score(e2),
                           ols._calculate_visibility_score(ellipse_mask(96, 128, 40, 60, 20, 10, 45).astype(bool))])

    # --- HybridSelector / ConfidenceManager (vla_system/*.py), loaded by path
    pkg = types.ModuleType("vla_system_ref")
    pkg.__path__ = [os.path.join(REF, "vla_system")]
    sys.modules["vla_system_ref"] = pkg
    cm = _load_by_path("vla_system_ref.confidence_manager", os.path.join(REF, "vla_system", "confidence_manager.py"))
    hs = _load_by_path("vla_system_ref.hybrid_selector", os.path.join(REF, "vla_system", "hybrid_selector.py"))
    geo, vla = [0.85, 0.65, 0.75], [0.8, 0.6, 0.7]
    c = cm.ConfidenceManager()
    conf = c.calculate_confidence(vla, geo)
    h = hs.HybridSelector(device="cpu")
    cands = [{"leaf_id": i + 1, "x": 10.0 * i, "y": 5.0 * i, "geometric_score": g} for i, g in enumerate(geo)]
    best = h.select_best_candidate(cands, geo, vla, conf)
    out["hyb_known"] = np.array([conf, best["leaf_id"], best["hybrid_score"], best["vla_weight"],
                                 best["geometric_weight"]])
    hr = np.random.default_rng(3)
    cases = []
    for n in (1, 2, 3, 5, 8):
        for _ in range(4):
            g_ = hr.random(n)
            v_ = hr.random(n)
            if _ == 3:
                v_[:] = 0.5  # constant VLA scores (LLaVA fallback, llava_processor.py:35-36)
            cmi = cm.ConfidenceManager()
            cf = float(cmi.calculate_confidence(list(v_), list(g_)))
            hsel = hs.HybridSelector(device="cpu")
            cd = [{"leaf_id": i} for i in range(n)]
            b = hsel.select_best_candidate(cd, list(g_), list(v_), cf)
            row = np.full(2 * 8 + 4, np.nan)
            row[0] = n
            row[1:1 + n] = g_
            row[9:9 + n] = v_
            row[17] = cf
            row[18] = b["leaf_id"]
            row[19] = b["hybrid_score"]
            cases.append(row)
    out["hyb_cases"] = np.array(cases)
    c2 = cm.ConfidenceManager()
    hist = [float(c2.calculate_confidence(list(hr.random(4)), list(hr.random(4)))) for _ in range(12)]
    out["conf_hist"] = np.array(hist)
    out["conf_running"] = np.array([c2.get_running_confidence(), float(c2.is_stable())])

    np.savez_compressed(os.path.join(HERE, "reference_vectors.npz"), **out)
    print("wrote", os.path.join(HERE, "reference_vectors.npz"), "with", len(out), "arrays")
    for k in ("cnn_param_count", "hyb_known", "vis", "cand_sm_out", "ml_scores", "patch_bool_is_none"):
        print(k, out[k].tolist() if out[k].size < 20 else out[k].shape)


if __name__ == "__main__":
    main()
