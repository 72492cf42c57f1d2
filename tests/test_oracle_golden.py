"""Pin the CPU oracle (oracle/lg_oracle.py) against vectors produced by the reference itself
(tests/golden/reference_vectors.npz, generator: tests/golden/make_golden.py) and against the known
answers recorded in SURVEY.md Appendix C."""
import os

import numpy as np
import pytest

from oracle import lg_oracle as O


def _sel(g):
    s = O.RefGraspPointSelector()
    s.set_camera_params(g["P"])
    return s


def test_camera_params(golden):
    s = _sel(golden)
    np.testing.assert_allclose([s.f_norm, s.camera_cx, s.camera_cy, s.baseline], golden["cam"], rtol=0, atol=0)


def test_accessibility_and_approach(golden):
    s = _sel(golden)
    m = golden["mask"]
    np.testing.assert_allclose(s._calculate_accessibility_score(m), golden["accessibility"], rtol=1e-13, atol=1e-15)
    np.testing.assert_allclose(s.calculate_approach_vector_score(m), golden["approach"], rtol=1e-13, atol=1e-15)


def test_appendix_c_planes_native():
    # native 1080x1440 frame, mask all ones
    s = O.RefGraspPointSelector()
    s.set_camera_params(np.array([[1750.68, 0, 707.87, -200.0], [0, 1749.7, 494.07, 0], [0, 0, 1, 0]]))
    assert s.baseline == pytest.approx(0.11424132337148993, rel=1e-15)
    m = np.ones((1080, 1440), np.uint8)
    a = s.calculate_approach_vector_score(m)
    c = s._calculate_accessibility_score(m)
    pts = [(0, 0), (707, 494), (708, 494), (1439, 1079), (100, 900)]
    exp_a = [0.8968928167807491, 0.9999998757211143, 0.9999999964435806, 0.8818036508971632, 0.9227963283113726]
    exp_c = [0.11829085959837016, 0.4006269499222909, 0.9640839112203109, 0.5701323341664967, 0.16625684091600654]
    for (x, y), ea, ec in zip(pts, exp_a, exp_c):
        assert a[y, x] == pytest.approx(ea, rel=1e-14)
        assert c[y, x] == pytest.approx(ec, rel=1e-13)


def test_gaussian_and_sobel(golden):
    np.testing.assert_array_equal(O.gaussian_kernel(5), golden["gaussian"])
    np.testing.assert_array_equal(O.SOBEL_X, golden["sobel_x"])
    np.testing.assert_array_equal(O.SOBEL_Y, golden["sobel_y"])
    assert O.gaussian_kernel(5)[2, 2] == pytest.approx(0.2298611, rel=1e-6)


def test_flatness(golden):
    dm = golden["depth"] * golden["mask"].astype(np.float32)
    np.testing.assert_allclose(O.smooth_depth(dm), golden["smooth"], rtol=2e-6, atol=1e-7)
    # tolerance: flatness = exp(-5 |grad|); summation order of torch's conv differs from ours
    np.testing.assert_allclose(O.flatness_map(dm), golden["flatness"], rtol=1e-4, atol=1e-7)
    dm2 = golden["depth2"] * golden["mask2"].astype(np.float32)
    np.testing.assert_allclose(O.flatness_map(dm2), golden["flatness2"], rtol=1e-4, atol=1e-7)


def test_flatness_plane_known_answer():
    yy, xx = np.mgrid[0:1080, 0:1440]
    d = (0.45 + 1e-4 * xx + 5e-5 * yy).astype(np.float32)
    f = O.flatness_map(d)
    assert f[500, 700] == pytest.approx(0.9955377578735352, rel=2e-6)
    assert f[0, 0] == pytest.approx(0.9999998211860657, rel=2e-6)
    assert f[0, 700] == pytest.approx(0.9960076212882996, rel=2e-6)


def test_valid_regions(golden):
    s = _sel(golden)
    v = s._get_valid_regions(golden["mask"], {"distance_map": golden["vr_dist"], "stem_penalty": golden["vr_stem"]})
    np.testing.assert_array_equal(v, golden["valid_regions"])


@pytest.mark.parametrize("tie_rule", ["total", "numpy"])
def test_candidate_points(golden, tie_rule):
    s = _sel(golden)
    sm = golden["cand_sm"]
    got = s._get_candidate_points(sm, np.ones_like(sm, bool), 6, 10, tie_rule)
    assert got == [tuple(r) for r in golden["cand_sm_out"].tolist()]
    assert got == [(76, 12), (50, 29), (37, 54), (94, 37), (3, 5), (76, 61)]  # Appendix C
    got2 = s._get_candidate_points(golden["cand_sm2"], golden["cand_valid2"], 20, 10, tie_rule)
    # fixture has >= 20 strictly positive distinct candidates => independent of the tie rule
    assert got2 == [tuple(r) for r in golden["cand_sm2_out"].tolist()]


def test_candidate_fall_through_rule():
    # Appendix C fall-through row: numpy's tie order is not portable; the build's rule is
    # (score desc, flat index desc): zero-score picks start from the bottom-right corner.
    s = O.RefGraspPointSelector()
    sm = np.zeros((40, 40))
    sm[20, 20], sm[20, 25] = 1.0, 0.9
    got = s._get_candidate_points(sm, np.ones_like(sm, bool), 4, 3)
    assert got[0] == (20, 20)
    assert got[1:] == [(39, 39), (32, 39), (25, 39)]


def test_patch_extraction(golden):
    s = O.RefGraspPointSelector
    for i, (x, y) in enumerate(golden["patch_pts"].tolist()):
        np.testing.assert_array_equal(s._extract_local_patch(golden["patch_plane"], x, y), golden["patch_np"][i])
        np.testing.assert_array_equal(s._extract_local_patch(golden["depth"], x, y), golden["patch_torch"][i])
    # Appendix B.7 behaviour recorded from the reference: bool mask + border patch -> None
    H, W = golden["mask"].shape
    for (x, y), is_none in zip(golden["patch_pts"].tolist(), golden["patch_bool_is_none"].tolist()):
        assert (x < 16 or y < 16 or x + 16 > W or y + 16 > H) == is_none


def test_cnn_forward(golden):
    p = O.cnn_closed_form_params(seed=0)
    assert sum(v.size for k, v in p.items() if "running" not in k) == int(golden["cnn_param_count"]) == 1258818
    x = O.synthetic_patches(20, seed=int(golden["cnn_x_seed"]))
    got = O.cnn_forward(p, x)
    np.testing.assert_allclose(got, golden["cnn_logits"], rtol=1e-4, atol=1e-5)
    import torch
    got64 = O.cnn_forward(p, x, dtype=torch.float64)
    np.testing.assert_allclose(got64, golden["cnn_logits_f64"], rtol=1e-9, atol=1e-11)


def test_ml_score_end_to_end(golden):
    s = _sel(golden)
    p = O.cnn_closed_form_params(seed=0)
    scores = {k[3:]: golden[k] for k in golden.files if k.startswith("ml_") and k not in
              ("ml_pts", "ml_scores", "ml_post_in", "ml_post_out")}
    for pt, exp in zip(golden["ml_pts"].tolist(), golden["ml_scores"].tolist()):
        feat = s.patch_features(golden["mask"], golden["depth"], scores, tuple(pt))
        got = s.ml_post(O.cnn_forward(p, feat[None])[0])
        assert got == pytest.approx(exp, rel=1e-5)
    for v, e in zip(golden["ml_post_in"], golden["ml_post_out"]):
        assert s.ml_post(v) == pytest.approx(e, rel=1e-6)
    assert s.ml_post(-2) == pytest.approx(0.6715530169831622, rel=1e-7)  # Appendix C
    assert s.ml_post(0) == pytest.approx(0.9525741268224333, rel=1e-7)
    assert s.ml_post(2) == pytest.approx(0.9949574219138471, rel=1e-7)


def test_3d_and_projection(golden):
    s = _sel(golden)
    np.testing.assert_allclose(s.get_3d_grasp_point((64, 48), golden["depth"]), golden["g3d"], rtol=1e-15)
    assert list(s._project_point_to_2d((0.013, -0.021, 0.47))) == golden["proj2d"].tolist()


def test_visibility(golden):
    from tests.golden.make_golden import ellipse_mask
    e = ellipse_mask(1080, 1440, 700, 500, 260, 140, 0).astype(bool)
    assert O.visibility_score(e) == pytest.approx(0.9503096005000047, rel=1e-14)
    e2 = e.copy()
    e2[0, 5] = True
    assert O.visibility_score(e2) == 0.0
    assert O.visibility_score(ellipse_mask(96, 128, 40, 60, 20, 10, 45).astype(bool)) == pytest.approx(
        float(golden["vis"][2]), rel=1e-14)


def test_cnn_attention_variants_match_reference():
    """'channel' / 'hybrid' / 'none' attention (model.py:30-60,108-121): oracle restatement vs the reference module's own
    outputs (tests/golden/make_golden_cnn_variants.py)."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "cnn_variant_vectors.npz"))
    x = O.synthetic_patches(int(g["n"]), seed=int(g["x_seed"]))
    for att in ("channel", "hybrid", "none"):
        p = O.cnn_closed_form_params(seed=1, attention_type=att)
        np.testing.assert_allclose(O.cnn_forward(p, x), g[f"logits_{att}"], rtol=1e-5, atol=1e-6, err_msg=att)
    # encoder_filters of the sweep (train_model_mlflow.py:177-182), paired like scripts/demo_mlflow_setup.py:44-49
    for name, filt, att in (("lightweight", (32, 64, 128), "spatial"), ("deep", (64, 128, 256, 512), "hybrid"),
                            ("wide", (128, 256, 512), "none")):
        p = O.cnn_closed_form_params(seed=2, attention_type=att, filters=filt)
        np.testing.assert_allclose(O.cnn_forward(p, x), g[f"logits_{name}"], rtol=1e-5, atol=1e-6, err_msg=name)
