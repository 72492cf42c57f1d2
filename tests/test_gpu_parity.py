"""GPU parity: the HIP path (through the C-ABI) against the CPU oracle on the same seeded inputs, against
the reference-generated golden vectors, and at BASELINE.json's full sizes.
Tolerances (BASELINE.json north_star): float planes within 1e-4 relative (atol 1e-6 absorbs exact zeros /
sign changes); integer outputs (distance-transform fixed point, stem, valid, candidate indices) bit-exact."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
from oracle import lg_oracle as O  # noqa: E402

RTOL, ATOL = 1e-4, 1e-6


@pytest.fixture(scope="module")
def L():
    import leafgrasp_amd

    assert torch.cuda.is_available()
    return leafgrasp_amd


@pytest.fixture(scope="module")
def sel(L):
    s = L.GraspPointSelector(torch.device("cuda:0"), load_model=False)
    return s


def _oracle(P, cnn=None):
    r = O.RefGraspPointSelector(cnn=cnn)
    r.set_camera_params(P)
    return r


def _compare_maps(sel, mask, depth, P, exact_dt=True):
    sel.set_camera_params(P)
    maps, valid, theta = sel.score_maps(torch.from_numpy(mask).cuda(), torch.from_numpy(depth).cuda())
    ref = _oracle(P)
    sc = ref._calculate_all_scores(mask, depth)
    if exact_dt:  # integer work: bit-exact
        np.testing.assert_array_equal(maps["distance_map"].cpu().numpy(), sc["distance_map"])
        np.testing.assert_array_equal(maps["stem_penalty"].cpu().numpy(), sc["stem_penalty"])
    if ref._last_angle is None:
        assert theta is None
    else:
        assert theta == pytest.approx(ref._last_angle, abs=1e-6)
    for k, v in sc.items():
        np.testing.assert_allclose(maps[k].cpu().numpy(), v, rtol=RTOL, atol=ATOL, err_msg=k)
    np.testing.assert_array_equal(valid.cpu().numpy().astype(bool), ref._get_valid_regions(mask, sc))
    return maps, valid, sc


def test_golden_planes_from_reference(sel, golden):
    """accessibility / approach / flatness straight against the reference's own outputs."""
    sel.set_camera_params(golden["P"])
    maps, _, _ = sel.score_maps(torch.from_numpy(golden["mask"]).cuda(), torch.from_numpy(golden["depth"]).cuda())
    np.testing.assert_allclose(maps["accessibility_map"].cpu().numpy(), golden["accessibility"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(maps["approach_score"].cpu().numpy(), golden["approach"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(maps["flatness_map"].cpu().numpy(), golden["flatness"], rtol=RTOL, atol=ATOL)
    maps2, _, _ = sel.score_maps(torch.from_numpy(golden["mask2"]).cuda(), torch.from_numpy(golden["depth2"]).cuda())
    np.testing.assert_allclose(maps2["flatness_map"].cpu().numpy(), golden["flatness2"], rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("shape,seed", [((96, 128), 0), ((270, 360), 1), ((135, 180), 2), ((101, 131), 3),
                                        ((200, 1028), 4), ((720, 1280), 5), ((1080, 1440), 6)])
def test_score_maps_vs_oracle(sel, shape, seed):
    H, W = shape
    labels, depth, P = O.synthetic_scene(H, W, seed)
    mask = (labels == 1).astype(np.uint8)
    _compare_maps(sel, mask, depth, P)


def test_score_maps_edge_cases(sel):
    H, W = 160, 224
    _, depth, P = O.synthetic_scene(H, W, 11)
    empty = np.zeros((H, W), np.uint8)
    full = np.ones((H, W), np.uint8)
    border = np.zeros((H, W), np.uint8)
    border[H - 70:, :90] = 1           # touches the bottom-left corner: stem penalty + border patches
    specks = np.zeros((H, W), np.uint8)
    specks[40:110, 60:170] = 1
    specks[5, 5] = specks[150, 200] = specks[151, 201] = 1   # extra components must not win the contour
    ring = np.zeros((H, W), np.uint8)
    yy, xx = np.mgrid[0:H, 0:W]
    rr = np.hypot(yy - 80, xx - 110)
    ring[(rr < 70) & (rr > 35)] = 1    # hole: outer contour only
    for m in (empty, full, border, specks, ring):
        _compare_maps(sel, m, depth, P)


def _window_case(rng, H, W, kind):
    yy, xx = np.mgrid[0:H, 0:W]
    m = np.zeros((H, W), np.uint8)

    def blob(cx, cy, a, b, ang):
        t = np.deg2rad(ang)
        u = (xx - cx) * np.cos(t) + (yy - cy) * np.sin(t)
        v = -(xx - cx) * np.sin(t) + (yy - cy) * np.cos(t)
        return (u / a) ** 2 + (v / b) ** 2 <= 1.0

    if kind == 0:      # one small leaf anywhere (also hanging over the frame border)
        m |= blob(rng.uniform(-10, W + 10), rng.uniform(-10, H + 10), rng.uniform(5, W / 4), rng.uniform(4, H / 4), rng.uniform(0, 180))
    elif kind == 1:    # two far components: the window spans both, the gap rows / columns have no leaf pixel
        m |= blob(rng.uniform(0, W / 3), rng.uniform(0, H / 3), rng.uniform(3, 20), rng.uniform(3, 20), rng.uniform(0, 180))
        m |= blob(rng.uniform(2 * W / 3, W), rng.uniform(2 * H / 3, H), rng.uniform(3, 20), rng.uniform(3, 20), rng.uniform(0, 180))
    elif kind == 2:    # ring: d_out has an interior maximum (the hole), d_in a thin band
        r = np.hypot(yy - rng.uniform(H / 3, 2 * H / 3), xx - rng.uniform(W / 3, 2 * W / 3))
        r0 = rng.uniform(6, min(H, W) / 3)
        m |= (r < r0) & (r > r0 * rng.uniform(0.3, 0.8))
    elif kind == 3:    # single pixels / thin lines (corners included)
        for _ in range(int(rng.integers(1, 4))):
            m[int(rng.integers(H)), int(rng.integers(W))] = 1
        if rng.random() < 0.5:
            m[int(rng.integers(H)), :] = 1
        if rng.random() < 0.3:
            m[0, 0] = 1
        if rng.random() < 0.3:
            m[H - 1, W - 1] = 1
    elif kind == 4:    # concave "C": the far point of d_out is not a frame corner of the window
        m[H // 4:3 * H // 4, W // 4:3 * W // 4] = 1
        m[H // 3:2 * H // 3, W // 3:3 * W // 4] = 0
    else:              # leaf touching one frame border
        side = int(rng.integers(4))
        cx = (0, W - 1, rng.uniform(0, W), rng.uniform(0, W))[side]
        cy = (rng.uniform(0, H), rng.uniform(0, H), 0, H - 1)[side]
        m |= blob(cx, cy, rng.uniform(5, W / 5), rng.uniform(5, H / 5), rng.uniform(0, 180))
    return m.astype(np.uint8)


# d_in comes from the two raster sweeps (LG_DT_SEARCH=0) or from the row search (lg_hrun_kernel + one-level / anchors-and-bands
# search); the library picks per frame and batch size.  Every form must give the oracle's integers: the options are read when a
# handle is created, so each form gets a selector of its own.
DT_FORMS = {"auto": {}, "sweeps": {"LG_DT_SEARCH": "0"}, "search1": {"LG_DT_SEARCH": "1", "LG_DT_SEARCH_ALGO": "1"},
            "search2_np4": {"LG_DT_SEARCH": "1", "LG_DT_SEARCH_ALGO": "3"}, "search2_np1": {"LG_DT_SEARCH": "1", "LG_DT_SEARCH_ALGO": "4"}}


@pytest.fixture(scope="module")
def dt_sels(L):
    out = {}
    for name, env in DT_FORMS.items():
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            out[name] = L.GraspPointSelector(torch.device("cuda:0"), load_model=False)
            out[name].set_camera_params(O.synthetic_scene(64, 64, 0)[2])
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
    return out


@pytest.mark.parametrize("form", list(DT_FORMS))
def test_windowed_sweeps_are_exact(dt_sels, form):
    """The distance transforms run on the tile-aligned window around the leaf's bounding box; max d_out outside it comes
    from the closed-form chamfer norm on the frame border.  Both must reproduce the full-frame two-pass transform of the
    oracle bit for bit: distance_map, max d_in and max d_out (the sdf normaliser, grasp_point_selector.py:531-533)."""
    sel = dt_sels[form]
    rng = np.random.default_rng(77)
    shapes = [(96, 128), (150, 333), (210, 520), (300, 1030), (64, 64), (131, 258)]
    n_windowed = 0
    for case in range(60):
        H, W = shapes[case % len(shapes)]
        mask = _window_case(rng, H, W, case % 6)
        if mask.sum() == 0:
            mask[H // 2, W // 2] = 1
        depth = np.full((H, W), 0.5, np.float32)
        maps, _, _ = sel.score_maps(torch.from_numpy(mask).cuda(), torch.from_numpy(depth).cuda())
        d_in = O.distance_transform(mask, 5)
        d_out = O.distance_transform(1 - mask, 5)
        msg = f"case {case} kind {case % 6} {H}x{W}"
        np.testing.assert_array_equal(maps["distance_map"].cpu().numpy(), d_in, err_msg=msg)
        mi, mo, win = sel.dt_maxima(0)
        assert mi == d_in.max(), msg
        assert mo == d_out.max(), f"{msg}: max d_out {mo} vs {d_out.max()} (window {win})"
        n_windowed += (win[1] - win[0]) * (win[3] - win[2]) < H * W
    assert n_windowed >= 20   # the windowed path was actually exercised


@pytest.mark.parametrize("form", ["auto", "search1", "search2_np4", "search2_np1"])
def test_row_search_on_hard_masks(dt_sels, form):
    """Masks the row search has to get right beyond the window cases: a frame with a single zero pixel, full-width bands (rows
    without any zero pixel), leaves wider than 64 words' worth of columns is covered at 4K; thin diagonal strips (every
    column's minimising row differs), a comb (many medial axes), a 17-frame batch mixing all of them with an empty and a full
    mask (those two go through the sweeps)."""
    sel = dt_sels[form]
    H, W = 200, 520
    yy, xx = np.mgrid[0:H, 0:W]
    masks = []
    m = np.ones((H, W), np.uint8); m[137, 411] = 0; masks.append(m)
    m = np.ones((H, W), np.uint8); m[0, 0] = 0; masks.append(m)
    m = np.ones((H, W), np.uint8); m[H - 1, W - 1] = 0; masks.append(m)
    m = np.zeros((H, W), np.uint8); m[40:160, :] = 1; masks.append(m)                     # full-width band
    m = np.zeros((H, W), np.uint8); m[:, 100:400] = 1; masks.append(m)                    # full-height band
    m = np.ones((H, W), np.uint8); m[:, 0] = 0; masks.append(m)
    m = np.ones((H, W), np.uint8); m[H - 1, :] = 0; masks.append(m)
    masks.append((np.abs((yy - 100) - 0.35 * (xx - 260)) < 23).astype(np.uint8))         # diagonal strip
    masks.append((np.abs((yy - 100) + 1.9 * (xx - 260)) < 60).astype(np.uint8))          # steep strip
    m = np.zeros((H, W), np.uint8); m[20:180, 30:490] = 1; m[20:150, 60:480:40] = 0; masks.append(m)   # comb
    m = ((xx % 7 != 0) | (yy % 5 != 0)).astype(np.uint8); masks.append(m)                # lattice of zero pixels
    rng = np.random.default_rng(5)
    masks.append((rng.random((H, W)) < 0.97).astype(np.uint8))
    masks.append((rng.random((H, W)) < 0.5).astype(np.uint8))
    masks.append((np.hypot(yy - 100, xx - 260) < 95).astype(np.uint8))
    masks.append(((np.hypot(yy - 100, (xx - 260) / 2.5) < 90) & (np.hypot(yy - 100, xx - 260) > 30)).astype(np.uint8))
    masks.append(np.zeros((H, W), np.uint8))
    masks.append(np.ones((H, W), np.uint8))
    depth = np.full((H, W), 0.5, np.float32)
    exp = [O.distance_transform(m, 5) for m in masks]
    for i, m in enumerate(masks):
        maps, _, _ = sel.score_maps(torch.from_numpy(m).cuda(), torch.from_numpy(depth).cuda())
        np.testing.assert_array_equal(maps["distance_map"].cpu().numpy(), exp[i], err_msg=f"mask {i}")
        assert sel.dt_maxima(0)[0] == exp[i].max(), f"mask {i}"
    mb, _, _ = sel.score_maps(torch.from_numpy(np.stack(masks)).cuda(), torch.from_numpy(np.stack([depth] * len(masks))).cuda())
    for i in range(len(masks)):
        np.testing.assert_array_equal(mb["distance_map"][i].cpu().numpy(), exp[i], err_msg=f"batched mask {i}")
        assert sel.dt_maxima(i)[0] == exp[i].max(), f"batched mask {i}"
    if form == "auto":
        # search or sweeps is decided per BATCH from sums over its frames (lg_bbox_kernel): twenty times the batch is past the
        # budget -- every frame is swept --, the small batch after it is searched again (the kernel resets its own sums)
        big = [masks[i % len(masks)] for i in range(20 * len(masks))]
        mb, _, _ = sel.score_maps(torch.from_numpy(np.stack(big)).cuda(), torch.from_numpy(np.stack([depth] * len(big))).cuda())
        for i in range(0, len(big), 7):
            np.testing.assert_array_equal(mb["distance_map"][i].cpu().numpy(), exp[i % len(masks)], err_msg=f"big batch, frame {i}")
            assert sel.dt_maxima(i)[0] == exp[i % len(masks)].max(), f"big batch, frame {i}"
        assert not any(sel.dt_form(i)[0] for i in range(0, len(big), 7)), "a batch past the budget is swept"
        del mb
        ms, _, _ = sel.score_maps(torch.from_numpy(np.stack(masks[:5])).cuda(), torch.from_numpy(np.stack([depth] * 5)).cuda())
        assert all(sel.dt_form(i)[0] for i in range(5)), "a small batch is searched"
        for i in range(5):
            np.testing.assert_array_equal(ms["distance_map"][i].cpu().numpy(), exp[i], err_msg=f"small batch after the big one, mask {i}")


def test_chamfer_init_dist0_is_a_parameter(L):
    """OpenCV's INIT_DIST0 shows in the transform of an image without a zero pixel: the isolation map (always: other_leaves == 0,
    grasp_point_selector.py:605-616) and dist_inside of an all-ones mask.  Which constant the pinned opencv-python 4.10.0.84 uses
    cannot be checked here: both INT_MAX >> 2 (default) and INT_MAX go through lg_params.chamfer_init_dist0
    and must give the oracle's planes for the same value; the two differ from each other by a few per cent."""
    s = L.GraspPointSelector(torch.device("cuda:0"), load_model=False)
    H, W = 270, 360
    labels, depth, P = O.synthetic_scene(H, W, 1)
    s.set_camera_params(P)
    iso = {}
    for init0 in (O.INIT_DIST0, 2 ** 31 - 1):
        s.params.chamfer_init_dist0 = init0
        for mask in ((labels == 1).astype(np.uint8), np.ones((H, W), np.uint8)):
            maps, valid, _ = s.score_maps(torch.from_numpy(mask).cuda(), torch.from_numpy(depth).cuda())
            ref = O.RefGraspPointSelector(init_dist0=init0)
            ref.set_camera_params(P)
            sc = ref._calculate_all_scores(mask, depth)
            np.testing.assert_array_equal(maps["distance_map"].cpu().numpy(), sc["distance_map"])
            for k, v in sc.items():
                np.testing.assert_allclose(maps[k].cpu().numpy(), v, rtol=RTOL, atol=ATOL, err_msg=f"{k} init {init0}")
            np.testing.assert_array_equal(valid.cpu().numpy().astype(bool), ref._get_valid_regions(mask, sc))
        iso[init0] = maps["isolation_map"].cpu().numpy()
    a, b = iso[O.INIT_DIST0], iso[2 ** 31 - 1]
    assert 0.005 < np.max(np.abs(a - b)) < 0.08          # the same ramp, a few per cent apart
    s.params.chamfer_init_dist0 = 12345                   # below INT_MAX >> 2: refused, the reference-style None triple
    assert s.select_grasp_point(torch.from_numpy(labels == 1).cuda(), torch.from_numpy(depth).cuda(), None) == (None, None, None)


def test_batch_equals_single(sel):
    H, W = 135, 180
    frames = [O.synthetic_scene(H, W, s) for s in range(5)]
    P = frames[0][2]
    sel.set_camera_params(P)
    masks = np.stack([(f[0] == 1).astype(np.uint8) for f in frames])
    depths = np.stack([f[1] for f in frames])
    mb, vb, _ = sel.score_maps(torch.from_numpy(masks).cuda(), torch.from_numpy(depths).cuda())
    for i in range(5):
        ms, vs, _ = sel.score_maps(torch.from_numpy(masks[i]).cuda(), torch.from_numpy(depths[i]).cuda())
        for k in ms:
            np.testing.assert_array_equal(mb[k][i].cpu().numpy(), ms[k].cpu().numpy(), err_msg=k)
        np.testing.assert_array_equal(vb[i].cpu().numpy(), vs.cpu().numpy())


@pytest.mark.parametrize("form", list(DT_FORMS))
def test_full_size_1080p_and_4k_distance_transform(dt_sels, form):
    """BASELINE configs 2 and 4: bit-exact chamfer transform at 1080p and 4K + size-independent properties."""
    sel = dt_sels[form]
    for (H, W), seed in (((1080, 1920), 7), ((2160, 3840), 8)):
        labels, depth, P = O.synthetic_scene(H, W, seed)
        mask = (labels == 1).astype(np.uint8)
        sel.set_camera_params(P)
        maps, valid, _ = sel.score_maps(torch.from_numpy(mask).cuda(), torch.from_numpy(depth).cuda())
        d = maps["distance_map"].cpu().numpy()
        np.testing.assert_array_equal(d, O.distance_transform(mask, 5))
        assert np.all(d[mask == 0] == 0) and np.all(d[mask == 1] >= 1.0)
        # 1-Lipschitz w.r.t. the chamfer metric along both axes
        assert np.max(np.abs(np.diff(d, axis=0))) <= 1.0 + 1e-6 and np.max(np.abs(np.diff(d, axis=1))) <= 1.0 + 1e-6
        tr = maps["traditional_score"].cpu().numpy()
        assert np.all(np.isfinite(tr))
        st = maps["stem_penalty"].cpu().numpy()
        assert set(np.unique(st)) <= {0.0, 1.0} and np.all(st[: H - H // 3 - 15] == 0)
        v = valid.cpu().numpy().astype(bool)
        assert np.all(v <= ((d > 20) & (mask > 0)))


def test_full_size_1080p_all_planes(sel):
    labels, depth, P = O.synthetic_scene(1080, 1920, 9)
    _compare_maps(sel, (labels == 1).astype(np.uint8), depth, P)


# ----------------------------------------------------------------------------- candidates
def test_candidates_golden(sel, golden):
    sm = golden["cand_sm"]
    got = sel._get_candidate_points(sm.astype(np.float32), np.ones_like(sm, bool), 6, 10)
    ref = O.RefGraspPointSelector()._get_candidate_points(sm.astype(np.float32), np.ones_like(sm, bool), 6, 10)
    assert got == ref
    assert got == [tuple(r) for r in golden["cand_sm_out"].tolist()]  # float32 rounding keeps the order here
    got2 = sel._get_candidate_points(golden["cand_sm2"].astype(np.float32), golden["cand_valid2"], 20, 10)
    assert got2 == [tuple(r) for r in golden["cand_sm2_out"].tolist()]


@pytest.mark.parametrize("seed", range(6))
def test_candidates_vs_oracle_bit_exact(sel, seed):
    rng = np.random.default_rng(seed)
    H, W = [(64, 96), (200, 333), (135, 180), (300, 500), (97, 65), (540, 960)][seed]
    yy, xx = np.mgrid[0:H, 0:W]
    sm = (0.7 * np.exp(-(((xx - W * 0.6) / (W * 0.3)) ** 2 + ((yy - H * 0.4) / (H * 0.3)) ** 2))
          + 0.3 * rng.random((H, W))).astype(np.float32)
    valid = rng.random((H, W)) > (0.2 if seed % 2 else 0.97)  # sparse valid => zero-score fall-through too
    if seed == 4:
        sm[:] = np.round(sm, 1)  # massive ties: total order (score desc, index desc) must hold
    k, md = (20, 10) if seed != 2 else (33, 4)
    got = sel._get_candidate_points(sm, valid, k, md)
    ref = O.RefGraspPointSelector()._get_candidate_points(sm, valid, k, md)
    assert got == ref


def test_candidates_large_min_distance(sel):
    """min_distance is caller-controlled (_get_candidate_points(min_distance=...), grasp_point_selector.py:447): a suppression
    window wider than 1024 tiles (2*300+1 px at 1080p) must still equal the oracle's greedy walk; a negative one is rejected."""
    rng = np.random.default_rng(11)
    H, W = 1080, 1920
    sm = rng.random((H, W), dtype=np.float32)
    valid = np.ones((H, W), bool)
    for k, md in ((8, 300), (6, 200), (20, 64)):
        assert sel._get_candidate_points(sm, valid, k, md) == O.RefGraspPointSelector()._get_candidate_points(sm, valid, k, md), (k, md)
    assert sel._get_candidate_points(sm, valid, 5, -3) == []   # LG_ERR_INVALID -> logged, empty list


def test_candidates_fall_through(sel):
    sm = np.zeros((40, 40), np.float32)
    sm[20, 20], sm[20, 25] = 1.0, 0.9
    got = sel._get_candidate_points(sm, np.ones_like(sm, bool), 4, 3)
    assert got == [(20, 20), (39, 39), (32, 39), (25, 39)]


# ----------------------------------------------------------------------------- patches + CNN
def test_patches_vs_oracle(sel):
    H, W = 135, 180
    labels, depth, P = O.synthetic_scene(H, W, 2)
    mask = (labels == 1).astype(np.uint8)
    sel.set_camera_params(P)
    maps, _, _ = sel.score_maps(torch.from_numpy(mask).cuda(), torch.from_numpy(depth).cuda())
    ref = _oracle(P)
    pts = [(90, 70), (3, 4), (179, 134), (16, 16), (100, 5)]
    got = sel.gather_patches(torch.from_numpy(mask).cuda(), torch.from_numpy(depth).cuda(), maps, pts).cpu().numpy()
    maps_np = {k: v.cpu().numpy() for k, v in maps.items()}
    for i, p in enumerate(pts):
        exp = ref.patch_features(mask, depth, maps_np, p)
        np.testing.assert_allclose(got[i], exp, rtol=1e-6, atol=1e-7)


def test_cnn_vs_reference_golden(sel, golden):
    params = O.cnn_closed_form_params(seed=0)
    sel.set_cnn_state_dict(params)
    x = O.synthetic_patches(20, seed=int(golden["cnn_x_seed"]))
    got = sel.cnn_forward(torch.from_numpy(x).cuda()).cpu().numpy()
    np.testing.assert_allclose(got, golden["cnn_logits"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(got, golden["cnn_logits_f64"], rtol=1e-4, atol=1e-5)
    # batch independence + odd batch sizes (not bit for bit: with fewer items than workgroups a layer's items are split along
    # the input channels, and the partial sums meet in another order -- lg_wino4_kernel, "the tail")
    got7 = sel.cnn_forward(torch.from_numpy(x[:7]).cuda()).cpu().numpy()
    np.testing.assert_allclose(got7, got[:7], rtol=1e-5, atol=1e-6)
    x2 = O.synthetic_patches(64, seed=9)
    np.testing.assert_allclose(sel.cnn_forward(torch.from_numpy(x2).cuda()).cpu().numpy(),
                               O.cnn_forward(params, x2), rtol=1e-4, atol=1e-5)
    sel.clear_cnn()


def test_cnn_items_split_along_the_input_channels(sel):
    """A layer with fewer items than the device has workgroups -- or with a last round that would leave most of them idle --
    splits those items along the input channels (2, 4 or 8 parts whose partial sums meet through L2).  Which layers split,
    and how, changes with the patch count: one frame's 20 patches (every layer), config 3's 640 (the tail of the 16 x 16 and
    8 x 8 layers), counts that leave ragged tile blocks.  Every count must give the logits of the reference network, and the
    same patch the same logit (to summation order) whatever batch it travels in; repeated calls must agree exactly (the
    parts' counters are reset by the kernel itself)."""
    params = O.cnn_closed_form_params(seed=0)
    sel.set_cnn_state_dict(params)
    x = torch.from_numpy(O.synthetic_patches(1111, seed=21)).cuda()
    big = sel.cnn_forward(x).cpu().numpy()
    want = O.cnn_forward(params, x[:48].cpu().numpy())
    np.testing.assert_allclose(big[:48], want, rtol=1e-4, atol=1e-5)
    for n in (1, 2, 3, 8, 20, 33, 48, 64, 100, 257, 640, 1000):
        got = sel.cnn_forward(x[:n]).cpu().numpy()
        np.testing.assert_allclose(got, big[:n], rtol=1e-5, atol=1e-6, err_msg=f"{n} patches")
        np.testing.assert_array_equal(sel.cnn_forward(x[:n]).cpu().numpy(), got, err_msg=f"{n} patches, second call")
    sel.clear_cnn()


def test_cnn_winograd_matches_direct(sel, monkeypatch):
    """Winograd conv layers (default) vs the direct implicit-GEMM kernels, layer by layer and all together, for even / odd /
    single patch counts (8x8 layers put several patches into one workgroup).  The switches are read when the model is
    loaded (lg_cnn_load), never on the per-call path."""
    params = O.cnn_closed_form_params(seed=0)
    x = torch.from_numpy(O.synthetic_patches(41, seed=5)).cuda()
    monkeypatch.setenv("LG_CNN_DIRECT", "1")
    sel.set_cnn_state_dict(params)
    want = sel.cnn_forward(x).cpu().numpy()
    monkeypatch.delenv("LG_CNN_DIRECT")
    for mask in (1, 2, 4, 8, 16, 32, 0x3f):
        monkeypatch.setenv("LG_CNN_WINO_MASK", str(mask))
        sel.set_cnn_state_dict(params)
        got = sel.cnn_forward(x).cpu().numpy()
        np.testing.assert_allclose(got, want, rtol=2e-5, atol=2e-6, err_msg=f"wino mask {mask}")
        for n in (1, 2, 7):
            np.testing.assert_allclose(sel.cnn_forward(x[:n]).cpu().numpy(), got[:n], rtol=1e-5, atol=1e-6, err_msg=f"mask {mask} n {n}")
    monkeypatch.delenv("LG_CNN_WINO_MASK")
    # the older F(2x2,3x3) Winograd form of every layer (LG_CNN_F23=1), kept as the second opinion on the default F(4x4,3x3)
    monkeypatch.setenv("LG_CNN_F23", "1")
    sel.set_cnn_state_dict(params)
    got23 = sel.cnn_forward(x).cpu().numpy()
    np.testing.assert_allclose(got23, want, rtol=2e-5, atol=2e-6, err_msg="F(2x2,3x3)")
    monkeypatch.delenv("LG_CNN_F23")
    sel.set_cnn_state_dict(params)
    np.testing.assert_allclose(sel.cnn_forward(x).cpu().numpy(), O.cnn_forward(params, x.cpu().numpy()), rtol=1e-4, atol=1e-5)
    sel.clear_cnn()


def _wide_range_params(seed):
    """The closed-form fill with a WIDE dynamic range in the BatchNorm terms that fold into the conv weights: gamma
    log-uniform in [0.25, 4], running_var log-uniform in [1e-2, 1] (per-channel scale gamma / sqrt(var + eps) from 0.25 to
    40; trained checkpoints sit inside that), running_mean up to +-0.5.  The benign fill of synthetic_inputs.py has
    gamma in [0.9, 1.1] and var in [0.8, 1.2]."""
    params = O.cnn_closed_form_params(seed=seed)
    rng = np.random.default_rng(1000 + seed)
    for k in list(params):
        if k.startswith("encoder") and k.endswith("running_var"):
            params[k] = np.exp(rng.uniform(np.log(1e-2), 0.0, params[k].shape)).astype(np.float32)
        elif k.startswith("encoder") and k.endswith("running_mean"):
            params[k] = rng.uniform(-0.5, 0.5, params[k].shape).astype(np.float32)
        elif k.startswith("encoder") and k.endswith(".weight") and params[k].ndim == 1:
            params[k] = np.exp(rng.uniform(np.log(0.25), np.log(4.0), params[k].shape)).astype(np.float32)
    return params


@pytest.mark.parametrize("seed", [0, 1])
def test_cnn_forms_hold_1e4_on_wide_dynamic_range_weights(sel, monkeypatch, seed):
    """Winograd F(4x4,3x3) loses digits with the dynamic range of the BN-folded weights.  All three forms
    of the conv layers -- F(4x4,3x3) (default), F(2x2,3x3), direct -- against the float64 oracle at the 1e-4 bar on a fill
    whose folded per-channel scales span 0.25 ... 40."""
    import torch as T
    params = _wide_range_params(seed)
    x = O.synthetic_patches(24, seed=11 + seed)
    ref = O.cnn_forward(params, x, dtype=T.float64)
    scale = np.abs(ref).max()
    errs = {}
    for name, env in (("direct", {"LG_CNN_DIRECT": "1"}), ("f23", {"LG_CNN_F23": "1"}), ("f43", {})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        sel.set_cnn_state_dict(params)
        for k in env:
            monkeypatch.delenv(k)
        got = sel.cnn_forward(T.from_numpy(x).cuda()).cpu().numpy()
        errs[name] = float(np.abs(got - ref).max() / scale)
    print("wide-range CNN: max |err| / max |logit| =", errs, "max |logit|", scale)
    sel.clear_cnn()
    assert all(e <= 1e-4 for e in errs.values()), errs


def test_cnn_attention_variants_vs_reference(sel):
    """The sweep's attention variants (model.py:30-60) through lg_cnn_load / lg_cnn_forward vs the reference's outputs."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "cnn_variant_vectors.npz"))
    x = torch.from_numpy(O.synthetic_patches(int(g["n"]), seed=int(g["x_seed"]))).cuda()
    for att in ("channel", "hybrid", "none"):
        sel.set_cnn_state_dict(O.cnn_closed_form_params(seed=1, attention_type=att))
        got = sel.cnn_forward(x).cpu().numpy()
        np.testing.assert_allclose(got, g[f"logits_{att}"], rtol=1e-4, atol=1e-5, err_msg=att)
        np.testing.assert_allclose(got, g[f"logits_f64_{att}"], rtol=1e-4, atol=1e-5, err_msg=att)
    # encoder_filters variants (32-channel stages zero-padded to 64; a fourth 4x4 stage; 512-channel head)
    for name, filt, att in (("lightweight", (32, 64, 128), "spatial"), ("deep", (64, 128, 256, 512), "hybrid"),
                            ("wide", (128, 256, 512), "none")):
        params = O.cnn_closed_form_params(seed=2, attention_type=att, filters=filt)
        sel.set_cnn_state_dict(params)
        got = sel.cnn_forward(x).cpu().numpy()
        np.testing.assert_allclose(got, g[f"logits_{name}"], rtol=1e-4, atol=1e-5, err_msg=name)
        np.testing.assert_allclose(got, g[f"logits_f64_{name}"], rtol=1e-4, atol=1e-5, err_msg=name)
        for n in (1, 3, 9):   # 4x4 / 8x8 stages put 8 / 2 patches into one workgroup: ragged counts
            np.testing.assert_allclose(sel.cnn_forward(x[:n]).cpu().numpy(), got[:n], rtol=1e-5, atol=1e-6, err_msg=f"{name} n={n}")
    with pytest.raises(Exception):
        sel.set_cnn_state_dict(O.cnn_closed_form_params(seed=2, filters=(48, 96, 192)))
    # back to the default model: the per-handle plan is rebuilt
    sel.set_cnn_state_dict(O.cnn_closed_form_params(seed=0))
    x20 = torch.from_numpy(O.synthetic_patches(20, seed=5)).cuda()
    np.testing.assert_allclose(sel.cnn_forward(x20).cpu().numpy(), O.cnn_forward(O.cnn_closed_form_params(seed=0), x20.cpu().numpy()),
                               rtol=1e-4, atol=1e-5)
    sel.clear_cnn()


def test_ml_score_golden(sel, golden):
    sel.set_camera_params(golden["P"])
    sel.set_cnn_state_dict(O.cnn_closed_form_params(seed=0))
    scores = {k[3:]: golden[k] for k in golden.files if k.startswith("ml_") and k not in
              ("ml_pts", "ml_scores", "ml_post_in", "ml_post_out")}
    scores["traditional_score"] = np.zeros_like(golden["ml_sdf_score"])
    scores = {k: torch.from_numpy(np.asarray(v, np.float32)).cuda() for k, v in scores.items()}
    fmask = torch.from_numpy(golden["mask"].astype(np.float32)).cuda()
    for pt, exp in zip(golden["ml_pts"].tolist(), golden["ml_scores"].tolist()):
        got = sel.get_ml_score(fmask, torch.from_numpy(golden["depth"]).cuda(), scores, tuple(pt))
        assert got == pytest.approx(exp, rel=1e-4)
    # bool mask + border patch -> None (behaviour recorded from the reference)
    bmask = torch.from_numpy(golden["mask"].astype(bool)).cuda()
    assert sel.get_ml_score(bmask, torch.from_numpy(golden["depth"]).cuda(), scores, (5, 7)) is None
    sel.clear_cnn()


# ----------------------------------------------------------------------------- the whole path
@pytest.mark.parametrize("shape,seed,with_cnn", [((192, 256), 3, True), ((270, 360), 1, True), ((270, 360), 4, False),
                                                 ((720, 1280), 5, True)])
def test_select_grasp_point_vs_oracle(sel, shape, seed, with_cnn):
    H, W = shape
    labels, depth, P = O.synthetic_scene(H, W, seed)
    mask = (labels == 1).astype(np.uint8)
    sel.set_camera_params(P)
    params = O.cnn_closed_form_params(seed=0)
    if with_cnn:
        sel.set_cnn_state_dict(params)
    else:
        sel.clear_cnn()
    ref = _oracle(P, cnn=(lambda x: O.cnn_forward(params, x)) if with_cnn else None)
    exp, dbg = ref.select_grasp_point(mask, depth, return_debug=True)
    got = sel.select_grasp_point(torch.from_numpy(mask.astype(bool)).cuda(), torch.from_numpy(depth).cuda(), None)
    assert got[0] == exp[0]
    np.testing.assert_allclose(got[1], exp[1], rtol=1e-5)
    np.testing.assert_allclose(got[2], exp[2], rtol=1e-5)
    # candidates themselves (integer indices) are bit-exact
    maps, valid, _ = sel.score_maps(torch.from_numpy(mask).cuda(), torch.from_numpy(depth).cuda())
    cands = sel._get_candidate_points(maps["traditional_score"], valid, 20, 10)
    assert cands == dbg["candidates"]
    sel.clear_cnn()


def _largest_leaf(labels):
    """The label with the largest area: the benchmark's notion of 'the chosen leaf' (a big interior leaf with >= 20
    strictly positive candidates at full size), unlike label 1, which later ellipses may occlude."""
    ids, counts = np.unique(labels[labels > 0], return_counts=True)
    return int(ids[np.argmax(counts)])


@pytest.mark.parametrize("seed", [22, 23, 21])
def test_select_grasp_point_1080p_with_cnn_vs_oracle(sel, seed):
    """BASELINE config 2 end to end at its full size (grasp_point_selector.py:184-253): 1080x1920, eight planes, top-20
    candidates, GraspPointCNN rescoring, 3-D point and pre-grasp point against the float64 oracle."""
    H, W = 1080, 1920
    labels, depth, P = O.synthetic_scene(H, W, seed)
    mask = (labels == _largest_leaf(labels)).astype(np.uint8)
    sel.set_camera_params(P)
    params = O.cnn_closed_form_params(seed=0)
    sel.set_cnn_state_dict(params)
    ref = _oracle(P, cnn=lambda x: O.cnn_forward(params, x))
    exp, dbg = ref.select_grasp_point(mask, depth, return_debug=True)
    assert len(dbg["candidates"]) == 20
    tr = dbg["scores"]["traditional_score"]
    if seed != 21:   # seed 21: the whole leaf lies under the stem penalty -> zero-score fall-through (the build's total order)
        assert all(tr[y, x] > 0.5 for (x, y) in dbg["candidates"])
    got = sel.select_grasp_point(torch.from_numpy(mask.astype(bool)).cuda(), torch.from_numpy(depth).cuda(), None)
    assert got[0] == exp[0]                                        # 2-D point: integer, exact
    np.testing.assert_allclose(got[1], exp[1], rtol=1e-5)
    np.testing.assert_allclose(got[2], exp[2], rtol=1e-5)
    maps, valid, _ = sel.score_maps(torch.from_numpy(mask).cuda(), torch.from_numpy(depth).cuda())
    assert sel._get_candidate_points(maps["traditional_score"], valid, 20, 10) == dbg["candidates"]
    for k, v in dbg["scores"].items():
        np.testing.assert_allclose(maps[k].cpu().numpy(), v, rtol=RTOL, atol=ATOL, err_msg=k)
    sel.clear_cnn()


def test_config3_per_gpu_share_b32_1080p(sel):
    """BASELINE config 3: 256 frames of 1080x1920 over 8 GPUs = 32 frames per GPU in one select_grasp_points_batch call.
    Every frame of the batch equals its single-frame call; four distinct scenes equal the float64 oracle (CNN included)."""
    H, W = 1080, 1920
    seeds = [31, 32, 33, 34, 35, 36, 37, 38]
    scenes = [O.synthetic_scene(H, W, s) for s in seeds]
    P = scenes[0][2]
    sel.set_camera_params(P)
    params = O.cnn_closed_form_params(seed=0)
    sel.set_cnn_state_dict(params)
    base_m = [sc[0] == _largest_leaf(sc[0]) for sc in scenes]
    order = [(i * 5 + 3) % 8 for i in range(32)]
    masks = torch.from_numpy(np.stack([base_m[i] for i in order])).cuda()
    depths = torch.from_numpy(np.stack([scenes[i][1] for i in order])).cuda()
    batch = sel.select_grasp_points_batch(masks, depths)
    assert len(batch) == 32
    singles = {}
    for j, i in enumerate(order):
        if i not in singles:
            singles[i] = sel.select_grasp_point(masks[j], depths[j], None)
        assert batch[j] == singles[i], f"frame {j} (scene {i})"
    ref = _oracle(P, cnn=lambda x: O.cnn_forward(params, x))
    for i in (0, 3, 5, 6):
        exp = ref.select_grasp_point(base_m[i].astype(np.uint8), scenes[i][1])
        got = singles[i]
        assert got[0] == exp[0], f"scene {i}"
        np.testing.assert_allclose(got[1], exp[1], rtol=1e-5)
        np.testing.assert_allclose(got[2], exp[2], rtol=1e-5)
    sel.clear_cnn()


def test_4k_all_planes_vs_oracle(sel):
    """BASELINE config 4 (2160x3840): all eight planes + valid against the oracle (float planes 1e-4, integer work exact)."""
    labels, depth, P = O.synthetic_scene(2160, 3840, 8)
    _compare_maps(sel, (labels == _largest_leaf(labels)).astype(np.uint8), depth, P)


# ----------------------------------------------------------------------------- the caller's ImageProcessor is honoured
@pytest.fixture(scope="module")
def golden3():
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_vectors_r3.npz"))


def test_smooth_depth_through_the_library(L, golden, golden3):
    """ImageProcessor.smooth_depth (image_processor.py:56-64) = lg_smooth_depth, against the REFERENCE's own outputs for
    Gaussian sizes 1, 3, 5 (the node's), 7 and the even size 4 ((H+1) x (W+1), as F.conv2d returns)."""
    dm = torch.from_numpy(golden["depth"] * golden["mask"].astype(np.float32)).cuda()
    for size, ref in ((5, golden["smooth"]), (1, golden3["smooth_1"]), (3, golden3["smooth_3"]), (7, golden3["smooth_7"]),
                      (4, golden3["smooth_4"])):
        got = L.ImageProcessor(96, 128, 21, size).smooth_depth(dm, torch.device("cuda:0"))
        assert got.is_cuda and tuple(got.shape) == ref.shape
        np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=2e-6, atol=1e-7, err_msg=f"size {size}")
    # larger sizes and ragged shapes against the oracle's restated smooth_depth (itself pinned by the vectors above)
    rng = np.random.default_rng(5)
    for (H, W), size in (((37, 131), 9), ((64, 64), 15), ((17, 300), 11), ((200, 70), 13)):
        d = rng.random((H, W)).astype(np.float32)
        got = L.ImageProcessor(H, W, 21, size).smooth_depth(torch.from_numpy(d), "cuda:0").cpu().numpy()
        np.testing.assert_allclose(got, O.smooth_depth(d, size), rtol=2e-6, atol=1e-7)
    with pytest.raises(L.LgError):   # torch: reflect padding must be smaller than the input
        L.ImageProcessor(8, 8, 21, 15).smooth_depth(torch.zeros(7, 7), "cuda:0")
    # the C entry point takes B frames per call: three frames at once equal the three single calls
    import ctypes as C
    from leafgrasp_amd import _lib
    from leafgrasp_amd.image_processor import _handle
    x = torch.from_numpy(rng.random((3, 45, 70)).astype(np.float32)).cuda()
    out = torch.empty_like(x)
    rc = _lib.lib.lg_smooth_depth(_handle(0), x.data_ptr(), 3, 45, 70, 5, out.data_ptr(), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    ip5 = L.ImageProcessor(45, 70, 21, 5)
    for b in range(3):
        np.testing.assert_array_equal(out[b].cpu().numpy(), ip5.smooth_depth(x[b], "cuda:0").cpu().numpy())
    assert _lib.lib.lg_smooth_depth(_handle(0), x.data_ptr(), 3, 45, 70, 16, out.data_ptr(), None) == _lib.LG_ERR_UNSUPPORTED


def test_flatness_follows_the_callers_image_processor(L, sel, golden, golden3):
    """_calculate_flatness_map smooths with the ImageProcessor handed to select_grasp_point (grasp_point_selector.py:
    635-657): sizes 1 / 3 / 7 against the reference's own planes, the whole call against the oracle built with the same
    size; an even size (shape mismatch in the reference's fusion), a size beyond the kernel's halo and a foreign kernel
    end in the logged None triple instead of silently scoring with size 5."""
    H, W = golden["mask"].shape
    sel.set_camera_params(golden["P"])
    m, d = torch.from_numpy(golden["mask"]).cuda(), torch.from_numpy(golden["depth"]).cuda()
    planes = {}
    for size, ref in ((1, golden3["flatness_1"]), (3, golden3["flatness_3"]), (5, golden["flatness"]), (7, golden3["flatness_7"])):
        maps, _, _ = sel.score_maps(m, d, L.ImageProcessor(H, W, 21, size))
        planes[size] = maps["flatness_map"].cpu().numpy()
        np.testing.assert_allclose(planes[size], ref, rtol=RTOL, atol=ATOL, err_msg=f"gaussian size {size}")
        sc = sel._calculate_all_scores(golden["mask"], d, L.ImageProcessor(H, W, 21, size))   # the reference's signature
        np.testing.assert_array_equal(sc["flatness_map"], planes[size])
    assert np.abs(planes[3] - planes[7]).max() > 1e-2          # (the sizes do differ: honouring them is observable)
    for seed, size, shape in ((3, 7, (270, 360)), (1, 3, (192, 256)), (5, 7, (720, 1280)), (2, 1, (135, 180))):
        labels, depth, P = O.synthetic_scene(*shape, seed)
        mask = (labels == _largest_leaf(labels)).astype(np.uint8)
        sel.set_camera_params(P)
        ref = O.RefGraspPointSelector(gaussian_size=size)
        ref.set_camera_params(P)
        exp, dbg = ref.select_grasp_point(mask, depth, return_debug=True)
        ip = L.ImageProcessor(shape[0], shape[1], 21, size)
        got = sel.select_grasp_point(torch.from_numpy(mask.astype(bool)).cuda(), torch.from_numpy(depth).cuda(), ip)
        assert got[0] == exp[0], (seed, size)
        np.testing.assert_allclose(got[1], exp[1], rtol=1e-5)
        np.testing.assert_allclose(got[2], exp[2], rtol=1e-5)
        maps, valid, _ = sel.score_maps(torch.from_numpy(mask).cuda(), torch.from_numpy(depth).cuda(), ip)
        for k, v in dbg["scores"].items():
            np.testing.assert_allclose(maps[k].cpu().numpy(), v, rtol=RTOL, atol=ATOL, err_msg=f"{k} size {size}")
        assert sel._get_candidate_points(maps["traditional_score"], valid, 20, 10) == dbg["candidates"]
    mb = torch.from_numpy(mask.astype(bool)).cuda()
    dd = torch.from_numpy(depth).cuda()
    for bad in (L.ImageProcessor(135, 180, 21, 4), L.ImageProcessor(135, 180, 21, 9)):
        assert sel.select_grasp_point(mb, dd, bad) == (None, None, None)
    foreign = L.ImageProcessor(135, 180, 21, 5)
    foreign.kernels["gaussian"] = torch.ones(5, 5) / 25.0
    assert sel.select_grasp_point(mb, dd, foreign) == (None, None, None)
    assert sel.select_grasp_point(mb, dd, L.ImageProcessor(135, 180, 21, 5))[0] is not None


# ----------------------------------------------------------------------------- threads (SURVEY 8b "Threading")
def test_two_handles_on_two_threads_equal_the_serial_results(L):
    """rospy runs every subscriber callback on its own thread (leaf_grasp_node_v3.py:104-107,185-205): two selector
    instances driven concurrently from two threads -- 1080p, CNN on, several calls each -- return exactly what the same
    calls return one after the other."""
    import threading
    H, W = 1080, 1920
    scenes = [O.synthetic_scene(H, W, 100 + s) for s in range(4)]
    P = scenes[0][2]
    params = O.cnn_closed_form_params(seed=0)
    sels = []
    for _ in range(2):
        s = L.GraspPointSelector(torch.device("cuda:0"), load_model=False)
        s.set_camera_params(P)
        s.set_cnn_state_dict(params)
        sels.append(s)
    work = []   # per thread: a list of (masks, depths) batches of different sizes
    for t in range(2):
        items = []
        for rep in range(6):
            idx = [(t + rep + j) % 4 for j in range(1 + (rep + t) % 3)]
            items.append((torch.from_numpy(np.stack([scenes[i][0] == _largest_leaf(scenes[i][0]) for i in idx])).cuda(),
                          torch.from_numpy(np.stack([scenes[i][1] for i in idx])).cuda()))
        work.append(items)
    serial = [[sels[t].select_grasp_points_batch(m, d) for m, d in work[t]] for t in range(2)]
    assert all(r[0] is not None for batch in serial[0] + serial[1] for r in batch)
    results, errors = [None, None], []
    start = threading.Barrier(2)

    def run(t):
        try:
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                start.wait()
                results[t] = [sels[t].select_grasp_points_batch(m, d) for m, d in work[t]]
        except Exception as e:  # noqa: BLE001
            errors.append(e)
    threads = [threading.Thread(target=run, args=(t,)) for t in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    assert results == serial


def test_second_thread_on_a_busy_handle_is_refused(L):
    """One call in flight per handle: while a thread is inside lg_select_grasp, another thread entering the SAME handle
    gets LG_ERR_BUSY (and the running call's results are untouched)."""
    import ctypes as C
    import threading
    from leafgrasp_amd import _lib
    H, W, B = 1080, 1920, 48
    labels, depth, P = O.synthetic_scene(H, W, 101)
    s = L.GraspPointSelector(torch.device("cuda:0"), load_model=False)
    s.set_camera_params(P)
    s.set_cnn_state_dict(O.cnn_closed_form_params(seed=0))
    m = torch.from_numpy(np.repeat((labels == _largest_leaf(labels))[None], B, 0)).cuda()
    d = torch.from_numpy(np.repeat(depth[None], B, 0)).cuda()
    exp = s.select_grasp_points_batch(m, d)
    torch.cuda.synchronize()
    got, seen = [], []
    inside = threading.Event()

    def run():
        inside.set()
        while len(got) < 4:
            try:
                got.append(s.select_grasp_points_batch(m, d))
            except _lib.LgError as e:   # the poller was inside the handle when this call arrived: the rule cuts both ways
                assert "another thread" in str(e)
    th = threading.Thread(target=run)
    th.start()
    inside.wait()
    n, ms = C.c_int(0), C.c_double(0)
    while th.is_alive():
        seen.append(_lib.lib.lg_profile_read(s._h, b"final", C.byref(n), C.byref(ms)))
    th.join()
    assert _lib.LG_ERR_BUSY in seen, "the poller never met the running call"
    assert set(seen) <= {_lib.LG_OK, _lib.LG_ERR_BUSY}
    assert got == [exp] * 4


def test_error_convention(L):
    s = L.GraspPointSelector(torch.device("cuda:0"), load_model=False)
    m = torch.zeros((64, 64), dtype=torch.bool).cuda()
    d = torch.zeros((64, 64)).cuda()
    assert s.select_grasp_point(m, d, None) == (None, None, None)  # f_norm unset -> reference raises inside -> None triple
    s.set_camera_params(np.array([[100.0, 0, 32, -10], [0, 100, 32, 0], [0, 0, 1, 0]]))
    assert s.select_grasp_point(m[:4, :4], d[:4, :4], None) == (None, None, None)  # unsupported shape -> logged, not raised
    assert s.get_ml_score(m, d, {}, (10, 10)) is None


# ----------------------------------------------------------------------------- batches, pipeline variant, robustness
def test_batched_select_equals_single_calls_1080p(L, sel, monkeypatch):
    """BASELINE config 3 per-GPU share in miniature: a batch of 1080p frames gives exactly the per-frame results."""
    H, W = 1080, 1920
    frames = [O.synthetic_scene(H, W, 100 + s) for s in range(3)]
    P = frames[0][2]
    sel.set_camera_params(P)
    sel.set_cnn_state_dict(O.cnn_closed_form_params(seed=0))
    order = [0, 1, 2, 1, 0, 2, 2, 0]
    masks = torch.from_numpy(np.stack([frames[i][0] == 1 for i in order])).cuda()
    depths = torch.from_numpy(np.stack([frames[i][1] for i in order])).cuda()
    batch = sel.select_grasp_points_batch(masks, depths)
    singles = [sel.select_grasp_point(masks[i], depths[i], None) for i in range(3)]
    for j, i in enumerate(order):
        assert batch[j] == singles[i]
    # the experimental sub-batch pipeline (multi-stream; switch read when the handle is created) must give identical results
    monkeypatch.setenv("LG_SUBBATCH", "3")
    piped = L.GraspPointSelector(torch.device("cuda:0"), load_model=False)
    monkeypatch.delenv("LG_SUBBATCH")
    piped.set_camera_params(P)
    piped.set_cnn_state_dict(O.cnn_closed_form_params(seed=0))
    assert piped.select_grasp_points_batch(masks, depths) == batch
    sel.clear_cnn()


def _same_triple(got, exp):
    """(xy, XYZ, preXYZ) triples with NaN-aware comparison of the floats."""
    assert (got[0] is None) == (exp[0] is None) and (got[2] is None) == (exp[2] is None), (got, exp)
    if exp[0] is None:
        return
    assert tuple(got[0]) == tuple(exp[0]), (got, exp)
    np.testing.assert_allclose(np.array(got[1], np.float64), np.array(exp[1], np.float64), rtol=1e-5, equal_nan=True)
    if exp[2] is not None:
        np.testing.assert_allclose(np.array(got[2], np.float64), np.array(exp[2], np.float64), rtol=1e-5, equal_nan=True)


def test_nan_depth_on_the_leaf_follows_the_reference(sel):
    """grasp_point_selector.py:262 multiplies depth by the mask and :451 multiplies the fused score by the validity mask: a NaN
    depth pixel ON the leaf makes flatness / traditional NaN over the 7 x 7 reach of the Gaussian and the Sobel, NaN * 0 stays
    NaN for pixels that are not valid, and np.argsort(...)[::-1] (:454) lists NaN pixels FIRST.  The planes, the validity
    mask, the candidate list (NaN first, then score, ties by flat index: DESIGN 2 quirk 3) and the triple must equal the
    oracle's; a grasp point whose own depth is NaN has NaN coordinates and no pre-grasp point (:817-819)."""
    H, W = 400, 520
    labels, depth, P = O.synthetic_scene(H, W, 5)
    ids, cnt = np.unique(labels[labels > 0], return_counts=True)
    mask = (labels == ids[np.argmax(cnt)]).astype(np.uint8)
    ys, xs = np.nonzero(mask)
    cy, cx = int(ys.mean()), int(xs.mean())
    corner = np.zeros((H, W), np.uint8)
    corner[H - 150:, W - 200:] = 1      # a leaf in the frame's last rows and columns: the NaN reach ends at pixel (H-1, W-1) itself
    for case, (mask, pts) in enumerate(((mask, [(cy, cx)]),
                                        (mask, [(cy - 30, cx - 20), (cy + 25, cx + 40), (ys.min(), xs[ys.argmin()])]),
                                        (corner, [(H - 1, W - 1), (H - 60, W - 70)]))):
        d = depth.copy()
        for (y, x) in pts:
            assert mask[y, x]
            d[y, x] = np.nan
        sel.set_camera_params(P)
        maps, valid, _ = sel.score_maps(torch.from_numpy(mask).cuda(), torch.from_numpy(d).cuda())
        ref = _oracle(P)
        with np.errstate(all="ignore"):
            sc = ref._calculate_all_scores(mask, d)
            exp_valid = ref._get_valid_regions(mask, sc)
        assert np.isnan(sc["traditional_score"]).sum() >= 49
        for k, v in sc.items():
            np.testing.assert_allclose(maps[k].cpu().numpy(), v, rtol=RTOL, atol=ATOL, equal_nan=True, err_msg=f"case {case} {k}")
        np.testing.assert_array_equal(valid.cpu().numpy().astype(bool), exp_valid)
        with np.errstate(all="ignore"):
            exp_c = ref._get_candidate_points(sc["traditional_score"], exp_valid, 20, 10)
            exp = ref.select_grasp_point(mask, d)
        got_c = sel._get_candidate_points(maps["traditional_score"], valid, 20, 10)
        assert [tuple(c) for c in got_c] == [tuple(c) for c in exp_c], f"case {case}"
        assert np.isnan(sc["traditional_score"][exp_c[0][1], exp_c[0][0]])     # a NaN pixel leads the list
        got = sel.select_grasp_point(torch.from_numpy(mask.astype(bool)).cuda(), torch.from_numpy(d).cuda(), None)
        _same_triple(got, exp)
        if case == 2:
            assert got[0] is not None and np.isnan(got[1][2]) and got[2] is None   # the chosen pixel's own depth is NaN


def test_non_finite_depth_off_the_leaf_is_ignored(sel):
    """DESIGN 2 quirk 9 (a deliberate deviation): the reference's depth * mask turns a NaN / infinite depth pixel OFF the leaf into
    NaN (NaN * 0), which then spreads through flatness and the fused score; here depth * mask is a selection -- tiles without a
    leaf pixel in stencil reach never read depth at all -- so such a pixel counts as the 0 every finite depth gives.  Asserted: the
    outputs equal those of the same frame with these pixels set to 1.0; the distance transform never touches depth."""
    H, W = 270, 360
    labels, depth, P = O.synthetic_scene(H, W, 1)
    sel.set_camera_params(P)
    mask = labels == 1
    ys, xs = np.nonzero(mask)
    d = depth.copy()
    off = ~mask
    d[100:140, 150:200][off[100:140, 150:200]] = np.nan
    d[10, 10] = np.inf
    d[ys.min() - 1, xs[ys.argmin()]] = -np.inf          # right next to the leaf: inside the stencil reach
    clean = np.where(off & ~np.isfinite(d), np.float32(1.0), d)
    assert np.isfinite(clean).all()
    m8 = torch.from_numpy(mask.astype(np.uint8)).cuda()
    maps, valid, _ = sel.score_maps(m8, torch.from_numpy(d).cuda())
    maps2, valid2, _ = sel.score_maps(m8, torch.from_numpy(clean).cuda())
    for k in maps:
        np.testing.assert_array_equal(maps[k].cpu().numpy(), maps2[k].cpu().numpy(), err_msg=k)
    np.testing.assert_array_equal(valid.cpu().numpy(), valid2.cpu().numpy())
    np.testing.assert_array_equal(maps["distance_map"].cpu().numpy(), O.distance_transform(mask.astype(np.uint8), 5))
    res = sel.select_grasp_point(torch.from_numpy(mask).cuda(), torch.from_numpy(d).cuda(), None)
    res2 = sel.select_grasp_point(torch.from_numpy(mask).cuda(), torch.from_numpy(clean).cuda(), None)
    _same_triple(res, res2)


def test_pcl_data_ends_in_the_logged_none_triple(sel):
    """get_3d_grasp_point's point-cloud branch reads the undefined self.width (grasp_point_selector.py:164-178): with pcl_data the
    reference raises inside select_grasp_point, logs and returns (None, None, None) -- so does the mirror."""
    H, W = 270, 360
    labels, depth, P = O.synthetic_scene(H, W, 1)
    sel.set_camera_params(P)
    m, d = torch.from_numpy(labels == 1).cuda(), torch.from_numpy(depth).cuda()
    assert sel.select_grasp_point(m, d, None)[0] is not None
    assert sel.select_grasp_point(m, d, None, pcl_data=np.zeros((H * W, 3), np.float32)) == (None, None, None)
    assert sel.select_grasp_point(m, d, None, pcl_data=[]) == (None, None, None)


def test_orientation_scratch_failure_hands_over_to_the_host_analysis(L, monkeypatch):
    """when the device-side contour scratch cannot be set up the call goes on with the host analysis of every frame --
    the failed allocation's (sticky) HIP error must not fail that very call, and the reason is readable (lg_orientation_note)."""
    from leafgrasp_amd._lib import lib
    H, W = 270, 360
    labels, depth, P = O.synthetic_scene(H, W, 1)
    m, d = torch.from_numpy(labels == 1).cuda(), torch.from_numpy(depth).cuda()
    ok = L.GraspPointSelector(torch.device("cuda:0"), load_model=False)
    ok.set_camera_params(P)
    exp = ok.select_grasp_point(m, d, None)
    assert lib.lg_orientation_note(ok._h) == b""
    monkeypatch.setenv("LG_ORIENT_FAIL", "1")
    s = L.GraspPointSelector(torch.device("cuda:0"), load_model=False)
    s.set_camera_params(P)
    got = s.select_grasp_point(m, d, None)           # the FIRST call after the failed set-up
    monkeypatch.delenv("LG_ORIENT_FAIL")
    assert got[0] is not None and got == exp
    assert b"LG_ORIENT_FAIL" in lib.lg_orientation_note(s._h)
    a1, a2 = ok.estimate_leaf_orientation((labels == 1).astype(np.uint8)), s.estimate_leaf_orientation((labels == 1).astype(np.uint8))
    assert a1 == a2 and a1[0] is not None


def test_gaussian_larger_than_the_frame_ends_in_the_none_triple(L):
    """torch's reflect padding refuses size // 2 >= min(H, W) ("Padding size should be less than the corresponding input
    dimension"): the reference ends in its logged None triple (grasp_point_selector.py:635-657, :251-253)."""
    s = L.GraspPointSelector(torch.device("cuda:0"), load_model=False)
    s.set_camera_params(O.synthetic_scene(64, 64, 0)[2])
    H, W = 8, 3
    m = torch.ones((H, W), dtype=torch.bool).cuda()
    d = torch.full((H, W), 0.5).cuda()
    assert s.select_grasp_point(m, d, L.ImageProcessor(H, W, 21, 7)) == (None, None, None)


def test_uint8_mask_with_255_values(sel):
    H, W = 270, 360
    labels, depth, P = O.synthetic_scene(H, W, 2)
    sel.set_camera_params(P)
    m1 = (labels == 1).astype(np.uint8)
    a, va, _ = sel.score_maps(torch.from_numpy(m1).cuda(), torch.from_numpy(depth).cuda())
    b, vb, _ = sel.score_maps(torch.from_numpy(m1 * 255).cuda(), torch.from_numpy(depth).cuda())
    for k in a:
        np.testing.assert_array_equal(a[k].cpu().numpy(), b[k].cpu().numpy(), err_msg=k)
    np.testing.assert_array_equal(va.cpu().numpy(), vb.cpu().numpy())


def test_optical_centre_pixel_precision(sel):
    """Regression (found by tools/stress_parity.py): x - cx cancels in float32 for the pixel next to the optical
    centre (r = 0.08 px here) unless the integer part is subtracted exactly first."""
    labels, depth, P = O.synthetic_scene(284, 537, 1072)
    _compare_maps(sel, (labels == 1).astype(np.uint8), depth, P)


def test_randomised_sweep_small(sel):
    """A slice of tools/stress_parity.py: odd sizes, multi-component masks, specks, border contact."""
    rng = np.random.default_rng(7)
    for case in range(16):
        H, W = int(rng.integers(40, 300)), int(rng.integers(40, 400))
        if case % 5 == 0:
            W = W // 4 * 4 + int(rng.integers(1, 4))
        labels, depth, P = O.synthetic_scene(H, W, 500 + case)
        kind = case % 4
        mask = [(labels == 1), (labels >= 1), (labels == 1) | (rng.random((H, W)) > 0.995),
                (labels == 2) | (labels == 3)][kind].astype(np.uint8)
        _compare_maps(sel, mask, depth, P)


@pytest.mark.parametrize("env", [{"LG_FINAL_PERSIST": "8"}, {"LG_FINAL_PERSIST": "3"}, {"LG_FINAL_TPW": "4"}, {"LG_SIDE_TAIL": "0"},
                                 {"LG_SIDE_TAIL": "2"}, {"LG_HOST_ORIENT": "1"}])
def test_launch_forms_of_the_plane_kernel_give_the_same_planes(env):
    """The plane kernel's other launch forms (resident workgroups walking the tiles, several consecutive tiles per workgroup) and
    the other placements of the side work (border maxima + stem bits behind the sweeps / on a third stream, orientation on the
    host) are A/B switches read once per process: each runs the planes-vs-oracle and end-to-end tests of this file in a process
    of its own (odd image sizes, border tiles, empty masks, 1080p with the CNN)."""
    import os
    import subprocess
    import sys

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sel = ("test_score_maps_vs_oracle or test_score_maps_edge_cases or test_batch_equals_single or test_full_size_1080p_all_planes "
           "or test_select_grasp_point_1080p_with_cnn_vs_oracle or test_candidates_vs_oracle_bit_exact")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(repo, "tests", "test_gpu_parity.py"), "-m", "gpu", "-x", "-q",
                        "-k", sel], env=dict(os.environ, **env), cwd=repo, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout
