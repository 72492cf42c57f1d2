"""Self-consistency of the C oracle's cv2 stand-ins (oracle/lg_oracle.c).  These rows are "parity
unpinned vs real OpenCV" (absent from the image); the checks below pin the restatement against
independent brute-force definitions and also establish the identities the HIP kernels rely on
(closed-form chamfer norm, no-source closed form)."""
import heapq
import math

import numpy as np
import pytest

from oracle import lg_oracle as O

A5, B5, C5 = 65536, 91750, 143976  # 16.16 fixed point of 1, 1.4, 2.1969
A3, B3 = 62587, 89738  # 0.955, 1.3693
INIT = (2 ** 31 - 1) >> 2


def _norm5(dx, dy):
    a, b = max(abs(dx), abs(dy)), min(abs(dx), abs(dy))
    return (a - 2 * b) * A5 + b * C5 if 2 * b <= a else (a - b) * C5 + (2 * b - a) * B5


def _norm3(dx, dy):
    a, b = max(abs(dx), abs(dy)), min(abs(dx), abs(dy))
    return (a - b) * A3 + b * B3


def _dijkstra(src, steps):
    H, W = src.shape
    dist = np.full((H, W), np.iinfo(np.int64).max, np.int64)
    pq = []
    for y, x in zip(*np.where(src == 0)):
        dist[y, x] = 0
        pq.append((0, y, x))
    heapq.heapify(pq)
    while pq:
        d, y, x = heapq.heappop(pq)
        if d > dist[y, x]:
            continue
        for dy, dx, w in steps:
            yy, xx = y + dy, x + dx
            if 0 <= yy < H and 0 <= xx < W and d + w < dist[yy, xx]:
                dist[yy, xx] = d + w
                heapq.heappush(pq, (d + w, yy, xx))
    return dist


STEPS5 = [(dy, dx, A5) for dy, dx in ((0, 1), (0, -1), (1, 0), (-1, 0))] + \
         [(dy, dx, B5) for dy in (-1, 1) for dx in (-1, 1)] + \
         [(dy, dx, C5) for dy, dx in ((1, 2), (1, -2), (-1, 2), (-1, -2), (2, 1), (2, -1), (-2, 1), (-2, -1))]
STEPS3 = [(dy, dx, A3) for dy, dx in ((0, 1), (0, -1), (1, 0), (-1, 0))] + \
         [(dy, dx, B3) for dy in (-1, 1) for dx in (-1, 1)]


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_chamfer_equals_graph_shortest_path_and_norm(seed):
    rng = np.random.default_rng(seed)
    H, W = 37, 53
    src = (rng.random((H, W)) > (0.02 if seed % 2 else 0.3)).astype(np.uint8)
    src[rng.integers(H), rng.integers(W)] = 0
    for ms, steps, norm in ((5, STEPS5, _norm5), (3, STEPS3, _norm3)):
        f32, fix = O.distance_transform(src, ms, return_fix=True)
        np.testing.assert_array_equal(fix.astype(np.int64), _dijkstra(src, steps))
        zs = np.argwhere(src == 0)
        for (y, x) in [(0, 0), (H - 1, W - 1), (H // 2, W // 3), (5, W - 2)]:
            assert fix[y, x] == min(norm(x - zx, y - zy) for zy, zx in zs)
        np.testing.assert_array_equal(f32, fix.astype(np.float32) * np.float32(1 / 65536))


@pytest.mark.parametrize("shape", [(20, 30), (33, 17), (64, 64)])
def test_chamfer_no_zero_pixel_closed_form(shape):
    # the degenerate isolation map: image with no zero pixel -> INIT + d*HV
    H, W = shape
    yy, xx = np.mgrid[0:H, 0:W]
    d = np.minimum(np.minimum(xx + 1, W - xx), np.minimum(yy + 1, H - yy))
    for ms, a in ((3, A3), (5, A5)):
        _, fix = O.distance_transform(np.ones((H, W), np.uint8), ms, return_fix=True)
        np.testing.assert_array_equal(fix.astype(np.int64), INIT + d * a)
    f3 = O.distance_transform(np.ones((H, W), np.uint8), 3)
    assert f3.max() == np.float32(INIT + math.ceil(min(H, W) / 2) * A3) * np.float32(1 / 65536)


def test_chamfer_tracks_euclid():
    m = np.ones((61, 61), np.uint8)
    m[30, 30] = 0
    d = O.distance_transform(m, 5)
    yy, xx = np.mgrid[0:61, 0:61]
    e = np.hypot(yy - 30, xx - 30)
    assert np.all(d[30, :] == np.abs(np.arange(61) - 30))  # exact on the axes
    assert np.max(np.abs(d - e) / np.maximum(e, 1)) < 0.03


def test_ellipse_se_and_dilate_bruteforce():
    se30, se31, se40 = O.ellipse_se(30), O.ellipse_se(31), O.ellipse_se(40)
    assert se30.shape == (30, 30) and se30[0].sum() == 1 and se30[15].sum() == 30 and se30[0, 15] == 1
    assert np.array_equal(se31, se31[::-1]) and np.array_equal(se31, se31[:, ::-1]) and se31[15].sum() == 31
    assert se40[20].sum() == 40
    rng = np.random.default_rng(0)
    src = (rng.random((50, 70)) > 0.97).astype(np.uint8)
    for se in (se30, se31):
        k = se.shape[0]
        a = k // 2
        exp = np.zeros_like(src)
        for y in range(50):
            for x in range(70):
                hit = 0
                for i, j in zip(*np.where(se)):
                    yy, xx = y + i - a, x + j - a
                    if 0 <= yy < 50 and 0 <= xx < 70 and src[yy, xx]:
                        hit = 1
                        break
                exp[y, x] = hit
        np.testing.assert_array_equal(O.dilate(src, se), exp)


def test_orientation_rotated_rectangle_and_largest_component():
    H, W = 200, 260
    yy, xx = np.mgrid[0:H, 0:W]
    for ang in (20.0, 75.0, 110.0, 160.0):
        t = np.deg2rad(ang)
        u = (xx - 130) * np.cos(t) + (yy - 100) * np.sin(t)
        v = -(xx - 130) * np.sin(t) + (yy - 100) * np.cos(t)
        m = ((np.abs(u) <= 80) & (np.abs(v) <= 25)).astype(np.uint8)
        m[5:9, 5:9] = 1  # a small second component must not win
        r = O.leaf_orientation_raw(m)
        assert abs(((r[0] - t + np.pi / 2) % np.pi) - np.pi / 2) < 0.02
        assert abs(r[1] - 160) < 4 and abs(r[2] - 50) < 4
        assert abs(r[3] - 130) < 1.5 and abs(r[4] - 100) < 1.5
    assert O.leaf_orientation_raw(np.zeros((10, 10), np.uint8)) is None
    one = np.zeros((10, 10), np.uint8)
    one[4, 6] = 1
    r = O.leaf_orientation_raw(one)
    assert r[1] == 0 and r[2] == 0 and (r[3], r[4]) == (6, 4)


def test_contour_area_is_pixel_centre_polygon():
    m = np.zeros((40, 50), np.uint8)
    m[10:30, 12:40] = 1
    m[15:20, 20:30] = 0  # a hole does not change the outer contour
    r = O.leaf_orientation_raw(m)
    assert r[5] == 27 * 19 and r[6] == 2 * (27 + 19)


def test_pareto_restatement():
    s = np.array([[1, 1, 1], [2, 0, 0], [1, 1, 1], [0, 0, 0], [0, 3, 0]], float)
    assert O.pareto_max(s).tolist() == [True, True, False, False, True]
