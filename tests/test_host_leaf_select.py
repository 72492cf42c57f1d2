"""The native host half of the leaf selection (lg_leaf_select_from_stats, the arithmetic inside lg_leaf_select_batch) against the
Python mirror of leaf_scorer.py:53-203 (OptimalLeafSelector._select_from_statistics) on random statistics rows: no device is
involved -- the entry point takes rows, not images.  Random rows exercise what real frames rarely do: ties in the scores, empty
tall / regular sets, Pareto sets of one, candidates on the border, float32 medians one ulp around their mean."""
import ctypes as C

import numpy as np
import pytest

from leafgrasp_amd._lib import LgLeafStat, lib
from leafgrasp_amd.leaf_scorer import OptimalLeafSelector


def _selector(cx, cy, f):
    ols = object.__new__(OptimalLeafSelector)          # no lg_create: the statistics -> id half needs no handle
    ols.camera_cx, ols.camera_cy, ols.f_norm = cx, cy, f
    ols._tall_leaves = []
    return ols


def _random_frame(rng, H, W):
    n = int(rng.integers(0, 40))
    ids = np.sort(rng.choice(np.arange(1, 3000), size=n, replace=False))
    rows = []
    for i in ids:
        area = int(rng.choice([rng.integers(1, 10000), rng.integers(10000, 200000)]))
        cx_, cy_ = rng.uniform(0, W - 1), rng.uniform(0, H - 1)
        if rng.random() < 0.2:                             # exact ties between candidates
            cx_, cy_ = float(W // 3), float(H // 3)
        md = np.float32(rng.uniform(0.2, 1.0)) if rng.random() < 0.8 else np.float32(0.5)
        mean_depth = float(md) * rng.uniform(0.9, 1.1)
        rows.append(dict(id=int(i), area=area, touches_border=bool(rng.random() < 0.3),
                         sum_x=float(np.round(cx_ * area)), sum_y=float(np.round(cy_ * area)),
                         sum_depth=mean_depth * area, sum_ray=rng.uniform(1.0, 1.3) * area, median_depth=md))
    ext = (int(rng.integers(0, H)), int(rng.integers(0, W))), (int(rng.integers(0, H)), int(rng.integers(0, W)))
    return rows, ext


def _native(rows, ext, H, W, cx, cy, f):
    arr = (LgLeafStat * max(1, len(rows)))()
    for k, r in enumerate(rows):
        arr[k].id, arr[k].area, arr[k].touches_border = r["id"], r["area"], int(r["touches_border"])
        arr[k].sum_x, arr[k].sum_y, arr[k].sum_depth, arr[k].sum_ray = r["sum_x"], r["sum_y"], r["sum_depth"], r["sum_ray"]
        arr[k].median_depth = float(r["median_depth"])
    e = (C.c_int32 * 4)(ext[0][0], ext[0][1], ext[1][0], ext[1][1])
    lid, nt = C.c_int32(0), C.c_int32(0)
    tall = (C.c_int32 * 64)()
    assert lib.lg_leaf_select_from_stats(arr, len(rows), e, H, W, cx, cy, f, C.byref(lid), tall, 64, C.byref(nt)) == 0
    return lid.value, [tall[k] for k in range(min(nt.value, 64))]


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_native_selection_equals_python_mirror(seed):
    rng = np.random.default_rng(seed)
    H, W = 1080, 1920
    cx, cy, f = 707.87, 494.07, 1750.68
    ols = _selector(cx, cy, f)
    n_pick = 0
    for _ in range(700):
        rows, ext = _random_frame(rng, H, W)
        want = ols._select_from_statistics([dict(r) for r in rows], ext, (H, W))
        got, tall = _native(rows, ext, H, W, cx, cy, f)
        assert got == (want if want is not None else -1), (rows, ext)
        if want is not None:
            n_pick += 1
            assert tall == ols.get_tall_leaves()
    assert n_pick > 300


def test_native_selection_edge_cases():
    H, W = 540, 720
    ols = _selector(360.0, 270.0, 600.0)
    assert _native([], ((0, 0), (0, 0)), H, W, 360.0, 270.0, 600.0)[0] == -1
    # the labels cover the whole frame: torch.unique(mask)[1:] drops the first one (leaf_scorer.py:32)
    rows = [dict(id=1, area=H * W - 20000, touches_border=True, sum_x=1e7, sum_y=1e7, sum_depth=0.5 * (H * W - 20000),
                 sum_ray=1.1 * (H * W - 20000), median_depth=np.float32(0.5)),
            dict(id=2, area=20000, touches_border=False, sum_x=360.0 * 20000, sum_y=270.0 * 20000, sum_depth=0.4 * 20000,
                 sum_ray=1.05 * 20000, median_depth=np.float32(0.4))]
    ext = ((10, 10), (500, 700))
    assert _native(rows, ext, H, W, 360.0, 270.0, 600.0)[0] == ols._select_from_statistics([dict(r) for r in rows], ext, (H, W)) == 2
    # 128 leaves: handed back (-2), the Python path decides
    many = [dict(id=k + 1, area=10, touches_border=False, sum_x=10.0, sum_y=10.0, sum_depth=5.0, sum_ray=11.0,
                 median_depth=np.float32(0.5)) for k in range(128)]
    assert _native(many, ext, H, W, 360.0, 270.0, 600.0)[0] == -2
