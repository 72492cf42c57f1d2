"""How far can the build's clutter-field semantics (exact Euclidean distance) move what the reference
consumes from skfmm.distance (leaf_scorer.py:67-71: only argmin / argmax)?  scikit-fmm is not installed, so
the comparison is against a RESTATED second-order fast-marching solver (oracle/lg_oracle.c::lg_fmm_distance).
Both sides are "parity unpinned"; this test documents the deviation on the synthetic scenes."""
import numpy as np
import pytest
from scipy import ndimage

from oracle import lg_oracle as O


def test_fmm_restatement_is_a_distance_field():
    m = np.zeros((41, 41), bool)
    m[20, 20] = True
    f = O.fmm_distance(m)
    yy, xx = np.mgrid[0:41, 0:41]
    e = np.hypot(yy - 20, xx - 20)
    assert np.all(f[20, :] == np.abs(np.arange(41) - 20))          # exact along the grid axes
    assert np.all(f >= e - 1e-9)                                    # upwind schemes over-estimate
    assert np.max(np.abs(f - e)[e > 10] / e[e > 10]) < 0.03         # second order: < 3 % in the far field
    assert O.fmm_distance(np.zeros((8, 8), bool)).max() == 0.0


@pytest.mark.parametrize("shape,seed", [((270, 360), 0), ((270, 360), 1), ((360, 480), 2), ((720, 1280), 2),
                                        ((720, 1280), 5), ((1080, 1440), 3)])
def test_edt_vs_fmm_extrema_and_selection(shape, seed):
    H, W = shape
    labels, depth, P = O.synthetic_scene(H, W, seed)
    mn_e, mx_e = O.clutter_extrema(labels, "edt")
    mn_f, mx_f = O.clutter_extrema(labels, "fmm")
    assert tuple(mn_e) == tuple(mn_f)                               # first leaf pixel either way
    dev = np.hypot(mx_e[0] - mx_f[0], mx_e[1] - mx_f[1])
    leaf = labels >= 1
    e = ndimage.distance_transform_edt(~leaf)
    # the FMM arg-max is (near-)optimal for the exact field too: its exact distance is within 1 % of the maximum
    assert e[tuple(mx_f)] >= 0.99 * e.max()
    assert dev <= 0.02 * np.hypot(H, W), dev
    a = O.RefOptimalLeafSelector("edt")
    b = O.RefOptimalLeafSelector("fmm")
    a.set_camera_params(P)
    b.set_camera_params(P)
    assert a.select_optimal_leaf(labels, depth) == b.select_optimal_leaf(labels, depth)
