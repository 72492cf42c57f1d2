"""The identities the device's sweep-free distance transform rests on (lg_hrun_kernel / lg_dtsearch_kernel / lg_dtanchor_kernel /
lg_dtlevel_kernel, csrc/lg_kernels.hip), checked on the CPU against the oracle's two-pass chamfer transform
(cv2.distanceTransform(mask, DIST_L2, 5), grasp_point_selector.py:266, :529):
  1. d(x, y) = min over rows y' of N(h[y'][x], |y - y'|) with h = distance to the row's nearest zero pixel and N the closed-form
     chamfer norm; rows outside [by0 - 1, by1 + 1] never matter;
  2. for two candidate rows r1 < r2, sign(N(h1, |y - r1|) - N(h2, |y - r2|)) never decreases with y, hence
  3. between two anchor rows with ANY minimisers a1, a2, every row has a minimiser in [min(a1, a2), max(a1, a2)].
NumPy emulation of the arithmetic only -- the kernels themselves are compared with the oracle in tests/test_gpu_parity.py."""
import numpy as np

from oracle import lg_oracle as O

A, B, C = 65536, 91750, 143976
CAP = 16383


def N(dx, dy):
    M, m = np.maximum(dx, dy), np.minimum(dx, dy)
    return np.where(2 * m <= M, (M - 2 * m) * A + m * C, (M - m) * C + (2 * m - M) * B)


def _cases(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for t in range(n):
        H, W = int(rng.integers(8, 72)), int(rng.integers(8, 72))
        yy, xx = np.mgrid[:H, :W]
        kind = t % 6
        if kind == 0:
            m = rng.random((H, W)) < rng.random()
        elif kind == 1:
            m = np.zeros((H, W), bool)
            y0, x0 = rng.integers(0, H - 3), rng.integers(0, W - 3)
            m[y0:y0 + rng.integers(2, H), x0:x0 + rng.integers(2, W)] = True
        elif kind == 2:
            th = rng.random() * np.pi
            u = (xx - W / 2) * np.cos(th) + (yy - H / 2) * np.sin(th)
            v = -(xx - W / 2) * np.sin(th) + (yy - H / 2) * np.cos(th)
            m = (u / (W / 2.1)) ** 2 + (v / (H / 5)) ** 2 < 1
        elif kind == 3:
            m = np.ones((H, W), bool)
            m[rng.integers(0, H), rng.integers(0, W)] = False
        elif kind == 4:
            m = np.ones((H, W), bool)
            m[rng.integers(0, H)] = False
        else:
            m = rng.random((H, W)) < 0.93
            m[:2] = True
        m = m.astype(np.uint8)
        if 0 < m.sum() < H * W:
            out.append(m)
    return out


def _runs(mask):
    H, W = mask.shape
    ys = np.nonzero(mask)[0]
    lo, hi = max(ys.min() - 1, 0), min(ys.max() + 1, H - 1)
    h = np.full((H, W), CAP, np.int64)
    for y in range(lo, hi + 1):
        z = np.nonzero(mask[y] == 0)[0]
        if z.size:
            h[y] = np.abs(np.arange(W)[:, None] - z[None, :]).min(1)
    return h, lo, hi, ys.min(), ys.max()


def test_chamfer_norm_is_the_max_of_its_four_linear_pieces():
    d = np.arange(0, 300)
    dx, dy = np.meshgrid(d, d)
    al, be, ga = C - 2 * A, C - B, 2 * B - C
    pieces = np.maximum(np.maximum(dx * A + dy * al, dx * al + dy * A), np.maximum(dx * be + dy * ga, dx * ga + dy * be))
    np.testing.assert_array_equal(pieces, N(dx, dy))


def test_row_search_equals_the_two_pass_transform():
    for mask in _cases(90, 0):
        ref = O.distance_transform(mask, 5, return_fix=True)[1].astype(np.int64) * (mask > 0)
        h, lo, hi, by0, by1 = _runs(mask)
        rows = np.arange(lo, hi + 1)
        got = np.zeros_like(ref)
        for y in range(by0, by1 + 1):
            got[y] = np.where(mask[y] > 0, N(h[lo:hi + 1], np.abs(rows - y)[:, None]).min(0), 0)
        np.testing.assert_array_equal(got, ref)


def test_candidate_rows_cross_once():
    ys = np.arange(-120, 200)
    for D in range(1, 40, 3):
        for h1 in range(0, 45, 2):
            f1 = N(np.abs(ys), h1)
            for h2 in range(0, 45, 2):
                s = np.sign(f1 - N(np.abs(ys - D), h2))
                assert not (np.diff(s) < 0).any(), (D, h1, h2)


def test_anchor_and_band_search_with_arbitrary_minimisers():
    rng = np.random.default_rng(3)
    for mask in _cases(60, 1):
        H, W = mask.shape
        ref = O.distance_transform(mask, 5, return_fix=True)[1].astype(np.int64) * (mask > 0)
        h, lo, hi, by0, by1 = _runs(mask)
        rows = np.arange(lo, hi + 1)
        wy0, wy1 = (by0 // 16) * 16, min(H, ((by1 + 16) // 16) * 16)
        got = np.zeros_like(ref)
        arg = {}
        for y in range(wy0, wy1, 8):
            cand = N(h[lo:hi + 1], np.abs(rows - y)[:, None])
            mn = cand.min(0)
            a = np.array([rows[rng.choice(np.nonzero(cand[:, x] == mn[x])[0])] for x in range(W)])   # any minimiser
            arg[y] = np.where(mask[y] > 0, a, y)
            got[y] = np.where(mask[y] > 0, mn, 0)
        for ya in range(wy0, wy1, 8):
            a1 = arg[ya]
            a2 = arg[ya + 8] if ya + 8 <= by1 else np.full(W, hi)
            for y in range(ya + 1, min(ya + 8, wy1)):
                for x in np.nonzero(mask[y])[0]:
                    r0, r1 = max(min(a1[x], a2[x]), lo), min(max(a1[x], a2[x]), hi)
                    got[y, x] = N(h[r0:r1 + 1, x], np.abs(np.arange(r0, r1 + 1) - y)).min()
        np.testing.assert_array_equal(got, ref)


def test_refinement_ladder_with_arbitrary_minimisers():
    """Identity 3 applied recursively: anchors every 32 rows, then the row half-way between two solved rows from the candidate
    rows between their minimisers, level by level -- any minimiser at every level.  (The device forms built on it in round 4 were
    exact and slower than anchors + bands, and were removed: profiles/NOTES_r04.md.)"""
    rng = np.random.default_rng(11)
    for mask in _cases(60, 2):
        H, W = mask.shape
        ref = O.distance_transform(mask, 5, return_fix=True)[1].astype(np.int64) * (mask > 0)
        h, lo, hi, by0, by1 = _runs(mask)
        rows = np.arange(lo, hi + 1)
        wy0, wy1 = (by0 // 16) * 16, min(H, ((by1 + 16) // 16) * 16)
        got = np.zeros_like(ref)
        arg = {}

        def solve(y, r0, r1):   # per column: minimum over candidate rows r0[x] .. r1[x], a random minimiser
            out, a = np.zeros(W, np.int64), np.full(W, y)
            for x in np.nonzero(mask[y])[0]:
                rr = np.arange(r0[x], r1[x] + 1)
                c = N(h[rr, x], np.abs(rr - y))
                out[x] = c.min()
                a[x] = rr[rng.choice(np.nonzero(c == c.min())[0])]
            return out, a
        for y in range(wy0, wy1, 32):
            got[y], arg[y] = solve(y, np.full(W, lo), np.full(W, hi))
        for S in (16, 8, 4, 2, 1):
            for y in range(wy0 + S, wy1, 2 * S):
                a1 = arg[y - S]
                a2 = arg[y + S] if y + S <= by1 else np.full(W, hi)
                got[y], arg[y] = solve(y, np.maximum(np.minimum(a1, a2), lo), np.minimum(np.maximum(a1, a2), hi))
        np.testing.assert_array_equal(got, ref)
