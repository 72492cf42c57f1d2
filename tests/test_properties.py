"""Property tests (hypothesis) for identities the HIP kernels rely on, checked on the CPU oracle:
  * greedy walk over the full descending argsort == 'repeat {arg-max over not-yet-suppressed pixels; suppress the
    Chebyshev ball of radius 2*min_distance}'  (the formulation lg_topk_kernel implements);
  * order-preserving float -> uint32 key used for the (score desc, index desc) total order;
  * radix-select bookkeeping = np.median for odd / even counts with duplicates."""
import numpy as np
from hypothesis import given, settings, strategies as st

from oracle import lg_oracle as O


def _iterative_masked_argmax(vs, top_k, md):
    H, W = vs.shape
    alive = np.ones((H, W), bool)
    flat = (vs + 0.0).ravel()
    out = []
    yy, xx = np.mgrid[0:H, 0:W]
    for _ in range(top_k):
        if not alive.any():
            break
        cand = np.where(alive.ravel())[0]
        best = cand[np.lexsort((-cand, -flat[cand]))[0]]          # score desc, flat index desc
        y, x = divmod(int(best), W)
        out.append((x, y))
        alive &= ~((np.abs(yy - y) <= 2 * md) & (np.abs(xx - x) <= 2 * md))
    return out


@settings(max_examples=40, deadline=None)
@given(st.integers(8, 40), st.integers(8, 48), st.integers(1, 12), st.integers(0, 6), st.integers(0, 2 ** 31 - 1),
       st.sampled_from(["smooth", "ties", "sparse", "negative"]))
def test_greedy_nms_equals_iterative_masked_argmax(H, W, k, md, seed, kind):
    rng = np.random.default_rng(seed)
    sm = rng.random((H, W)).astype(np.float32)
    valid = np.ones((H, W), bool)
    if kind == "ties":
        sm = np.round(sm, 1)
    elif kind == "sparse":
        valid = rng.random((H, W)) > 0.95
    elif kind == "negative":
        sm = sm - np.float32(0.7)
        valid = rng.random((H, W)) > 0.3
    ref = O.RefGraspPointSelector()._get_candidate_points(sm, valid, k, md)
    assert ref == _iterative_masked_argmax(np.asarray(sm * valid), k, md)
    for i, (x, y) in enumerate(ref):           # spacing invariant of the reference's used-window test
        for (x2, y2) in ref[:i]:
            assert max(abs(x - x2), abs(y - y2)) > 2 * md


def _orderable(f32):
    b = np.asarray(f32, np.float32).view(np.uint32).astype(np.uint64)
    return np.where(b & 0x80000000, (~b) & 0xFFFFFFFF, b | 0x80000000)


@settings(max_examples=50, deadline=None)
@given(st.lists(st.floats(-1e6, 1e6, width=32, allow_nan=False), min_size=2, max_size=50))
def test_orderable_key_is_monotone(vals):
    v = np.array(vals, np.float32) + np.float32(0.0)     # -0.0 -> +0.0 as the kernels do
    k = _orderable(v)
    order_v = np.argsort(v, kind="stable")
    assert np.all(np.diff(k[order_v].astype(np.int64)) >= 0)
    assert np.all((np.diff(v[order_v]) == 0) == (np.diff(k[order_v].astype(np.int64)) == 0))


@settings(max_examples=50, deadline=None)
@given(st.integers(1, 200), st.integers(0, 2 ** 31 - 1), st.booleans())
def test_radix_select_bookkeeping_matches_np_median(n, seed, dup):
    rng = np.random.default_rng(seed)
    d = rng.random(n).astype(np.float32)
    if dup:
        d = np.round(d, 1)
    keys = np.sort(_orderable(d))
    lo_rank = (n - 1) // 2
    prefix, rank = 0, lo_rank
    cur = keys
    for p in (3, 2, 1, 0):                                  # 4 x 8-bit digits, most significant first
        digit = (cur >> np.uint64(8 * p)) & np.uint64(0xFF)
        hist = np.bincount(digit.astype(np.int64), minlength=256)
        acc = np.cumsum(hist)
        b = int(np.searchsorted(acc, rank, side="right"))
        rank -= int(acc[b - 1]) if b else 0
        cur = cur[digit == b]
        prefix |= b << (8 * p)
    above = len(cur) - rank - 1                             # equal keys ranked above the selected one
    key_lo = np.uint64(prefix)
    bigger = keys[keys > key_lo]
    key_hi = key_lo if (above > 0 or n % 2 == 1) else bigger.min()

    def k2f(k):
        k = np.uint32(k)
        b = (k & np.uint32(0x7FFFFFFF)) if (k & np.uint32(0x80000000)) else ~k
        return np.array([b], np.uint32).view(np.float32)[0]

    lo, hi = k2f(key_lo), k2f(key_hi)
    got = lo if n % 2 == 1 else np.float32(np.float32(lo + hi) / np.float32(2.0))
    assert got == np.median(d)
