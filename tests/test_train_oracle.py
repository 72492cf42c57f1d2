"""CPU tests of the training row: the oracle's restated optimisation step against the fixture produced by the
reference module + torch.optim.Adam (tests/golden/make_golden_train.py), and the host-side layout / helper logic."""
import os

import numpy as np
import pytest
import torch

import synthetic_inputs as S
from oracle import lg_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = (("spatial", (64, 128, 256), 8), ("none", (32, 64, 128), 6), ("hybrid", (64, 128, 256), 6), ("channel", (32, 64, 128), 8))


def assert_close_robust(a, b, rtol, atol, max_bad=0.03, l2=0.05, what=""):
    """Elementwise agreement except for a small fraction of elements, plus a bound on the relative L2 error.  A training
    step is not a continuous function of its inputs: an activation within rounding of a ReLU / max-pool decision flips
    in one implementation and not the other (fp32 vs fp64 torch differ the same way -- see the float64 test below), and a
    flip moves the whole gradient of its BatchNorm channel.  Such channels are rare; a wrong kernel fails both bounds."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    bad = np.abs(a - b) > atol + rtol * np.abs(b)
    assert bad.mean() <= max_bad, (what, float(bad.mean()), float(np.abs(a - b).max()))
    den = np.linalg.norm(b)
    assert np.linalg.norm(a - b) <= l2 * den + atol * np.sqrt(a.size), (what, float(np.linalg.norm(a - b)), float(den))


def noisy_bias(k):
    """Biases in front of a BatchNorm (encoder convs .0 / .3, classifier Linears .0 / .4 / .8): mathematically zero gradient."""
    return k.endswith(".bias") and "attention" not in k and k.split(".")[-2] in ("0", "3", "4", "8")


@pytest.fixture(scope="module")
def tv():
    return np.load(os.path.join(HERE, "golden", "train_vectors.npz"))


def case_inputs(att, filt, n):
    params = S.cnn_closed_form_params(seed=3, attention_type=att, filters=filt)
    x = S.synthetic_patches(n, seed=11)
    y = np.array([(i * 5 + 1) % 3 == 0 for i in range(n)], np.float32)
    return params, x, y


@pytest.mark.parametrize("att,filt,n", CASES)
def test_oracle_train_step_matches_reference_fixture(tv, att, filt, n):
    tag = f"{att}_{len(filt)}x{filt[0]}"
    params, x, y = case_inputs(att, filt, n)
    np.testing.assert_array_equal(y, tv[f"{tag}_labels"])
    names = [str(k) for k in tv[f"{tag}_names"]]
    pick = tv[f"{tag}_pick"]
    st = None
    for step in range(2):
        r = O.cnn_train_step(params, x, y, masks=None, opt_state=st)
        tol = 2e-6 if step == 0 else 5e-5   # the second step sees parameters that went through one Adam update
        assert r["loss"] == pytest.approx(float(tv[f"{tag}_loss{step}"]), rel=tol)
        np.testing.assert_allclose(r["logits"], tv[f"{tag}_logits{step}"], rtol=10 * tol, atol=10 * tol)
        assert r["grad_norm"] == pytest.approx(float(tv[f"{tag}_gnorm{step}"]), rel=20 * tol)
        gt = np.array([np.linalg.norm(r["grads"][k].astype(np.float64)) for k in names])
        assert_close_robust(gt, tv[f"{tag}_gtnorm{step}"], 1e-4 if step == 0 else 5e-3, 1e-5, 0.0 if step == 0 else 0.1, what="gtnorm")
        gs = np.stack([r["grads"][k].reshape(-1)[pick[i]] for i, k in enumerate(names)])
        assert_close_robust(gs, tv[f"{tag}_gsample{step}"], 1e-4 if step == 0 else 1e-2, 2e-5 if step == 0 else 2e-3,
                            0.0 if step == 0 else 0.03, what="gsample")
        params, st = r["params"], r["opt_state"]
    buf = np.concatenate([params[k].reshape(-1) for k in params if "running_" in k])
    # running means carry 0.1 * (conv / linear bias), and those biases take noise-sign Adam steps (see below)
    np.testing.assert_allclose(buf, tv[f"{tag}_buffers"], rtol=1e-4, atol=2e-4)
    ms = np.stack([st["exp_avg"][k].reshape(-1)[pick[i]] for i, k in enumerate(names)])
    vs = np.stack([st["exp_avg_sq"][k].reshape(-1)[pick[i]] for i, k in enumerate(names)])
    ps = np.stack([params[k].reshape(-1)[pick[i]] for i, k in enumerate(names)])
    assert_close_robust(ms, tv[f"{tag}_msample"], 5e-3, 2e-6, what="exp_avg")
    assert_close_robust(vs, tv[f"{tag}_vsample"], 1e-2, 1e-10, what="exp_avg_sq")
    # parameters whose gradient is rounding noise (biases in front of a BatchNorm) take Adam steps of noise sign:
    # up to 2 * lr per step apart; everything else agrees tightly
    noisy = np.array([noisy_bias(k) for k in names])
    assert_close_robust(ps[~noisy], tv[f"{tag}_psample"][~noisy], 1e-4, 2e-5, 0.01, 1e-3, what="params")
    np.testing.assert_allclose(ps[noisy], tv[f"{tag}_psample"][noisy], atol=2 * 2 * 0.0005 + 1e-6)


def test_oracle_dropout_masks_scale_and_zero():
    """keep masks multiply the activations: an all-ones mask equals no mask; a zeroed classifier mask kills the logit's
    dependence on the input (the logit becomes classifier.12.bias)."""
    params, x, y = case_inputs("spatial", (64, 128, 256), 4)
    widths = [64, 128, 256, 256, 128, 64]
    ones = [np.ones((4, w), np.float32) for w in widths]
    a = O.cnn_train_step(params, x, y, masks=None, apply_update=False)
    b = O.cnn_train_step(params, x, y, masks=ones, apply_update=False)
    np.testing.assert_array_equal(a["logits"], b["logits"])
    ones[-1] = np.zeros((4, 64), np.float32)
    c = O.cnn_train_step(params, x, y, masks=ones, apply_update=False)
    np.testing.assert_allclose(c["logits"], float(params["classifier.12.bias"][0]), rtol=0, atol=1e-7)
    assert np.all(c["grads"]["encoder.0.0.weight"] == 0)


def test_parameter_layout_is_the_reference_modules_order(tv):
    from leafgrasp_amd.trainer import dropout_layout, parameter_layout
    for att, filt, _ in CASES:
        tag = f"{att}_{len(filt)}x{filt[0]}"
        assert [k for k, _ in parameter_layout(filt, att)[0]] == [str(k) for k in tv[f"{tag}_names"]]
        n_buf = sum(int(np.prod(s)) for _, s in parameter_layout(filt, att)[1])
        assert n_buf == tv[f"{tag}_buffers"].size
    assert dropout_layout((64, 128, 256)) == [(64, 0.3), (128, 0.3), (256, 0.3), (256, 0.5), (128, 0.5), (64, 0.4)]


def test_normalize_and_analyze_helpers():
    from leafgrasp_amd.trainer import analyze_predictions, normalize_data
    rng = np.random.default_rng(0)
    d = torch.from_numpy(rng.random((10, 1, 32, 32)).astype(np.float32))
    s = torch.from_numpy(rng.random((10, 7, 32, 32)).astype(np.float32) * np.arange(1, 8, dtype=np.float32)[None, :, None, None])
    out = normalize_data(d, s)
    assert abs(float(out["depth_patches"].mean())) < 1e-5 and float(out["depth_patches"].std()) == pytest.approx(1.0, rel=1e-5)
    np.testing.assert_allclose(out["score_patches"].mean(dim=(0, 2, 3)).numpy(), 0, atol=1e-5)
    np.testing.assert_allclose(out["score_patches"].std(dim=(0, 2, 3)).numpy(), 1, rtol=1e-4)
    assert tuple(out["stats"]["score_mean"].shape) == (1, 7, 1, 1)
    m = analyze_predictions(torch.tensor([[2.0], [0.7], [0.2], [-1.0]]), torch.tensor([1.0, 0.0, 1.0, 0.0]))
    assert m["confusion_matrix"] == {"true_positive": 1, "false_positive": 1, "false_negative": 1, "true_negative": 1}
    assert m["precision"] == 50.0 and m["recall"] == 50.0 and m["f1_score"] == 50.0


def test_trainer_has_no_cpu_fallback():
    from leafgrasp_amd.trainer import GraspTrainer
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        GraspTrainer(torch.device("cpu"))


def test_prepare_features_follows_the_training_script():
    """train_model.py:167-180: unsqueeze, normalize_data, cat([depth, mask, scores])."""
    from leafgrasp_amd.trainer import normalize_data, prepare_features
    rng = np.random.default_rng(1)
    data = {"depth_patches": torch.from_numpy(rng.random((6, 32, 32)).astype(np.float32)),
            "mask_patches": torch.from_numpy((rng.random((6, 32, 32)) > 0.5).astype(np.float32)),
            "score_patches": torch.from_numpy(rng.random((6, 7, 32, 32)).astype(np.float32)),
            "labels": torch.tensor([1, 0, 1, 0, 0, 1])}
    f, y, stats = prepare_features(data)
    assert tuple(f.shape) == (6, 9, 32, 32) and y.dtype == torch.float32
    n = normalize_data(data["depth_patches"].unsqueeze(1), data["score_patches"])
    torch.testing.assert_close(f[:, 0:1], n["depth_patches"])
    torch.testing.assert_close(f[:, 1], data["mask_patches"])
    torch.testing.assert_close(f[:, 2:], n["score_patches"])
    assert set(stats) == {"depth_mean", "depth_std", "score_mean", "score_std"}


def test_host_helpers_match_the_reference_training_script():
    """EarlyStopping (train_model.py:11-39), normalize_data (:41-62) and analyze_predictions (:64-99) against vectors produced
    by the reference's own functions (tests/golden/make_golden_train.py)."""
    from leafgrasp_amd.trainer import EarlyStopping, analyze_predictions, normalize_data
    hv = np.load(os.path.join(HERE, "golden", "train_host_vectors.npz"))

    class Model:
        def __init__(self):
            self.w = torch.zeros(1)

        def state_dict(self):
            return {"w": self.w.clone()}

        def load_state_dict(self, sd):
            self.w = sd["w"].clone()
    for i, seq in enumerate(hv["es_seqs"]):
        es = EarlyStopping(patience=15 if i % 2 == 0 else 5, min_delta=0.001, restore_best_weights=True)
        m, stop = Model(), -1
        for epoch, v in enumerate(seq):
            m.w = torch.tensor([float(epoch)])
            if es.step(float(v), epoch, m):
                stop = epoch
                break
        assert stop == int(hv["es_stop"][i]) and es.best_epoch == int(hv["es_best"][i][0]) and es.best_loss == hv["es_best"][i][1]
        assert float(m.w[0]) == hv["es_restored"][i]
    assert (hv["es_stop"] >= 0).sum() >= 5 and (hv["es_stop"] < 0).sum() >= 3     # both outcomes are in the fixture
    nd = normalize_data(torch.from_numpy(hv["nd_depth_in"]), torch.from_numpy(hv["nd_score_in"]))
    np.testing.assert_array_equal(nd["depth_patches"].numpy(), hv["nd_depth"])
    np.testing.assert_array_equal(nd["score_patches"].numpy(), hv["nd_score"])
    st = nd["stats"]
    np.testing.assert_array_equal(np.concatenate([st["depth_mean"].reshape(-1).numpy(), st["depth_std"].reshape(-1).numpy(),
                                                  st["score_mean"].reshape(-1).numpy(), st["score_std"].reshape(-1).numpy()]),
                                  hv["nd_stats"])
    ap = analyze_predictions(torch.from_numpy(hv["ap_outputs"]), torch.from_numpy(hv["ap_labels"]))
    got = [ap["positive_accuracy"], ap["negative_accuracy"], ap["precision"], ap["recall"], ap["f1_score"],
           ap["confusion_matrix"]["true_positive"], ap["confusion_matrix"]["false_positive"],
           ap["confusion_matrix"]["false_negative"], ap["confusion_matrix"]["true_negative"]]
    np.testing.assert_array_equal(np.array(got, np.float64), hv["ap_metrics"])


def test_plateau_scheduler_equals_torch():
    """PlateauScheduler against torch.optim.lr_scheduler.ReduceLROnPlateau(mode='min', factor=0.5, patience=5, min_lr=1e-6)
    (train_model.py:223-229) on seeded loss sequences."""
    from leafgrasp_amd.trainer import PlateauScheduler

    class T:
        lr = 0.0005
    rng = np.random.default_rng(4)
    reduced = 0
    for trial in range(8):
        seq = 1.0 / (1.0 + 0.2 * np.arange(80)) + 0.2 + rng.normal(0, 0.003 * (1 + trial % 3), 80)
        if trial % 2:
            seq[15:] = seq[15]
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.Adam([p], lr=0.0005)
        ref = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, mode="min", factor=0.5, patience=5, min_lr=1e-6)
        t = T()
        t.lr = 0.0005
        mine = PlateauScheduler(t, factor=0.5, patience=5, min_lr=1e-6)
        for v in seq:
            ref.step(float(v))
            mine.step(float(v))
            assert t.lr == opt.param_groups[0]["lr"]
        reduced += t.lr < 0.0005
    assert reduced >= 4      # the plateau sequences do reduce the rate


def test_plateau_scheduler_property():
    """Hypothesis: any loss sequence, any factor / patience -- PlateauScheduler tracks torch's ReduceLROnPlateau exactly."""
    from hypothesis import given, settings, strategies as st
    from leafgrasp_amd.trainer import PlateauScheduler

    class T:
        lr = 0.0

    @settings(max_examples=60, deadline=None)
    @given(st.lists(st.floats(min_value=1e-3, max_value=10.0, allow_nan=False), min_size=1, max_size=60),
           st.sampled_from([0.1, 0.5, 0.9]), st.integers(min_value=0, max_value=6), st.sampled_from([1e-6, 1e-4, 1e-3]))
    def check(seq, factor, patience, min_lr):
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.Adam([p], lr=0.0005)
        ref = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, mode="min", factor=factor, patience=patience, min_lr=min_lr)
        t = T()
        t.lr = 0.0005
        mine = PlateauScheduler(t, factor=factor, patience=patience, min_lr=min_lr)
        for v in seq:
            ref.step(v)
            mine.step(v)
            assert t.lr == opt.param_groups[0]["lr"]
    check()


def test_early_stopping_property():
    """Hypothesis: the mirror's decisions equal a direct restatement of train_model.py:21-39 on any sequence."""
    from hypothesis import given, settings, strategies as st
    from leafgrasp_amd.trainer import EarlyStopping

    class M:
        def __init__(self):
            self.v = 0

        def state_dict(self):
            return {"v": torch.tensor(self.v)}

        def load_state_dict(self, sd):
            self.v = int(sd["v"])

    @settings(max_examples=80, deadline=None)
    @given(st.lists(st.floats(min_value=0.0, max_value=2.0, allow_nan=False), min_size=1, max_size=50),
           st.integers(min_value=1, max_value=8), st.sampled_from([0.0, 0.001, 0.05]))
    def check(seq, patience, min_delta):
        es, m = EarlyStopping(patience=patience, min_delta=min_delta), M()
        best, best_epoch, counter, stop = None, None, 0, None
        for epoch, v in enumerate(seq):
            m.v = epoch
            got = es.step(v, epoch, m)
            if best is None:
                best, best_epoch, want = v, epoch, False
            elif v > best - min_delta:
                counter += 1
                want = counter >= patience
            else:
                best, best_epoch, counter, want = v, epoch, 0, False
            assert got == want
            if got:
                stop = epoch
                break
        assert es.best_epoch == best_epoch and es.best_loss == best
        if stop is not None:
            assert m.v == best_epoch      # the best epoch's weights are back
    check()
