"""Pins the CPU oracle (oracle/) against golden vectors produced by the unmodified reference
(oracle/make_golden.py).  CPU only.  Tolerances are fp32 rounding-level: the oracle sums in a
different order than ATen's vectorised kernels, nothing more."""
import numpy as np
import pytest

from conftest import E2ES, KATS, load_golden
from oracle import oracle as O

RTOL, ATOL = 2e-5, 2e-7


@pytest.mark.parametrize("name", KATS)
def test_kat_forward_grad_adam(orc, name):
    g = load_golden(name)
    lr, wd = float(g["lr"]), float(g["wd"])
    st = O.new_state(g["U0"], g["V0"])
    for k in range(int(g["n_steps"])):
        u, i, j, z = g[f"u{k}"], g[f"i{k}"], g[f"j{k}"], g[f"z{k}"]
        p, loss, dU, dV = orc.grad(st["U"], st["V"], u, i, j, z)
        np.testing.assert_allclose(p, g[f"p{k}"], rtol=RTOL, atol=ATOL)
        assert abs(loss - float(g[f"loss{k}"])) <= 1e-5 * max(1.0, abs(loss))
        np.testing.assert_allclose(dU, g[f"dU{k}"], rtol=1e-4, atol=1e-8)
        np.testing.assert_allclose(dV, g[f"dV{k}"], rtol=1e-4, atol=1e-8)
        # Adam on the *golden* gradient isolates T5 from T4's summation order
        for nm, gr in (("U", g[f"dU{k}"]), ("V", g[f"dV{k}"])):
            orc.adam(st[nm], st["m" + nm], st["v" + nm], gr, k + 1, lr=lr, wd=wd)
            np.testing.assert_allclose(st[nm], g[f"{nm}{k + 1}"], rtol=1e-6, atol=1e-7)
            np.testing.assert_allclose(st["m" + nm], g[f"m{nm}{k + 1}"], rtol=1e-5, atol=1e-10)
            np.testing.assert_allclose(st["v" + nm], g[f"v{nm}{k + 1}"], rtol=1e-5, atol=1e-14)


def test_kat_saturation_semantics(orc):
    """|x| large → p rounds to 0/1, loss term clamps at 100, gradient exactly 0 (SURVEY §7 hard part 4)."""
    g = load_golden("kat_saturated_d8.npz")
    p, term = orc.forward(g["U0"], g["V0"], g["u0"], g["i0"], g["j0"], g["z0"])
    sat = (g["p0"] == 0.0) | (g["p0"] == 1.0)
    assert sat.sum() >= 5, "fixture must exercise saturation"
    np.testing.assert_array_equal(p[sat], g["p0"][sat])
    wrong = sat & (g["p0"] != g["z0"].astype(np.float32))
    assert wrong.any()
    np.testing.assert_array_equal(term[wrong], np.float32(100.0))


@pytest.mark.parametrize("name", E2ES)
def test_e2e_training_matches_reference(orc, name):
    g = load_golden(name)
    lr, wd, E = float(g["lr"]), float(g["wd"]), int(g["epochs"])
    st = O.new_state(g["U0"], g["V0"])
    vd = g["val_data"]
    step = 0
    for e in range(E):
        s = g["epoch_stream"][e]
        bl = orc.train_steps(st, s[:, 0], s[:, 1], s[:, 2], s[:, 3], 64, step, lr=lr, wd=wd)
        step += len(bl)
        assert abs(O.epoch_losses(bl) - g["train_losses"][e]) < 2e-6
        vl, _, _ = orc.eval_batches(st["U"], st["V"], vd[:, 0], vd[:, 1], vd[:, 2], vd[:, 3], 64)
        assert abs(O.epoch_losses(vl) - g["val_losses"][e]) < 2e-6
    assert step == int(g["adam_step"])
    np.testing.assert_allclose(st["U"], g["U_final"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(st["V"], g["V_final"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(st["mU"], g["mU_final"], rtol=0, atol=2e-7)
    np.testing.assert_allclose(st["vV"], g["vV_final"], rtol=1e-3, atol=1e-12)
    td = g["test_data"]
    tl, corr, _ = orc.eval_batches(g["U_final"], g["V_final"], td[:, 0], td[:, 1], td[:, 2], td[:, 3], 64)
    assert abs(O.epoch_losses(tl) - float(g["test_loss"])) < 2e-6
    assert corr.sum() / len(td) == pytest.approx(float(g["test_acc"]), abs=1e-12)


@pytest.mark.parametrize("name", E2ES)
def test_metrics_match_reference(orc, name):
    g = load_golden(name)
    U, V, X, s = g["U_final"], g["V_final"], g["X"], float(g["s"])
    assert O.reconstruction_error(U, V, X, s) == pytest.approx(float(g["rec_error"]), abs=2e-6)
    r = O.alpha_and_norm_ratios(U, V, X)
    names = ["alpha", "norm_X", "norm_ratio", "rec_scaled", "pearson_mean", "pearson_std", "spearman_mean",
             "spearman_std", "svd_err", "slopes", "correlations", "spearman_scores", "rec_scaled_per_row",
             "alpha_per_row"]
    for nm, v in zip(names, r):
        ref = g["m14_" + nm]
        v = np.asarray(v, dtype=np.float64)
        assert v.shape == ref.shape, nm
        scale = max(1.0, float(np.max(np.abs(ref))) if ref.size else 1.0)
        np.testing.assert_allclose(v, ref, rtol=0, atol=5e-5 * scale, err_msg=nm)
    gl, ga = O.ground_truth_metrics(g["test_data"], X)
    assert gl == pytest.approx(float(g["gt_loss"]), abs=1e-6)
    assert ga == pytest.approx(float(g["gt_acc"]), abs=1e-12)
    # the C UV^T pass agrees with the numpy restatement
    rs, err2, ref2 = orc.uvt_stats(U, V, X, s)
    assert np.sqrt(err2 / ref2) == pytest.approx(float(g["rec_error"]), abs=2e-6)
    alpha = rs[:, 0].sum() / (rs[:, 1].sum() + 1e-8)
    assert alpha == pytest.approx(float(g["m14_alpha"]), abs=5e-5 * max(1.0, abs(float(g["m14_alpha"]))))
