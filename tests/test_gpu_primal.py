"""GPU parity tests of the primal path: HIP library (through the C ABI) vs the golden fixtures captured
from the reference and vs the NumPy oracle on seeded inputs.  Bar: 1e-5 relative (BASELINE.json), written
as TOL below; most quantities agree far tighter and the assertions say how tight.
"""

from __future__ import annotations

import numpy as np
import pytest
from conftest import PRIMAL_CASES, relerr, signed_targets

import neolssvm_oracle as orc

pytestmark = pytest.mark.gpu

TOL = 1e-5  # north-star tolerance on fitted weights and LOO residuals


@pytest.fixture(scope="module")
def hp():
    import neo_ls_svm_amd as pkg

    pkg.default_context()  # raises loudly if the HIP library or the GPU is missing
    return pkg


@pytest.mark.parametrize("name", PRIMAL_CASES)
def test_featuremap_matches_reference_and_oracle(name, golden_loader, hp):
    g = golden_loader(name)
    phi = hp.featuremap(g["Xq"], g["shift"], g["scale"], g["B"])
    assert phi.shape == (g["Xq"].shape[0], int(g["D"]) + 1)
    assert relerr(phi[:8], g["phi_q"]) < 1e-13
    assert relerr(phi, orc.feature_map(g["Xq"], g["shift"], g["scale"], g["B"])) < 1e-13
    assert np.all(phi[:, -1] == 1.0)


def test_featuremap_ragged_shapes(hp):
    """n, d, D that are not multiples of any tile; one row; huge arguments for sincos."""
    rng = np.random.default_rng(11)
    for n, d, D in [(1, 1, 1), (3, 5, 7), (129, 17, 130), (257, 33, 255), (1000, 130, 384)]:
        X = rng.standard_normal((n, d)) * 3
        shift, scale = rng.standard_normal(d), rng.uniform(0.5, 2, d) * rng.choice([-1, 1], d)
        B = rng.standard_normal((d, D))
        phi = hp.featuremap(X, shift, scale, B)
        assert relerr(phi, orc.feature_map(X, shift, scale, B)) < 1e-12
    X = rng.standard_normal((64, 4)) * 1e6  # |t| ~ 1e6: needs a real argument reduction
    B = rng.standard_normal((4, 16))
    phi = hp.featuremap(X, np.zeros(4), np.ones(4), B)
    ref = orc.feature_map(X, np.zeros(4), np.ones(4), B)
    assert relerr(phi, ref) < 1e-9


def test_fused_decision_function_with_arguments_straddling_the_short_sincos_range(hp):
    """k_featuremap_gemv picks the short or the library sincos per WAVE: one outlier row whose arguments lie on both sides
    of 2^30 (and one with |t| far beyond it) must not disturb the 16-lane reduction of the rows that share its wave."""
    rng = np.random.default_rng(12)
    n, d, D = 300, 3, 200
    X = rng.standard_normal((n, d))
    B = rng.standard_normal((d, D))
    B[0, ::2] *= 1e-3  # the outlier row's arguments: ~1e6 in the even columns, ~1e9-1e10 (>= 2^30) in the odd ones
    X[17, 0] = 3.0e9
    X[211] = (-7.0e12, 2.5e11, 1.0e10)
    beta = rng.standard_normal(D + 1) + 1j * rng.standard_normal(D + 1)
    shift, scale = np.zeros(d), np.ones(d)
    t17 = np.abs(X[17] @ B)
    assert t17.min() < 2.0**30 < t17.max()
    yhat, _ = hp.primal_predict(X, shift, scale, B, beta=beta)
    phi = hp.featuremap(X, shift, scale, B)  # the unfused kernel (per-thread choice, no cross-lane step)
    ref = np.real(phi @ beta)
    ok = np.ones(n, bool)
    ok[[17, 211]] = False
    assert relerr(yhat[ok], ref[ok]) < 1e-13
    assert relerr(yhat[ok], np.real(orc.feature_map(X[ok], shift, scale, B) @ beta)) < 1e-12
    assert np.max(np.abs(yhat[~ok] - ref[~ok])) < 1e-9 * np.sqrt(D)  # huge arguments: same library sincos in both kernels


@pytest.mark.parametrize("name", PRIMAL_CASES)
def test_gram_matches_oracle(name, golden_loader, hp):
    g = golden_loader(name)
    y = signed_targets(g)
    A, b = hp.gram(g["X"], y, g["s"], g["shift"], g["scale"], g["B"])
    phi = orc.feature_map(g["X"], g["shift"], g["scale"], g["B"])
    A0, b0, _ = orc.primal_gram(phi, y, g["s"])
    assert relerr(A, A0) < 1e-12
    assert relerr(b, b0) < 1e-12
    assert np.array_equal(A, A.conj().T)  # exactly Hermitian, like (A + A^H) / 2 in the reference
    if "A_over_c" in g:
        assert relerr(A * phi.size, g["A_over_c"]) < 1e-12


@pytest.mark.parametrize("name", PRIMAL_CASES[:3])
def test_rotation_kernel_matches_oracle(name, golden_loader, hp):
    """K4 in isolation with a random unitary-free Q: U = Re(P v), Gm = |P|^2 (3M complex product on the GPU)."""
    g = golden_loader(name)
    rng = np.random.default_rng(1)
    D1 = int(g["D"]) + 1
    Q = (rng.standard_normal((D1, D1)) + 1j * rng.standard_normal((D1, D1))) / np.sqrt(D1)
    v = rng.standard_normal(D1) + 1j * rng.standard_normal(D1)
    X = g["X"][:700]
    U, Gm = hp.rotate(X, g["shift"], g["scale"], g["B"], Q, v)
    P = orc.feature_map(X, g["shift"], g["scale"], g["B"]) @ Q
    assert relerr(U, np.real(P * v[None, :])) < 1e-12
    assert relerr(Gm, P.real**2 + P.imag**2) < 1e-12


@pytest.mark.parametrize("name", PRIMAL_CASES)
def test_primal_fit_matches_reference_fixture(name, golden_loader, hp):
    g = golden_loader(name)
    y, is_clf = signed_targets(g), g["task"] == "clf"
    r = hp.primal_fit(g["X"], y, g["s"], g["shift"], g["scale"], g["B"], is_clf)
    assert np.array_equal(r["gammas"], g["gammas"])
    assert relerr(r["lam"], g["lam"]) < 1e-9
    assert relerr(r["loo_errors_gammas"], g["loo_errors_gammas"]) < TOL
    # argmin equality is reported separately from the numeric bar (SURVEY.md section 7 "argmin fragility")
    assert r["opt"] == int(g["opt"])
    assert r["gamma"] == float(g["gamma"])
    assert relerr(r["beta"], g["beta"]) < TOL
    assert relerr(r["loo_residuals"], g["loo_residuals"]) < TOL
    assert relerr(r["loo_yhat"], g["loo_yhat"]) < TOL
    assert relerr(r["loo_leverage"], g["loo_leverage"]) < TOL
    assert relerr(r["loo_std"], g["loo_std"]) < TOL
    assert relerr(r["residuals"], g["residuals"]) < TOL
    assert abs(r["loo_error"] - float(g["loo_error"])) < TOL * abs(float(g["loo_error"]))
    assert abs(r["loo_score"] - float(g["loo_score"])) < 1e-9
    if g["L"].size:
        iu = np.triu_indices(r["L"].shape[0])
        assert relerr(r["L"][iu], g["L"][iu]) < TOL


@pytest.mark.parametrize("name", PRIMAL_CASES)
def test_primal_fit_at_reference_index_is_tight(name, golden_loader, hp):
    """Forced to the reference's grid index the whole result agrees to ~1e-9, far inside the bar."""
    g = golden_loader(name)
    y, is_clf = signed_targets(g), g["task"] == "clf"
    r = hp.primal_fit(g["X"], y, g["s"], g["shift"], g["scale"], g["B"], is_clf, gamma_index=int(g["opt"]))
    assert r["opt"] == int(g["opt"])
    assert relerr(r["beta"], g["beta"]) < 1e-7
    assert relerr(r["loo_residuals"], g["loo_residuals"]) < 1e-8
    assert relerr(r["loo_errors_gammas"], g["loo_errors_gammas"]) < 1e-8


@pytest.mark.parametrize("name", PRIMAL_CASES)
def test_primal_predict_matches_reference_fixture(name, golden_loader, hp):
    g = golden_loader(name)
    y, is_clf = signed_targets(g), g["task"] == "clf"
    r = hp.primal_fit(g["X"], y, g["s"], g["shift"], g["scale"], g["B"], is_clf)
    yq, sq = hp.primal_predict(g["Xq"], g["shift"], g["scale"], g["B"], beta=r["beta"], L=r["L"])
    assert relerr(yq, g["decision_function"]) < TOL
    assert relerr(sq, g["predict_std"]) < TOL
    # inference from the reference's own weights isolates K1 + K8
    yq2, _ = hp.primal_predict(g["Xq"], g["shift"], g["scale"], g["B"], beta=g["beta"])
    assert relerr(yq2, g["decision_function"]) < 1e-11


def test_primal_fit_device_resident_inputs_and_chunking(golden_loader, hp):
    """Inputs already in HBM give the same bits as host inputs; a tiny workspace forces several row chunks."""
    g = golden_loader("primal_reg_n5000_d16_D256_w")
    ctx = hp.default_context()
    y = signed_targets(g)
    r0 = hp.primal_fit(g["X"], y, g["s"], g["shift"], g["scale"], g["B"], False)
    dX, dy, ds = ctx.to_device(g["X"]), ctx.to_device(y), ctx.to_device(g["s"])
    r1 = hp.primal_fit(dX, dy, ds, g["shift"], g["scale"], g["B"], False)
    for k in ("beta", "loo_residuals", "loo_errors_gammas", "residuals", "loo_std"):
        assert np.array_equal(r0[k], r1[k]), k
    ctx._check(ctx.lib.nls_set_workspace_limit(ctx.handle, 64 << 20))
    try:
        r2 = hp.primal_fit(g["X"], y, g["s"], g["shift"], g["scale"], g["B"], False)
    finally:
        ctx._check(ctx.lib.nls_set_workspace_limit(ctx.handle, 0))
    assert r2["timings"]["row_chunk"] < 5000
    assert relerr(r2["beta"], r0["beta"]) < 1e-9
    assert relerr(r2["loo_residuals"], r0["loo_residuals"]) < 1e-9
    assert r2["opt"] == r0["opt"]


def test_gram_tile_orders_are_bit_identical(golden_loader, monkeypatch):
    """The Gram kernel's workgroup order (plain / XCD-contiguous / XCD patches, ``NLS_GRAM_ORDER``) decides which L2 sees which operand panel,
    never the arithmetic: one slab slot per (row split, tile), fixed-order reduction - A and b are bit-identical in all three."""
    import neo_ls_svm_amd as pkg

    rng = np.random.default_rng(11)
    for n, d, D in ((9000, 16, 700), (3000, 8, 100), (5000, 8, 384), (40_000, 8, 1100)):  # 6 / 1 / 3 / 9 tile rows: partial, single and several patch bands
        X = rng.standard_normal((n, d))
        y = rng.standard_normal(n)
        s = rng.uniform(0.5, 2.0, n)
        B = pkg.orf_frequencies(d, D) * 0.3
        out = {}
        for order in ("plain", "contiguous", "patch"):
            monkeypatch.setenv("NLS_GRAM_ORDER", order)
            ctx = pkg.Context(0)
            out[order] = pkg.gram(X, y, s, np.zeros(d), np.ones(d), B, ctx=ctx)
            ctx.close()
        for order in ("contiguous", "patch"):
            assert np.array_equal(out[order][0], out["plain"][0]) and np.array_equal(out[order][1], out["plain"][1]), (order, D)
        if D == 700:
            A, b = orc.primal_gram(orc.feature_map(X, np.zeros(d), np.ones(d), B), y, s)[:2]
            assert relerr(out["patch"][0], A) < 1e-12 and relerr(out["patch"][1], b) < 1e-12


def test_returned_residuals_are_those_of_the_returned_beta(golden_loader, hp):
    """``residuals`` is Re(phi beta) - y of the RETURNED beta - the Cholesky re-solve when ``L`` is requested - as the reference computes it
    (``_neo_ls_svm.py:176-182``): equal to rounding to the decision function of that beta, and closer to the reference's fixture than the
    sweep table's column (``residuals_from_sweep=True``: the eigendecomposition's beta at gamma*, cond * eps away).  Without ``L`` the
    returned beta IS the eigendecomposition's, and the column is its residual vector."""
    for name in ("primal_reg_n3000_d20_D256", "primal_clf_n3000_d16_D256_wz"):
        g = golden_loader(name)
        y = signed_targets(g)
        clf = g["task"] == "clf"

        def residuals_of(beta):
            yhat, _ = hp.primal_predict(g["X"], g["shift"], g["scale"], g["B"], beta=beta)
            e = yhat - y
            return np.where(y * e > 0, 0.0, e) if clf else e  # residual clipping of the classifier (_neo_ls_svm.py:180-182)

        scale = max(np.max(np.abs(g["residuals"])), 1e-300)
        r = hp.primal_fit(g["X"], y, g["s"], g["shift"], g["scale"], g["B"], clf)
        assert np.max(np.abs(residuals_of(r["beta"]) - r["residuals"])) <= 2e-13 * scale, name
        assert np.max(np.abs(r["residuals"] - g["residuals"])) <= 1e-11 * scale, name
        rs = hp.primal_fit(g["X"], y, g["s"], g["shift"], g["scale"], g["B"], clf, residuals_from_sweep=True)
        assert np.array_equal(rs["beta"], r["beta"])
        assert np.max(np.abs(rs["residuals"] - r["residuals"])) <= 1e-9 * scale, name
        re = hp.primal_fit(g["X"], y, g["s"], g["shift"], g["scale"], g["B"], clf, want_L=False)
        assert np.max(np.abs(residuals_of(re["beta"]) - re["residuals"])) <= 1e-9 * scale, name
        assert np.array_equal(re["residuals"], rs["residuals"])  # both are the sweep's column


def test_sigma_grid_matches_reference(golden_loader, hp):
    """gamma x sigma grid of SURVEY.md 8(d): B / sigma_k with the 32-point sub-grid gammas[::33]."""
    sg = golden_loader("sigma_grid_reg_n3000")
    g = golden_loader(sg["base"])
    gam = hp.gamma_grid(1024)[::33]
    for k, sigma in enumerate(sg["sigmas"]):
        r = hp.primal_fit(g["X"], g["y"], g["s"], g["shift"], g["scale"], g["B"] / sigma, False, gammas=gam, want_L=False)
        assert relerr(r["loo_errors_gammas"], sg["loo_errors"][k]) < TOL


@pytest.mark.parametrize("G", [32, 64])
def test_small_grid_sweep_equals_the_tile_sweep(G, golden_loader, hp, monkeypatch):
    """Grids of <= 64 points take the streaming sweep ``k_sweep_small<2>`` / ``<4>`` (the gamma x sigma grid's 32 points); ``NLS_SWEEP_SMALL=0``
    (read per call) sends the same fit through the 128-wide tile kernel it replaces: error curve, objective, selection and row outputs must
    agree to rounding - and both with the reference's fixture curve on the sub-grid."""
    for name in ("primal_reg_n3000_d20_D256", "primal_clf_n3000_d16_D256_wz"):
        g = golden_loader(name)
        y, clf = signed_targets(g), g["task"] == "clf"
        step = 1024 // G
        gam = hp.gamma_grid(1024)[::step][:G]
        monkeypatch.setenv("NLS_SWEEP_SMALL", "1")
        a = hp.primal_fit(g["X"], y, g["s"], g["shift"], g["scale"], g["B"], clf, gammas=gam)
        monkeypatch.setenv("NLS_SWEEP_SMALL", "0")
        b = hp.primal_fit(g["X"], y, g["s"], g["shift"], g["scale"], g["B"], clf, gammas=gam)
        monkeypatch.delenv("NLS_SWEEP_SMALL")
        assert a["opt"] == b["opt"]
        for k in ("loo_errors_gammas", "objective", "loo_residuals", "loo_leverage", "loo_std", "beta"):
            assert relerr(a[k], b[k]) < 1e-12, (name, k, relerr(a[k], b[k]))
        assert relerr(a["loo_errors_gammas"], g["loo_errors_gammas"][::step][:G]) < TOL


def test_sigma_grid_driver(golden_loader, hp):
    """The gamma x sigma driver reproduces the reference's per-sigma error tables and picks the joint minimum."""
    sg = golden_loader("sigma_grid_reg_n3000")
    g = golden_loader(sg["base"])
    r = hp.primal_fit_sigma_grid(g["X"], g["y"], g["s"], g["shift"], g["scale"], g["B"], False, sg["sigmas"])
    assert r["loo_errors"].shape == (3, 32)
    assert relerr(r["loo_errors"], sg["loo_errors"]) < TOL
    k, gi = np.unravel_index(np.argmin(sg["loo_errors"]), sg["loo_errors"].shape)
    assert (r["sigma_index"], r["gamma_index"]) == (k, gi)
    assert r["best"]["opt"] == gi and r["best"]["beta"].shape == (257,)
    # rank-sharded sigmas (two "ranks" run in sequence here) merge to the same table
    parts = [hp.primal_fit_sigma_grid(g["X"], g["y"], g["s"], g["shift"], g["scale"], g["B"], False, sg["sigmas"], rank=rk, world=2)
             for rk in range(2)]  # fmt: skip
    merged = np.where(np.isnan(parts[0]["loo_errors"]), parts[1]["loo_errors"], parts[0]["loo_errors"])
    assert relerr(merged, sg["loo_errors"]) < TOL


def test_sigma_grid_edge_cases(golden_loader, hp):
    """``nls_primal_fit_grid`` on the edges: one sigma; one gamma; a classifier with zero-weight rows; unsorted sigmas (visiting order is by
    |ln sigma|, the table is indexed as given); argument errors come back as ValueError before any GPU work."""
    sys_path_tests = str(__import__("pathlib").Path(__file__).resolve().parent)
    import sys

    if sys_path_tests not in sys.path:
        sys.path.insert(0, sys_path_tests)
    import _grid_reference_driver as ref

    ctx = hp.default_context()
    for name in ("primal_reg_n3000_d20_D256", "primal_clf_n3000_d16_D256_wz"):
        g = golden_loader(name)
        y, clf = signed_targets(g), g["task"] == "clf"
        args = (g["X"], y, g["s"], g["shift"], g["scale"])
        for sig, gam in (([1.3], hp.gamma_grid(1024)[::33]), ([2.0, 0.5, 1.0, 0.8], hp.gamma_grid(64)), ([0.9, 1.1], np.array([1e-3]))):
            sig = np.asarray(sig)

            def fit(Bs, finish_below, gam=gam):
                return hp.primal_fit(*args, Bs, clf, gammas=gam, ctx=ctx, finish_below=finish_below)

            want = ref.grid(fit, g["B"], sig, gam)
            got = hp.primal_fit_sigma_grid(*args, g["B"], clf, sig, gammas=gam, ctx=ctx)
            assert got["loo_errors"].shape == (len(sig), len(gam))
            assert np.array_equal(got["loo_errors"], want["loo_errors"]) and np.array_equal(got["objective"], want["objective"])
            assert (got["sigma_index"], got["gamma_index"], got["finished_count"]) == (want["sigma_index"], want["gamma_index"], want["finished_count"])
            for k in ("beta", "loo_residuals", "loo_std", "residuals"):
                assert np.array_equal(got["best"][k], want["best"][k]), (name, k)
            assert got["best"]["loo_score"] == want["best"]["loo_score"]
    g = golden_loader("primal_reg_n3000_d20_D256")
    args = (g["X"], g["y"], g["s"], g["shift"], g["scale"], g["B"], False)
    for bad in ([], [1.0, -2.0], [0.0], [np.inf]):
        with pytest.raises(ValueError):
            hp.primal_fit_sigma_grid(*args, bad, ctx=ctx)
    with pytest.raises(ValueError):  # rank outside the world
        hp.primal_fit_sigma_grid(*args, [1.0, 2.0], ctx=ctx, rank=2, world=2)
    with pytest.raises(ValueError):  # the merge context must not be the fitting context
        hp.primal_fit_sigma_grid(*args, [1.0, 2.0], ctx=ctx, rank=0, world=2, merge_ctx=ctx)


@pytest.mark.parametrize("task", ["reg", "clf"])
def test_primal_fit_vs_oracle_seeded(task, hp):
    """Seeded synthetic problem of the BASELINE generator at a size the oracle finishes in seconds."""
    rng = np.random.default_rng(5)
    n, d, D = 6000, 32, 384
    X = rng.standard_normal((n, d))
    w = rng.standard_normal(d) / np.sqrt(d)
    y = np.sin(X @ w) + 0.1 * rng.standard_normal(n) if task == "reg" else np.where(X @ w + 0.3 * rng.standard_normal(n) > 0, 1.0, -1.0)
    s = rng.uniform(0.5, 2.0, n)
    B = hp.orf_frequencies(d, D) * 0.3
    shift, scale = X.mean(0), X.std(0)
    o = orc.primal_fit_streamed(X, y, s, shift, scale, B, task == "clf")
    r = hp.primal_fit(X, y, s, shift, scale, B, task == "clf", gamma_index=o["opt"])
    assert relerr(r["loo_errors_gammas"], o["loo_errors_gammas"]) < 1e-7
    assert relerr(r["beta"], o["beta"]) < TOL
    assert relerr(r["loo_residuals"], o["loo_residuals"]) < 1e-7
    assert relerr(r["loo_leverage"], o["loo_leverage"]) < 1e-7
    assert relerr(r["loo_std"], o["loo_std"]) < 1e-7
    assert abs(r["loo_score"] - o["loo_score"]) < 1e-9
    r_free = hp.primal_fit(X, y, s, shift, scale, B, task == "clf")
    assert abs(r_free["opt"] - o["opt"]) <= 1  # near-flat curves may flip by one grid step


def test_argument_errors_raise_like_the_reference(hp):
    rng = np.random.default_rng(0)
    X, y = rng.standard_normal((50, 3)), rng.standard_normal(50)
    B = rng.standard_normal((3, 8))
    with pytest.raises(ValueError):  # zero scale: _affine_feature_map.py:53
        hp.primal_fit(X, y, np.ones(50), np.zeros(3), np.array([1.0, 0.0, 1.0]), B, False)
    with pytest.raises(ValueError):  # weights summing to zero
        hp.primal_fit(X, y, np.zeros(50), np.zeros(3), np.ones(3), B, False)
    with pytest.raises(ValueError):
        hp.featuremap(X, np.zeros(3), np.ones(3), rng.standard_normal((4, 8)))


def test_predict_std_factor_handle(golden_loader, hp):
    """predict_std from an explicit factor handle (U^-1 kept on the device) equals the on-the-fly path; an edited L needs
    a new handle - nothing is recognised by address."""
    g = golden_loader("primal_reg_n3000_d20_D256")
    y = signed_targets(g)
    ctx = hp.default_context()
    r = hp.primal_fit(g["X"], y, g["s"], g["shift"], g["scale"], g["B"], False)
    L = np.ascontiguousarray(r["L"])
    _, s1 = hp.primal_predict(g["Xq"], g["shift"], g["scale"], g["B"], L=L)
    f = hp.Factor(ctx, L)
    _, s2 = hp.primal_predict(g["Xq"][:100], g["shift"], g["scale"], g["B"], factor=f)
    y3, s3 = hp.primal_predict(g["Xq"], g["shift"], g["scale"], g["B"], beta=r["beta"], factor=f)
    assert np.array_equal(s1[:100], s2) and np.array_equal(s1, s3) and relerr(s1, g["predict_std"]) < TOL
    assert relerr(y3, g["decision_function"]) < TOL
    L *= 2.0  # in-place edit: the on-the-fly path sees it, the handle (a snapshot) does not
    _, s4 = hp.primal_predict(g["Xq"], g["shift"], g["scale"], g["B"], L=L)
    _, s5 = hp.primal_predict(g["Xq"], g["shift"], g["scale"], g["B"], factor=f)
    assert relerr(s4, s1 / 2) < 1e-12 and np.array_equal(s5, s1)
    f2 = hp.Factor(ctx, L)
    _, s6 = hp.primal_predict(g["Xq"], g["shift"], g["scale"], g["B"], factor=f2)
    assert relerr(s6, s1 / 2) < 1e-12
    f.close()
    f2.close()
    with pytest.raises(ValueError):
        hp.primal_predict(g["Xq"], g["shift"], g["scale"], g["B"], factor=f)
    with pytest.raises(ValueError):  # wrong feature count
        hp.primal_predict(g["Xq"], g["shift"], g["scale"], g["B"][:, :100], factor=hp.Factor(ctx, L))


def test_cholesky_failure_raises_linalgerror(golden_loader, hp):
    """rocsolver_zpotrf info > 0 -> numpy.linalg.LinAlgError, as scipy's cho_factor raises in the reference (:177)."""
    g = golden_loader("primal_reg_n2000_d48_D32")
    y = signed_targets(g)
    with pytest.raises(np.linalg.LinAlgError):  # gamma * c I + A is indefinite for a hugely negative gamma
        hp.primal_fit(g["X"], y, g["s"], g["shift"], g["scale"], g["B"], False, gammas=np.array([-1e9]))
    r = hp.primal_fit(g["X"], y, g["s"], g["shift"], g["scale"], g["B"], False)  # the context is usable afterwards
    assert relerr(r["beta"], g["beta"]) < TOL


def test_sweep_only_and_finish_below(golden_loader, hp):
    g = golden_loader("primal_reg_n3000_d20_D256")
    y = signed_targets(g)
    full = hp.primal_fit(g["X"], y, g["s"], g["shift"], g["scale"], g["B"], False)
    so = hp.primal_fit(g["X"], y, g["s"], g["shift"], g["scale"], g["B"], False, sweep_only=True)
    assert not so["finished"] and "beta" not in so and "L" not in so and so["opt"] == full["opt"]
    assert np.array_equal(so["loo_errors_gammas"], full["loo_errors_gammas"]) and np.array_equal(so["lam"], full["lam"])
    assert so["timings"]["cholesky"] == 0.0 and so["timings"]["residuals"] == 0.0
    best = full["objective"][full["opt"]]
    lo = hp.primal_fit(g["X"], y, g["s"], g["shift"], g["scale"], g["B"], False, finish_below=best)  # not strictly below
    hi = hp.primal_fit(g["X"], y, g["s"], g["shift"], g["scale"], g["B"], False, finish_below=best * (1 + 1e-9))
    assert not lo["finished"] and hi["finished"] and np.array_equal(hi["beta"], full["beta"])


def test_workspace_limit_and_release(golden_loader, hp):
    """An explicit workspace limit is a hard bound for buffers that cannot be chunked (the EVD's n x n matrices), and
    nls_ws_release returns the arena."""
    ctx = hp.Context(0)
    try:
        g = golden_loader("primal_reg_n3000_d20_D256")
        r = hp.primal_fit(g["X"], signed_targets(g), g["s"], g["shift"], g["scale"], g["B"], False, ctx=ctx)
        assert r["finished"]
        held = ctx.release_workspace(min_bytes=1 << 20)
        assert held < 64 << 20
        assert ctx.release_workspace() == 0
        ctx._check(ctx.lib.nls_set_workspace_limit(ctx.handle, 8 << 20))
        A = np.eye(3000)
        with pytest.raises(ValueError):  # 72 MB for one 3000 x 3000 matrix > 8 MB (+ slack)
            hp.eigh(A, ctx=ctx)
        ctx._check(ctx.lib.nls_set_workspace_limit(ctx.handle, 0))
        lam, _ = hp.eigh(A[:200, :200], ctx=ctx)
        assert np.allclose(lam, 1.0)
    finally:
        ctx.close()


def test_eigh_of_badly_scaled_matrices(hp):
    """Entries far outside [1e-100, 1e100] go through the driver-level scaling (LAPACK's zheev scales at the driver level too)."""
    rng = np.random.default_rng(3)
    M = rng.standard_normal((150, 150)) + 1j * rng.standard_normal((150, 150))
    A = M @ M.conj().T / 150
    lam0 = np.linalg.eigvalsh(A)
    for f in (1e-200, 1e200):
        lam, Q = hp.eigh(A * f)
        assert np.max(np.abs(lam / f - lam0)) <= 1e-11 * lam0[-1]
        assert np.max(np.abs(Q.conj().T @ Q - np.eye(150))) <= 1e-11


def test_single_rank_rccl_communicator(golden_loader, hp):
    """A one-rank native RCCL communicator: the library's collectives (all-reduce of the packed Gram, broadcast,
    all-gather) really run through librccl on the device buffers and leave the results bit-identical."""
    g = golden_loader("primal_reg_n3000_d20_D256")
    y = signed_targets(g)
    r0 = hp.primal_fit(g["X"], y, g["s"], g["shift"], g["scale"], g["B"], False)
    ctx = hp.Context(0)
    try:
        ctx.comm_init(ctx.comm_unique_id(), 0, 1)
        assert np.array_equal(ctx.comm_allreduce([1.5, -2.0], "max"), [1.5, -2.0])
        ctx.comm_barrier()
        r1 = hp.primal_fit(g["X"], y, g["s"], g["shift"], g["scale"], g["B"], False, ctx=ctx)
        for k in ("beta", "loo_residuals", "loo_errors_gammas", "lam", "L"):
            assert np.array_equal(r0[k], r1[k]), k
        ctx.comm_destroy()
    finally:
        ctx.close()


def test_failure_handling_entry_points_on_real_rccl(golden_loader, hp, monkeypatch):
    """The failure contract's RCCL calls - ``ncclCommAbort``, ``ncclCommGetAsyncError``, the status vote's all-reduce, the polling wait - on the
    REAL librccl (a one-rank communicator: the only kind one GPU allows): a local fault goes through the vote and leaves the communicator
    joined; ``nls_comm_abort`` gives it up, the context then refuses sharded work until it joins a new one; results are unchanged throughout."""
    from neo_ls_svm_amd._lib import NlsError

    g = golden_loader("primal_reg_n3000_d20_D256")
    y = signed_targets(g)

    def fit(ctx):
        return hp.primal_fit(g["X"], y, g["s"], g["shift"], g["scale"], g["B"], False, ctx=ctx)

    ctx = hp.Context(0)
    try:
        assert ctx.comm_state == "none"
        ctx.comm_init(ctx.comm_unique_id(), 0, 1)
        ctx.comm_set_timeout(60.0)
        assert ctx.comm_state == "joined"
        r1 = fit(ctx)
        for site in ("prepare", "gram", "evd", "backtransform", "sweep", "select", "cholesky"):
            monkeypatch.setenv("NLS_FAULT_INJECT", f"{site}:0")
            with pytest.raises(NlsError, match=f"injected fault at '{site}'"):
                fit(ctx)
            assert ctx.comm_state == "joined", site  # left through the vote: nothing pending, communicator intact
        monkeypatch.setenv("NLS_FAULT_INJECT", "cholesky:0:3")
        with pytest.raises(np.linalg.LinAlgError):
            fit(ctx)
        monkeypatch.delenv("NLS_FAULT_INJECT")
        r2 = fit(ctx)
        for k in ("beta", "loo_residuals", "loo_errors_gammas", "residuals", "L"):
            assert np.array_equal(r1[k], r2[k]), k
        ctx.comm_abort()
        assert ctx.comm_state == "aborted"
        with pytest.raises(NlsError, match="aborted"):
            fit(ctx)
        with pytest.raises(NlsError, match="aborted"):
            ctx.comm_allreduce([1.0])
        ctx.comm_init(ctx.comm_unique_id(), 0, 1)  # a new communicator
        assert ctx.comm_state == "joined"
        r3 = fit(ctx)
        assert np.array_equal(r1["beta"], r3["beta"]) and np.array_equal(r1["L"], r3["L"])
        ctx.comm_destroy()
        assert ctx.comm_state == "none"
    finally:
        ctx.close()


@pytest.mark.parametrize("name", ["primal_reg_n3000_d20_D256", "primal_clf_n3000_d16_D256_wz"])
def test_compressed_sweep_equals_direct_products(name, golden_loader, hp, monkeypatch):
    """The 1024-point sweep through 128 Chebyshev nodes (default) against the direct n x (D+1) x G products."""
    g = golden_loader(name)
    y, is_clf = signed_targets(g), g["task"] == "clf"
    rc = hp.primal_fit(g["X"], y, g["s"], g["shift"], g["scale"], g["B"], is_clf)
    monkeypatch.setenv("NLS_SWEEP_DIRECT", "1")
    rd = hp.primal_fit(g["X"], y, g["s"], g["shift"], g["scale"], g["B"], is_clf)
    assert rc["opt"] == rd["opt"]
    assert relerr(rc["loo_errors_gammas"], rd["loo_errors_gammas"]) < 1e-12
    assert relerr(rc["objective"], rd["objective"]) < 1e-12
    for k in ("loo_residuals", "loo_leverage", "loo_std"):
        assert relerr(rc[k], rd[k]) < 1e-11, k
    assert np.array_equal(rc["beta"], rd["beta"])  # the re-solve does not depend on the sweep


def test_c_abi_edge_cases_of_the_round2_entry_points(golden_loader, hp):
    """Device-pointer L for nls_factor_create, the communicator / hook exclusivity, size limits of the utility collective."""
    import ctypes as C

    g = golden_loader("primal_reg_n2000_d48_D32")
    y = signed_targets(g)
    ctx = hp.Context(0)
    try:
        r = hp.primal_fit(g["X"], y, g["s"], g["shift"], g["scale"], g["B"], False, ctx=ctx)
        L = np.ascontiguousarray(r["L"])
        D = L.shape[0] - 1
        dL = ctx.to_device(L.view(np.float64))  # the factor may already be resident
        h = C.c_void_p()
        ctx._check(ctx.lib.nls_factor_create(ctx.handle, C.c_void_p(dL.ptr), D, C.byref(h)))
        f = hp.Factor.__new__(hp.Factor)
        f.ctx, f.D, f.handle = ctx, D, h
        _, s_dev = hp.primal_predict(g["Xq"], g["shift"], g["scale"], g["B"], factor=f, ctx=ctx)
        _, s_host = hp.primal_predict(g["Xq"], g["shift"], g["scale"], g["B"], L=L, ctx=ctx)
        assert np.array_equal(s_dev, s_host) and relerr(s_dev, g["predict_std"]) < TOL
        f.close()
        assert ctx.lib.nls_factor_destroy(ctx.handle, h) != 0  # already destroyed: not a live handle any more
        with pytest.raises(ValueError):
            hp.Factor(ctx, np.eye(3)[:2])  # not square
        # a context with a native communicator refuses a hook, and the utility collective has a size limit
        ctx.comm_init(ctx.comm_unique_id(), 0, 1)
        with pytest.raises(ValueError):
            ctx.set_allreduce(lambda p, c: None, 0, 2)
        with pytest.raises(ValueError):
            ctx.comm_allreduce(np.zeros(9000))
        assert np.array_equal(ctx.comm_allreduce(np.arange(5.0), "sum"), np.arange(5.0))
        ctx.comm_destroy()
        ctx.set_allreduce(None, 0, 1)  # fine again once the communicator is gone
        with pytest.raises(ValueError):
            ctx.comm_init(b"short", 0, 1)
    finally:
        ctx.close()


@pytest.mark.parametrize("name", ["primal_reg_n3000_d20_D256", "primal_clf_n3000_d16_D256_wz"])
@pytest.mark.parametrize("gi", [0, 1023])
def test_beta_is_the_cholesky_resolve(name, gi, golden_loader, hp):
    """``_neo_ls_svm.py:176-178``: beta = cho_solve(cho_factor(gamma* C + A), b).  At the SMALL-gamma edge of the grid (index 0:
    gamma = 1e-6, the worst-conditioned system of the sweep) and at the large one, the returned pair must satisfy
    beta == cho_solve(L_, b) to rounding, and a fit that does not ask for L_ (beta from the eigendecomposition, positive definiteness
    checked on the eigenvalues) must agree with it far inside the parity bar."""
    import scipy.linalg as sla

    g = golden_loader(name)
    y, is_clf = signed_targets(g), g["task"] == "clf"
    r = hp.primal_fit(g["X"], y, g["s"], g["shift"], g["scale"], g["B"], is_clf, gamma_index=gi)
    _, b = hp.gram(g["X"], y, g["s"], g["shift"], g["scale"], g["B"])
    beta_ref = sla.cho_solve((r["L"], False), b)
    assert np.linalg.norm(r["beta"] - beta_ref) <= 1e-10 * np.linalg.norm(beta_ref)
    r2 = hp.primal_fit(g["X"], y, g["s"], g["shift"], g["scale"], g["B"], is_clf, gamma_index=gi, want_L=False)
    assert "L" not in r2
    assert np.linalg.norm(r2["beta"] - r["beta"]) <= 1e-7 * np.linalg.norm(r["beta"])
    for k in ("loo_residuals", "loo_leverage", "loo_std", "loo_errors_gammas"):
        assert np.array_equal(r2[k], r[k]), k
    # residuals belong to the RETURNED beta (the re-solve with L, the eigendecomposition's without): the two agree to cond * eps
    assert np.max(np.abs(r2["residuals"] - r["residuals"])) <= 1e-7 * np.max(np.abs(r["residuals"]))


@pytest.mark.gpu
def test_large_factor_outputs_come_from_the_pool_and_can_be_page_locked(hp):
    """``L_`` of at least 64 MB is handed out from the pool of host mappings, page-locked once through the context after ``pin_large_outputs(True)``; the factor that arrives in it
    (asynchronous copies beside the factorisation) is the one a plain pageable buffer receives; buffers are recycled once nothing holds them."""
    import gc

    import scipy.linalg as sla
    from neo_ls_svm_amd import _hostpool as pool

    pool.release()
    pool.pin_large_outputs(True)  # (default False: with the unaligned start of the pooled arrays the staged copies are as fast)
    rng = np.random.default_rng(5)
    n, d, D = 6000, 12, 2100  # (D + 1)^2 complex = 71 MB
    X = rng.standard_normal((n, d))
    y = np.sin(X @ rng.standard_normal(d)) + 0.1 * rng.standard_normal(n)
    s = np.ones(n)
    shift, scale, B = X.mean(0), X.std(0), hp.orf_frequencies(d, D) * 0.5
    ctx = hp.default_context()
    r1 = hp.primal_fit(X, y, s, shift, scale, B, False, ctx=ctx)
    assert type(r1["L"].base).__name__ == "_Lease" and len(pool._registered) == 1
    addr = r1["L"].__array_interface__["data"][0]
    iu = np.triu_indices(D + 1)
    L1 = r1["L"][iu].copy()
    A, b = hp.gram(X, y, s, shift, scale, B, ctx=ctx)
    beta_ref = sla.cho_solve((r1["L"], False), b)  # the downloaded factor reproduces beta
    assert relerr(r1["beta"], beta_ref) < 1e-9
    del r1
    gc.collect()
    r2 = hp.primal_fit(X, y, s, shift, scale, B, False, ctx=ctx)
    assert r2["L"].__array_interface__["data"][0] == addr and len(pool._registered) == 1  # recycled, still page-locked
    assert np.array_equal(r2["L"][iu], L1)
    del r2
    gc.collect()
    pool.pin_large_outputs(False)
    r3 = hp.primal_fit(X, y, s, shift, scale, B, False, ctx=ctx)  # the recycled mapping stays page-locked; results as before
    assert np.array_equal(r3["L"][iu], L1)
    del r3
    gc.collect()
    pool.release()
    assert pool._pooled_bytes() == 0 and not pool._registered
    pool.pin_large_outputs("reuse")  # pageable the first time, page-locked from the second hand-out on
    r4 = hp.primal_fit(X, y, s, shift, scale, B, False, ctx=ctx)
    assert not pool._registered and np.array_equal(r4["L"][iu], L1)
    del r4
    gc.collect()
    r5 = hp.primal_fit(X, y, s, shift, scale, B, False, ctx=ctx)
    assert len(pool._registered) == 1 and np.array_equal(r5["L"][iu], L1)
    del r5
    gc.collect()
    pool.release()
    pool.pin_large_outputs(False)  # (the default)
    assert pool._pooled_bytes() == 0 and not pool._registered
