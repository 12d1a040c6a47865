"""Worker of tests/test_gpu_group.py: ONE process, N "devices" that are all GPU 0, joined through the test stand-in for librccl
(NLS_RCCL_LIB must point at tests/csrc/_shim/librccl.so.1 before the library first asks for a communicator: one library per process).

modes (argv[1]; argv[2] = world):
  fit        hotpath.primal_fit / primal_predict on a Group == the same calls on one Context (regression + classification, ragged row blocks)
  estimator  NeoLSSVM(devices=[0] * world) reproduces the reference fixtures like the single-device estimator does, and its own single-device twin to 1e-9
  grid       gamma x sigma grid: C driver on one context == the Python reference driver bit for bit; Group (sigma-sharded) == one context
  big        D = 4096 (4097 = 8 * 512 + 1 eigenvector columns over the ranks), one- or two-stage EVD per NLS_EVD
  fail       an injected ncclBroadcast failure (every member's) surfaces as NlsError from the group call, no hang
  fault      ONE member fails locally (NLS_FAULT_INJECT=site:rank[:code]): the group call returns that member's error at the next status vote,
             at once; the same communicator carries the next call
  lost       ONE member's RCCL call fails (NLS_SHIM_FAIL_RANK / _CALL): the group's abort flag releases the others at once (not the
             deadline); the group joins a new communicator at its next call
Prints "OK" on success."""

from __future__ import annotations

import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "oracle"), str(ROOT / "tests")]

import numpy as np  # noqa: E402


def rel(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return float(np.max(np.abs(a - b)) / max(float(np.max(np.abs(b))), 1e-300))


def problem(n=5003, d=12, D=200, clf=False, seed=123):
    import neo_ls_svm_amd as hp

    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, d))
    w = rng.standard_normal(d) / np.sqrt(d)
    y = np.where(X @ w > 0, 1.0, -1.0) if clf else np.sin(X @ w) + 0.1 * rng.standard_normal(n)
    s = rng.uniform(0.2, 2.0, n)
    B = hp.orf_frequencies(d, D) * 0.4
    return X, y, s, rng.standard_normal(d) * 0.1, rng.uniform(0.8, 1.2, d), B


def compare_fits(r, r1, D1, tol=1e-8):
    assert r["opt"] == r1["opt"], (r["opt"], r1["opt"])
    assert rel(r["beta"], r1["beta"]) < tol
    assert rel(r["lam"], r1["lam"]) < 1e-9
    assert rel(r["loo_errors_gammas"], r1["loo_errors_gammas"]) < 1e-10
    iu = np.triu_indices(D1)
    assert rel(r["L"][iu], r1["L"][iu]) < 1e-9
    for k in ("loo_residuals", "loo_leverage", "loo_std", "residuals", "loo_yhat"):
        assert r[k].shape == r1[k].shape, k
        assert rel(r[k], r1[k]) < tol, (k, rel(r[k], r1[k]))
    assert abs(r["loo_score"] - r1["loo_score"]) < 1e-10


def mode_fit(world):
    import neo_ls_svm_amd as hp

    grp = hp.Group([0] * world)
    assert grp.size == world and len(grp.contexts) == world
    solo = hp.Context(0)
    for clf in (False, True):
        X, y, s, shift, scale, B = problem(clf=clf)  # 5003 rows: ragged blocks
        D1 = B.shape[1] + 1
        r1 = hp.primal_fit(X, y, s, shift, scale, B, clf, ctx=solo)
        for _ in range(2):  # twice: the second call runs on warm workspaces / the same communicator
            r = hp.primal_fit(X, y, s, shift, scale, B, clf, ctx=grp)
            compare_fits(r, r1, D1)
        r = hp.primal_fit_sharded(X, y, s, shift, scale, B, clf, devices=[0] * world, want_L=False)  # the process-wide group of the tuple
        assert "L" not in r and rel(r["beta"], r1["beta"]) < 1e-8
        # inference: query rows sharded over the ranks, fewer rows than ranks included
        for m in sorted({1, max(world - 1, 1), 1001}):
            Xq = X[:m] * 1.01
            y1, s1 = hp.primal_predict(Xq, shift, scale, B, beta=r1["beta"], L=r1["L"], ctx=solo)
            yg, sg = hp.primal_predict(Xq, shift, scale, B, beta=r1["beta"], L=r1["L"], ctx=grp)
            assert rel(yg, y1) < 1e-11 and rel(sg, s1) < 1e-9, (m, rel(yg, y1), rel(sg, s1))
            fac = hp.GroupFactor(grp, r1["L"])
            yf, sf = hp.primal_predict(Xq, shift, scale, B, beta=r1["beta"], factor=fac, ctx=grp)
            yo, _ = hp.primal_predict(Xq, shift, scale, B, beta=r1["beta"], ctx=grp)
            fac.close()
            assert rel(yf, y1) < 1e-11 and rel(sf, s1) < 1e-9 and rel(yo, y1) < 1e-11
    # a rank whose whole row block has zero weight; a forced gamma index; the curve only (no re-solve); the generalised-EVD branch
    # (a full complexity matrix: its Cholesky factor and the triangular solves are replicated on every rank)
    X, y, s, shift, scale, B = problem(n=4000, d=10, D=96)
    s0 = s.copy()
    s0[: 4000 // max(world, 2)] = 0.0
    for kw in ({"gamma_index": 300}, {"sweep_only": True}, {"complexity_matrix": hp.exact_complexity_matrix(hp.orf_frequencies(10, 96) * 0.4)}):
        for sw in (s, s0):
            a = hp.primal_fit(X, y, sw, shift, scale, B, False, ctx=solo, **kw)
            b = hp.primal_fit(X, y, sw, shift, scale, B, False, ctx=grp, **kw)
            assert a["opt"] == b["opt"] and rel(b["loo_errors_gammas"], a["loo_errors_gammas"]) < 1e-9, kw
            if not kw.get("sweep_only"):
                assert rel(b["beta"], a["beta"]) < 1e-7 and rel(b["loo_residuals"], a["loo_residuals"]) < 1e-7, kw
                assert rel(b["loo_std"], a["loo_std"]) < 1e-7 and abs(b["loo_score"] - a["loo_score"]) < 1e-9, kw
            else:
                assert "beta" not in b and not b["finished"]
    # argument errors come back as ValueError, before any collective
    X, y, s, shift, scale, B = problem(n=world - 1) if world > 1 else problem(n=1)
    if world > 1:
        try:
            hp.primal_fit(X, y, s, shift, scale, B, False, ctx=grp)
        except ValueError as exc:
            assert "at least" in str(exc)
        else:
            raise AssertionError("n < world accepted")
    solo.close()
    grp.close()


def mode_estimator(world):
    from conftest import PRIMAL_CASES, load_golden

    import neo_ls_svm_amd as hp

    TOL = 1e-5
    for name in ("primal_reg_n3000_d20_D256", "primal_reg_n5000_d16_D256_w", "primal_clf_n3000_d16_D256_wz"):
        assert name in PRIMAL_CASES
        g = load_golden(name)
        sw = g["s"] if bool(g["has_weights"]) else None

        def make(**kw):
            return hp.NeoLSSVM(primal_feature_map=hp.OrthogonalRandomFourierFeatures(num_features=int(g["D"])), dual=False, **kw)

        m1 = make(device=0).fit(g["X"], g["y"], sample_weight=sw)
        m = make(devices=[0] * world).fit(g["X"], g["y"], sample_weight=sw)
        assert isinstance(m._ctx(), hp.Group) and m._ctx().size == world
        # against the reference's fixture (what the single-device estimator test asserts) ...
        assert m.γ_ == float(g["gamma"])
        for got, key in ((m.β̂_, "beta"), (m.loo_residuals_, "loo_residuals"), (m.loo_ŷ_, "loo_yhat"), (m.loo_leverage_, "loo_leverage"),
                         (m.loo_std_, "loo_std"), (m.residuals_, "residuals"), (m.loo_errors_γs_, "loo_errors_gammas")):  # fmt: skip
            assert rel(got, g[key]) < TOL, (name, key)
        assert rel(m.decision_function(g["Xq"]), g["decision_function"]) < TOL
        assert rel(m.predict_std(g["Xq"]), g["predict_std"]) < TOL
        # ... and against its single-device twin to 1e-9 (same pre-step, same normal equations up to the order of the row-block sums)
        iu = np.triu_indices(int(g["D"]) + 1)
        assert np.array_equal(m.primal_feature_map_.B_, m1.primal_feature_map_.B_)
        assert m.γ_ == m1.γ_
        for a, b, what in ((m.β̂_, m1.β̂_, "beta"), (m.L_[0][iu], m1.L_[0][iu], "L"), (m.loo_residuals_, m1.loo_residuals_, "loo_residuals"),
                           (m.loo_leverage_, m1.loo_leverage_, "loo_leverage"), (m.loo_std_, m1.loo_std_, "loo_std"),
                           (m.residuals_, m1.residuals_, "residuals"), (m.loo_errors_γs_, m1.loo_errors_γs_, "curve")):  # fmt: skip
            assert rel(a, b) < 1e-9, (name, what, rel(a, b))
        assert abs(m.loo_score_ - m1.loo_score_) < 1e-10
        assert rel(m.predict_std(g["Xq"]), m1.predict_std(g["Xq"])) < 1e-9
        assert rel(m.decision_function(g["Xq"]), m1.decision_function(g["Xq"])) < 1e-9
        if g["task"] == "clf":
            assert np.array_equal(m.predict(g["Xq"]), g["predict"])
            assert np.allclose(m.predict_proba(g["Xq"]), m1.predict_proba(g["Xq"]), atol=1e-9)
        q = m.predict_quantiles(g["Xq"][:50], quantiles=(0.1, 0.5, 0.9))
        q1 = m1.predict_quantiles(g["Xq"][:50], quantiles=(0.1, 0.5, 0.9))
        assert np.allclose(q, q1, rtol=1e-6, atol=1e-8)
    # the dual path does not shard: it runs on the group's first device ("replicas only") and equals the single-device fit exactly
    g = load_golden("dual_reg_n300_d12")
    md = hp.NeoLSSVM(dual=True, devices=[0] * world).fit(g["X"], g["y"])
    m1 = hp.NeoLSSVM(dual=True, device=0).fit(g["X"], g["y"])
    assert np.array_equal(md.α̂_, m1.α̂_) and rel(md.α̂_, g["alpha"]) < TOL
    assert np.array_equal(np.asarray(md.predict_std(g["Xq"])), np.asarray(m1.predict_std(g["Xq"])))


def mode_grid(world):
    import _grid_reference_driver as ref
    from conftest import load_golden

    import neo_ls_svm_amd as hp

    sg = load_golden("sigma_grid_reg_n3000")
    g = load_golden(sg["base"])
    gam = hp.gamma_grid(1024)[::33]
    sig = np.concatenate([sg["sigmas"], [0.7, 1.3, 1.0, 0.55, 1.9]])  # 8 sigmas: several finish, several do not
    solo = hp.Context(0)
    args = (g["X"], g["y"], g["s"], g["shift"], g["scale"])

    def fit(Bs, finish_below):
        return hp.primal_fit(*args, Bs, False, gammas=gam, ctx=solo, finish_below=finish_below)

    want = ref.grid(fit, g["B"], sig, gam)
    got = hp.primal_fit_sigma_grid(*args, g["B"], False, sig, gammas=gam, ctx=solo)
    assert np.array_equal(got["loo_errors"], want["loo_errors"]) and np.array_equal(got["objective"], want["objective"])
    assert (got["sigma_index"], got["gamma_index"], got["finished_count"]) == (want["sigma_index"], want["gamma_index"], want["finished_count"])
    assert rel(got["loo_errors"][:3], sg["loo_errors"]) < 1e-5  # the reference-captured table
    b, wb = got["best"], want["best"]
    assert b is not None and b["opt"] == wb["opt"] == got["gamma_index"]
    iu = np.triu_indices(g["B"].shape[1] + 1)
    for k in ("beta", "loo_residuals", "loo_leverage", "loo_std", "residuals", "loo_errors_gammas", "objective", "lam"):
        assert np.array_equal(b[k], wb[k]), k
    assert np.array_equal(b["L"][iu], wb["L"][iu]) and b["loo_score"] == wb["loo_score"]
    assert 1 <= got["finished_count"] < len(sig)
    # sigma-sharded over "ranks" run in sequence, unmerged: NaN rows for the other ranks' sigmas, best only on the owner
    for rk in range(2):
        p = hp.primal_fit_sigma_grid(*args, g["B"], False, sig, gammas=gam, ctx=solo, rank=rk, world=2)
        w2 = ref.grid(fit, g["B"], sig, gam, rank=rk, world=2)
        assert np.array_equal(np.isnan(p["loo_errors"]), np.isnan(w2["loo_errors"]))
        assert np.array_equal(np.nan_to_num(p["objective"]), np.nan_to_num(w2["objective"]))
        assert (p["sigma_index"], p["gamma_index"]) == (w2["sigma_index"], w2["gamma_index"])
        assert (p["best"] is None) == (w2["best"] is None)
    # the group deals the sigmas over its devices and hands back the winner's full result, wherever it was fitted
    if world > 1:
        grp = hp.Group([0] * world)
        for _ in range(2):
            gg = hp.primal_fit_sigma_grid(*args, g["B"], False, sig, gammas=gam, ctx=grp)
            assert np.array_equal(gg["loo_errors"], want["loo_errors"]) and np.array_equal(gg["objective"], want["objective"])
            assert (gg["sigma_index"], gg["gamma_index"]) == (want["sigma_index"], want["gamma_index"])
            bb = gg["best"]
            assert bb is not None
            for k in ("beta", "loo_residuals", "loo_leverage", "loo_std", "residuals", "loo_errors_gammas", "objective", "lam"):
                assert np.array_equal(bb[k], wb[k]), k
            assert np.array_equal(bb["L"][iu], wb["L"][iu]) and bb["loo_score"] == wb["loo_score"]
        # an exact tie between two sigmas (the same sigma twice): every deployment names the smaller index
        sig_t = np.array([0.6, sig[want["sigma_index"]], 1.7, sig[want["sigma_index"]], 0.9])
        gt = hp.primal_fit_sigma_grid(*args, g["B"], False, sig_t, gammas=gam, ctx=grp)
        g1 = hp.primal_fit_sigma_grid(*args, g["B"], False, sig_t, gammas=gam, ctx=solo)
        assert gt["sigma_index"] == 1 and gt["best"] is not None
        assert np.array_equal(gt["objective"], g1["objective"])
        assert np.array_equal(gt["best"]["beta"], wb["beta"])
        # row sharding afterwards still works on the same group (the contexts were only "solo" for the duration of the grid)
        r = hp.primal_fit(*args, g["B"], False, gammas=gam, ctx=grp)
        r1 = hp.primal_fit(*args, g["B"], False, gammas=gam, ctx=solo)
        assert rel(r["beta"], r1["beta"]) < 1e-8 and r["opt"] == r1["opt"]
        grp.close()
    solo.close()


def mode_big(world):
    import neo_ls_svm_amd as hp

    n, d, D = 24_000, 32, 4096
    X, y, s, shift, scale, B = problem(n=n, d=d, D=D, seed=5)
    grp = hp.Group([0] * world)
    solo = hp.Context(0)
    r1 = hp.primal_fit(X, y, s, shift, scale, B, False, ctx=solo)
    r = hp.primal_fit(X, y, s, shift, scale, B, False, ctx=grp)
    compare_fits(r, r1, D + 1, tol=1e-7)
    kind = grp.evd_stage_ms()["kind"]
    import os

    assert ("two-stage" in kind) == (os.environ.get("NLS_EVD") == "twostage"), kind
    solo.close()
    grp.close()


def mode_fail(world):
    import neo_ls_svm_amd as hp
    from neo_ls_svm_amd._lib import NlsError

    grp = hp.Group([0] * world)
    X, y, s, shift, scale, B = problem()
    try:
        hp.primal_fit(X, y, s, shift, scale, B, False, ctx=grp)
    except NlsError as exc:
        assert "rank" in str(exc) and "roadcast" in str(exc), str(exc)
    else:
        raise AssertionError("the injected ncclBroadcast failure did not surface")
    grp.close()


def _fit_equals_solo(hp, grp, clf=False):
    X, y, s, shift, scale, B = problem(clf=clf)
    solo = hp.Context(0)
    r1 = hp.primal_fit(X, y, s, shift, scale, B, clf, ctx=solo)
    solo.close()
    r = hp.primal_fit(X, y, s, shift, scale, B, clf, ctx=grp)
    compare_fits(r, r1, B.shape[1] + 1)


def mode_fault(world):
    import os
    import time

    import numpy.linalg as npl

    import neo_ls_svm_amd as hp
    from neo_ls_svm_amd._lib import NlsError

    spec = os.environ["NLS_FAULT_INJECT"]
    site, bad, *code = spec.split(":")
    linalg = bool(code) and code[0] == "3"
    grp = hp.Group([0] * world)
    X, y, s, shift, scale, B = problem()
    t0 = time.monotonic()
    try:
        hp.primal_fit(X, y, s, shift, scale, B, False, ctx=grp)
    except (NlsError, npl.LinAlgError) as exc:
        waited, msg = time.monotonic() - t0, str(exc)
        assert waited < 45.0, f"{waited:.1f} s: the deadline, not the vote, ended the call"
        assert isinstance(exc, npl.LinAlgError if linalg else NlsError), (type(exc), msg)
        # the group reports the member that failed of its own accord, with that member's message
        assert f"rank {bad} of {world}" in msg and "injected fault" in msg and f"'{site}'" in msg, msg
    else:
        raise AssertionError(f"the injected fault at {spec} did not surface")
    assert all(c.comm_state == "joined" for c in grp.contexts)  # everybody left at the same vote: the communicator is intact
    del os.environ["NLS_FAULT_INJECT"]
    _fit_equals_solo(hp, grp)
    grp.close()


def mode_lost(world):
    import os
    import time

    import neo_ls_svm_amd as hp
    from neo_ls_svm_amd._lib import NlsError

    bad = int(os.environ["NLS_SHIM_FAIL_RANK"])
    grp = hp.Group([0] * world)
    X, y, s, shift, scale, B = problem()
    t0 = time.monotonic()
    try:
        hp.primal_fit(X, y, s, shift, scale, B, False, ctx=grp)
    except NlsError as exc:
        waited, msg = time.monotonic() - t0, str(exc)
        # NLS_COMM_TIMEOUT_S is 90 s in this test: returning within seconds shows the abort flag, not the deadline, released the members
        assert waited < 45.0, f"{waited:.1f} s"
        assert f"rank {bad} of {world}" in msg and "injected" in msg, msg
    else:
        raise AssertionError("the lost RCCL call did not surface")
    states = [c.comm_state for c in grp.contexts]
    assert states[bad] == "aborted", states
    del os.environ["NLS_SHIM_FAIL_RANK"], os.environ["NLS_SHIM_FAIL_CALL"]
    _fit_equals_solo(hp, grp)  # the group joins its members to a new communicator first
    assert all(c.comm_state == "joined" for c in grp.contexts)
    grp.close()


if __name__ == "__main__":
    mode, world = sys.argv[1], int(sys.argv[2])
    {"fit": mode_fit, "estimator": mode_estimator, "grid": mode_grid, "big": mode_big, "fail": mode_fail, "fault": mode_fault, "lost": mode_lost}[mode](world)
    print("OK", flush=True)
