"""sklearn-compatible ``NeoLSSVM`` whose solver section runs on the MI355X.

Mirrors the reference estimator's public surface (``_neo_ls_svm.py:43-821``): constructor parameters,
``fit / decision_function / predict / predict_std / predict_proba / score`` and the fitted attributes the
reference sets (same names, including the Greek ones, so code written against upstream keeps working).
What differs is *where* the work happens:

    validation, task inference, primal/dual switch        here, as ``_neo_ls_svm.py:335-376``
    supervised affine pre-step + ORF frequency matrix     host NumPy (``_prestep.py``, ``hotpath.orf_frequencies``)
    feature map, Gram, EVD, gamma sweep, Cholesky, sigma  GPU, one C-ABI call (``nls_primal_fit`` / ``nls_dual_fit``)
    inference                                             GPU (``nls_primal_predict`` / ``nls_dual_predict``)
    isotonic calibration, conformal split                 host sklearn, as ``_neo_ls_svm.py:405-441``
    predict_quantiles / predict_interval                  ONE fused GPU predict (yhat + sigma) + the host LP layer ``conformal.py``

Fitted state lives in NumPy attributes (beta, L, shift/scale/A) so the estimator pickles and predicts
without the context that fitted it.
"""

from __future__ import annotations

import time

import numpy as np
from sklearn.base import BaseEstimator, clone
from sklearn.isotonic import IsotonicRegression
from sklearn.metrics import accuracy_score, r2_score
from sklearn.model_selection import train_test_split
from sklearn.utils.validation import check_array, check_consistent_length, check_is_fitted, check_X_y

from . import _prestep, hotpath
from .conformal import conformal_delta_quantiles
from ._lib import Group, default_context, default_group

__all__ = ["NeoLSSVM", "AffineFeatureMap", "AffineNormalizer", "AffineSeparator", "RandomFourierFeatures", "OrthogonalRandomFourierFeatures"]


def _affine_params(fm):
    """(shift, scale, A) of a fitted affine map: the fitted ``*_`` attribute when present, else the constructor parameter
    of the same name - how the reference reads them (``_affine_feature_map.py:49-51,77-79``).  None when ``fm`` is not an
    affine map in that sense."""
    out = []
    for name in ("shift", "scale", "A"):
        if hasattr(fm, name + "_"):
            out.append(getattr(fm, name + "_"))
        elif hasattr(fm, name):
            out.append(getattr(fm, name))
        else:
            return None
    if out[0] is None or out[1] is None:
        return None
    return tuple(out)


class AffineFeatureMap(BaseEstimator):
    """(x - shift) diag(1 / scale) A with GIVEN parameters: reference ``_affine_feature_map.py:17-136``.  ``fit`` validates as the reference
    does (``:52-69``); subclasses learn the parameters.  ``append_features=True`` (``:26-38,90-91``) appends the mapped columns to the given
    ones - honoured by ``transform`` / ``inverse_transform`` / ``get_feature_names_out`` and hence by the dual path (which consumes
    ``transform``'s output); a random-feature map folds (shift, scale, A) into its projection and refuses an appending map."""

    def __init__(self, *, scale, shift, A=None, append_features=False):
        self.scale = scale
        self.shift = shift
        self.A = A
        self.append_features = append_features

    def fit(self, X, y=None, sample_weight=None, ctx=None):
        X = check_array(X, dtype=np.float64)
        self.n_features_in_ = X.shape[1]
        shift, scale, A = _affine_params(self)
        scale, shift = np.reshape(scale, (-1, X.shape[1])), np.reshape(shift, (-1, X.shape[1]))  # incompatible sizes raise here
        if np.any(scale == 0) or not np.all(np.isfinite(scale)):
            raise ValueError("The scale must be finite and non-zero")
        if not np.all(np.isfinite(shift)):
            raise ValueError("The shift must be finite")
        if A is not None and (np.shape(A)[0] != X.shape[1] or not np.all(np.isfinite(A))):
            raise ValueError("The matrix A must be finite with rows equal to the number of features in X")
        return self

    def _appends(self, A):
        return bool(getattr(self, "append_features", False)) and A is not None

    def transform(self, X):
        """Affine map only (used by the dual path: ``_neo_ls_svm.py:394,668``); n x r output, host GEMM, with the
        reference's memory-order switch (``_affine_feature_map.py:81-89``); ``append_features``: [X, mapped X] (``:90-91``)."""
        X = check_array(X, dtype=np.float64)
        shift, scale, A = _affine_params(self)
        shift, scale = np.reshape(shift, (1, -1)), np.reshape(scale, (1, -1))
        if A is None:
            return (X - shift) / scale
        As = A / scale.T
        out = X @ As - shift @ As if A.shape[1] < A.shape[0] else (X - shift) @ As
        return np.hstack((X, out)) if self._appends(A) else out

    def inverse_transform(self, X_transformed):
        """Approximate inverse (``_affine_feature_map.py:101-114``): the given columns themselves when they were appended, else the
        pseudo-inverse of A undone, then scale and shift."""
        Xt = check_array(X_transformed, dtype=np.float64)
        shift, scale, A = _affine_params(self)
        if self._appends(A):
            return Xt[:, : np.shape(A)[0]].copy()
        if A is not None:
            Xt = Xt @ np.linalg.pinv(np.asarray(A, dtype=np.float64))
        return Xt * np.reshape(scale, (1, -1)) + np.reshape(shift, (1, -1))

    def get_feature_names_out(self, input_features=None):
        """Output column names (``_affine_feature_map.py:116-133``): ``<name>_shifted_scaled`` without a matrix, else one shared
        ``<all names>_affine_map`` label per mapped column; the input names first when they are appended."""
        from sklearn.utils.validation import _check_feature_names_in

        _, _, A = _affine_params(self)
        names = np.asarray(_check_feature_names_in(self, input_features), dtype=object)
        if A is None:
            return np.asarray([f"{n}_shifted_scaled" for n in names], dtype=object)
        mapped = np.asarray([f"{','.join(names)}_affine_map"] * np.shape(A)[1], dtype=object)
        return np.hstack((names, mapped)) if self._appends(A) else mapped


class AffineNormalizer(AffineFeatureMap):
    """Supervised shift / scale from per-bin weighted medians and deviations, A = None: reference ``_affine_normalizer.py:25-117``
    (the n-proportional bin statistics run on the GPU, ``nls_bin_stats``)."""

    def __init__(self, *, append_features=False, device=0):
        self.append_features = append_features  # (A is None for a normaliser: nothing is ever appended, as upstream)
        self.device = device

    def fit(self, X, y, sample_weight=None, ctx=None):
        X, y = check_X_y(X, y, dtype=np.float64)
        ctx = ctx or default_context(int(self.device))
        self.shift_, self.scale_ = _prestep.fit_affine_normalizer(
            X, np.asarray(y, dtype=np.float64), sample_weight, stats=lambda A, lab, w: hotpath.bin_stats(A, lab, w, ctx=ctx),
            unique=lambda t: hotpath.rank_codes(t, ctx=ctx),
        )
        self.A_ = None
        self.n_features_in_ = X.shape[1]
        return self


class AffineSeparator(AffineFeatureMap):
    """(x - shift) diag(1/scale) A with supervised shift/scale/A: reference ``_affine_separator.py:54-210``."""

    def __init__(self, *, append_features=False, rank_threshold=2e-2, edge_sample_size=384, edge_search_multiplier=4, random_state=42, device=0):
        self.append_features = append_features
        self.rank_threshold = rank_threshold
        self.edge_sample_size = edge_sample_size
        self.edge_search_multiplier = edge_search_multiplier
        self.random_state = random_state
        self.device = device

    def fit(self, X, y, sample_weight=None, ctx=None, _validated=False):
        """``ctx``: the context (hence GPU) that runs the bin statistics; default: the context of ``self.device``.
        ``NeoLSSVM.fit`` hands its own context down so that X is uploaded once, to the estimator's device, and - having validated X, y
        itself - sets the private ``_validated`` (a second finiteness pass over a 1 GB matrix is 40 ms); every other caller is validated here."""
        if not (_validated and isinstance(X, np.ndarray) and X.dtype == np.float64 and X.ndim == 2 and X.flags.c_contiguous):
            X, y = check_X_y(X, y, dtype=np.float64)
        ctx = ctx or default_context(int(self.device))

        def unique(t):  # the quantiser's np.unique(y, return_inverse=True) as a radix sort on the GPU (nls_rank_codes)
            return hotpath.rank_codes(t, ctx=ctx)

        def normalizer(Xa, ya, swa):  # per-bin weighted medians / deviations on the GPU (nls_bin_stats_labels)
            return _prestep.fit_affine_normalizer(Xa, ya, swa, stats=lambda A, lab, w: hotpath.bin_stats(A, lab, w, ctx=ctx), unique=unique)

        self.shift_, self.scale_, self.A_ = _prestep.fit_affine_separator(
            X,
            y,
            sample_weight,
            normalizer=normalizer,
            unique=unique,
            rank_threshold=self.rank_threshold,
            edge_sample_size=self.edge_sample_size,
            edge_search_multiplier=self.edge_search_multiplier,
            random_state=self.random_state,
        )
        self.n_features_in_ = X.shape[1]
        return self


def _fit_affine(fm, X, y, sample_weight, ctx, _validated=False):
    """Fit a (possibly caller-supplied) affine map; this package's classes take the context so that X is uploaded once.
    The fitted object must expose (shift, scale, A) - as parameters or fitted attributes, the reference's convention
    (``_affine_feature_map.py:49-51``) - or it is not an affine map this library can fold into its feature map."""
    if isinstance(fm, AffineSeparator):
        fitted = fm.fit(X, y, sample_weight, ctx=ctx, _validated=_validated)
    else:
        fitted = fm.fit(X, y, sample_weight, ctx=ctx) if isinstance(fm, AffineFeatureMap) else fm.fit(X, y, sample_weight)
    fitted = fm if fitted is None else fitted
    if _affine_params(fitted) is None:
        raise TypeError(f"{type(fm).__name__} is not an affine map: after fit it exposes no shift / scale / A (parameters or fitted attributes)")
    return fitted


class RandomFourierFeatures(BaseEstimator):
    """phi(x) = [exp(-i Z^T A^T ((x - shift)/scale)) / sqrt(D), 1]: reference ``_feature_maps.py:117-203``.

    ``fit`` fits the affine map (default: ``AffineSeparator()``, ``_feature_maps.py:66``; any affine map is honoured - this
    package's ``AffineFeatureMap`` / ``AffineNormalizer`` / ``AffineSeparator``, upstream's own, or a caller's: ``fit(X, y,
    sample_weight)`` and shift / scale / A as parameters or fitted attributes) and folds the frequency matrix Z into its A (``:143-150``); ``transform`` evaluates the map on the GPU
    (``nls_featuremap``).  ``orthogonal=False``: Z = ``RandomState(seed).randn(d', D)`` (``:120-127``); ``orthogonal=True``: each
    block of d' columns replaced by the Q of its QR and the columns rescaled by chi(d') draws (``:209-223``).
    """

    orthogonal_default = False

    def __init__(self, affine_feature_map=None, num_features=512, random_state=42, exact_complexity=False, orthogonal=None):
        self.affine_feature_map = affine_feature_map
        self.num_features = num_features
        self.random_state = random_state
        self.exact_complexity = exact_complexity
        self.orthogonal = orthogonal

    def fit(self, X, y=None, sample_weight=None, ctx=None, _validated=False):
        afm = _as_own_affine_map(self.affine_feature_map)
        self.affine_feature_map_ = _fit_affine(AffineSeparator() if afm is None else clone(afm), X, y, sample_weight, ctx, _validated)
        shift, scale, A = _affine_params(self.affine_feature_map_)
        if A is not None and getattr(self.affine_feature_map_, "append_features", False):
            raise TypeError("an affine map with append_features=True cannot sit inside a random-feature map: the map folds (shift, scale, A) into its "
                            "projection B = A Z (`_feature_maps.py:147-150`) and has no place for the appended columns")
        A = None if A is None else np.asarray(A, dtype=np.float64)
        d_in = A.shape[1] if A is not None else np.asarray(X).shape[1]
        orthogonal = self.orthogonal_default if self.orthogonal is None else bool(self.orthogonal)
        if orthogonal:
            self.Z_ = hotpath.orf_frequencies(d_in, self.num_features, self.random_state)
        else:
            gen = self.random_state if isinstance(self.random_state, np.random.RandomState) else np.random.RandomState(self.random_state)
            self.Z_ = gen.randn(d_in, self.num_features)
        # The fold A Z is a small product (d x r by r x D).  On the BLAS's default thread count (64 on the GPU box) its worker threads keep
        # spinning for tens of milliseconds after the call, and the solver call that follows - a host thread that launches thousands of
        # kernels and waits on the stream a handful of times - took 84 instead of 41 ms at n = 1e5 (profiles/r05_c2_idle.md: not the idle GPU).
        with _prestep.blas_threads(8):
            self.B_ = A @ self.Z_ if A is not None else self.Z_
        self.shift_, self.scale_ = np.ravel(np.asarray(shift, dtype=np.float64)), np.ravel(np.asarray(scale, dtype=np.float64))
        self.n_features_in_ = np.asarray(X).shape[1]
        return self

    @property
    def map_params(self):
        return self.shift_, self.scale_, self.B_

    @property
    def complexity_matrix(self):
        """Identity - the reference's fast diagonal approximation, the only one it reaches (``_feature_maps.py:129-135``) -
        or, with ``exact_complexity=True``, the exact matrix of ``_ztz_prod_sinc_zmz``'s slow branch (``:46-55``), which
        sends the solver down the generalised-EVD branch (``_neo_ls_svm.py:122-124``)."""
        if self.exact_complexity:
            return hotpath.exact_complexity_matrix(self.Z_)
        return np.eye(self.num_features + 1)

    def transform(self, X, ctx=None):
        shift, scale, B = self.map_params
        return hotpath.featuremap(check_array(X, dtype=np.float64), shift, scale, B, ctx=ctx)


class OrthogonalRandomFourierFeatures(RandomFourierFeatures):
    """Orthogonal random Fourier features (the default primal map): reference ``_feature_maps.py:206-223``."""

    orthogonal_default = True


def _as_own_feature_map(fm):
    """The primal plug-in point (``_neo_ls_svm.py:62-75,380-394``).  Honoured: this package's ``RandomFourierFeatures`` /
    ``OrthogonalRandomFourierFeatures`` (with any affine map inside), and upstream's two classes of the same names, which are
    translated by their public parameters (``num_features``, ``random_state``, ``affine_feature_map``).  Anything else - a
    feature map whose transform this library does not implement - is refused: silently fitting a different model is worse."""
    if isinstance(fm, RandomFourierFeatures):
        return fm
    name = type(fm).__name__
    if name in ("RandomFourierFeatures", "OrthogonalRandomFourierFeatures") and hasattr(fm, "num_features"):
        cls = OrthogonalRandomFourierFeatures if name.startswith("Orthogonal") else RandomFourierFeatures
        afm = getattr(fm, "affine_feature_map", None)
        return cls(affine_feature_map=_as_own_affine_map(afm), num_features=int(fm.num_features), random_state=getattr(fm, "random_state", 42))
    raise TypeError(
        f"primal_feature_map must be 'auto', a RandomFourierFeatures / OrthogonalRandomFourierFeatures of neo_ls_svm_amd (or upstream's "
        f"classes of those names); got {name}: this library evaluates exp(-i T) / sqrt(D) maps only and will not substitute another model"
    )


def _as_own_affine_map(afm):
    """Upstream's ``AffineSeparator`` / ``AffineNormalizer`` carry numba-jitted pre-steps this package restates itself: they are
    translated by their public parameters; every other affine map (upstream's fixed ``AffineFeatureMap``, a caller's own) is
    passed through and fitted as is."""
    if afm is None or isinstance(afm, AffineFeatureMap):
        return afm
    name, module = type(afm).__name__, type(afm).__module__
    if module.startswith("neo_ls_svm.") and name == "AffineSeparator":
        keys = ("append_features", "rank_threshold", "edge_sample_size", "edge_search_multiplier", "random_state")
        return AffineSeparator(**{k: getattr(afm, k) for k in keys if hasattr(afm, k)})
    if module.startswith("neo_ls_svm.") and name == "AffineNormalizer":
        return AffineNormalizer(append_features=bool(getattr(afm, "append_features", False)))
    if callable(getattr(afm, "fit", None)):
        return afm  # checked for shift / scale / A once fitted (_fit_affine)
    raise TypeError(f"{name} is not an affine map: it has no fit(X, y, sample_weight)")


def _series_like(values, X_in):
    if hasattr(X_in, "dtypes") and hasattr(X_in, "index"):
        try:
            import pandas as pd
        except ImportError:
            return values
        return pd.Series(values, index=X_in.index)
    return values


class NeoLSSVM(BaseEstimator):
    """Neo LS-SVM with the fit/predict hot path on an MI355X (drop-in for ``neo_ls_svm.NeoLSSVM``)."""

    def __init__(
        self,
        *,
        primal_feature_map="auto",
        dual_feature_map="auto",
        dual="auto",
        estimator_type="auto",
        random_state=42,
        device=0,
        devices=None,
        release_workspace=False,
    ):
        self.primal_feature_map = primal_feature_map
        self.dual_feature_map = dual_feature_map
        self.dual = dual
        self.random_state = random_state
        self.estimator_type = estimator_type
        self.device = device
        self.devices = devices
        self.release_workspace = release_workspace

    # ---- sklearn plumbing -------------------------------------------------------------------
    def __sklearn_tags__(self):
        tags = super().__sklearn_tags__()
        tags.target_tags.required = True
        kind = getattr(self, "_estimator_type", None) or (None if self.estimator_type == "auto" else self.estimator_type)
        if kind == "classifier":
            from sklearn.utils import ClassifierTags

            tags.estimator_type = "classifier"
            tags.classifier_tags = ClassifierTags(multi_class=False)
        elif kind == "regressor":
            from sklearn.utils import RegressorTags

            tags.estimator_type = "regressor"
            tags.regressor_tags = RegressorTags()
        return tags

    def _ctx(self):
        """Where this estimator computes: the context of ``device``, or - ``devices=[...]`` with more than one GPU - the process-wide
        ``Group`` of those devices: the primal fit, ``decision_function`` and ``predict_std`` then shard their rows over the GPUs inside ONE
        library call each (the surface is identical at 1 and 8 GPUs); the pre-step statistics and the dual path (its n x n
        eigendecomposition does not shard: "replicas only") run on the first device."""
        devs = getattr(self, "devices", None)
        if devs is not None and len(devs) > 1:
            return default_group(devs)
        return default_context(int(devs[0] if devs else self.device))

    @staticmethod
    def _ctx0(ctx):
        return ctx.contexts[0] if isinstance(ctx, Group) else ctx

    # The inverse Cholesky factor predict_std needs lives on the device between calls as an explicit handle owned by
    # this estimator: created on the first predict_std after a fit, dropped on refit, never pickled.
    def _drop_factor(self):
        f = self.__dict__.pop("_factor", None)
        if f is not None:
            f.close()

    def _factor_for(self, ctx):
        f = self.__dict__.get("_factor")
        if f is None or f.ctx is not ctx or not f.handle:
            self._drop_factor()
            f = self.__dict__["_factor"] = (hotpath.GroupFactor if isinstance(ctx, Group) else hotpath.Factor)(ctx, self.L_[0])
        return f

    def __getstate__(self):
        state = super().__getstate__() if hasattr(super(), "__getstate__") else self.__dict__.copy()
        state = dict(state)
        state.pop("_factor", None)
        return state

    # ---- fit --------------------------------------------------------------------------------
    def fit(self, X, y, sample_weight=None):
        """Fit this predictor (reference ``fit``: ``_neo_ls_svm.py:327-442``)."""
        X, y = check_X_y(X, y, dtype=(np.float64, np.float32), ensure_min_samples=2)
        X = np.ascontiguousarray(X, dtype=np.float64)  # the GPU path computes in float64 throughout
        y = np.ravel(np.asarray(y))
        self.n_features_in_ = X.shape[1]
        self.y_dtype_ = y.dtype
        sw = np.ones(y.shape, np.float64) if sample_weight is None else np.ravel(np.asarray(sample_weight)).astype(np.float64)
        check_consistent_length(y, sw)
        # Task inference (:351-373).
        unique_y = np.unique(y)
        inferred = None
        if len(unique_y) == 2:  # noqa: PLR2004
            inferred = "classifier"
        elif np.issubdtype(y.dtype, np.number) or np.issubdtype(y.dtype, np.datetime64) or np.issubdtype(y.dtype, np.timedelta64):
            inferred = "regressor"
        self._estimator_type = inferred if self.estimator_type == "auto" else self.estimator_type
        if self._estimator_type == "classifier":
            self.classes_ = unique_y
            y_ = np.ones(y.shape, dtype=np.float64)
            y_[y == self.classes_[0]] = -1
        elif self._estimator_type == "regressor":
            y_ = y.astype(np.float64)
        else:
            raise ValueError("Target type not supported")
        is_clf = self._estimator_type == "classifier"
        self._drop_factor()  # the device copy of a previous fit's inverse factor
        self.dual_ = bool(X.shape[0] <= 1024 if self.dual == "auto" else self.dual)  # noqa: PLR2004
        self.primal_ = not self.dual_
        ctx = self._ctx()
        wall = {}
        t0 = time.perf_counter()
        if self.primal_:
            fm = OrthogonalRandomFourierFeatures() if isinstance(self.primal_feature_map, str) else _as_own_feature_map(self.primal_feature_map)
            with ctx.hold(X):  # one upload of X serves the pre-step's bin statistics and the solver (a group: every rank uploads its own row block)
                self.primal_feature_map_ = clone(fm).fit(X, y_, sw, ctx=self._ctx0(ctx), _validated=True)
                shift, scale, B = self.primal_feature_map_.map_params
                Cm = self.primal_feature_map_.complexity_matrix if self.primal_feature_map_.exact_complexity else None
                wall["prestep"] = time.perf_counter() - t0
                t0 = time.perf_counter()
                r = hotpath.primal_fit(X, y_, sw, shift, scale, B, is_clf, ctx=ctx, complexity_matrix=Cm)
            self.β̂_, self.γ_ = r["beta"], r["gamma"]
            self.loo_leverage_ = r["loo_leverage"]
        else:
            nz = sw > 0
            X, y_, sw = X[nz], y_[nz], sw[nz]
            sep = AffineSeparator() if isinstance(self.dual_feature_map, str) else _as_own_affine_map(self.dual_feature_map)
            if sep is None:
                raise TypeError("dual_feature_map must be 'auto' or an affine map")
            self.dual_feature_map_ = _fit_affine(clone(sep), X, y_, sw, self._ctx0(ctx), _validated=True)
            self.X_ = np.ascontiguousarray(self._dual_transform(X))
            wall["prestep"] = time.perf_counter() - t0
            t0 = time.perf_counter()
            r = hotpath.dual_fit(self.X_, y_, sw, is_clf, ctx=self._ctx0(ctx))
            self.α̂_, self.γ_ = r["alpha"], r["gamma"]
        wall["solver"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        # Attributes the reference's solver sets as side effects (:146-187 / :270-323).
        self.γs_ = r["gammas"]
        self.loo_errors_γs_ = r["loo_errors_gammas"]
        self.loo_residuals_ = r["loo_residuals"]
        self.loo_ŷ_ = y_ + r["loo_residuals"]
        self.loo_error_ = r["loo_error"]
        self.loo_score_ = r["loo_score"]
        self.L_ = (r["L"], False)
        self.residuals_ = r["residuals"]
        self.loo_std_ = r["loo_std"]
        self.fit_timings_ = r["timings"]
        # The context's workspace is a caching arena (it only grows between calls: ~100 GB after a c3-size fit, so that the
        # next fit allocates nothing).  release_workspace=True hands the large fit-only buffers back at the end of every fit
        # for processes that share the GPU - at a price: freeing and re-allocating ~180 GB costs ~3 s per c3-size fit
        # (profiles/r02_profile_fit.log), which is why it is opt-in; Context.release_workspace() does the same on demand.
        if self.release_workspace:
            ctx.release_workspace(min_bytes=256 << 20)
        # Isotonic probability calibration on the LOO predictions (:406-412).
        if is_clf:
            self.predict_proba_calibrator_ = IsotonicRegression(out_of_bounds="clip", y_min=0, y_max=1, increasing=True)
            target = np.zeros_like(y_)
            target[y_ == np.max(y_)] = 1.0
            self.predict_proba_calibrator_.fit(self.loo_ŷ_, target, sw)
        # Two-level conformal calibration split of the LOO quantities (:414-430).
        (
            self.nonconformity_calib_l1_,
            self.nonconformity_calib_l2_,
            self.ŷ_calib_l1_,
            self.ŷ_calib_l2_,
            self.residuals_calib_l1_,
            self.residuals_calib_l2_,
            self.sample_weight_calib_l1_,
            self.sample_weight_calib_l2_,
        ) = train_test_split(
            self.loo_std_,
            self.loo_ŷ_,
            self.loo_residuals_,
            sw,
            train_size=min(1440, max(1024, (X.shape[0] * 2) // 3), X.shape[0] - 1),
            random_state=self.random_state,
        )
        # Conformal predictors are fitted lazily per requested quantile tuple (:431-441).
        self.conformal_l1_ = {"Δŷ": {}, "Δŷ/ŷ": {}}
        self.conformal_l2_ = {"Δŷ": {}, "Δŷ/ŷ": {}}
        wall["calibration"] = time.perf_counter() - t0
        self.fit_wall_ = wall  # seconds: pre-step (affine map + frequency matrix) / solver call / attributes + calibration split
        return self

    def _dual_transform(self, Xa):
        """X -> X_ of the dual path (``_neo_ls_svm.py:394,668``) through the fitted affine map (its own ``transform`` when it has one)."""
        fm = self.dual_feature_map_
        if callable(getattr(fm, "transform", None)):
            return np.asarray(fm.transform(Xa), dtype=np.float64)
        shift, scale, A = _affine_params(fm)
        return AffineFeatureMap(scale=scale, shift=shift, A=A).transform(Xa)

    # ASCII aliases of the Greek attribute names.
    @property
    def beta_(self):
        return self.β̂_

    @property
    def alpha_(self):
        return self.α̂_

    @property
    def gamma_(self):
        return self.γ_

    # ---- inference ----------------------------------------------------------------------------
    def _check_X(self, X):
        check_is_fitted(self)
        Xa = check_array(X, dtype=(np.float64, np.float32))
        if Xa.shape[1] != self.n_features_in_:
            raise ValueError(f"X has {Xa.shape[1]} features, but NeoLSSVM is expecting {self.n_features_in_} features as input")
        return np.ascontiguousarray(Xa, dtype=np.float64)

    def decision_function(self, X):
        """yhat(X): primal Re(phi(X) beta), dual k(X, X_) alpha + 1'alpha (``_neo_ls_svm.py:655-681``)."""
        Xa = self._check_X(X)
        if self.primal_:
            shift, scale, B = self.primal_feature_map_.map_params
            yhat, _ = hotpath.primal_predict(Xa, shift, scale, B, beta=self.β̂_, ctx=self._ctx())
        else:
            Xq = np.ascontiguousarray(self._dual_transform(Xa))
            yhat, _ = hotpath.dual_predict(Xq, self.X_, alpha=self.α̂_, ctx=self._ctx0(self._ctx()))
        return _series_like(yhat, X)

    def predict_std(self, X):
        """Bayesian predictive standard deviation from the stored Cholesky factor (``:452-487``)."""
        Xa = self._check_X(X)
        if self.primal_:
            shift, scale, B = self.primal_feature_map_.map_params
            ctx = self._ctx()
            _, sigma = hotpath.primal_predict(Xa, shift, scale, B, ctx=ctx, factor=self._factor_for(ctx))
        else:
            Xq = np.ascontiguousarray(self._dual_transform(Xa))
            _, sigma = hotpath.dual_predict(Xq, self.X_, L=self.L_[0], ctx=self._ctx0(self._ctx()))
        return _series_like(sigma, X)

    def _yhat_sigma(self, Xa):
        """decision_function and predict_std from ONE pass over X (one feature-map evaluation)."""
        if self.primal_:
            shift, scale, B = self.primal_feature_map_.map_params
            ctx = self._ctx()
            return hotpath.primal_predict(Xa, shift, scale, B, beta=self.β̂_, ctx=ctx, factor=self._factor_for(ctx))
        Xq = np.ascontiguousarray(self._dual_transform(Xa))
        return hotpath.dual_predict(Xq, self.X_, alpha=self.α̂_, L=self.L_[0], ctx=self._ctx0(self._ctx()))

    def predict_quantiles(self, X, *, quantiles=(0.025, 0.5, 0.975), priority="accuracy"):
        """Conformally calibrated quantiles (``:554-624``): [m x q] for a regressor, [m x q x 2] class probabilities
        for a classifier; a DataFrame when X is one."""
        Xa = self._check_X(X)
        yhat, sigma = self._yhat_sigma(Xa)
        delta = conformal_delta_quantiles(self, yhat, sigma, quantiles, priority)
        out = yhat[:, None] + delta
        is_clf = self._estimator_type == "classifier"
        if is_clf:
            pos = np.hstack([self.predict_proba_calibrator_.transform(out[:, j])[:, None] for j in range(out.shape[1])])
            out = np.dstack([1 - pos[:, ::-1], pos])
        elif not np.issubdtype(self.y_dtype_, np.integer):
            out = out.astype(self.y_dtype_)
        if hasattr(X, "dtypes") and hasattr(X, "index"):
            import pandas as pd

            if is_clf:
                neg = pd.DataFrame(out[:, :, 0], index=X.index, columns=quantiles)
                posd = pd.DataFrame(out[:, :, 1], index=X.index, columns=quantiles)
                df = pd.concat([neg, posd], axis=0, keys=self.classes_, names=["class", X.index.name])
            else:
                df = pd.DataFrame(out, index=X.index, columns=quantiles)
            df.columns.name = "quantile"
            return df
        return out

    def predict_interval(self, X, *, coverage=0.95):
        """Conformally calibrated central interval (``:636-646``): the (1-c)/2 and 1-(1-c)/2 quantiles, coverage first."""
        lb = (1 - coverage) / 2
        return self.predict_quantiles(X, quantiles=(lb, 1 - lb), priority="coverage")

    def predict(self, X, *, coverage=None, quantiles=None):
        """Point predictions, or an interval / quantiles when asked (``:719-762``)."""
        assert coverage is None or quantiles is None
        if coverage is not None:
            return self.predict_interval(X, coverage=coverage)
        if quantiles is not None:
            return self.predict_quantiles(X, quantiles=quantiles)
        yhat = np.asarray(self.decision_function(X))
        if self._estimator_type == "classifier":
            sgn = np.sign(yhat)
            sgn[sgn == 0] = -1
            out = self.classes_[((sgn + 1) // 2).astype(np.intp)]
        else:
            out = yhat
        if not np.issubdtype(self.y_dtype_, np.integer):
            out = out.astype(self.y_dtype_)
        return _series_like(out, X)

    def predict_proba(self, X):
        """Isotonically calibrated class probabilities (classifier) or the prediction (regressor): ``:772-799``."""
        yhat = np.asarray(self.decision_function(X))
        if self._estimator_type == "classifier":
            pos = self.predict_proba_calibrator_.transform(yhat)
            proba = np.hstack([1 - pos[:, None], pos[:, None]])
            if hasattr(X, "dtypes") and hasattr(X, "index"):
                import pandas as pd

                return pd.DataFrame(proba, index=X.index, columns=self.classes_)
            return proba
        out = yhat if np.issubdtype(self.y_dtype_, np.integer) else yhat.astype(self.y_dtype_)
        return _series_like(out, X)

    def score(self, X, y, sample_weight=None):
        """Accuracy or R^2 (``:801-817``)."""
        yhat = np.asarray(self.predict(X))
        if self._estimator_type == "classifier":
            return accuracy_score(y, yhat, sample_weight=sample_weight)
        return r2_score(np.asarray(y).astype(np.float64), yhat.astype(np.float64), sample_weight=sample_weight)
