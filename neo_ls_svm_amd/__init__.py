"""MI355X-native implementation of the neo-ls-svm fit/predict hot path.

Host code is Python + NumPy marshalling over a ctypes C ABI (``include/neolssvm_hip.h``) into
hand-written HIP kernels for gfx950 (feature map, Gram, eigendecompositions, sweeps, Cholesky factorisations); rocBLAS serves the blocked
back-transformations and rocSOLVER one triangular inverse behind ``predict_std``.  No CPU fallback.
"""

from ._lib import Context, DeviceArray, Factor, Group, GroupFactor, NlsError, default_context, default_group, load_library, set_default_context  # noqa: F401
from ._hostpool import pin_large_outputs  # noqa: F401
from ._hostpool import reserve as reserve_factor_outputs  # noqa: F401
from .hotpath import (  # noqa: F401
    dual_fit,
    dual_predict,
    eigh,
    exact_complexity_matrix,
    featuremap,
    gamma_grid,
    gram,
    orf_frequencies,
    primal_fit,
    primal_fit_sharded,
    primal_fit_sigma_grid,
    primal_predict,
    rotate,
    stedc,
    tridiagonalize,
    cholesky,
    twostage_stage,
)

from .estimator import (  # noqa: E402,F401
    AffineFeatureMap,
    AffineNormalizer,
    AffineSeparator,
    NeoLSSVM,
    OrthogonalRandomFourierFeatures,
    RandomFourierFeatures,
)

__all__ = [
    "NeoLSSVM",
    "OrthogonalRandomFourierFeatures",
    "RandomFourierFeatures",
    "AffineSeparator",
    "AffineNormalizer",
    "AffineFeatureMap",
    "set_default_context",
    "pin_large_outputs",
    "reserve_factor_outputs",
    "Context",
    "Group",
    "GroupFactor",
    "default_group",
    "DeviceArray",
    "Factor",
    "NlsError",
    "default_context",
    "load_library",
    "featuremap",
    "gram",
    "primal_fit",
    "primal_fit_sharded",
    "primal_fit_sigma_grid",
    "primal_predict",
    "dual_fit",
    "dual_predict",
    "gamma_grid",
    "orf_frequencies",
    "exact_complexity_matrix",
    "eigh",
    "tridiagonalize",
    "stedc",
    "cholesky",
    "twostage_stage",
]
