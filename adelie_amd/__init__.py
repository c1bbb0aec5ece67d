"""adelie_amd — an MI355X-native group-elastic-net path solver behind adelie's Python API.

Mirrors the user-facing surface of JamesYang007/adelie for the ``grpnet`` hot path only
(``solver.grpnet``, ``cv.cv_grpnet``, ``matrix.dense`` / ``matrix.snp_unphased`` / ``matrix.kronecker_eye``,
``glm.gaussian`` / ``glm.binomial`` / ``glm.multigaussian`` / ``glm.multinomial``, the naive State objects).  All numerics run in hand-written HIP
kernels for gfx950 behind the C ABI declared in ``include/adelie_hip.h``.
"""
from . import configs
from . import constraint
from . import glm
from . import matrix
from . import state
from . import solver
from . import diagnostic
from . import cv
from . import data
from . import io
try:  # the estimator wrapper needs scikit-learn's base classes; everything else does not
    from . import sklearn
except ImportError:  # pragma: no cover
    pass
from .configs import set_configs
from .cv import cv_grpnet
from .solver import gaussian_cov, grpnet

__version__ = "0.1.0"
