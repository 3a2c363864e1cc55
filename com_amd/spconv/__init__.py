"""Drop-in for the ``spconv`` names the reference imports through
``pcdet/utils/spconv_utils.py:3-6`` (``import spconv.pytorch as spconv``): every name the reference's
model files use (SURVEY.md section 8b "Import seam 1") is exported here, backed by the gfx950 HIP
library instead of spconv's CUDA kernels.
"""
from . import conv, utils  # noqa: F401
from .conv import SparseConv3d, SparseConvolution, SparseInverseConv3d, SubMConv3d  # noqa: F401
from .core import SparseConvTensor  # noqa: F401
from .modules import SparseModule, SparseSequential  # noqa: F401

import sys as _sys

pytorch = _sys.modules[__name__]  # `spconv.pytorch` spelling of spconv 2.x
