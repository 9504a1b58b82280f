"""DiffVC's `model` package: `from model import DiffVC` (DiffVC/inference.ipynb), `from model.diffusion import Diffusion`."""
from .diffusion import Diffusion, GradLogPEstimator  # noqa: F401
from .vc import DiffVC, FwdDiffusion  # noqa: F401
