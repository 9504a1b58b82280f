"""Decoder half of DiffVC's `model` package: `from model.diffusion import Diffusion, GradLogPEstimator`."""
from .diffusion import Diffusion, GradLogPEstimator  # noqa: F401
