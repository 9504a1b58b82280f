"""speech-backbones_amd -- MI355X-native Grad-TTS / DiffVC diffusion-decoder sampling path.

Import by name (the directory name carries a hyphen):  importlib.import_module("speech-backbones_amd").
The drop-in `model` package (same class names / signatures / state_dict as Grad-TTS/model) lives in
``speech-backbones_amd/model``; put ``speech-backbones_amd`` on sys.path to let the reference's own
``inference.py`` / ``train.py`` pick it up as ``from model import GradTTS``.
"""
from . import _lib  # noqa: F401
from ._lib import Plan, Vocoder, Encoder, PostNetPlan, PREC_BF16, PREC_BF16_STORE, PREC_BF16X3, PREC_F16F8, RangeError, euler_step, mas_maximum_path  # noqa: F401

__all__ = ["Plan", "Vocoder", "Encoder", "PostNetPlan", "PREC_BF16", "PREC_BF16_STORE", "PREC_BF16X3", "PREC_F16F8", "RangeError", "euler_step", "mas_maximum_path"]
