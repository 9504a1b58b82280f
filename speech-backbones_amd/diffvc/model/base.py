"""BaseModule of DiffVC (DiffVC/model/base.py) -- identical surface to Grad-TTS's."""
from ...model.base import BaseModule  # noqa: F401
