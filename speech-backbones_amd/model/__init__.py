"""Drop-in `model` package: same import surface as Grad-TTS/model/__init__.py (`from model import GradTTS`)."""
from .tts import GradTTS  # noqa: F401
