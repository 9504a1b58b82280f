"""Drop-in for Grad-TTS/hifi-gan/ (the vocoder Grad-TTS/inference.py:55-61,81 runs right after the decoder):
`env.AttrDict`, `models.Generator`, `xutils.{init_weights, get_padding}`.  With this directory on sys.path in place of
`./hifi-gan/`, `from env import AttrDict; from models import Generator as HiFiGAN` work unchanged."""
