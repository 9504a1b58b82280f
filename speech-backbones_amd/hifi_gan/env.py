"""env.AttrDict -- Grad-TTS/hifi-gan/env.py:7-10 (a dict whose keys are attributes; holds hifigan-config.json)."""


class AttrDict(dict):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.__dict__ = self
