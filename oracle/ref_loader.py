"""TEST INFRASTRUCTURE ONLY.  Import the reference's own Python (read-only) when /root/reference is mounted.

Used by tests/golden/make_golden.py and tests/test_oracle_vs_reference.py to pin the restatement.
Never used on the GPU box (the reference tree does not exist there) and never by product code.
"""
import os
import sys
import types

REF_ROOT = "/root/reference"


def available():
    return os.path.isdir(os.path.join(REF_ROOT, "Grad-TTS", "model"))


def _purge(prefix):
    saved = {}
    for k in [k for k in sys.modules if k == prefix or k.startswith(prefix + ".")]:
        saved[k] = sys.modules.pop(k)
    return saved


def load_gradtts():
    """Return the reference `model` package of Grad-TTS (tts.py, diffusion.py, ...).

    `model.monotonic_align` expects an in-place Cython build next to the read-only sources
    (monotonic_align/__init__.py:5); a shim module backed by oracle/_ref is registered instead."""
    if not available():
        raise RuntimeError("/root/reference is not mounted")
    saved = _purge("model")
    from . import mas
    shim = types.ModuleType("model.monotonic_align")
    shim.maximum_path = mas.maximum_path_ref
    path = os.path.join(REF_ROOT, "Grad-TTS")
    sys.path.insert(0, path)
    try:
        sys.modules["model.monotonic_align"] = shim
        import model  # noqa: F401  (the reference package)
        import model.diffusion  # noqa: F401
        import model.text_encoder  # noqa: F401
        mods = {k: v for k, v in sys.modules.items() if k == "model" or k.startswith("model.")}
    finally:
        sys.path.remove(path)
        _purge("model")          # keep the name `model` free for the product's drop-in package
        sys.modules.update(saved)
    return types.SimpleNamespace(pkg=mods["model"], GradTTS=mods["model.tts"].GradTTS,
                                 diffusion=mods["model.diffusion"], tts=mods["model.tts"],
                                 utils=mods["model.utils"], text_encoder=mods["model.text_encoder"])


def load_diffvc():
    """Return DiffVC's reference `model.diffusion` / `model.modules` (torchaudio/librosa stubbed:
    only FastGL in model/utils.py:10-12 needs them, and it is off the sampling path)."""
    if not os.path.isdir(os.path.join(REF_ROOT, "DiffVC", "model")):
        raise RuntimeError("/root/reference is not mounted")
    saved = _purge("model")
    stubs = {}
    for name in ("torchaudio", "librosa", "librosa.filters"):
        if name not in sys.modules:
            stubs[name] = types.ModuleType(name)
            sys.modules[name] = stubs[name]
    if "librosa.filters" in stubs:
        stubs["librosa.filters"].mel = lambda *a, **k: None
    path = os.path.join(REF_ROOT, "DiffVC")
    sys.path.insert(0, path)
    try:
        import model  # noqa: F401
        import model.diffusion  # noqa: F401
        import model.encoder  # noqa: F401
        import model.postnet  # noqa: F401
        mods = {k: v for k, v in sys.modules.items() if k == "model" or k.startswith("model.")}
    finally:
        sys.path.remove(path)
        _purge("model")
        sys.modules.update(saved)
        for name in stubs:
            sys.modules.pop(name, None)
    return types.SimpleNamespace(diffusion=mods["model.diffusion"], modules=mods["model.modules"],
                                 vc=mods.get("model.vc"), encoder=mods["model.encoder"], postnet=mods["model.postnet"])


def load_hifigan():
    """Return the reference HiFi-GAN `models` / `env` modules (Grad-TTS/hifi-gan/; top-level names `models`, `env`,
    `xutils` exactly as inference.py:19-20 imports them after sys.path.append('./hifi-gan/'))."""
    path = os.path.join(REF_ROOT, "Grad-TTS", "hifi-gan")
    if not os.path.isdir(path):
        raise RuntimeError("/root/reference is not mounted")
    names = ("models", "env", "xutils")
    saved = {k: sys.modules.pop(k) for k in names if k in sys.modules}
    stubs = {}
    try:
        import matplotlib  # noqa: F401  (xutils.py:5-8 imports it for plotting helpers)
    except Exception:
        for name in ("matplotlib", "matplotlib.pylab"):
            stubs[name] = types.ModuleType(name)
            sys.modules[name] = stubs[name]
        stubs["matplotlib"].use = lambda *a, **k: None
        stubs["matplotlib"].pylab = stubs["matplotlib.pylab"]
    sys.path.insert(0, path)
    try:
        import models  # noqa: F401
        import env  # noqa: F401
        mods = {k: sys.modules[k] for k in names}
    finally:
        sys.path.remove(path)
        for k in names:
            sys.modules.pop(k, None)
        sys.modules.update(saved)
        for name in stubs:
            sys.modules.pop(name, None)
    return types.SimpleNamespace(models=mods["models"], env=mods["env"], Generator=mods["models"].Generator,
                                 AttrDict=mods["env"].AttrDict)
