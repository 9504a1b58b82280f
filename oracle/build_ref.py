"""TEST INFRASTRUCTURE ONLY.  Compile the reference's own native component where it lies.

Source : /root/reference/Grad-TTS/model/monotonic_align/core.pyx  (never copied into the repo)
Outputs: oracle/_ref/core.<abi>.so   (+ generated C under oracle/_ref/build/; all git-ignored)

The reference's own recipe is `python setup.py build_ext --inplace` (monotonic_align/setup.py:7-11), which
writes next to the read-only source; this recipe runs the same two steps (cython, then the C compiler)
with explicit output paths instead.
"""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
PYX = "/root/reference/Grad-TTS/model/monotonic_align/core.pyx"


def ref_so_path():
    return os.path.join(HERE, "_ref", "core" + sysconfig.get_config_var("EXT_SUFFIX"))


def build(force=False):
    out = ref_so_path()
    if not os.path.exists(PYX):
        return out if os.path.exists(out) else None
    if os.path.exists(out) and not force and os.path.getmtime(out) >= os.path.getmtime(PYX):
        return out
    import numpy
    bdir = os.path.join(HERE, "_ref", "build")
    os.makedirs(bdir, exist_ok=True)
    c_file = os.path.join(bdir, "core.c")
    subprocess.check_call([sys.executable, "-m", "cython", "-3", PYX, "-o", c_file],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    cmd = ["gcc", "-O2", "-fPIC", "-shared", "-w", "-I", sysconfig.get_paths()["include"],
           "-I", numpy.get_include(), c_file, "-o", out]
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
