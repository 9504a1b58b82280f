"""Build libgradtts_gfx950.so in-tree with hipcc (gfx950 only; cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["plan.hip", "conv_mfma.hip", "conv_ws.hip", "conv_up.hip", "conv_up_ws.hip", "attn.hip", "misc.hip", "pack.hip", "mas.hip", "vc.hip", "glue.hip", "voc.hip", "enc.hip", "train.hip", "train_norm.hip", "train_wgrad.hip", "train_attn.hip", "train_elem.hip", "train_inglu.hip", "postnet.hip", "ubench.hip"]
HEADERS = ["common.h", "kernels.h", "conv1d.h", os.path.join("..", "..", "include", "gradtts_abi.h")]
# The SLP vectoriser packs the GroupNorm / Mish / split arithmetic of the conv prologue into v_pk_*_f32.  Beside MFMAs a
# packed f32 op costs more issue time than the two scalar ops it replaces (MI355X guide; measured here: -1.6 % per U-Net call).
PER_FILE_FLAGS = {"conv_mfma.hip": ["-fno-slp-vectorize"], "conv_ws.hip": ["-fno-slp-vectorize"], "conv_up.hip": ["-fno-slp-vectorize"], "conv_up_ws.hip": ["-fno-slp-vectorize"]}      # (attn.hip: measured the other way, 185 vs 173 us -- it is VALU-bound and profits from the packing)
LIB = os.path.join(HERE, os.environ.get("GTTS_LIB_NAME", "libgradtts_gfx950.so"))


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    bdir = os.path.join(HERE, "build", os.path.basename(LIB).replace(".so", ""))
    os.makedirs(bdir, exist_ok=True)
    procs = []
    for src in SOURCES:
        obj = os.path.join(bdir, src.replace(".hip", ".o"))
        objs.append(obj)
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
        if src not in os.environ.get("GTTS_NO_PERFILE", "").split(","):      # A/B builds
            cmd += PER_FILE_FLAGS.get(src, [])
        cmd += os.environ.get("GTTS_EXTRA_FLAGS", "").split()
        if src in ("conv_mfma.hip", "conv_ws.hip"):
            cmd += os.environ.get("GTTS_CONV_FLAGS", "").split()       # A/B builds of the convolution only
        cmd += ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, pr in procs:
        out, _ = pr.communicate()
        if pr.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError("hipcc failed on %s" % src)
        if verbose and out:
            sys.stderr.write(out.decode())
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    subprocess.check_call(cmd)
    return LIB


# Reference variant for the GPU test suite: the fused GroupNorm finalize with the textbook agent-scope release / acquire fences
# (common.h, GTTS_FENCED_FINALIZE).  Only the two files that contain the hand-off are recompiled; everything else links from the
# product's objects.  tests/test_gpu_fenced.py compares the two libraries bit for bit.
FENCED_LIB = os.path.join(HERE, "libgtts_fenced.so")
FENCED_SOURCES = ["conv_mfma.hip", "conv_ws.hip"]


def build_fenced(force=False):
    product = build()
    deps = [os.path.join(CSRC, f) for f in FENCED_SOURCES + HEADERS] + [product, os.path.abspath(__file__)]
    if not force and os.path.exists(FENCED_LIB) and all(os.path.getmtime(d) <= os.path.getmtime(FENCED_LIB) for d in deps):
        return FENCED_LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    pdir = os.path.join(HERE, "build", os.path.basename(product).replace(".so", ""))
    fdir = os.path.join(HERE, "build", "libgtts_fenced")
    os.makedirs(fdir, exist_ok=True)
    procs, objs = [], []
    for src in SOURCES:
        if src not in FENCED_SOURCES:
            objs.append(os.path.join(pdir, src.replace(".hip", ".o")))
            continue
        obj = os.path.join(fdir, src.replace(".hip", ".o"))
        objs.append(obj)
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-DGTTS_FENCED_FINALIZE=1"]
        cmd += PER_FILE_FLAGS.get(src, []) + ["-c", os.path.join(CSRC, src), "-o", obj]
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, pr in procs:
        out, _ = pr.communicate()
        if pr.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError("hipcc failed on %s (fenced variant)" % src)
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", FENCED_LIB] + objs)
    return FENCED_LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_fenced(force="--force" in sys.argv))
