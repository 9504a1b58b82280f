"""CPU: the C-ABI library loads and exports exactly the symbols include/gradtts_abi.h declares; host-only entry
points (plan creation, layouts, argument validation) behave; no compute call is made (no GPU here)."""
import ctypes
import os
import re
import subprocess

import pytest

from conftest import ROOT, pkg


def _declared():
    src = open(os.path.join(ROOT, "include", "gradtts_abi.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gtts_[a-z_0-9]+)\s*\(", src)))


def test_header_symbols_are_exported():
    S = pkg()
    assert os.path.exists(S._lib.LIB_PATH), "run __graft_entry__.build() first"
    out = subprocess.check_output(["nm", "-D", "--defined-only", S._lib.LIB_PATH]).decode()
    exported = set(re.findall(r"\bT (gtts_[a-z_0-9]+)", out))
    declared = _declared()
    assert len(declared) >= 15
    assert set(declared) <= exported, sorted(set(declared) - exported)
    lib = ctypes.CDLL(S._lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name)


def test_cfg_struct_layout_matches_the_header(tmp_path):
    """The ctypes mirror of gtts_unet_cfg (and the copy in INTEGRATION.md) must have the C compiler's layout of the
    struct in include/gradtts_abi.h: compile the header with gcc and compare size and every field offset."""
    S = pkg()
    fields = [f[0] for f in S._lib.UnetCfg._fields_]
    prog = ['#include <stdio.h>', '#include <stddef.h>', '#include "gradtts_abi.h"', 'int main(void) {',
            '  printf("%zu\\n", sizeof(gtts_unet_cfg));']
    prog += ['  printf("%%zu\\n", offsetof(gtts_unet_cfg, %s));' % f for f in fields]
    prog += ['  return 0; }']
    src = tmp_path / "cfg.c"
    src.write_text("\n".join(prog))
    exe = tmp_path / "cfg"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    nums = [int(x) for x in subprocess.check_output([str(exe)]).decode().split()]
    assert nums[0] == ctypes.sizeof(S._lib.UnetCfg)
    assert nums[1:] == [getattr(S._lib.UnetCfg, f).offset for f in fields]
    # the binding stub shown to reference maintainers lists the same fields in the same order
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    stub = doc[doc.index("class _Cfg"):doc.index("_L.gtts_packed_weight_bytes.restype")]
    assert re.findall(r'\("([a-z_]+)", ctypes', stub) == fields


def test_plan_layout_and_validation():
    S = pkg()
    from oracle import gradtts_oracle as O
    for n_spks, n in ((1, 172), (247, 176)):
        p = S.Plan(n_spks=n_spks)
        layout = p.param_layout()
        sd = O.make_estimator_state(n_spks=n_spks)
        assert [k for k, _ in layout] == list(sd.keys())            # reference registration order
        assert all(tuple(sd[k].shape) == s for k, s in layout)
        assert len(layout) == n
        assert p.packed_bytes() > 30e6
        assert p.workspace_bytes(16, 1024) > p.workspace_bytes(2, 64) > 0
    p = S.Plan()
    with pytest.raises(RuntimeError):
        p.workspace_bytes(1, 30)          # T % 4 != 0
    with pytest.raises(RuntimeError):
        S.Plan(groups=4)
    with pytest.raises(RuntimeError):
        S.Plan(dim=48)
    assert S.Plan(keep_intermediates=True).workspace_bytes(2, 64) > S.Plan().workspace_bytes(2, 64)


def test_conv_kernel_plan_option():
    """gtts_unet_cfg.conv_ws (ABI 3): default by network width, never for the single-pass bf16 modes; the sub-batch stream
    default follows it; the choice changes the GroupNorm partial-slot layout and therefore the workspace, never the weights."""
    import copy
    S = pkg()
    grad, vc = S.Plan(), S.Plan(dim=256, arch=1)
    assert grad.conv_ws is False and grad._nstreams == 3
    assert vc.conv_ws is True and vc._nstreams == 2
    forced = S.Plan(conv_ws=True)
    assert forced.conv_ws is True and forced._nstreams == 2 and int(forced.cfg.conv_ws) == 1
    assert S.Plan(conv_ws=True, precision=S.PREC_BF16_STORE).conv_ws is False
    assert S.Plan(conv_ws=True, streams=3)._nstreams == 3
    assert forced.packed_bytes() == grad.packed_bytes()
    assert [k for k, _ in forced.param_layout()] == [k for k, _ in grad.param_layout()]
    assert forced.workspace_bytes(2, 64) > 0 and grad.workspace_bytes(2, 64) > 0
    assert copy.deepcopy(forced).conv_ws is True          # plans rebuild from their constructor arguments


def test_no_cpu_fallback():
    import torch
    S = pkg()
    p = S.Plan()
    with pytest.raises(RuntimeError, match="HIP device"):
        p.estimator_forward(None, torch.zeros(1, 80, 32), torch.ones(1, 1, 32), torch.zeros(1, 80, 32), torch.ones(1))
    with pytest.raises(RuntimeError, match="HIP device"):
        p.reverse_diffusion(None, torch.zeros(1, 80, 32), torch.ones(1, 1, 32), torch.zeros(1, 80, 32), 2)


def test_abi_contract_surface():
    """ABI v2: side streams are registered by the caller; enqueueing calls take a non-const plan; diagnostics are not
    compiled into the product library (no GTTS_SKIP_OPS / trace entry points)."""
    S = pkg()
    L = S._lib.lib()
    assert L.gtts_abi_version() == 6
    p = S.Plan(streams=0)
    assert L.gtts_plan_set_streams(p._h, None, 0) == 0
    assert L.gtts_plan_set_streams(p._h, None, 3) != 0            # null stream array
    assert b"null" in L.gtts_last_error()
    assert L.gtts_plan_set_streams(p._h, None, 9) != 0
    # registering side streams grows the workspace (one slice per sub-batch) -- pure host arithmetic
    w0 = p.workspace_bytes(16, 1024)
    arr = (ctypes.c_void_p * 3)(1, 2, 3)          # opaque handles: only stored until a sampler call uses them
    try:
        rc = L.gtts_plan_set_streams(p._h, arr, 3)
    except Exception:          # pragma: no cover
        rc = -1
    if rc == 0:                # event creation needs a HIP device; on a CPU-only box the call fails cleanly instead
        assert p.workspace_bytes(16, 1024) >= w0
        assert L.gtts_plan_set_streams(p._h, None, 0) == 0
    out = subprocess.check_output(["nm", "-D", "--defined-only", S._lib.LIB_PATH]).decode()
    assert "gtts_debug_trace" not in out
    blob = open(S._lib.LIB_PATH, "rb").read()
    assert b"GTTS_SKIP_OPS" not in blob and b"GTTS_STREAMS" not in blob and b"GTTS_MAS_KERNEL" not in blob
    # gtts_bcast_weights validates its arguments before touching RCCL
    assert L.gtts_bcast_weights(None, 0, 0, None, None) == S._lib.lib().gtts_bcast_weights(None, 0, 0, None, None) != 0


def test_mas_cpu_twin_bit_exact():
    """gtts_mas_maximum_path_cpu (host C++, the any-device behaviour of monotonic_align/__init__.py:8-23) against the
    golden paths of the reference, the plain-C oracle and -- where present -- the compiled reference Cython."""
    import numpy as np
    import torch
    from conftest import golden
    from oracle import gradtts_oracle as O
    from oracle import mas as MAS
    S = pkg()
    g = golden("mas.npz")
    path = S.mas_maximum_path(torch.from_numpy(g["value"]), torch.from_numpy(g["mask"]).float())
    assert path.dtype == torch.float32 and not path.is_cuda
    assert torch.equal(path.to(torch.uint8), torch.from_numpy(g["path"]))
    MA = __import__("importlib").import_module("speech-backbones_amd.model.monotonic_align")
    # (the last two: more tokens than frames -- a degenerate band, core.pyx:18; still a function of the input alone)
    for b, tx, ty in [(4, 31, 90), (3, 1, 7), (2, 50, 50), (5, 60, 333), (2, 300, 1000), (3, 40, 25), (2, 300, 120)]:
        gen = torch.Generator().manual_seed(b * 1000 + tx)
        value = torch.randn(b, tx, ty, generator=gen) * 4
        xl = torch.randint(1, tx + 1, (b,), generator=gen)
        xl[0] = tx
        yl = torch.randint(1, ty + 1, (b,), generator=gen)
        if tx <= ty:
            yl = torch.maximum(yl, xl)
        yl[0] = ty
        mask = (O.sequence_mask(xl, tx).unsqueeze(-1) * O.sequence_mask(yl, ty).unsqueeze(1)).float()
        got = MA.maximum_path(value, mask)
        assert torch.equal(got, MAS.maximum_path_port(value, mask))
        if MAS.ref_available():
            assert torch.equal(got, MAS.maximum_path_ref(value, mask))


def test_oracle_is_test_infrastructure_only():
    """Nothing under the package or under tools/ imports oracle/; bench.py touches it only inside its CPU-baseline legs (the functions
    whose result is a `cpu_baseline` / `cpu_ms` entry), never for weights, inputs or the measured path."""
    import ast
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for top in ("speech-backbones_amd", "tools"):
        for dirpath, _, files in os.walk(os.path.join(root, top)):
            for fn in files:
                if fn.endswith(".py"):
                    src = open(os.path.join(dirpath, fn)).read()
                    assert "from oracle" not in src and "import oracle" not in src, os.path.join(dirpath, fn)
    tree = ast.parse(open(os.path.join(root, "bench.py")).read())
    allowed = {"cpu_baseline", "cpu_baseline_vc", "bench_hifigan", "mas"}

    def walk(node, fn_stack):
        if isinstance(node, (ast.FunctionDef, ast.AsyncFunctionDef)):
            fn_stack = fn_stack + [node.name]
        if isinstance(node, ast.ImportFrom) and (node.module or "").split(".")[0] == "oracle":
            assert fn_stack and fn_stack[-1] in allowed, "bench.py imports oracle in %s" % (fn_stack or ["<module>"])
        if isinstance(node, ast.Import):
            assert all(a.name.split(".")[0] != "oracle" for a in node.names)
        for ch in ast.iter_child_nodes(node):
            walk(ch, fn_stack)
    walk(tree, [])


def test_batch_invariant_kernels_do_not_rely_on_implicit_fp_contraction():
    """Results must not depend on how utterances are batched, and the batch size picks between template instances of the 3x3
    kernels (regular / half-height tiles, eight- / three-wave persistent form).  Those instances agree bit for bit only if the
    compiler fuses the same multiply-add pairs in each; with -ffp-contract=fast it did not always.  The
    two kernel files therefore switch implicit contraction off and spell every fused multiply-add out; this test keeps it so."""
    csrc = os.path.join(ROOT, "speech-backbones_amd", "csrc")
    for name in ("conv_mfma.hip", "conv_ws.hip"):
        src = open(os.path.join(csrc, name)).read()
        head = src[:src.index("namespace gtts {")]
        assert "#pragma clang fp contract(off)" in head, name
        code = re.sub(r"//[^\n]*", "", src)
        # the statistics and prologue expressions that used to be written as a * b + c
        assert not re.search(r"\+=\s*\w+(\[\w+\])?\s*\*\s*\w+(\[\w+\])?\s*;", code), name
        assert "fmaf(" in code, name
    mish = open(os.path.join(csrc, "common.h")).read()
    body = mish[mish.index("float mish_f(float x)"):]
    assert "#pragma clang fp contract(off)" in body[:body.index("}")]
