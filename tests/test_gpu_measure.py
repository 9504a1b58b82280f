"""GPU tests (-m gpu) of the ABI-4 measurement / batching entry points: the measured-ceiling micro-benchmarks return sane numbers,
the un-traced timeline records every launch of a sampler call on the stream it ran on, and the one-launch batched weight pack
(gtts_pack_batch) is bit-identical to the per-weight packs it replaces."""
import ctypes
import importlib

import pytest
import torch

from oracle import gradtts_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def S():
    assert torch.cuda.is_available()
    return importlib.import_module("speech-backbones_amd")


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def test_measured_ceilings_are_sane(S, dev):
    c = S._lib.measured_ceilings(dev, seconds=0.5)
    # random operands are throttled by the power budget, zero operands are not; nothing exceeds the quoted peaks by more than noise
    assert 800.0 < c["mfma_bf16_tflops_random"] < c["mfma_bf16_tflops_zero"] < 2700.0
    assert 800.0 < c["mfma_bf16_16x16x32_tflops_random"] < 2700.0
    assert 2000.0 < c["hbm_copy_gbs"] < 8200.0 and 2000.0 < c["hbm_triad_gbs"] < 8200.0 and 2000.0 < c["hbm_read_gbs"] < 8200.0
    assert 300.0 < c["gemm_bf16_tflops_random_hipblaslt"] < 2700.0


def test_ubench_argument_checks(S, dev):
    L = S._lib.lib()
    buf = torch.zeros(4096, dtype=torch.float32, device=dev)
    fl = ctypes.c_double(0.0)
    p, st = S._lib._ptr, S._lib._stream
    assert L.gtts_ubench_mfma(None, ctypes.c_size_t(8192), p(buf), 4, 10, ctypes.byref(fl), st()) != 0
    assert L.gtts_ubench_mfma(p(buf), ctypes.c_size_t(16), p(buf), 4, 10, ctypes.byref(fl), st()) != 0
    assert L.gtts_ubench_hbm(p(buf), p(buf), p(buf), ctypes.c_size_t(6), 0, 4, ctypes.byref(fl), st()) != 0
    assert L.gtts_ubench_hbm(p(buf), None, p(buf), ctypes.c_size_t(4096), 1, 4, ctypes.byref(fl), st()) != 0
    assert L.gtts_ubench_hbm(p(buf), p(buf), p(buf), ctypes.c_size_t(4096), 0, 4, ctypes.byref(fl), st()) == 0
    torch.cuda.synchronize()
    assert fl.value == 4096 * 8


@pytest.mark.parametrize("streams", [0, 3])
def test_timeline_records_every_launch_on_its_stream(S, dev, streams):
    plan = S.Plan(streams=streams)
    blob = plan.pack(O.make_estimator_state(seed=0), dev)
    inp = O.make_inputs(6, 64, seed=2)
    z, m, mu = (inp[k].to(dev) for k in ("z", "mask", "mu"))
    ref = plan.reverse_diffusion(blob, z, m, mu, 2)
    plan.profile(2)
    out = plan.reverse_diffusion(blob, z, m, mu, 2)
    torch.cuda.synchronize()
    recs = plan.profile_timeline()
    plan.profile(False)
    assert torch.equal(out, ref)                                   # recording events changes nothing
    parts = 3 if streams == 3 else 1
    used = sorted({r[1] for r in recs})
    assert used == ([0, 1, 2, 3] if streams == 3 else [0])
    # the same launches per sub-batch and step on every sub-batch stream; time_mlp and xt = z * mask once on the call's stream
    side = [r for r in recs if r[1] != 0] if streams == 3 else recs
    assert len(side) % (2 * parts) == 0 and len(side) // (2 * parts) >= 60
    assert all(r[3] >= r[2] for r in recs)
    assert plan.profile_timeline() == []                            # the record was cleared


def test_batched_pack_is_bit_identical_to_single_packs(S, dev):
    be = S._lib
    g = torch.Generator().manual_seed(0)
    specs, singles = [], []
    for kind, ci, co, t, shape in (("3x3", 64, 128, False, (128, 64, 3, 3)), ("3x3", 128, 64, True, (128, 64, 3, 3)),
                                   ("1x1", 128, 384, False, (384, 128, 1, 1)), ("1x1", 384, 128, True, (384, 128, 1, 1)),
                                   ("dn", 64, 64, False, (64, 64, 3, 3)), ("dn_T", 64, 64, False, (64, 64, 3, 3)),
                                   ("up", 128, 128, False, (128, 128, 4, 4))):
        w = torch.randn(*shape, generator=g).to(dev)
        specs.append((w, ci, co, t, kind))
    be.new_pack_generation()
    for w, ci, co, t, kind in specs:
        singles.append(be._packed_weight(w, ci, co, t, kind).clone())
    be.new_pack_generation()
    be.prepack(specs)
    torch.cuda.synchronize()
    for (w, ci, co, t, kind), one in zip(specs, singles):
        got = be._packed_weight(w, ci, co, t, kind)                 # a cache hit on the batched blob
        assert got.data_ptr() != one.data_ptr() and torch.equal(got, one), kind
    # the blobs are refilled in place at the next call and follow edits made through .data
    with torch.no_grad():
        specs[0][0].data.mul_(2.0)
    be.new_pack_generation()
    fresh = be._packed_weight(specs[0][0].clone(), 64, 128, False, "3x3").clone()
    be.new_pack_generation()
    be.prepack(specs)
    assert torch.equal(be._packed_weight(specs[0][0], 64, 128, False, "3x3"), fresh)
