"""CPU: the oracle restatement reproduces the committed outputs of the reference's own modules."""
import numpy as np
import pytest
import torch

from conftest import golden
from oracle import gradtts_oracle as O
from oracle import mas as MAS


def _wsum(sd):
    return float(sum(float(v.double().abs().sum()) for v in sd.values()))


def _t(a):
    return torch.from_numpy(np.asarray(a))


def test_fixture_weights_reproducible():
    g = golden("est_1spk.npz")
    sd = O.make_estimator_state(seed=int(g["seed"]))
    assert abs(_wsum(sd) - float(g["wsum"])) <= 1e-6 * float(g["wsum"])


def test_estimator_single_speaker_matches_reference_output():
    g = golden("est_1spk.npz")
    sd = O.make_estimator_state(seed=int(g["seed"]))
    est = O.estimator_forward(sd, _t(g["z"]), _t(g["mask"]), _t(g["mu"]), _t(g["t"]))
    # same ops, same library, same machine class: identical up to thread-count noise
    assert torch.allclose(est, _t(g["est"]), rtol=0, atol=2e-5)


def test_estimator_multispeaker_matches_reference_output():
    g = golden("est_3ch.npz")
    sd = O.make_estimator_state(seed=int(g["seed"]), n_spks=4)
    assert abs(_wsum(sd) - float(g["wsum"])) <= 1e-6 * float(g["wsum"])
    est = O.estimator_forward(sd, _t(g["z"]), _t(g["mask"]), _t(g["mu"]), _t(g["t"]), _t(g["spk"]))
    assert torch.allclose(est, _t(g["est"]), rtol=0, atol=2e-5)


def test_reverse_diffusion_ode_matches_reference_output():
    g = golden("rd_ode.npz")
    sd = O.make_estimator_state(seed=int(g["seed"]))
    out = O.reverse_diffusion(sd, _t(g["z"]), _t(g["mask"]), _t(g["mu"]), int(g["n"]))
    ref = _t(g["out"])
    assert (out - ref).abs().max() <= 1e-5 * ref.abs().max()


def test_reverse_diffusion_sde_matches_reference_output():
    g = golden("rd_sde.npz")
    sd = O.make_estimator_state(seed=int(g["seed"]))
    out = O.reverse_diffusion(sd, _t(g["z"]), _t(g["mask"]), _t(g["mu"]), int(g["n"]), stoc=True,
                              noise=_t(g["noise"]))
    ref = _t(g["out"])
    assert (out - ref).abs().max() <= 1e-5 * ref.abs().max()


def test_masked_frames_are_zero_and_stats_include_padding():
    # quirk: GroupNorm statistics include masked frames -> changing padded input changes output
    sd = O.make_estimator_state(seed=0)
    inp = O.make_inputs(2, 32)
    t = torch.full((2,), 0.5)
    est = O.estimator_forward(sd, inp["z"], inp["mask"], inp["mu"], t)
    assert float((est * (1 - inp["mask"])).abs().max()) == 0.0


def test_mas_port_matches_compiled_reference_golden():
    g = golden("mas.npz")
    path = MAS.maximum_path_port(_t(g["value"]), _t(g["mask"]).float())
    assert torch.equal(path.to(torch.uint8), _t(g["path"]))


@pytest.mark.skipif(not MAS.ref_available(), reason="oracle/_ref not built and /root/reference absent")
def test_mas_port_equals_compiled_reference_random():
    g = torch.Generator().manual_seed(11)
    for b, tx, ty in [(4, 31, 90), (3, 1, 7), (2, 50, 50), (6, 13, 200)]:
        value = torch.randn(b, tx, ty, generator=g) * 4
        xl = torch.randint(1, tx + 1, (b,), generator=g)
        yl = torch.maximum(torch.randint(1, ty + 1, (b,), generator=g), xl)   # t_y >= t_x (defined regime)
        mask = (O.sequence_mask(xl, tx).unsqueeze(-1) * O.sequence_mask(yl, ty).unsqueeze(1)).float()
        assert torch.equal(MAS.maximum_path_port(value, mask), MAS.maximum_path_ref(value, mask))


def test_utils_match_reference_outputs():
    g = golden("utils.npz")
    assert np.array_equal(O.sequence_mask(_t(g["lens"]), 9).numpy(), g["seqmask"])
    assert [O.fix_len_compatibility(n) for n in range(0, 20)] == list(g["fixlen"])
    assert np.array_equal(O.generate_path(_t(g["dur"]), _t(g["pmask"])).numpy(), g["path"])


def test_diffvc_oracle_matches_reference_outputs():
    from oracle import diffvc_oracle as V
    g = golden("vc_dim64.npz")
    sd = V.make_state(dim_base=64, dim_cond=128, use_ref_t=True, seed=int(g["seed"]))
    assert abs(_wsum(sd) - float(g["wsum"])) <= 1e-6 * float(g["wsum"])
    est = V.estimator_forward(sd, _t(g["z"]), _t(g["mask"]), _t(g["mean"]), _t(g["xt_ref"]), _t(g["ref_mask"]), _t(g["c"]),
                              _t(g["t"]))
    assert torch.allclose(est, _t(g["est"]), rtol=0, atol=2e-5)
    for mode in ("pf", "em", "ml"):
        out = V.reverse_diffusion(sd, _t(g["z"]), _t(g["mask"]), _t(g["mean"]), _t(g["ref"]), _t(g["ref_mask"]),
                                  _t(g["mean_ref"]), _t(g["c"]), 3, mode, noise=_t(g["noise"]))
        ref = _t(g["out_" + mode])
        assert (out - ref).abs().max() <= 1e-5 * ref.abs().max()
