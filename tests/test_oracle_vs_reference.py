"""CPU, build container only: live comparison of the restatement with the reference's Python."""
import pytest
import torch

from oracle import gradtts_oracle as O
from oracle import ref_loader

pytestmark = pytest.mark.skipif(not ref_loader.available(), reason="/root/reference not mounted")


@pytest.fixture(scope="module")
def ref():
    return ref_loader.load_gradtts()


def test_state_dict_layout_equals_reference(ref):
    for n_spks in (1, 5):
        dec = ref.diffusion.Diffusion(80, 64, n_spks, 64, 0.05, 20.0, 1000)
        want = {k[len("estimator."):]: tuple(v.shape) for k, v in dec.state_dict().items()}
        got = {k: tuple(v.shape) for k, v in O.make_estimator_state(n_spks=n_spks).items()}
        assert got == want


def test_estimator_bit_identical_with_taps(ref):
    torch.manual_seed(0)
    sd = O.make_estimator_state(seed=7)
    dec = ref.diffusion.Diffusion(80, 64, 1, 64, 0.05, 20.0, 1000)
    dec.estimator.load_state_dict(sd, strict=True)
    inp = O.make_inputs(2, 48, seed=3)
    t = torch.tensor([0.15, 0.95])
    taps = {}
    with torch.no_grad():
        a = dec.estimator(inp["z"], inp["mask"], inp["mu"], t)
    b = O.estimator_forward(sd, inp["z"], inp["mask"], inp["mu"], t, taps=taps)
    assert torch.equal(a, b)
    assert "downs.0.0.b1.raw" in taps and "ups.1.3.out" in taps and "final_block.raw" in taps


def test_reverse_diffusion_bit_identical(ref):
    sd = O.make_estimator_state(seed=1)
    dec = ref.diffusion.Diffusion(80, 64, 1, 64, 0.05, 20.0, 1000)
    dec.estimator.load_state_dict(sd, strict=True)
    inp = O.make_inputs(1, 32, seed=4)
    with torch.no_grad():
        a = dec(inp["z"], inp["mask"], inp["mu"], 3)
    b = O.reverse_diffusion(sd, inp["z"], inp["mask"], inp["mu"], 3)
    assert torch.equal(a, b)
