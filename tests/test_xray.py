"""xvr_amd.xray.XrayPreparation against vectors produced by the reference's own ``_preprocess_xray``
(/root/reference/src/xvr/io/xray.py:93-129; tests/golden/make_golden_xray.py) -- bit for bit: the pipeline is a handful of
elementwise and reduction ops in a fixed order."""
from pathlib import Path

import numpy as np
import pytest
import torch

from xvr_amd.xray import XrayPreparation

GOLD = np.load(Path(__file__).parent / "golden" / "xvr_reference_xray.npz")


def _check(device):
    for k in range(int(GOLD["n_single"])):
        crop, background, linearize, _odd = (int(v) for v in GOLD[f"c{k}_cfg"])
        prep = XrayPreparation(trim=crop, subtract_background=bool(background), linearize=bool(linearize))
        got = prep(torch.from_numpy(GOLD[f"c{k}_in"]).to(device)).cpu()
        want = torch.from_numpy(GOLD[f"c{k}_out"])
        assert got.shape == want.shape, (k, got.shape, want.shape)
        if device == "cpu":
            assert torch.equal(got, want), (k, (got - want).abs().max())
        else:   # (the device's log differs from the host's in the last place)
            assert torch.allclose(got, want, rtol=2e-6, atol=2e-7), (k, (got - want).abs().max())
    frames = torch.from_numpy(GOLD["m_in"]).to(device)
    for j, how in enumerate(("max", "sum", 3, None)):
        got = XrayPreparation(trim=4, frames=how)(frames).cpu()
        want = torch.from_numpy(GOLD[f"m{j}_out"])
        assert got.shape == want.shape and torch.allclose(got, want, rtol=2e-6, atol=2e-6 if how == "sum" else 2e-7), (how,)


def test_xray_preparation_reproduces_the_reference_vectors():
    _check("cpu")


@pytest.mark.gpu
def test_xray_preparation_on_the_device():
    _check("cuda")


def test_xray_preparation_edge_cases():
    x = torch.rand(1, 1, 3, 16, 16)
    assert XrayPreparation(frames=lambda t: t.mean(dim=2))(x).shape == (1, 1, 16, 16)
    with pytest.raises(ValueError, match="unknown frame reduction"):
        XrayPreparation(frames="median")(x)
    with pytest.raises(ValueError):
        XrayPreparation(trim=16)(x[:, :, 0])
    with pytest.raises(ValueError):
        XrayPreparation()(torch.zeros(16, 16))
    flat = XrayPreparation(linearize=False)(torch.full((1, 1, 8, 8), 7.0))     # a constant image: 0 / 1e-6, not NaN
    assert torch.equal(flat, torch.zeros(1, 1, 8, 8))


def test_registrar_records_the_preparation_it_applied():
    """parameters.pt's "xray" block (/root/reference/src/xvr/registrar/base.py:367-373) names what was really done to the pixels."""
    from xvr_amd.registrar import Registrar

    reg = Registrar.__new__(Registrar)
    reg.crop, reg.xray_preparation = 6, XrayPreparation(trim=6, subtract_background=True, linearize=True, frames="sum")
    assert reg.xray_block("a.dcm") == {"filename": "a.dcm", "crop": 6, "subtract_background": True, "linearize": True, "reducefn": "sum"}
    raw = torch.rand(1, 1, 40, 40)
    assert torch.equal(reg.prepare_xray(raw), reg.xray_preparation(raw))
