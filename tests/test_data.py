"""Host-side data plumbing: NIfTI reading + RAS canonicalisation, HU -> density, phantoms."""
import gzip
import struct

import numpy as np
import pytest
import torch

from xvr_amd.data import make_phantom, read, read_nifti, transform_hu_to_density


def _write_nifti(path, data, affine, dtype="<i2", slope=1.0, inter=0.0):
    hdr = bytearray(352)
    struct.pack_into("<i", hdr, 0, 348)
    struct.pack_into("<8h", hdr, 40, 3, *data.shape, 1, 1, 1, 1)
    code = {"<i2": 4, "<f4": 16, "<u1": 2}[dtype]
    struct.pack_into("<hh", hdr, 70, code, np.dtype(dtype).itemsize * 8)
    struct.pack_into("<8f", hdr, 76, 1.0, *np.linalg.norm(affine[:3, :3], axis=0), 0, 0, 0, 0)
    struct.pack_into("<3f", hdr, 108, 352.0, slope, inter)
    struct.pack_into("<hh", hdr, 252, 0, 1)
    for r in range(3):
        struct.pack_into("<4f", hdr, 280 + 16 * r, *affine[r])
    hdr[344:348] = b"n+1\0"
    payload = bytes(hdr) + np.asarray(data, dtype=dtype).tobytes(order="F")
    opener = gzip.open if str(path).endswith(".gz") else open
    with opener(path, "wb") as f:
        f.write(payload)


def test_nifti_roundtrip_and_ras_canonicalisation(tmp_path):
    rng = np.random.default_rng(0)
    data = rng.integers(-1000, 2000, size=(5, 6, 7)).astype(np.int16)
    # an LPS-oriented volume: x and y flipped, anisotropic spacing, an offset
    affine = np.array([[-0.8, 0, 0, 40.0], [0, -0.9, 0, 55.0], [0, 0, 2.5, -30.0], [0, 0, 0, 1.0]])
    path = tmp_path / "ct.nii.gz"
    _write_nifti(path, data, affine)
    out, aff = read_nifti(path)
    assert out.shape == (5, 6, 7) and out.dtype == np.float32
    assert np.all(np.diag(aff)[:3] > 0), "canonical RAS+: positive diagonal"
    assert np.allclose(np.abs(np.diag(aff)[:3]), [0.8, 0.9, 2.5], atol=1e-6)
    # the same physical point keeps its value: voxel (1,2,3) of the file
    world = affine @ np.array([1, 2, 3, 1.0])
    ijk = np.linalg.solve(aff, world)[:3].round().astype(int)
    assert out[tuple(ijk)] == data[1, 2, 3]
    # axis permutation (file stored as z, x, y)
    perm_aff = np.array([[0, 1.1, 0, 1.0], [0, 0, 1.2, 2.0], [1.3, 0, 0, 3.0], [0, 0, 0, 1.0]])
    p2 = tmp_path / "perm.nii"
    _write_nifti(p2, data, perm_aff, dtype="<f4", slope=2.0, inter=1.0)
    out2, aff2 = read_nifti(p2)
    assert out2.shape == (6, 7, 5) and np.allclose(np.diag(aff2)[:3], [1.1, 1.2, 1.3], atol=1e-6)
    w = perm_aff @ np.array([4, 5, 6, 1.0])
    ijk = np.linalg.solve(aff2, w)[:3].round().astype(int)
    assert abs(out2[tuple(ijk)] - (2.0 * data[4, 5, 6] + 1.0)) < 1e-3


def test_read_from_file_builds_a_centred_density_subject(tmp_path):
    data = np.full((8, 8, 8), -1000, dtype=np.int16)
    data[2:6, 2:6, 2:6] = 100
    data[3:5, 3:5, 3:5] = 900
    path = tmp_path / "ct.nii.gz"
    _write_nifti(path, data, np.diag([1.5, 1.5, 1.5, 1.0]))
    sub = read(str(path), orientation="PA", bone_attenuation_multiplier=2.0)
    assert sub.volume.shape == (8, 8, 8) and sub.orientation == "PA"
    assert float(sub.density.min()) == 0.0 and abs(float(sub.density.max()) - 1.0) < 1e-6
    assert np.allclose(sub.get_center(), (0.0, 0.0, 0.0), atol=1e-5)
    assert torch.allclose(torch.diag(sub.affine)[:3], torch.tensor([1.5, 1.5, 1.5]))


def test_transform_hu_to_density_piecewise():
    """the definition (oracle) and the loader's one-off host pass agree; the per-step product function has no CPU path"""
    from oracle.data_restated import transform_hu_to_density as oracle_hu
    from xvr_amd.data import _density_at_load

    hu = torch.tensor([[-1000.0, -800.0, -500.0, 0.0, 350.0, 351.0, 1000.0]]).reshape(1, 1, 7).expand(2, 2, 7)
    d = oracle_hu(hu, 3.0)
    assert torch.equal(d, _density_at_load(hu, 3.0))
    g = torch.Generator().manual_seed(5)
    big = torch.rand(9, 8, 7, generator=g) * 2800 - 1100
    assert torch.equal(oracle_hu(big, 4.2), _density_at_load(big, 4.2))
    with pytest.raises(RuntimeError, match="no CPU path"):
        transform_hu_to_density(hu, 3.0)
    assert torch.isclose(d[0, 0, 0], d[0, 0, 2]) and d[0, 0, 0] == 0  # air -> soft-tissue minimum
    assert d[0, 0, 6] == 1.0 and d[0, 0, 5] > d[0, 0, 4]                # bone scaled, normalised to [0, 1]


def test_phantom_is_seeded_and_labelled():
    a, la = make_phantom(24, n_labels=4, seed=3)
    b, _ = make_phantom(24, n_labels=4, seed=3)
    assert torch.equal(a, b) and a.min() >= 0 and a.max() <= 1
    assert set(la.unique().tolist()) <= {0.0, 1.0, 2.0, 3.0} and la.max() >= 1


def test_training_checkpoint_has_the_reference_schema(tmp_path):
    """NNNN.pth: the keys, file name pattern and reload rules of /root/reference/src/xvr/model/trainer.py:318-332 and
    /root/reference/src/xvr/model/utils.py:132-150,176-183."""
    from datetime import datetime

    from xvr_amd.training import CHECKPOINT_KEYS, load_checkpoint, restore_from_checkpoint, save_checkpoint

    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 13))
    opt = torch.optim.Adam(net.parameters(), lr=3e-4)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda s: 1.0 / (1 + s))
    for _ in range(3):
        net(torch.randn(4, 8)).square().mean().backward()
        opt.step()
        sched.step()
        opt.zero_grad()
    config = dict(volpath="ct.nii.gz", sdd=1020.0, height=128, delx=2.1764375, renderer="trilinear", batch_size=116,
                  n_grad_accum_itrs=4, lr=3e-4)
    path, nxt = save_checkpoint(tmp_path, net, opt, sched, itr=1000, model_number=7, config=config)
    assert path.name == "0007.pth" and nxt == 8
    raw = torch.load(path, weights_only=False)
    assert tuple(raw) == CHECKPOINT_KEYS and isinstance(raw["date"], datetime) and raw["config"] == config

    ckpt, start_itr, number = load_checkpoint(path, reuse_optimizer=True)
    assert (start_itr, number) == (1000, 7)
    assert load_checkpoint(path, reuse_optimizer=False)[1:] == (0, 0) and load_checkpoint(None) == (None, 0, 0)
    net2 = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 13))
    opt2 = torch.optim.Adam(net2.parameters(), lr=1.0)
    sched2 = torch.optim.lr_scheduler.LambdaLR(opt2, lambda s: 1.0 / (1 + s))
    restore_from_checkpoint(ckpt, net2, opt2, sched2, reuse_optimizer=True)
    x = torch.randn(5, 8)
    assert torch.equal(net(x), net2(x))
    assert opt2.state_dict()["state"][0]["step"] == opt.state_dict()["state"][0]["step"]
    assert sched2.last_epoch == sched.last_epoch == 3 and abs(opt2.param_groups[0]["lr"] - opt.param_groups[0]["lr"]) < 1e-12
    torch.save({"model_state_dict": {}}, tmp_path / "bad.pth")
    with pytest.raises(KeyError, match="missing"):
        load_checkpoint(tmp_path / "bad.pth")
